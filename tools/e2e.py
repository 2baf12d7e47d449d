#!/usr/bin/env python3
"""End-to-end rate of an all-key-frame stream: compressed bytes in host memory -> MD5 of every decoded frame.

SURVEY.md 8(d) asks for this figure beside the kernel-only one, 8(f)1 for the feeder that makes it possible: key
frames are independently decodable, so T host threads run the entropy decoder (the C feeder releases the GIL) on
different frames, each writing the IR straight into a slot's pinned staging; the main thread uploads a batch,
launches the pixel path, and while the next batch is being parsed downloads the previous one, whose frames the same
pool hashes.  Everything the kernel-only number leaves out is in here: entropy decode, H2D of the dense IR
(3.3 B/px), D2H of the visible planes (1.5 B/px), MD5.

    python tools/e2e.py [frames] [batch] [threads] [MD5 on the device: 1 / 0]
"""
import hashlib
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(P, device=0, fixture="kf_1920x1080", nframes=1024, batch=128, threads=None, device_md5=None):
    from vp8_testlib import ivf_path, golden_md5
    w, h, frames = P.read_ivf(ivf_path(fixture))
    gold = golden_md5(fixture)
    nsrc = len(frames)
    threads = threads or max(1, min(32, (os.cpu_count() or 2) - 1))      # (more than 32 feeders gain nothing, 64 lose: the box runs under a 16-CPU quota)
    nbatch = (nframes + batch - 1) // batch
    ctx = P.Vp8Hip(device)
    # three sets of slots and frame buffers: batch k+1 is parsed while k is decoded and k-1 is downloaded and hashed
    ctx.configure(w, h, 3 * batch, 3 * batch)
    import ctypes
    c_void_p, c_size_t = ctypes.c_void_p, ctypes.c_size_t

    # pinned staging of the slots in the device form of include/vp8_ir.h: header, records, block stream, MVs
    maps = [ctx.ir_map_compact(s) for s in range(3 * batch)]   # created by the main thread
    counts = [0] * (3 * batch)                          # blocks the feeder wrote per slot
    h2d_bytes = [0]
    parsers = [P.Parser() for _ in range(threads)]
    free = list(range(threads))

    def parse(slot, data):
        k = free.pop()                                 # list.pop / append are atomic under the GIL
        ps = parsers[k]
        hdr, _ = ps.begin(data)
        ph, pm, pb, pv, cap = maps[slot]
        nb, _ = ps.decode_mbs_compact(pm, pb, cap, pv)
        counts[slot] = nb
        ctypes.memmove(ph, ctypes.byref(hdr), 64)
        ps.swap(hdr)
        free.append(k)
        return hdr.frame_type

    # whole frame buffers come back into pinned host memory (torch is only the allocator here): ONE asynchronous copy per batch on
    # a stream of its own (vp8hip_frames_download_async), two host sets -- one being filled, one being hashed
    import numpy as np
    import torch
    L = ctx.L
    L.vp8hip_frame_stride.restype = ctypes.c_size_t
    L.vp8hip_frame_stride.argtypes = [c_void_p]
    L.vp8hip_frames_fetch_async.argtypes = [c_void_p, ctypes.c_int, ctypes.c_int, c_void_p, c_void_p]
    L.vp8hip_download_wait.argtypes = [c_void_p]
    stride = L.vp8hip_frame_stride(ctx.h)
    pinned = torch.empty((2, batch, stride), dtype=torch.uint8, pin_memory=True)
    host = pinned.numpy()
    # the digests come with the frames, computed on the device (a frame per lane, vp8_md5.hip), where a row is a whole number of
    # MD5 blocks; the host's cores are the feeder's then
    if device_md5 is None:
        device_md5 = True
    pinned_dig = torch.zeros((2, batch, 16), dtype=torch.uint8, pin_memory=True)
    dig = pinned_dig.numpy()

    def md5_frame(k, i):
        return P.frame_md5(host[k, i], ctx.g, w, h)

    pool = ThreadPoolExecutor(threads)
    hpool = ThreadPoolExecutor(max(4, threads // 2))    # hashing has its own workers: it must not queue behind the feeder
    bad = 0
    t_parse = t_gpu = t_out = 0.0

    def submit_parse(b):
        base = (b % 3) * batch
        n = min(batch, nframes - b * batch)
        return [pool.submit(parse, base + i, frames[(b * batch + i) % nsrc]) for i in range(n)]

    def launch(b, n):
        base = (b % 3) * batch
        for i in range(n):
            nb = counts[base + i]
            ctx._chk(L.vp8hip_ir_upload_compact(ctx.h, base + i, nb), "vp8hip_ir_upload_compact")
            h2d_bytes[0] += nb * 32 + ctx.nmb * 128
        ctx.decode([(base + i, base + i, None) for i in range(n)], P.STAGE_ALL)

    hashing = None                                     # (batch, futures) whose digests are still being computed

    def collect():
        nonlocal bad, hashing
        if hashing is not None:
            b, futs = hashing
            for i, f in enumerate(futs):
                if f.result() != gold[(b * batch + i) % nsrc]:
                    bad += 1
            hashing = None

    def arrived(b, n):
        """Batch b's copy has landed in host set b & 1: hash it (after the digests of batch b-1... of the set's previous user are in)."""
        nonlocal hashing, bad
        ctx._chk(L.vp8hip_download_wait(ctx.h), "vp8hip_download_wait")
        collect()
        if device_md5:
            for i in range(n):
                if dig[b & 1, i].tobytes().hex() != gold[(b * batch + i) % nsrc]:
                    bad += 1
        else:
            hashing = (b, [hpool.submit(md5_frame, b & 1, i) for i in range(n)])

    t0 = time.perf_counter()
    pending = submit_parse(0)
    prev = None
    for b in range(nbatch):
        ta = time.perf_counter()
        for f in pending:
            assert f.result() == 0, "end-to-end probe wants key frames"
        n = len(pending)
        if b + 1 < nbatch:
            # the feeder goes on with the next batch at once (slot set (b+1)%3 was last used by batch b-2, which came back an
            # iteration ago) while this thread uploads and launches batch b and collects batch b-1
            pending = submit_parse(b + 1)
        tb = time.perf_counter()
        launch(b, n)
        tc = time.perf_counter()
        if prev is not None:
            arrived(*prev)                             # (its copy ran beside this batch's uploads)
        ctx._chk(L.vp8hip_frames_fetch_async(ctx.h, (b % 3) * batch, n, host[b & 1].ctypes.data,
                                             dig[b & 1].ctypes.data if device_md5 else None), "vp8hip_frames_fetch_async")
        prev = (b, n)
        td = time.perf_counter()
        t_parse += tb - ta; t_gpu += tc - tb; t_out += td - tc
    arrived(*prev)
    collect()
    elapsed = time.perf_counter() - t0
    pool.shutdown()
    hpool.shutdown()
    for ps in parsers:
        ps.close()
    ctx_nmb = ctx.nmb
    ctx.close()
    return {"workload": f"{fixture}.ivf looped to {nframes} key frames, compressed input in host memory -> per-frame MD5 "
                        f"(entropy decode on {threads} host threads, H2D of the IR, pixel path, D2H of the frames, MD5 "
                        f"{'on the device, a frame per lane' if device_md5 else 'on the host'})",
            "md5_on": "device" if device_md5 else "host",
            "Mpix_s": round(nframes * w * h / elapsed / 1e6, 1), "frames_per_s": round(nframes / elapsed, 1),
            "host_threads": threads, "frames": nframes, "frames_per_launch": batch, "md5_mismatches": bad,
            "h2d_bytes_per_pixel": round(h2d_bytes[0] / (nframes * w * h), 3),
            "h2d_bytes_per_pixel_dense_ir": round(ctx_nmb * 864 / (w * h), 3),
            "main_thread_s": {"waiting_for_feeder": round(t_parse, 3), "upload_and_launch": round(t_gpu, 3),
                              "waiting_for_the_previous_batch_to_arrive": round(t_out, 3)}}


if __name__ == "__main__":
    from vp8_testlib import load_package
    a = [int(x) for x in sys.argv[1:]]
    out = run(load_package(), 0, nframes=a[0] if a else 1024, batch=a[1] if len(a) > 1 else 128,
              threads=a[2] if len(a) > 2 else None, device_md5=bool(a[3]) if len(a) > 3 else None)
    import json
    print(json.dumps(out))
