#!/usr/bin/env python3
"""End-to-end rate of an all-key-frame stream: compressed bytes in host memory -> MD5 of every decoded frame.

SURVEY.md 8(d) asks for this figure beside the kernel-only one, 8(f)1 for the feeder that makes it possible: key
frames are independently decodable, so T host threads run the entropy decoder (the C feeder releases the GIL) on
different frames, each writing the IR straight into a slot's pinned staging; the main thread uploads a batch,
launches the pixel path, and while the next batch is being parsed downloads the previous one, whose frames the same
pool hashes.  Everything the kernel-only number leaves out is in here: entropy decode, H2D of the dense IR
(3.3 B/px), D2H of the visible planes (1.5 B/px), MD5.

    python tools/e2e.py [frames] [batch] [threads]
"""
import hashlib
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(P, device=0, fixture="kf_1920x1080", nframes=1024, batch=128, threads=None):
    from vp8_testlib import ivf_path, golden_md5
    w, h, frames = P.read_ivf(ivf_path(fixture))
    gold = golden_md5(fixture)
    nsrc = len(frames)
    threads = threads or max(1, min(64, (os.cpu_count() or 2) - 1))
    nbatch = (nframes + batch - 1) // batch
    ctx = P.Vp8Hip(device)
    # three sets of slots and frame buffers: batch k+1 is parsed while k is decoded and k-1 is downloaded and hashed
    ctx.configure(w, h, 3 * batch, 3 * batch)
    maps = [ctx.ir_map(s) for s in range(3 * batch)]   # pinned staging, created by the main thread
    parsers = [P.Parser() for _ in range(threads)]
    free = list(range(threads))
    import ctypes

    def parse(slot, data):
        k = free.pop()                                 # list.pop / append are atomic under the GIL
        ps = parsers[k]
        hdr, _ = ps.begin(data)
        ph, pm, pc, pv = maps[slot]
        ps.decode_mbs(pm, pc, pv)
        ctypes.memmove(ph, ctypes.byref(hdr), 64)
        ps.swap(hdr)
        free.append(k)
        return hdr.frame_type

    # whole frame buffers come back into pinned host memory (torch is only the allocator here): one contiguous copy
    # per frame at PCIe speed instead of three strided ones into pageable memory
    import numpy as np
    import torch
    fsz = ctx.g.frame_size
    pinned = torch.empty((batch, fsz), dtype=torch.uint8, pin_memory=True)
    host = pinned.numpy()

    def md5_frame(i):
        return P.frame_md5(host[i], ctx.g, w, h)

    pool = ThreadPoolExecutor(threads)
    hpool = ThreadPoolExecutor(max(4, threads // 2))    # hashing has its own workers: it must not queue behind the feeder
    bad = 0
    t_parse = t_gpu = t_out = t_first = t_d2h = 0.0

    def submit_parse(b):
        base = (b % 3) * batch
        n = min(batch, nframes - b * batch)
        return [pool.submit(parse, base + i, frames[(b * batch + i) % nsrc]) for i in range(n)]

    def launch(b, n):
        base = (b % 3) * batch
        for i in range(n):
            ctx.upload(base + i)
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

    def download(b, n):
        """D2H of batch b into the pinned buffer (after its digests of the previous round are in), hashing started."""
        nonlocal hashing, t_first, t_d2h
        collect()
        base = (b % 3) * batch
        t1 = time.perf_counter()
        for i in range(n):                                                # synchronous D2H, in order
            ctx._chk(ctx.L.vp8hip_frame_download(ctx.h, base + i, 1, host[i].ctypes.data, None, None, 0, 0), "download")
            if i == 0:
                t2 = time.perf_counter()         # the first download waits for the batch's kernels
        t3 = time.perf_counter()
        t_first += t2 - t1; t_d2h += t3 - t2
        hashing = (b, [hpool.submit(md5_frame, i) for i in range(n)])

    t0 = time.perf_counter()
    pending = submit_parse(0)
    prev = None
    for b in range(nbatch):
        ta = time.perf_counter()
        for f in pending:
            assert f.result() == 0, "end-to-end probe wants key frames"
        n = len(pending)
        tb = time.perf_counter()
        # the previous batch comes back BEFORE this one's uploads are queued on the (in-order) stream: the download
        # then only waits for kernels that had a whole iteration to finish, and this batch's H2D overlaps the feeder
        if prev is not None:
            download(*prev)
        tc = time.perf_counter()
        launch(b, n)
        if b + 1 < nbatch:
            pending = submit_parse(b + 1)              # set (b+1)%3: last used by batch b-2, downloaded already
        prev = (b, n)
        td = time.perf_counter()
        t_parse += tb - ta; t_out += tc - tb; t_gpu += td - tc
    download(*prev)
    collect()
    elapsed = time.perf_counter() - t0
    pool.shutdown()
    hpool.shutdown()
    for ps in parsers:
        ps.close()
    ctx.close()
    return {"workload": f"{fixture}.ivf looped to {nframes} key frames, compressed input in host memory -> per-frame MD5 "
                        f"(entropy decode on {threads} host threads, H2D of the IR, pixel path, D2H, MD5)",
            "Mpix_s": round(nframes * w * h / elapsed / 1e6, 1), "frames_per_s": round(nframes / elapsed, 1),
            "host_threads": threads, "frames": nframes, "frames_per_launch": batch, "md5_mismatches": bad,
            "main_thread_s": {"waiting_for_feeder": round(t_parse, 3), "upload_and_launch": round(t_gpu, 3),
                              "download": round(t_out, 3),
                              "of_which_waiting_for_kernels": round(t_first, 3), "of_which_d2h": round(t_d2h, 3)}}


if __name__ == "__main__":
    from vp8_testlib import load_package
    a = [int(x) for x in sys.argv[1:]]
    out = run(load_package(), 0, nframes=a[0] if a else 1024, batch=a[1] if len(a) > 1 else 128,
              threads=a[2] if len(a) > 2 else None)
    import json
    print(json.dumps(out))
