"""Dev aid (GPU), an experiment of round 6 that did NOT pay: decode_to_md5's gate as a pipeline -- batch k hashed on the fetch stream BESIDE the
decode of batch k+1 into a second set of frame buffers, the decode launched with three quarters of its waves so that the hash kernel's
waves have registers to live in.  16,384 frames a batch: 166.5 ms per batch against 114.7 with the two one after the other
(gpurun_out/r6b): the hash is a chain of dependent loads, and beside a kernel that keeps the memory system busy every link of it
takes longer than the decode saves.
   python3 tools/hash_pipe_time.py [frames]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import load_stream
from vp8_testlib import load_package


def hash_pipeline_probe(P, device, fixture, W, H, F, steps=4):
    """decode_to_md5's gate as a PIPELINE (round 6): a batch's MD5s are a serial chain per frame -- 37 ms whatever the batch, 80 ms for
    16,384 frames read from their tiles -- on 256 of the chip's 1024 SIMDs, so batch k is hashed (vp8hip_frames_fetch_async: a stream
    of its own) BESIDE the decode of batch k+1 into a second set of frame buffers; the decode is launched with three quarters of its
    waves (VP8HIP_SIMT_WAVES), which leaves the hash kernel's waves registers to live in.  Reported: the steady-state time per batch,
    every digest of every timed batch compared with the reference listing.  What it stands for: md5_utils.c:70-141, 167-245 behind
    vp8dx_get_raw_frame, as vpxdec --md5 runs them frame after frame."""
    import numpy as np
    from vp8_testlib import golden_md5
    gold = golden_md5(fixture)
    keep = {k: os.environ.get(k) for k in ("VP8HIP_SIMT_WAVES", "VP8HIP_MD5_PACK_FROM")}
    os.environ["VP8HIP_SIMT_WAVES"] = "768"
    os.environ["VP8HIP_MD5_PACK_FROM"] = "100000000"          # (from the tiles: the packed copy -- 51 GB -- does not fit beside two sets)
    ctx = P.Vp8Hip(device)
    try:
        ctx.configure(W, H, 2 * F, F)
        nsrc, _ = load_stream(P, ctx, fixture, F, 0)
        L = ctx.L
        L.vp8hip_frames_fetch_async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.vp8hip_download_wait.argtypes = [ctypes.c_void_p]
        jobs = [(P.Job * F)(), (P.Job * F)()]
        for b in range(2):
            for i in range(F):
                jobs[b][i].ir_slot, jobs[b][i].dst_fb = i, b * F + i
                for k in range(4):
                    jobs[b][i].ref_fb[k] = -1
        dig = [np.zeros(16 * F, np.uint8), np.zeros(16 * F, np.uint8)]
        bad = 0

        def check(b):
            return sum(1 for i in range(F) if dig[b][16 * i:16 * i + 16].tobytes().hex() != gold[i % nsrc])

        def batch(k, first):
            nonlocal bad
            b = k & 1
            ctx.decode_array(jobs[b], F, P.STAGE_ALL)
            if not first:                       # the batch before: its hash ran beside this decode's predecessor
                ctx._chk(L.vp8hip_download_wait(ctx.h), "wait")
                bad += check(b ^ 1)
            ctx._chk(L.vp8hip_frames_fetch_async(ctx.h, b * F, F, None, dig[b].ctypes.data), "fetch")
        batch(0, True); batch(1, False)         # (warm-up: the tiled forms are allocated, the pipeline is full)
        ctx.sync()
        t0 = time.perf_counter()
        for k in range(2, 2 + steps):
            batch(k, False)
        ctx._chk(L.vp8hip_download_wait(ctx.h), "wait")
        ctx.sync()
        dt = (time.perf_counter() - t0) / steps
        bad += check((1 + steps) & 1)
        mem = sum(ctx.memory_usage().values())
        return {"ms_per_step": round(dt * 1e3, 3), "Mpix_s": round(F * W * H / dt / 1e6, 1), "md5_mismatches": bad,
                "digests_checked": F * (steps + 1), "frames_per_batch": F, "device_GB": round(mem / 1e9, 2),
                "what": f"steady state of: decode batch k+1 ({F} frames, 768 of 1024 luma / chroma wave pairs) into one of two frame-buffer sets "
                        f"WHILE batch k's {F} frames are hashed from their tiles on a stream of their own; every digest of every batch checked"}
    finally:
        ctx.close()
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v




F = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
print(hash_pipeline_probe(load_package(), 0, "kf_1920x1080", 1920, 1080, F))
