"""Dev aid (GPU, a library built with -DVP8_STAMPS: tools/variant.sh stamps -DVP8_STAMPS -- python3 tools/kf_diag.py ...): what the
waves of vp8_keyframe_kernel / vp8_interframe_kernel do.

    kf_diag.py waves  key|inter [frames]     how long the luma and the chroma waves run (cycles: min / median / max)
    kf_diag.py stamps key|inter [frames]     shares of a step by phase, wave 0 of either role
    kf_diag.py sched  key [frames]           where the waves ran and in which role (the pairing per SIMD)
"""
import collections
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from vp8_testlib import load_package, ivf_path  # noqa: E402

P = load_package()
what = sys.argv[1] if len(sys.argv) > 1 else "waves"
kind = sys.argv[2] if len(sys.argv) > 2 else "key"
n = int(sys.argv[3]) if len(sys.argv) > 3 else (8192 if kind == "key" else 4096)


def setup():
    """The bench's launch: n key frames (kf_1920x1080 looped), or n copies of an inter frame of p_dense_1920x1080, each job with
    its own IR slot, reference and destination."""
    ctx = P.Vp8Hip(0)
    parser = P.Parser()
    jobs = (P.Job * n)()
    if kind == "key":
        w, h, frames = P.read_ivf(ivf_path("kf_1920x1080"))
        ctx.configure(w, h, n, n)
        for i, data in enumerate(frames):
            hdr, _ = ctx.parse_into_slot_compact(parser, data, i)
            parser.swap(hdr)
        for i in range(len(frames), n):
            ctx.ir_copy(i, i % len(frames))
        for i in range(n):
            jobs[i].ir_slot, jobs[i].dst_fb = i, i
    else:
        name, k = "p_dense_1920x1080", 2
        w, h, frames = P.read_ivf(ivf_path(name))
        ctx.configure(w, h, 4 + 2 * n, 2 + n)
        for data in frames[:k]:
            ctx.sync()
            hdr, _ = ctx.parse_into_slot_compact(parser, data, 0)
            r = parser.refs
            ctx.decode([(0, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx) if hdr.frame_type else None)], P.STAGE_ALL)
            ctx.sync()
            parser.swap(hdr)
        hdr, _ = ctx.parse_into_slot_compact(parser, frames[k], 1)
        r = parser.refs
        for i in range(n):
            ctx.ir_copy(2 + i, 1)
            ctx.L.vp8hip_frame_copy(ctx.h, 4 + 2 * i, r.lst_idx)
            jobs[i].ir_slot, jobs[i].dst_fb = 2 + i, 5 + 2 * i
            jobs[i].ref_fb[1], jobs[i].ref_fb[2], jobs[i].ref_fb[3] = 4 + 2 * i, r.gld_idx, r.alt_idx
    for _ in range(2):
        ctx.decode_array(jobs, n, 7)
    ctx.sync()
    return ctx, jobs


NW = 16 + 16384 + 4 * 4096
ctx, jobs = setup()
L = ctx.L
if not hasattr(L, "vp8hip_debug_sched"):
    raise SystemExit("this library was not built with -DVP8_STAMPS (tools/variant.sh stamps -DVP8_STAMPS -- ...)")
L.vp8hip_debug_sched.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
L.vp8hip_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]


def sched_log(st):
    buf = (ctypes.c_uint * NW)()
    L.vp8hip_debug_sched(ctx.h, buf, NW)
    nw = st.workgroups * 2
    return buf, [(buf[16 + 16384 + 4 * b], buf[16 + 16384 + 4 * b + 1], buf[16 + 16384 + 4 * b + 2] & 0xff,
                  buf[16 + 16384 + 4 * b + 2] >> 8, buf[16 + 16384 + 4 * b + 3]) for b in range(nw)]


if what == "waves":
    for rep in range(2):
        ctx.decode_array(jobs, n, 7); ctx.sync(); st = ctx.stats()
        _, log = sched_log(st)
        role = np.array([e[2] for e in log])
        dur = np.array([(e[4] >> 10) * 1024 for e in log], dtype=np.float64)
        for rl, nm in ((0, "luma"), (1, "chroma")):
            d = dur[role == rl] / 1e6
            print(f"{kind} launch {rep}: {nm:6s} waves {len(d)}: Mcycles min {d.min():.1f} median {np.median(d):.1f} max {d.max():.1f}   "
                  f"(recon interval {st.recon_ms:.2f} ms)")
elif what == "stamps":
    buf = (ctypes.c_ulonglong * 16)()
    L.vp8hip_debug_stamps(ctx.h, 0, buf); L.vp8hip_debug_stamps(ctx.h, 1, buf)      # (clear what the warm-up left)
    ctx.decode_array(jobs, n, 7); ctx.sync(); st = ctx.stats()
    names = ["loop overhead", "row start, late phase 0, record request", "step setup (gates, line above, read-back)",
             "fetch residuals + queue next phase (+ prepare next MB)", "prediction + add", "loop filter (+ row stores, chroma)",
             "drain next phase (luma) / stores+rotate+drain (chroma)", "bottom rows, context, end of step", "fetch residuals",
             "prepare next MB (luma)", "row stores + rotate (luma)", "chroma: lf_block_row", "chroma: row stores", "",
             "drain: waiting for the coefficients (vmcnt)"]
    for which, kn in ((0, "luma role"), (1, "chroma role")):
        L.vp8hip_debug_stamps(ctx.h, which, buf)
        tot = sum(buf)
        print(f"{kn}: {tot} cycles in wave 0 ({st.recon_ms:.2f} ms recon interval)")
        for i, v in enumerate(buf):
            if v:
                print(f"   [{i:2d}] {100.0 * v / tot:5.1f} %  {v:12d} cyc  {names[i] if i < len(names) else ''}")
elif what == "sched":
    for rep in range(4):
        ctx.decode_array(jobs, n, 7); ctx.sync(); st = ctx.stats()
        buf, log = sched_log(st)
        by_simd = collections.defaultdict(list)
        for hw, xcc, role, seen, item in log:
            by_simd[((hw >> 4) & 3) | (((hw >> 8) & 0xff) << 2) | ((xcc & 15) << 10)].append(role)
        combos = collections.Counter(tuple(sorted(v)) for v in by_simd.values())
        print(f"launch {rep}: kernel {st.recon_ms:.2f} ms, waves {len(log)}, distinct SIMDs {len(by_simd)}, role combinations per SIMD: {dict(combos)}; "
              f"work counters {buf[0]} {buf[1]}")
ctx.close()
