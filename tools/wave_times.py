"""Dev aid (GPU, diagnostic library built with -DVP8_STAMPS copied over lib/libvp8hip.so): how long the luma and the chroma waves of
   vp8_keyframe_kernel / vp8_interframe_kernel run.  usage: wave_times.py key|inter [frames]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from vp8_testlib import load_package, ivf_path
P = load_package()
kind = sys.argv[1] if len(sys.argv) > 1 else "key"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (8192 if kind == "key" else 4096)
ctx = P.Vp8Hip(0)
parser = P.Parser()
jobs = (P.Job * n)()
if kind == "key":
    w, h, frames = P.read_ivf(ivf_path("kf_1920x1080")); ctx.configure(w, h, n, n)
    for i, data in enumerate(frames):
        hdr = ctx.parse_into_slot(parser, data, i); parser.swap(hdr); ctx.upload(i)
    for i in range(len(frames), n): ctx.ir_copy(i, i % len(frames))
    for i in range(n): jobs[i].ir_slot, jobs[i].dst_fb = i, i
else:
    name, k = "p_dense_1920x1080", 2
    w, h, frames = P.read_ivf(ivf_path(name)); ctx.configure(w, h, 4 + 2 * n, 2 + n)
    for data in frames[:k]:
        hdr = ctx.parse_into_slot(parser, data, 0); ctx.upload(0); r = parser.refs
        ctx.decode([(0, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx) if hdr.frame_type else None)], P.STAGE_ALL); ctx.sync(); parser.swap(hdr)
    hdr = ctx.parse_into_slot(parser, frames[k], 1); ctx.upload(1); r = parser.refs
    for i in range(n):
        ctx.ir_copy(2 + i, 1); ctx.L.vp8hip_frame_copy(ctx.h, 4 + 2 * i, r.lst_idx)
        jobs[i].ir_slot, jobs[i].dst_fb = 2 + i, 5 + 2 * i
        jobs[i].ref_fb[1], jobs[i].ref_fb[2], jobs[i].ref_fb[3] = 4 + 2 * i, r.gld_idx, r.alt_idx
for _ in range(3): ctx.decode_array(jobs, n, 7)
ctx.sync()
for rep in range(2):
    ctx.decode_array(jobs, n, 7); ctx.sync(); st = ctx.stats()
    NW = 16 + 16384 + 4 * 4096
    buf = (ctypes.c_uint * NW)()
    ctx.L.vp8hip_debug_sched.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    ctx.L.vp8hip_debug_sched(ctx.h, buf, NW)
    nw = st.workgroups * 2
    role = np.array([buf[16 + 16384 + 4 * b + 2] & 0xff for b in range(nw)])
    dur = np.array([(buf[16 + 16384 + 4 * b + 3] >> 10) * 1024 for b in range(nw)], dtype=np.float64)
    for rl, nm in ((0, "luma"), (1, "chroma")):
        d = dur[role == rl] / 1e6
        print(f"{kind} launch {rep}: {nm:6s} waves {len(d)}: Mcycles min {d.min():.1f} median {np.median(d):.1f} max {d.max():.1f}   (recon interval {st.recon_ms:.2f} ms)")
ctx.close()
