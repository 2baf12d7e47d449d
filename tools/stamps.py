"""Dev aid (GPU, diagnostic library built with -DVP8_STAMPS): shares of a lane-per-row kernel step by phase.
   cp libvpx.opencl_amd/lib/var/libvp8hip_stamps.so libvpx.opencl_amd/lib/libvp8hip.so; VP8HIP_SIMT_LGG=3 python3 tools/stamps.py [frames=1024]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from vp8_testlib import load_package, ivf_path
P = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w, h, frames = P.read_ivf(ivf_path("kf_1920x1080"))
ctx = P.Vp8Hip(0); ctx.configure(w, h, n, n)
parser = P.Parser()
for i, data in enumerate(frames):
    hdr = ctx.parse_into_slot(parser, data, i); parser.swap(hdr); ctx.upload(i)
for i in range(len(frames), n): ctx.ir_copy(i, i % len(frames))
jobs = (P.Job * n)()
for i in range(n): jobs[i].ir_slot, jobs[i].dst_fb = i, i
L = ctx.L
L.vp8hip_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 16)()
ctx.decode_array(jobs, n, 7); ctx.sync()
L.vp8hip_debug_stamps(ctx.h, 0, buf); L.vp8hip_debug_stamps(ctx.h, 1, buf)
ctx.decode_array(jobs, n, 7); ctx.sync(); st = ctx.stats()
names = {0: ["loop overhead", "prefetches, next macroblock's tables, queue of its luma blocks 0-7", "chroma prediction + add of the previous macroblock, history",
             "DPP fetch, row starts, line above", "transform rounds of luma blocks 0-7", "queue of the next phase + luma prediction + add (2 x 2 block rows)",
             "transform rounds of luma blocks 8-15 and of chroma", "hand-over to the next iteration", ""],
         1: ["loop overhead / idle step", "DPP fetch + history", "descriptor, limits, loads issued, read-back", "luma staging into LDS (waits for the loads)", "luma filter (both passes)",
             "luma outputs (stores / holds)", "chroma staging", "chroma filter", "chroma outputs"]}
for which, kn in ((0, "recon"), (1, "loop filter")):
    L.vp8hip_debug_stamps(ctx.h, which, buf)
    tot = sum(buf)
    print(f"{kn}: {tot} cycles in wave 0 ({(st.recon_ms, st.lf_ms)[which]:.2f} ms kernel)")
    for i, v in enumerate(buf):
        if v: print(f"   [{i}] {100.0 * v / tot:5.1f} %  {v / 1094:9.0f} cyc/step  {names[which][i] if i < len(names[which]) else ''}")
ctx.close()
