"""Dev aid (GPU, diagnostic library built with -DVP8_STAMPS): shares of a step of the fused key-frame kernel by phase.
   VP8HIP_LIB_OVERRIDE is not a thing: copy lib/var/libvp8hip_stamps.so over lib/libvp8hip.so on the GPU box first."""
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
L.vp8hip_debug_stamps(ctx.h, 0, buf)
ctx.decode_array(jobs, n, 7); ctx.sync(); st = ctx.stats()
names = ["loop overhead", "top: prefetches, prepare next MB, queue its phase 0", "chroma (rest: swaps, setup)", "this-step setup (DPP, row start, gates)",
         "drain phase 0", "luma loop rest (queue next phase)", "drain phases 1, 2", "hand-over to next iteration", "chroma recon (both planes)",
         "chroma loop filter + stores", "luma prediction + add (4 block rows)", "luma loop filter (lf_block_row x4)", "luma stores + rotate", "luma bottom rows / row end"]
L.vp8hip_debug_stamps(ctx.h, 0, buf)
tot = sum(buf)
print(f"fused={st.fused}: {tot} cycles in wave 0 ({st.recon_ms:.2f} ms kernel, lf {st.lf_ms:.2f})")
for i, v in enumerate(buf):
    if v: print(f"   [{i:2d}] {100.0 * v / tot:5.1f} %  {v:12d} cyc  {names[i] if i < len(names) else ''}")
ctx.close()
