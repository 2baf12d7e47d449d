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
names = ["loop overhead", "row start, late phase 0, descriptor request", "step setup (gates, line above, read-back)", "fetch residuals + queue next phase (+ prepare next MB)",
         "prediction + add", "loop filter (+ row stores, chroma)", "drain next phase (luma) / stores+rotate+drain (chroma)", "bottom rows, context, end of step", "fetch residuals", "prepare next MB (luma)", "row stores + rotate (luma)", "chroma: lf_block_row", "chroma: row stores", "", "drain: waiting for the coefficients (vmcnt)"]
for which, kn in ((0, "luma kernel"), (1, "chroma kernel")):
    L.vp8hip_debug_stamps(ctx.h, which, buf)
    tot = sum(buf)
    print(f"{kn} fused={st.fused}: {tot} cycles in wave 0 ({st.recon_ms:.2f} ms for both kernels)")
    for i, v in enumerate(buf):
        if v: print(f"   [{i:2d}] {100.0 * v / tot:5.1f} %  {v:12d} cyc  {names[i] if i < len(names) else ''}")
ctx.close()
