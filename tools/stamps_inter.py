"""Dev aid (GPU, diagnostic library built with -DVP8_STAMPS copied over lib/libvp8hip.so): shares of a step of vp8_interframe_kernel
   by phase, on the bench's inter-frame probe.  usage: stamps_inter.py [jobs]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from vp8_testlib import load_package, ivf_path
P = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
name, k = "p_dense_1920x1080", 2
w, h, frames = P.read_ivf(ivf_path(name))
ctx = P.Vp8Hip(0)
ctx.configure(w, h, 4 + 2 * n, 2 + n)
parser = P.Parser()
for data in frames[:k]:
    hdr = ctx.parse_into_slot(parser, data, 0); ctx.upload(0); r = parser.refs
    ctx.decode([(0, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx) if hdr.frame_type else None)], P.STAGE_ALL); ctx.sync(); parser.swap(hdr)
hdr = ctx.parse_into_slot(parser, frames[k], 1); ctx.upload(1); r = parser.refs
jobs = (P.Job * n)()
for i in range(n):
    ctx.ir_copy(2 + i, 1)
    ctx.L.vp8hip_frame_copy(ctx.h, 4 + 2 * i, r.lst_idx)
    jobs[i].ir_slot, jobs[i].dst_fb = 2 + i, 5 + 2 * i
    jobs[i].ref_fb[1], jobs[i].ref_fb[2], jobs[i].ref_fb[3] = 4 + 2 * i, r.gld_idx, r.alt_idx
L = ctx.L
L.vp8hip_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 16)()
for _ in range(3): ctx.decode_array(jobs, n, 7)
ctx.sync()
L.vp8hip_debug_stamps(ctx.h, 0, buf)
ctx.decode_array(jobs, n, 7); ctx.sync(); st = ctx.stats()
names = ["loop overhead", "row start, late phase 0, descriptor request", "step setup (gates, line above, read-back)", "fetch residuals + queue next phase (+ prepare next MB)",
         "prediction + add", "loop filter (+ row stores, chroma)", "drain next phase (luma) / stores+rotate+drain (chroma)", "bottom rows, context, end of step", "fetch residuals", "prepare next MB (luma)", "row stores + rotate (luma)", "chroma: lf_block_row", "chroma: row stores", "", "drain: waiting for the coefficients (vmcnt)"]
for which, kn in ((0, "luma role"), (1, "chroma role")):
    L.vp8hip_debug_stamps(ctx.h, which, buf)
    tot = sum(buf)
    print(f"{kn} fused={st.fused}: {tot} cycles in wave 0 ({st.recon_ms:.2f} ms recon interval)")
    for i, v in enumerate(buf):
        if v: print(f"   [{i:2d}] {100.0 * v / tot:5.1f} %  {v:12d} cyc  {names[i] if i < len(names) else ''}")
ctx.close()
