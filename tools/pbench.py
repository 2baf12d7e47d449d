#!/usr/bin/env python3
"""Dev tool: throughput of the pixel path on REAL inter frames.  Decodes tests/golden/p_1920x1080.ivf up to frame k
the normal way, then launches N jobs that all decode frame k+1 from the same references into different buffers."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from vp8_testlib import load_package, ivf_path
P = load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
name = sys.argv[2] if len(sys.argv) > 2 else "p_1920x1080"
w, h, frames = P.read_ivf(ivf_path(name))
ctx = P.Vp8Hip(0)
ctx.configure(w, h, N + 4, 2)
parser = P.Parser()
for i, data in enumerate(frames[:5]):
    hdr = ctx.parse_into_slot(parser, data, 0)
    ctx.upload(0)
    r = parser.refs
    refs = (r.lst_idx, r.gld_idx, r.alt_idx)
    ctx.decode([(0, r.new_idx, refs if hdr.frame_type else None)], 7)
    ctx.sync()
    parser.swap(hdr)
hdr = ctx.parse_into_slot(parser, frames[5], 1)
ctx.upload(1)
r = parser.refs
refs = (r.lst_idx, r.gld_idx, r.alt_idx)
jobs = (P.Job * N)()
for i in range(N):
    jobs[i].ir_slot, jobs[i].dst_fb = 1, 4 + i
    jobs[i].ref_fb[1], jobs[i].ref_fb[2], jobs[i].ref_fb[3] = refs
ctx.decode_array(jobs, N, 7); ctx.sync()
rr = ll = ee = 0.0
for _ in range(3):
    ctx.decode_array(jobs, N, 7); ctx.sync(); st = ctx.stats(); rr += st.recon_ms / 3; ll += st.lf_ms / 3; ee += st.extend_ms / 3
tot = rr + ll + ee
print(f"{name} frame 5 x{N}: recon {rr:.2f} lf {ll:.2f} extend {ee:.2f} ms -> {N*w*h/tot/1e3:.0f} Mpix/s kernels only "
      f"(frame type {hdr.frame_type}, waves/wg {st.recon_waves})")
ctx.close()
