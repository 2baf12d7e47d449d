"""Dev aid (GPU): one launch of n frames for a profiler pass.  python3 tools/pmc_one.py <stages> <frames> [fixture] [shared|""] [raster]
   shared: all frames decode the fixture's few IR slots (no per-frame copy of the IR: far fewer API calls, which rocprofv3
   survives at 8192 frames; the coefficient reads then hit the caches -- SQ counters only)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from vp8_testlib import load_package, ivf_path
P = load_package()
stage = int(sys.argv[1]); n = int(sys.argv[2])
fx = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] else "kf_1920x1080"
shared = len(sys.argv) > 4 and sys.argv[4] == "shared"
w, h, frames = P.read_ivf(ivf_path(fx))
ns = len(frames)
ctx = P.Vp8Hip(0); ctx.configure(w, h, n, ns if shared else n)
parser = P.Parser()
for i, data in enumerate(frames):
    hdr, _ = ctx.parse_into_slot_compact(parser, data, i); parser.swap(hdr)
if not shared:
    for i in range(ns, n): ctx.ir_copy(i, i % ns)
jobs = (P.Job * n)()
for i in range(n): jobs[i].ir_slot, jobs[i].dst_fb = (i % ns if shared else i), i
ctx.decode_array(jobs, n, stage); ctx.sync()
if len(sys.argv) > 5 and sys.argv[5] == "raster":      # ... and the raster form of every frame (vp8_detile_kf_kernel + vp8_extend_kernel)
    ctx.frames_to_raster(0, n); ctx.sync()
print("done", stage, n)
