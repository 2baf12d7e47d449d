"""Dev aid (GPU): launches of n copies of an inter frame of a WRITTEN 1080p stream in which about 80 % of the macroblocks stand still
or move by whole pixels (tests/test_gpu_whole_pixel.py's generator): what the prediction kernels' copy lists buy.  Run under
rocprofv3 for the per-kernel counters (tools/profile_round6.sh ... whole).
   python3 tools/whole_pixel_time.py [jobs] [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from vp8_testlib import load_package
from test_gpu_whole_pixel import whole_pixel_sequence
P = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w, h = 1920, 1080
t0 = time.time()
frames = whole_pixel_sequence(w, h, 21, keep_split=False)
print(f"stream written in {time.time() - t0:.1f} s: {[len(f) for f in frames]} bytes")
ctx = P.Vp8Hip(0)
ctx.configure(w, h, 4 + 2 * n, 2 + n)
parser = P.Parser()
for data in frames[:2]:
    hdr = ctx.parse_into_slot(parser, data, 0); ctx.upload(0)
    r = parser.refs
    ctx.decode([(0, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx) if hdr.frame_type else None)], P.STAGE_ALL); ctx.sync()
    parser.swap(hdr)
hdr = ctx.parse_into_slot(parser, frames[2], 1); ctx.upload(1)
r = parser.refs
fwd, back = (P.Job * n)(), (P.Job * n)()
for i in range(n):
    ctx.ir_copy(2 + i, 1)
    ctx.L.vp8hip_frame_copy(ctx.h, 4 + 2 * i, r.lst_idx)
    fwd[i].ir_slot, fwd[i].dst_fb = 2 + i, 5 + 2 * i
    fwd[i].ref_fb[1], fwd[i].ref_fb[2], fwd[i].ref_fb[3] = 4 + 2 * i, r.gld_idx, r.alt_idx
    back[i].ir_slot, back[i].dst_fb = 2 + i, 4 + 2 * i
    back[i].ref_fb[1], back[i].ref_fb[2], back[i].ref_fb[3] = 5 + 2 * i, r.gld_idx, r.alt_idx
ctx.decode_array(fwd, n, P.STAGE_ALL); ctx.sync()
t = time.perf_counter()
for _ in range(reps): ctx.decode_array(fwd, n, P.STAGE_ALL)
ctx.sync()
print(f"{n} jobs, references in raster form: {(time.perf_counter() - t) / reps * 1e3:.2f} ms per launch (pred_tiles {ctx.stats().pred_tiles})")
ctx.decode_array(back, n, P.STAGE_ALL); ctx.decode_array(fwd, n, P.STAGE_ALL); ctx.sync()
t = time.perf_counter()
for _ in range(reps):
    ctx.decode_array(back, n, P.STAGE_ALL); ctx.decode_array(fwd, n, P.STAGE_ALL)
ctx.sync()
print(f"{n} jobs, chained (tiles): {(time.perf_counter() - t) / (2 * reps) * 1e3:.2f} ms per launch (pred_tiles {ctx.stats().pred_tiles})")
ctx.close()
