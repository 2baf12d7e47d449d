"""Dev aid (GPU): launches of n real 1080p P frames CHAINED -- every launch predicts from the frames the launch before wrote, as n
streams in lock step do --, with the references read as tiles (vp8hip_set_pred_tiles 1) and through their raster form (0).
   [VP8HIP_LIB=...] python3 tools/inter_chain_time.py [jobs] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from vp8_testlib import load_package, ivf_path, golden_md5
P = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
name, k = "p_dense_1920x1080", 2
w, h, frames = P.read_ivf(ivf_path(name))
gold = golden_md5(name)
ctx = P.Vp8Hip(0)
ctx.configure(w, h, 4 + 2 * n, 2 + n)
parser = P.Parser()
for data in frames[:k]:
    hdr = ctx.parse_into_slot(parser, data, 0); ctx.upload(0)
    r = parser.refs
    ctx.decode([(0, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx) if hdr.frame_type else None)], P.STAGE_ALL); ctx.sync()
    parser.swap(hdr)
hdr = ctx.parse_into_slot(parser, frames[k], 1); ctx.upload(1)
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
ok = ctx.frames_md5(5 + 2 * (n // 2), 1)[0] == gold[k]
t = time.perf_counter()
for _ in range(reps): ctx.decode_array(fwd, n, P.STAGE_ALL)
ctx.sync()
print(f"{os.environ.get('VP8HIP_LIB', 'product')}: {n} jobs: references in raster form: {(time.perf_counter() - t) / reps * 1e3:.2f} ms per launch (md5 {'ok' if ok else 'DIFFERS'})")
for mode in (1, 0, 1):
    ctx._chk(ctx.L.vp8hip_set_pred_tiles(ctx.h, mode), "set_pred_tiles")
    ctx.decode_array(back, n, P.STAGE_ALL); ctx.decode_array(fwd, n, P.STAGE_ALL); ctx.sync()
    t = time.perf_counter()
    for _ in range(reps):
        ctx.decode_array(back, n, P.STAGE_ALL); ctx.decode_array(fwd, n, P.STAGE_ALL)
    ctx.sync()
    dt = (time.perf_counter() - t) / (2 * reps) * 1e3
    print(f"   chained, pred_tiles={mode} (stats.pred_tiles {ctx.stats().pred_tiles}): {dt:.2f} ms per launch")
ctx.close()
