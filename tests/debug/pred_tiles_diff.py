"""Dev aid (GPU): which macroblocks differ between the tile-reading predictor and the oracle on test_gpu_pred_tiles' random IR.
   python3 tests/debug/pred_tiles_diff.py W H VERSION FTYPE SEED"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["VP8HIP_RECON"] = "simt"
from vp8_testlib import load_package, oracle_decode, synth_ir
P = load_package()
w, h, version, ftype, seed = (int(v) for v in sys.argv[1:6])
ctx = P.Vp8Hip(0); ctx.configure(w, h, 5, 4); g = ctx.g
refs = []
for k in range(3):
    hdr, mbs, coef, mvs = synth_ir(w, h, 500 + 11 * seed + k, inter=False, dense=0.4)
    o = np.zeros(g.frame_size, np.uint8); oracle_decode(hdr, mbs, coef, mvs, o, (None, None, None), 7)
    ctx.fill_slot(1 + k, hdr, mbs, coef, mvs); refs.append(o)
ctx.decode([(1 + k, 1 + k, None) for k in range(3)], 7)
hdr, mbs, coef, mvs = synth_ir(w, h, seed * 5 + w + 7 * version, inter=True, version=version, filter_type=ftype, dense=(0.2, 0.6, 0.35)[seed], big=seed == 1)
stage = int(sys.argv[6]) if len(sys.argv) > 6 else 1          # 1: reconstruction only (no loop filter: differences stay in their macroblock)
o = np.zeros(g.frame_size, np.uint8); oracle_decode(hdr, mbs, coef, mvs, o, tuple(refs), stage)
ctx.fill_slot(0, hdr, mbs, coef, mvs)
ctx._chk(ctx.L.vp8hip_set_pred_tiles(ctx.h, 1), "x")
ctx.decode([(0, 0, (1, 2, 3))], stage)
print("stats fused/pred_tiles", ctx.stats().fused, ctx.stats().pred_tiles)
got = ctx.download_full(0)
cols = (w + 15) // 16
for name, off, stride, W, H, sz in (("Y", g.y_off, g.y_stride, g.aligned_w, g.aligned_h, 16), ("U", g.u_off, g.uv_stride, g.aligned_w // 2, g.aligned_h // 2, 8),
                                    ("V", g.v_off, g.uv_stride, g.aligned_w // 2, g.aligned_h // 2, 8)):
    pa = np.lib.stride_tricks.as_strided(got[off:], shape=(H, W), strides=(stride, 1))
    pb = np.lib.stride_tricks.as_strided(o[off:], shape=(H, W), strides=(stride, 1))
    d = pa != pb
    ys, xs = np.nonzero(d)
    bad = sorted(set((int(y) // sz, int(x) // sz) for y, x in zip(ys, xs)))
    print(name, "differing macroblocks:", len(bad))
    for r, c in bad[:12]:
        i = r * cols + c
        mv = mvs[i, 0]
        dd = d[r * sz:(r + 1) * sz, c * sz:(c + 1) * sz]
        print(f"  mb ({r},{c}) idx {i} unit {i // 64} lane {i % 64}: y_mode {mbs[i,0]} ref {mbs[i,2]} flags {mbs[i,3]} mv(row,col) {tuple(int(v) for v in mv)} "
              f"-> x0 {c*sz + (int(mv[1])>>3)} y0 {r*sz + (int(mv[0])>>3)}; wrong rows {sorted(set(np.nonzero(dd)[0].tolist()))} cols {sorted(set(np.nonzero(dd)[1].tolist()))}")
ctx.close()
