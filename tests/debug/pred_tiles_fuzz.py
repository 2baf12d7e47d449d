"""Dev aid (GPU): many random inter frames through the tile-reading predictor (and the raster one) against the oracle, sizes and seeds
beyond what tests/test_gpu_pred_tiles.py runs every time.   python3 tests/debug/pred_tiles_fuzz.py [cases] [first seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["VP8HIP_RECON"] = "simt"
from vp8_testlib import bordered_area_equal, load_package, oracle_decode, synth_ir
P = load_package()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad = 0
for k in range(cases):
    rng = np.random.default_rng(seed0 + k)
    w = int(rng.integers(2, 24)) * 16 - int(rng.integers(0, 16)); h = int(rng.integers(1, 14)) * 16 - int(rng.integers(0, 16))
    w, h = max(w, 17), max(h, 2)
    version = int(rng.integers(0, 4)); ftype = int(rng.integers(0, 2))
    ctx = P.Vp8Hip(0); ctx.configure(w, h, 6, 4); g = ctx.g
    refs = []
    for j in range(3):
        hdr, mbs, coef, mvs = synth_ir(w, h, 9000 + 7 * (seed0 + k) + j, inter=False, dense=0.4)
        o = np.zeros(g.frame_size, np.uint8); oracle_decode(hdr, mbs, coef, mvs, o, (None, None, None), 7)
        ctx.fill_slot(1 + j, hdr, mbs, coef, mvs); refs.append(o)
    ctx.decode([(1 + j, 1 + j, None) for j in range(3)], 7)
    hdr, mbs, coef, mvs = synth_ir(w, h, seed0 + k, inter=True, version=version, filter_type=ftype, dense=float(rng.random()) * 0.6, big=bool(rng.integers(0, 2)))
    if rng.random() < 0.5:                  # many whole-pixel vectors
        one = (mbs[:, 2] != 0) & (mbs[:, 0] != 9)
        sel = one & (rng.random(len(mbs)) < 0.6)
        mvs[sel] = (mvs[sel] // 8) * 8
    o = np.zeros(g.frame_size, np.uint8); oracle_decode(hdr, mbs, coef, mvs, o, tuple(refs), 7)
    ctx.fill_slot(0, hdr, mbs, coef, mvs)
    for mode, dst in ((2, 0), (0, 4)):
        ctx._chk(ctx.L.vp8hip_set_pred_tiles(ctx.h, mode), "x")
        ctx.decode([(0, dst, (1, 2, 3))], 7)
        d = bordered_area_equal(ctx.download_full(dst), o, g)
        if d:
            bad += 1
            print(f"case {k} seed {seed0 + k}: {w}x{h} version {version} filter {ftype} pred_tiles {mode}: {d}")
    ctx.close()
print(f"{cases} cases, {bad} mismatching launches")
