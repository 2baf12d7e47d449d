#!/usr/bin/env python3
"""Debug aid: one-MB inter frames (pure prediction, no residual) HIP vs oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vp8_testlib import load_package, oracle_decode, synth_ir, random_frame
P = load_package()
ctx = P.Vp8Hip(0)
w = h = 16
ctx.configure(w, h, 4, 1)
g = ctx.g
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    hdr, mbs, coef, mvs = synth_ir(w, h, seed, inter=True, version=0, dense=0.0, segmented=False)
    refs_np = [random_frame(g, 100 + seed * 3 + k) for k in range(3)]
    for k in range(3):
        ctx.upload_frame(1 + k, refs_np[k])
    ctx.fill_slot(0, hdr, mbs, coef, mvs)
    o = np.zeros(g.frame_size, np.uint8)
    oracle_decode(hdr, mbs, coef, mvs, o, tuple(refs_np), 1)
    ctx.decode([(0, 0, (1, 2, 3))], 1)
    got = ctx.download_full(0)
    m = mbs[0]
    line = f"seed {seed}: y_mode {m[0]} ref {m[2]} flags {m[3]} part {m[5]} mv0 {tuple(mvs[0,0])}"
    for (off, st, n, nm) in ((g.y_off, g.y_stride, 16, "Y"), (g.u_off, g.uv_stride, 8, "U"), (g.v_off, g.uv_stride, 8, "V")):
        a = np.lib.stride_tricks.as_strided(got[off:], (n, n), (st, 1))
        b = np.lib.stride_tricks.as_strided(o[off:], (n, n), (st, 1))
        d = a != b
        if d.any():
            line += f" | {nm} bad rows {sorted(set(np.nonzero(d)[0].tolist()))} cols {sorted(set(np.nonzero(d)[1].tolist()))}"
    print(line)
    if seed == int(os.environ.get('DBG_SEED', '-1')):
        a = np.lib.stride_tricks.as_strided(got[g.y_off:], (16, 16), (g.y_stride, 1))
        b = np.lib.stride_tricks.as_strided(o[g.y_off:], (16, 16), (g.y_stride, 1))
        rfb = refs_np[m[2] - 1]
        rr = np.lib.stride_tricks.as_strided(rfb[g.y_off - 4 * g.y_stride - 4:], (24, 24), (g.y_stride, 1))
        print("got\n", a[:8, :8], "\nwant\n", b[:8, :8], "\nref (-4..)\n", rr[:14, :14])
    if m[0] == 9:
        print("   mvs", [tuple(x) for x in mvs[0]])
ctx.close()
