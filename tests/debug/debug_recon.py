#!/usr/bin/env python3
"""Debug aid: recon stage of the HIP path vs the oracle on one fixture, first mismatching MBs with their modes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vp8_testlib import load_package, ivf_path, oracle_decode

name = sys.argv[1] if len(sys.argv) > 1 else "kf_odd_67x45"
nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
P = load_package()
w, h, frames = P.read_ivf(ivf_path(name))
parser = P.Parser()
ctx = P.Vp8Hip(0)
obufs = None
for fi, data in enumerate(frames[:nframes]):
    hdr, changed, mbs, coef, mvs = P.parse_to_numpy(parser, data)
    if changed:
        ctx.configure(hdr.width, hdr.height, 4, 1)
        g = ctx.g
        obufs = [np.zeros(g.frame_size, np.uint8) for _ in range(4)]
    r = parser.refs
    refs = (r.lst_idx, r.gld_idx, r.alt_idx)
    for idx in set(refs) - {r.new_idx}:
        ctx.upload_frame(idx, obufs[idx])
    ctx.fill_slot(0, hdr, mbs, coef, mvs)
    o = np.zeros(g.frame_size, np.uint8)
    oracle_decode(hdr, mbs, coef, mvs, o, tuple(obufs[i] for i in refs), 1)
    ctx.decode([(0, r.new_idx, refs)], 1)
    got = ctx.download_full(r.new_idx)
    cols, rows = g.aligned_w // 16, g.aligned_h // 16
    bad = []
    for (off, st, n, nm) in ((g.y_off, g.y_stride, 16, "Y"), (g.u_off, g.uv_stride, 8, "U"), (g.v_off, g.uv_stride, 8, "V")):
        for rr in range(rows):
            for cc in range(cols):
                a = np.lib.stride_tricks.as_strided(got[off + rr * n * st + cc * n:], (n, n), (st, 1))
                b = np.lib.stride_tricks.as_strided(o[off + rr * n * st + cc * n:], (n, n), (st, 1))
                if not np.array_equal(a, b):
                    bad.append((rr, cc, nm, a.copy(), b.copy()))
    print(f"frame {fi}: type {hdr.frame_type} {len(bad)} bad MB-planes of {rows*cols*3}; bad rows {sorted(set(b[0] for b in bad))}")
    seen = 0
    for rr, cc, nm, a, b in sorted(bad, key=lambda t: (t[0] * 2 + t[1], t[2]))[:6]:
        m = mbs[rr * cols + cc]
        print(f"  MB r={rr} c={cc} plane {nm}: y_mode {m[0]} uv_mode {m[1]} ref {m[2]} flags {m[3]} "
              f"seg {m[4]} bmodes {list(m[40:56])}")
        d = (a != b)
        print("   diff rows:", sorted(set(np.nonzero(d)[0].tolist())), "cols:", sorted(set(np.nonzero(d)[1].tolist())))
        if seen < 2:
            print("   got:\n", a, "\n   want:\n", b)
            seen += 1
    # continue from the oracle's full decode so later frames have good references
    full = np.zeros(g.frame_size, np.uint8)
    oracle_decode(hdr, mbs, coef, mvs, full, tuple(obufs[i] for i in refs), 7)
    obufs[r.new_idx][:] = full
    parser.swap(hdr)
ctx.close()
