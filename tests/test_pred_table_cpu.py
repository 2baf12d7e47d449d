"""CPU: the selector table of the branch-free 4x4 intra predictor (libvpx.opencl_amd/csrc/hip/vp8_pred_sel.inc, written by
tools/gen_pred_sel.py; used by pred4x4_net, vp8_simt_prims.hip.h) against the oracle's vp8_intra4x4_predict
(vp8/common/reconintra4x4.c:16-303).

The device computes a POOL from the block's edge -- F[k] = (P[k-1] + 2 P[k] + P[k+1] + 2) >> 2 and G[k] = (P[k] + P[k+1] + 1) >> 1
over P = { L3 L3 L2 L1 L0 TL A0 .. A7 A7 }, by v_lerp_u8 pairs -- and a predicted row is perm(pair 0) | perm(pair 1) | perm(pair 2)
with the selectors of the table.  Here the same arithmetic in numpy (v_perm_b32 and v_lerp_u8 restated from the ISA's description),
the committed table, every directional mode + B_DC_PRED, random and flat edges; and the table is what the generator writes."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np

from vp8_testlib import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "libvpx.opencl_amd", "csrc", "hip", "vp8_pred_sel.inc")


def table():
    words = [int(w, 16) for w in re.findall(r"0x([0-9a-f]{8})u", open(INC).read())]
    assert len(words) == 11 * 12
    return np.array(words, np.uint32).reshape(11, 4, 3)


def perm(hi, lo, sel):
    """v_perm_b32: byte k of the result is byte sel[k] of { lo (0..3), hi (4..7) }; 0x0c selects zero"""
    pool = list(lo) + list(hi)
    out = []
    for k in range(4):
        s = (int(sel) >> (8 * k)) & 0xff
        out.append(0 if s == 0x0c else pool[s])
        assert s == 0x0c or s < 8
    return out


def pool_of(above, left, tl):
    P = [left[3], left[3], left[2], left[1], left[0], tl] + list(above) + [above[7]]      # 15 entries; P[15] = A7 too (E3's last byte)
    P.append(above[7])
    F = [0] * 16
    G = [0] * 16
    for k in range(16):
        pm = P[k - 1] if k else 0                          # M0 = E0 << 8: a zero byte in front
        pn = P[k + 1] if k + 1 < 16 else 0                 # N3 = E3 >> 8
        F[k] = ((((pm + pn) >> 1) + P[k] + 1) >> 1)        # lerp(lerp(M, N, 0), E, 1)
        G[k] = (P[k] + pn + 1) >> 1
    return F, G


def test_table_is_what_the_generator_writes(tmp_path):
    before = open(INC).read()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_pred_sel.py")], check=True, stdout=subprocess.DEVNULL)
    assert open(INC).read() == before


def test_every_mode_against_the_oracle():
    O = oracle()
    T = table()
    rng = np.random.default_rng(3)
    vp = ctypes.c_void_p
    for trial in range(400):
        ctx = rng.integers(0, 256, 13).astype(np.uint8)
        if trial % 5 == 0:
            ctx[:] = ctx[0]
        if trial % 7 == 0:
            ctx[:] = rng.choice([0, 255], 13)
        above, left, tl = ctx[:8].copy(), ctx[8:12].copy(), int(ctx[12])
        F, G = pool_of([int(v) for v in above], [int(v) for v in left], tl)
        dc = (int(above[:4].sum()) + int(left.sum()) + 4) >> 3
        for mode in (0, 2, 3, 4, 5, 6, 7, 8, 9):
            lo0 = [dc] * 4 if mode == 0 else F[0:4]          # entry 0: the dword C in place of F[0..3]
            pairs = ((F[4:8], lo0), ([F[12], F[13], G[8], G[9]], F[8:12]), (G[4:8], G[0:4]))
            got = np.zeros((4, 4), np.uint8)
            for r in range(4):
                row = [0, 0, 0, 0]
                for k, (hi, lo) in enumerate(pairs):
                    row = [a | b for a, b in zip(row, perm(hi, lo, T[mode, r, k]))]
                got[r] = row
            want = np.zeros((4, 4), np.uint8)
            O.vp8o_intra4x4_predict(vp(above.ctypes.data), vp(left.ctypes.data), ctypes.c_ubyte(tl), ctypes.c_int(mode),
                                    vp(want.ctypes.data), ctypes.c_int(4))
            assert np.array_equal(got, want), (trial, mode, got, want)


def test_macroblock_entries():
    """entry 0 with C = the line above is V_PRED's row, entry 10 over the left column H_PRED's rows (reconintra.c:162-187);
    entry 1 (B_TM_PRED: arithmetic, merged by a select) is a copy of entry 0"""
    T = table()
    C, col = [11, 22, 33, 44], [5, 6, 7, 8]
    assert np.array_equal(T[0], T[1])
    for r in range(4):
        assert perm([0] * 4, C, T[0, r, 0]) == C and T[0, r, 1] == 0x0c0c0c0c and T[0, r, 2] == 0x0c0c0c0c
        assert perm([0] * 4, col, T[10, r, 2]) == [col[r]] * 4 and T[10, r, 0] == 0x0c0c0c0c and T[10, r, 1] == 0x0c0c0c0c
