"""CPU: the oracle's RTCD-level functions against the REAL reference's `_c` functions
(oracle/_ref/libvpxref.so, built from /root/reference by oracle/Makefile) on seeded random inputs.
Skipped where the reference build is absent."""
import ctypes
import os

import numpy as np
import pytest

from vp8_testlib import ORACLE_LIB, REF_LIB, oracle

pytestmark = pytest.mark.skipif(not os.path.exists(REF_LIB), reason="oracle/_ref not built (needs /root/reference)")

vp = ctypes.c_void_p
ci = ctypes.c_int


@pytest.fixture(scope="module")
def libs():
    O = oracle()
    R = ctypes.CDLL(REF_LIB)
    return O, R


def rnd_coefs(rng, n, dense=True, big=False):
    mag = 2047 if big else 120
    q = rng.integers(-mag, mag + 1, size=(n, 16)).astype(np.int16)
    if not dense:
        q[rng.random((n, 16)) < 0.7] = 0
    return q


def test_dequant_idct_add(libs):
    O, R = libs
    rng = np.random.default_rng(1)
    for big in (False, True):
        q = rnd_coefs(rng, 3000, big=big)
        dq = rng.integers(4, 158 if not big else 2000, size=(3000, 16)).astype(np.int16)
        pred = rng.integers(0, 256, size=(3000, 4, 8)).astype(np.uint8)
        for i in range(3000):
            qa, qb = q[i].copy(), q[i].copy()
            da, db = pred[i].copy(), pred[i].copy()
            R.vp8_dequant_idct_add_c(vp(qa.ctypes.data), vp(dq[i].ctypes.data), vp(da.ctypes.data), ci(8))
            O.vp8o_dequant_idct_add(vp(qb.ctypes.data), vp(dq[i].ctypes.data), vp(db.ctypes.data), ci(8))
            assert (da == db).all() and (qa == qb).all()


def test_dc_only_and_walsh(libs):
    O, R = libs
    rng = np.random.default_rng(2)
    R.vp8_dc_only_idct_add_c.argtypes = [ctypes.c_short, vp, ci, vp, ci]
    O.vp8o_dc_only_idct_add.argtypes = [ctypes.c_short, vp, ci, vp, ci]
    for i in range(2000):
        dc = int(rng.integers(-32768, 32768))
        pred = rng.integers(0, 256, size=(4, 8)).astype(np.uint8)
        a, b = pred.copy(), pred.copy()
        R.vp8_dc_only_idct_add_c(dc, a.ctypes.data, 8, a.ctypes.data, 8)
        O.vp8o_dc_only_idct_add(dc, b.ctypes.data, 8, b.ctypes.data, 8)
        assert (a == b).all()
    for i in range(2000):
        y2 = rng.integers(-32768, 32768, size=16).astype(np.int16) if i % 2 else rnd_coefs(rng, 1, big=True)[0]
        a, b = np.zeros(400, np.int16), np.zeros(400, np.int16)
        R.vp8_short_inv_walsh4x4_c(vp(y2.ctypes.data), vp(a.ctypes.data))
        O.vp8o_short_inv_walsh4x4(vp(y2.ctypes.data), vp(b.ctypes.data))
        assert (a == b).all()
        R.vp8_short_inv_walsh4x4_1_c(vp(y2.ctypes.data), vp(a.ctypes.data))
        O.vp8o_short_inv_walsh4x4_1(vp(y2.ctypes.data), vp(b.ctypes.data))
        assert (a == b).all()


def test_block_drivers(libs):
    O, R = libs
    rng = np.random.default_rng(3)
    for i in range(300):
        q = rnd_coefs(rng, 25, dense=bool(i % 2)).reshape(-1)
        eobs = rng.integers(0, 17, size=25).astype(np.int8)
        # make coefficients consistent with "eob <= 1 means DC only"
        for b in range(25):
            if eobs[b] <= 1:
                q[b * 16 + 1: b * 16 + 16] = 0
        dq = np.full(16, int(rng.integers(4, 158)), np.int16)
        dq[0] = int(rng.integers(4, 158))
        fa = rng.integers(0, 256, size=(16, 32)).astype(np.uint8)
        fb = fa.copy()
        qa, qb = q.copy(), q.copy()
        R.vp8_dequant_idct_add_y_block_c(vp(qa.ctypes.data), vp(dq.ctypes.data), vp(fa.ctypes.data), ci(32), vp(eobs.ctypes.data))
        O.vp8o_dequant_idct_add_y_block(vp(qb.ctypes.data), vp(dq.ctypes.data), vp(fb.ctypes.data), ci(32), vp(eobs.ctypes.data))
        assert (fa == fb).all() and (qa == qb).all()
        ua = rng.integers(0, 256, size=(8, 16)).astype(np.uint8)
        va = rng.integers(0, 256, size=(8, 16)).astype(np.uint8)
        ub, vb = ua.copy(), va.copy()
        R.vp8_dequant_idct_add_uv_block_c(vp(qa.ctypes.data + 512), vp(dq.ctypes.data), vp(ua.ctypes.data), vp(va.ctypes.data), ci(16), vp(eobs.ctypes.data + 16))
        O.vp8o_dequant_idct_add_uv_block(vp(qb.ctypes.data + 512), vp(dq.ctypes.data), vp(ub.ctypes.data), vp(vb.ctypes.data), ci(16), vp(eobs.ctypes.data + 16))
        assert (ua == ub).all() and (va == vb).all()


@pytest.mark.parametrize("w,h,suffix", [(4, 4, "4x4"), (8, 8, "8x8"), (8, 4, "8x4"), (16, 16, "16x16")])
def test_subpixel_predictors(libs, w, h, suffix):
    O, R = libs
    rng = np.random.default_rng(4)
    O.vp8o_sixtap_predict.argtypes = [vp, ci, ci, ci, vp, ci, ci, ci]
    O.vp8o_bilinear_predict.argtypes = [vp, ci, ci, ci, vp, ci, ci, ci]
    for trial in range(12):
        src = rng.integers(0, 256, size=(32, 48)).astype(np.uint8)
        if trial % 3 == 0:
            src = (src > 127).astype(np.uint8) * 255      # extreme edges: exercises the pass-1 clamp
        sp = src.ctypes.data + 8 * 48 + 8
        for xo in range(8):
            for yo in range(8):
                for kind in ("sixtap", "bilinear"):
                    if kind == "sixtap" and xo == 0 and yo == 0:
                        pass   # the C code is still well defined (identity taps)
                    a = np.zeros((16, 16), np.uint8)
                    b = np.zeros((16, 16), np.uint8)
                    getattr(R, f"vp8_{kind}_predict{suffix}_c")(vp(sp), ci(48), ci(xo), ci(yo), vp(a.ctypes.data), ci(16))
                    getattr(O, f"vp8o_{kind}_predict")(sp, 48, xo, yo, b.ctypes.data, 16, w, h)
                    assert (a == b).all(), (kind, xo, yo)


def test_intra4x4(libs):
    O, R = libs
    rng = np.random.default_rng(5)
    for i in range(600):
        src = rng.integers(0, 256, size=(8, 16)).astype(np.uint8)
        for mode in range(10):
            a, b = src.copy(), src.copy()
            R.vp8_intra4x4_predict_c(vp(a.ctypes.data + 2 * 16 + 4), ci(16), ci(mode), vp(a.ctypes.data + 2 * 16 + 4), ci(16))
            O.vp8o_intra4x4_predict_ptr(vp(b.ctypes.data + 2 * 16 + 4), ci(16), ci(mode), vp(b.ctypes.data + 2 * 16 + 4), ci(16))
            assert (a == b).all(), mode


class RefLfi(ctypes.Structure):      # loop_filter_info, vp8/common/loopfilter.h:51-57
    _fields_ = [("mblim", vp), ("blim", vp), ("lim", vp), ("hev_thr", vp)]


class OraLfi(ctypes.Structure):
    _fields_ = [("mblim", ctypes.c_ubyte), ("blim", ctypes.c_ubyte), ("lim", ctypes.c_ubyte), ("hev_thr", ctypes.c_ubyte)]


def test_loop_filters(libs):
    O, R = libs
    rng = np.random.default_rng(6)
    for i in range(400):
        level = int(rng.integers(1, 64))
        sharp = int(rng.integers(0, 8))
        ftype = int(rng.integers(0, 2))
        ol = OraLfi()
        O.vp8o_lf_limits(ci(sharp), ci(level), ci(ftype), ctypes.byref(ol))
        arrs = [np.full(16, v, np.uint8) for v in (ol.mblim, ol.blim, ol.lim, ol.hev_thr)]
        rl = RefLfi(*[a.ctypes.data for a in arrs])
        # smooth-ish content so the masks actually open
        base = rng.integers(0, 256)
        y = np.clip(base + rng.integers(-12, 13, size=(48, 64)), 0, 255).astype(np.uint8)
        u = np.clip(base + rng.integers(-12, 13, size=(24, 32)), 0, 255).astype(np.uint8)
        v = np.clip(base + rng.integers(-12, 13, size=(24, 32)), 0, 255).astype(np.uint8)
        if i % 5 == 0:
            y = rng.integers(0, 256, size=(48, 64)).astype(np.uint8)
        for fn in ("mbv", "bv", "mbh", "bh"):
            ya, ua, va = y.copy(), u.copy(), v.copy()
            yb, ub, vb = y.copy(), u.copy(), v.copy()
            getattr(R, f"vp8_loop_filter_{fn}_c")(vp(ya.ctypes.data + 16 * 64 + 16), vp(ua.ctypes.data + 8 * 32 + 8),
                                                  vp(va.ctypes.data + 8 * 32 + 8), ci(64), ci(32), ctypes.byref(rl))
            getattr(O, f"vp8o_loop_filter_{fn}")(vp(yb.ctypes.data + 16 * 64 + 16), vp(ub.ctypes.data + 8 * 32 + 8),
                                                 vp(vb.ctypes.data + 8 * 32 + 8), ci(64), ci(32), ctypes.byref(ol))
            assert (ya == yb).all() and (ua == ub).all() and (va == vb).all(), fn
        for rfn, ofn, lim in (("vp8_loop_filter_simple_vertical_edge_c", "vp8o_loop_filter_simple_mbv", ol.mblim),
                              ("vp8_loop_filter_simple_horizontal_edge_c", "vp8o_loop_filter_simple_mbh", ol.mblim),
                              ("vp8_loop_filter_bvs_c", "vp8o_loop_filter_simple_bv", ol.blim),
                              ("vp8_loop_filter_bhs_c", "vp8o_loop_filter_simple_bh", ol.blim)):
            ya, yb = y.copy(), y.copy()
            la = np.full(16, lim, np.uint8)
            getattr(R, rfn)(vp(ya.ctypes.data + 16 * 64 + 16), ci(64), vp(la.ctypes.data))
            getattr(O, ofn)(vp(yb.ctypes.data + 16 * 64 + 16), ci(64), ctypes.c_ubyte(lim))
            assert (ya == yb).all(), rfn


def test_quant_tables(libs):
    O, R = libs

    class Hdr(ctypes.Structure):
        pass
    from vp8_testlib import load_package
    P = load_package()
    for q in range(128):
        for d in (-15, -3, 0, 4, 15):
            h = P.FrameHdr()
            h.base_qindex = q
            h.y1dc_delta_q = h.y2dc_delta_q = h.y2ac_delta_q = h.uvdc_delta_q = h.uvac_delta_q = d
            out = (ctypes.c_short * 6)()
            O.vp8o_mb_dequant(ctypes.byref(h), ci(0), out)
            exp = [R.vp8_dc_quant(q, d), R.vp8_ac_yquant(q), R.vp8_dc2quant(q, d), R.vp8_ac2quant(q, d),
                   R.vp8_dc_uv_quant(q, d), R.vp8_ac_uv_quant(q, d)]
            assert list(out) == exp


# ---- output-side post-processing (vp8/common/postproc.c; the reference build has CONFIG_POSTPROC 1) ----
def _planes(rng, rows, cols, border=24):
    """a plane with a border all round; the second one is flat-ish so that the filters actually fire"""
    h, w = rows + 2 * border, cols + 2 * border
    noisy = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
    base = rng.integers(0, 256, size=(h // 8 + 1, w // 8 + 1)).astype(np.int32)
    flat = np.kron(base, np.ones((8, 8), np.int32))[:h, :w]
    flat = np.clip(flat + rng.integers(-3, 4, size=(h, w)), 0, 255).astype(np.uint8)
    return noisy, flat, border, w


@pytest.mark.parametrize("rows,cols", [(16, 16), (48, 80), (144, 176)])
def test_postproc_filters(libs, rows, cols):
    O, R = libs
    libc = ctypes.CDLL(None)
    rng = np.random.default_rng(8)
    for trial in range(6):
        for src in _planes(rng, rows, cols)[:2]:
            _, _, b, w = _planes(rng, rows, cols)
            off = b * w + b
            for flimit in (-3, 0, 2, 7, 30, 300):
                a = np.zeros_like(src)
                o = np.zeros_like(src)
                R.vp8_post_proc_down_and_across_c(vp(src.ctypes.data + off), vp(a.ctypes.data + off), ci(w), ci(w), ci(rows), ci(cols), ci(flimit))
                O.vp8o_post_proc_down_and_across(vp(src.ctypes.data + off), vp(o.ctypes.data + off), ci(w), ci(w), ci(rows), ci(cols), ci(flimit))
                assert (a[b:b + rows, b:b + cols] == o[b:b + rows, b:b + cols]).all(), ("down_and_across", flimit)
            for flimit in (0, 50, 533, 3000, 100000):
                a = src.copy()
                o = np.zeros_like(src)
                R.vp8_mbpost_proc_across_ip_c(vp(a.ctypes.data + off), ci(w), ci(rows), ci(cols), ci(flimit))
                O.vp8o_mbpost_proc_across(vp(src.ctypes.data + off), vp(o.ctypes.data + off), ci(w), ci(rows), ci(cols), ci(flimit))
                assert (a[b:b + rows, b:b + cols] == o[b:b + rows, b:b + cols]).all(), ("across_ip", flimit)
                seed = int(rng.integers(1, 1 << 30))
                libc.srand(seed)
                rv_offset = libc.rand() & 63
                libc.srand(seed)
                a = src.copy()
                R.vp8_mbpost_proc_down_c(vp(a.ctypes.data + off), ci(w), ci(rows), ci(cols), ci(flimit))
                O.vp8o_mbpost_proc_down(vp(src.ctypes.data + off), vp(o.ctypes.data + off), ci(w), ci(rows), ci(cols), ci(flimit), ci(rv_offset))
                assert (a[b:b + rows, b:b + cols] == o[b:b + rows, b:b + cols]).all(), ("down", flimit)


def test_postproc_noise_and_strengths(libs):
    O, R = libs
    libc = ctypes.CDLL(None)
    rng = np.random.default_rng(9)
    for trial in range(10):
        rows, cols, w = 40, 72, 96
        src = rng.integers(0, 256, size=(rows, w)).astype(np.uint8)
        if trial % 2:
            src[:, :36] = rng.integers(0, 6, size=(rows, 36))         # near black / near white: the clamps
            src[:, 36:] = rng.integers(250, 256, size=(rows, w - 36))
        noise = rng.integers(-20, 21, size=3072).astype(np.int8)
        clamp = int(-noise.min())
        cl = np.full(16, clamp, np.int8)
        seed = int(rng.integers(1, 1 << 30))
        libc.srand(seed)
        offs = np.array([libc.rand() & 0xff for _ in range(rows)], np.uint8)
        libc.srand(seed)
        a, o = src.copy(), src.copy()
        R.vp8_plane_add_noise_c(vp(a.ctypes.data), vp(noise.ctypes.data), vp(cl.ctypes.data), vp(cl.ctypes.data), vp(cl.ctypes.data),
                                ctypes.c_uint(cols), ctypes.c_uint(rows), ci(w))
        O.vp8o_plane_add_noise(vp(o.ctypes.data), vp(noise.ctypes.data), ci(clamp), ci(cols), ci(rows), ci(w), vp(offs.ctypes.data))
        assert (a == o).all()
