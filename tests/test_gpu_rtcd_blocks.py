"""GPU: the reference's per-block RTCD entries as the product exports them (include/vp8_rtcd.h part 2, set by
vpx_rtcd()) against the oracle's restatements of the reference's `_c` functions, called through the function pointers of
the table with the reference's argument lists.  The same seeded inputs (and shapes) as tests/test_oracle_vs_ref.py, which
pins those restatements to the reference compiled here."""
import ctypes

import numpy as np
import pytest

from vp8_testlib import oracle

pytestmark = pytest.mark.gpu

vp = ctypes.c_void_p
ci = ctypes.c_int


class Lfi(ctypes.Structure):         # struct loop_filter_info, vp8/common/loopfilter.h:51-57
    _fields_ = [("mblim", vp), ("blim", vp), ("lim", vp), ("hev_thr", vp)]


class OraLfi(ctypes.Structure):
    _fields_ = [("mblim", ctypes.c_ubyte), ("blim", ctypes.c_ubyte), ("lim", ctypes.c_ubyte), ("hev_thr", ctypes.c_ubyte)]


class Blockd(ctypes.Structure):      # leading members of BLOCKD, vp8/common/blockd.h:186-192
    _fields_ = [("qcoeff_base", vp), ("qcoeff_offset", ci), ("dqcoeff_base", vp), ("dqcoeff_offset", ci)]


@pytest.fixture(scope="module")
def table(pkg):
    """name -> callable through the RTCD function pointer (not the _hip symbol): what a caller of the table sees."""
    H = pkg.load_host()
    H.vpx_rtcd()

    def entry(name, *argtypes):
        ptr = ctypes.c_void_p.in_dll(H, name)
        assert ptr.value, f"vpx_rtcd() left {name} unset"
        return ctypes.CFUNCTYPE(None, *argtypes)(ptr.value)
    return entry


def rnd_coefs(rng, n, dense=True, big=False):
    mag = 2047 if big else 120
    q = rng.integers(-mag, mag + 1, size=(n, 16)).astype(np.int16)
    if not dense:
        q[rng.random((n, 16)) < 0.7] = 0
    return q


def test_dequant_and_idct(table):
    O = oracle()
    rng = np.random.default_rng(1)
    f = table("vp8_dequant_idct_add", vp, vp, vp, ci)
    for big in (False, True):
        q = rnd_coefs(rng, 150, big=big)
        dq = rng.integers(4, 158 if not big else 2000, size=(150, 16)).astype(np.int16)
        pred = rng.integers(0, 256, size=(150, 4, 8)).astype(np.uint8)
        for i in range(150):
            qa, qb = q[i].copy(), q[i].copy()
            da, db = pred[i].copy(), pred[i].copy()
            f(qa.ctypes.data, dq[i].ctypes.data, da.ctypes.data, 8)
            O.vp8o_dequant_idct_add(vp(qb.ctypes.data), vp(dq[i].ctypes.data), vp(db.ctypes.data), ci(8))
            assert (da == db).all() and (qa == qb).all()
    # vp8_dequantize_b over the leading members of BLOCKD, with non-zero offsets
    f = table("vp8_dequantize_b", ctypes.POINTER(Blockd), vp)
    for i in range(50):
        q = rnd_coefs(rng, 25, big=bool(i & 1)).reshape(-1)
        dqc = rng.integers(4, 2000, size=16).astype(np.int16)
        out = np.zeros(400, np.int16)
        blk = int(rng.integers(0, 25))
        b = Blockd(q.ctypes.data, 16 * blk, out.ctypes.data, 16 * blk)
        f(ctypes.byref(b), dqc.ctypes.data)
        exp = np.zeros(400, np.int16)
        exp[16 * blk: 16 * blk + 16] = (q[16 * blk: 16 * blk + 16].astype(np.int32) * dqc).astype(np.int16)
        assert (out == exp).all()
    # vp8_short_idct4x4llm: separate predictor and destination
    f = table("vp8_short_idct4x4llm", vp, vp, ci, vp, ci)
    O.vp8o_short_idct4x4llm.argtypes = [vp, vp, ci, vp, ci]
    for i in range(150):
        inp = rnd_coefs(rng, 1, big=bool(i & 1))[0]
        pred = rng.integers(0, 256, size=(4, 16)).astype(np.uint8)
        a = np.full((4, 8), 7, np.uint8)
        b = a.copy()
        f(inp.ctypes.data, pred.ctypes.data, 16, a.ctypes.data, 8)
        O.vp8o_short_idct4x4llm(inp.ctypes.data, pred.ctypes.data, 16, b.ctypes.data, 8)
        assert (a == b).all()


def test_dc_only_and_walsh(table):
    O = oracle()
    rng = np.random.default_rng(2)
    f = table("vp8_dc_only_idct_add", ctypes.c_short, vp, ci, vp, ci)
    O.vp8o_dc_only_idct_add.argtypes = [ctypes.c_short, vp, ci, vp, ci]
    for i in range(150):
        dc = int(rng.integers(-32768, 32768))
        pred = rng.integers(0, 256, size=(4, 8)).astype(np.uint8)
        a, b = pred.copy(), pred.copy()
        f(dc, a.ctypes.data, 8, a.ctypes.data, 8)
        O.vp8o_dc_only_idct_add(dc, b.ctypes.data, 8, b.ctypes.data, 8)
        assert (a == b).all()
    w = table("vp8_short_inv_walsh4x4", vp, vp)
    w1 = table("vp8_short_inv_walsh4x4_1", vp, vp)
    for i in range(150):
        y2 = rng.integers(-32768, 32768, size=16).astype(np.int16) if i % 2 else rnd_coefs(rng, 1, big=True)[0]
        a = rng.integers(-100, 100, size=400).astype(np.int16)      # everything but the 16 DC slots must survive
        b = a.copy()
        w(y2.ctypes.data, a.ctypes.data)
        O.vp8o_short_inv_walsh4x4(vp(y2.ctypes.data), vp(b.ctypes.data))
        assert (a == b).all()
        w1(y2.ctypes.data, a.ctypes.data)
        O.vp8o_short_inv_walsh4x4_1(vp(y2.ctypes.data), vp(b.ctypes.data))
        assert (a == b).all()


def test_block_drivers(table):
    O = oracle()
    rng = np.random.default_rng(3)
    fy = table("vp8_dequant_idct_add_y_block", vp, vp, vp, ci, vp)
    fuv = table("vp8_dequant_idct_add_uv_block", vp, vp, vp, vp, ci, vp)
    for i in range(100):
        q = rnd_coefs(rng, 25, dense=bool(i % 2)).reshape(-1)
        eobs = rng.integers(0, 17, size=25).astype(np.int8)
        for b in range(25):
            if eobs[b] <= 1:
                q[b * 16 + 1: b * 16 + 16] = 0
        dq = np.full(16, int(rng.integers(4, 158)), np.int16)
        dq[0] = int(rng.integers(4, 158))
        fa = rng.integers(0, 256, size=(16, 32)).astype(np.uint8)
        fb = fa.copy()
        qa, qb = q.copy(), q.copy()
        fy(qa.ctypes.data, dq.ctypes.data, fa.ctypes.data, 32, eobs.ctypes.data)
        O.vp8o_dequant_idct_add_y_block(vp(qb.ctypes.data), vp(dq.ctypes.data), vp(fb.ctypes.data), ci(32), vp(eobs.ctypes.data))
        assert (fa == fb).all() and (qa == qb).all()
        ua = rng.integers(0, 256, size=(8, 16)).astype(np.uint8)
        va = rng.integers(0, 256, size=(8, 16)).astype(np.uint8)
        ub, vb = ua.copy(), va.copy()
        fuv(qa.ctypes.data + 512, dq.ctypes.data, ua.ctypes.data, va.ctypes.data, 16, eobs.ctypes.data + 16)
        O.vp8o_dequant_idct_add_uv_block(vp(qb.ctypes.data + 512), vp(dq.ctypes.data), vp(ub.ctypes.data), vp(vb.ctypes.data),
                                         ci(16), vp(eobs.ctypes.data + 16))
        assert (ua == ub).all() and (va == vb).all() and (qa == qb).all()


@pytest.mark.parametrize("w,h,suffix", [(4, 4, "4x4"), (8, 8, "8x8"), (8, 4, "8x4"), (16, 16, "16x16")])
def test_subpixel_predictors(table, w, h, suffix):
    O = oracle()
    rng = np.random.default_rng(4)
    O.vp8o_sixtap_predict.argtypes = [vp, ci, ci, ci, vp, ci, ci, ci]
    O.vp8o_bilinear_predict.argtypes = [vp, ci, ci, ci, vp, ci, ci, ci]
    for trial in range(3):
        src = rng.integers(0, 256, size=(32, 48)).astype(np.uint8)
        if trial == 0:
            src = (src > 127).astype(np.uint8) * 255      # extreme edges: exercises the pass-1 clamp
        sp = src.ctypes.data + 8 * 48 + 8 + trial          # every alignment of the source pointer
        for kind in ("sixtap", "bilinear"):
            f = table(f"vp8_{kind}_predict{suffix}", vp, ci, ci, ci, vp, ci)
            for xo in range(8):
                for yo in range(8):
                    a = np.full((16, 24), 9, np.uint8)
                    b = a.copy()
                    f(sp, 48, xo, yo, a.ctypes.data, 24)
                    getattr(O, f"vp8o_{kind}_predict")(sp, 48, xo, yo, b.ctypes.data, 24, w, h)
                    assert (a == b).all(), (kind, xo, yo)


def test_copy_mem(table):
    rng = np.random.default_rng(7)
    for name, w, h in (("vp8_copy_mem16x16", 16, 16), ("vp8_copy_mem8x8", 8, 8), ("vp8_copy_mem8x4", 8, 4)):
        f = table(name, vp, ci, vp, ci)
        src = rng.integers(0, 256, size=(20, 40)).astype(np.uint8)
        dst = np.full((20, 24), 3, np.uint8)
        f(src.ctypes.data + 2 * 40 + 5, 40, dst.ctypes.data + 1 * 24 + 2, 24)
        exp = np.full((20, 24), 3, np.uint8)
        exp[1:1 + h, 2:2 + w] = src[2:2 + h, 5:5 + w]
        assert (dst == exp).all()


def test_intra_predictors(table):
    O = oracle()
    rng = np.random.default_rng(5)
    f4 = table("vp8_intra4x4_predict", vp, ci, ci, vp, ci)
    for i in range(40):
        src = rng.integers(0, 256, size=(8, 16)).astype(np.uint8)
        for mode in range(10):
            a, b = src.copy(), src.copy()
            f4(a.ctypes.data + 2 * 16 + 4, 16, mode, a.ctypes.data + 2 * 16 + 4, 16)      # in place, as the decoder calls it
            O.vp8o_intra4x4_predict_ptr(vp(b.ctypes.data + 2 * 16 + 4), ci(16), ci(mode), vp(b.ctypes.data + 2 * 16 + 4), ci(16))
            assert (a == b).all(), mode
    ys = table("vp8_build_intra_predictors_mby_s_px", vp, ci, ci, ci, ci)
    yp = table("vp8_build_intra_predictors_mby_px", vp, ci, ci, ci, ci, vp)
    uvs = table("vp8_build_intra_predictors_mbuv_s_px", vp, vp, ci, ci, ci, ci)
    uvp = table("vp8_build_intra_predictors_mbuv_px", vp, vp, ci, ci, ci, ci, vp, vp)
    O.vp8o_build_intra_predictors_plane_s.argtypes = [vp, ci, ci, ci, ci, ci]
    for i in range(12):
        for mode in range(5):            # DC, V, H, TM, and B_PRED (which must leave the block alone)
            for up, left in ((0, 0), (1, 0), (0, 1), (1, 1)):
                y = rng.integers(0, 256, size=(20, 40)).astype(np.uint8)
                u = rng.integers(0, 256, size=(12, 24)).astype(np.uint8)
                v = rng.integers(0, 256, size=(12, 24)).astype(np.uint8)
                ya, ua, va = y.copy(), u.copy(), v.copy()
                yb, ub, vb = y.copy(), u.copy(), v.copy()
                oy, ou = 2 * 40 + 8, 2 * 24 + 8
                if mode < 4:
                    O.vp8o_build_intra_predictors_plane_s(yb.ctypes.data + oy, 40, 16, mode, up, left)
                    O.vp8o_build_intra_predictors_plane_s(ub.ctypes.data + ou, 24, 8, mode, up, left)
                    O.vp8o_build_intra_predictors_plane_s(vb.ctypes.data + ou, 24, 8, mode, up, left)
                pred = np.full(384, 5, np.uint8)
                yp(ya.ctypes.data + oy, 40, mode, up, left, pred.ctypes.data)
                uvp(ua.ctypes.data + ou, va.ctypes.data + ou, 24, mode, up, left, pred.ctypes.data + 256, pred.ctypes.data + 320)
                assert (ya == y).all() and (ua == u).all() and (va == v).all()
                if mode < 4:
                    assert (pred[:256].reshape(16, 16) == yb[2:18, 8:24]).all()
                    assert (pred[256:320].reshape(8, 8) == ub[2:10, 8:16]).all()
                    assert (pred[320:].reshape(8, 8) == vb[2:10, 8:16]).all()
                else:
                    assert (pred == 5).all()
                ys(ya.ctypes.data + oy, 40, mode, up, left)
                uvs(ua.ctypes.data + ou, va.ctypes.data + ou, 24, mode, up, left)
                assert (ya == yb).all() and (ua == ub).all() and (va == vb).all(), (mode, up, left)


def test_loop_filters(table):
    O = oracle()
    rng = np.random.default_rng(6)
    for i in range(60):
        level = int(rng.integers(1, 64))
        sharp = int(rng.integers(0, 8))
        ftype = int(rng.integers(0, 2))
        ol = OraLfi()
        O.vp8o_lf_limits(ci(sharp), ci(level), ci(ftype), ctypes.byref(ol))
        arrs = [np.full(16, v, np.uint8) for v in (ol.mblim, ol.blim, ol.lim, ol.hev_thr)]
        rl = Lfi(*[a.ctypes.data for a in arrs])
        base = rng.integers(0, 256)
        y = np.clip(base + rng.integers(-12, 13, size=(48, 64)), 0, 255).astype(np.uint8)
        u = np.clip(base + rng.integers(-12, 13, size=(24, 32)), 0, 255).astype(np.uint8)
        v = np.clip(base + rng.integers(-12, 13, size=(24, 32)), 0, 255).astype(np.uint8)
        if i % 5 == 0:
            y = rng.integers(0, 256, size=(48, 64)).astype(np.uint8)
        yo, co = 16 * 64 + 16, 8 * 32 + 8
        for fn in ("mbv", "bv", "mbh", "bh"):
            f = table(f"vp8_loop_filter_{fn}", vp, vp, vp, ci, ci, ctypes.POINTER(Lfi))
            ya, ua, va = y.copy(), u.copy(), v.copy()
            yb, ub, vb = y.copy(), u.copy(), v.copy()
            f(ya.ctypes.data + yo, ua.ctypes.data + co, va.ctypes.data + co, 64, 32, ctypes.byref(rl))
            getattr(O, f"vp8o_loop_filter_{fn}")(vp(yb.ctypes.data + yo), vp(ub.ctypes.data + co), vp(vb.ctypes.data + co),
                                                 ci(64), ci(32), ctypes.byref(ol))
            assert (ya == yb).all() and (ua == ub).all() and (va == vb).all(), fn
            if i % 10 == 0:              # luma only: the reference filters chroma `if (u_ptr)` (loopfilter_filters.c:330)
                ya = y.copy()
                f(ya.ctypes.data + yo, None, None, 64, 32, ctypes.byref(rl))
                assert (ya == yb).all(), fn
        for fn, lim in (("mbv", ol.mblim), ("mbh", ol.mblim), ("bv", ol.blim), ("bh", ol.blim)):
            f = table(f"vp8_loop_filter_simple_{fn}", vp, ci, vp)
            ya, yb = y.copy(), y.copy()
            la = np.full(16, lim, np.uint8)
            f(ya.ctypes.data + yo, 64, la.ctypes.data)
            getattr(O, f"vp8o_loop_filter_simple_{fn}")(vp(yb.ctypes.data + yo), ci(64), ctypes.c_ubyte(lim))
            assert (ya == yb).all(), fn
