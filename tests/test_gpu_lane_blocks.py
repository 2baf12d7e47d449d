"""GPU: the LANE-PER-ROW arithmetic, block by block, in the reference's terms.  The kernels that carry the benchmark
(vp8_keyframe_kernel; vp8_recon_simt_kernel / vp8_loopfilter_simt_*) are built from per-lane primitives -- packed-byte intra 4x4
predictors, the inverse transform on packed 16-bit pairs, the signed 8.8 loop filter streamed block row by block row -- that
the RTCD entries of tests/test_gpu_rtcd_blocks.py (wave-per-row primitives) do not reach.  include/vp8hip.h exports them one
block / macroblock per lane (vp8hip_lane_*); here they meet the oracle's restatements of the reference's per-block functions
(pinned to the reference compiled here by tests/test_oracle_vs_ref.py): vp8_loop_filter_{mbv,bv,mbh,bh}[_simple] in
vp8_loop_filter_frame's order (vp8/common/loopfilter.c:259-299), vp8_intra4x4_predict (reconintra4x4.c:16),
vp8_dequant_idct_add_c (dequantize.c:29)."""
import ctypes

import numpy as np
import pytest

from vp8_testlib import oracle

pytestmark = pytest.mark.gpu
vp, ci = ctypes.c_void_p, ctypes.c_int


class OraLfi(ctypes.Structure):
    _fields_ = [("mblim", ctypes.c_ubyte), ("blim", ctypes.c_ubyte), ("lim", ctypes.c_ubyte), ("hev_thr", ctypes.c_ubyte)]


@pytest.fixture(scope="module")
def L(pkg):
    lib = pkg.load_hip()
    lib.vp8hip_lane_loop_filter_mbs.argtypes = [vp, vp, vp, ci]
    lib.vp8hip_lane_intra4x4.argtypes = [vp, vp, vp, ci]
    lib.vp8hip_lane_dequant_idct_add.argtypes = [vp, vp, vp, vp, ci]
    return lib


@pytest.mark.parametrize("seed,noise", [(1, 3), (2, 12), (3, 40), (4, 255), (5, 255)])
def test_loop_filter_macroblocks(L, seed, noise):
    """64-lane steps of the streamed loop filter (lf_block_row: lf_mbedge / lf_inner / lf_simple on packed pairs, the
    row <-> column-pair shuffles, the left-context fix-up) on random macroblocks with random limits and edge gates."""
    O = oracle()
    rng = np.random.default_rng(seed)
    n = 64 * 40 + 17                        # a last wave with idle lanes
    base = rng.integers(0, 256, size=(n, 1, 1))
    grad = rng.integers(-3, 4, size=(n, 1, 1)) * np.arange(20).reshape(1, 20, 1) + rng.integers(-3, 4, size=(n, 1, 1)) * np.arange(20).reshape(1, 1, 20)
    step = (np.arange(20).reshape(1, 1, 20) >= rng.integers(0, 20, size=(n, 1, 1))) * rng.integers(-30, 31, size=(n, 1, 1))
    src = np.clip(base + grad + step + rng.integers(-noise, noise + 1, size=(n, 20, 20)), 0, 255).astype(np.uint8)
    if seed == 5:                           # black and white only: every difference is 0 or 255 (the masks' saturated differences)
        src = np.where(src > 127, 255, 0).astype(np.uint8)
    par = np.zeros((n, 8), np.uint8)
    lfi = OraLfi()
    for i in range(n):
        O.vp8o_lf_limits(ci(int(rng.integers(0, 8))), ci(int(rng.integers(1, 64))), ci(int(rng.integers(0, 2))), ctypes.byref(lfi))
        par[i, :4] = (lfi.mblim, lfi.blim, lfi.lim, lfi.hev_thr)
        par[i, 4:7] = rng.integers(0, 2, size=3)
        par[i, 7] = rng.random() < 0.3
    got = np.zeros_like(src)
    assert L.vp8hip_lane_loop_filter_mbs(vp(src.ctypes.data), vp(got.ctypes.data), vp(par.ctypes.data), n) == 0
    want = src.copy()
    dummy = np.zeros((24, 24), np.uint8)
    du = vp(dummy.ctypes.data + 8 * 24 + 8)
    for i in range(n):
        y = vp(want[i].ctypes.data + 4 * 20 + 4)
        lf = OraLfi(*[int(v) for v in par[i, :4]])
        mbv, inner, mbh, simple = (int(v) for v in par[i, 4:8])
        if simple:                          # loopfilter.c:284-299: luma only, macroblock edges with mblim, inner ones with blim
            if mbv: O.vp8o_loop_filter_simple_mbv(y, ci(20), ctypes.c_ubyte(lf.mblim))
            if inner: O.vp8o_loop_filter_simple_bv(y, ci(20), ctypes.c_ubyte(lf.blim))
            if mbh: O.vp8o_loop_filter_simple_mbh(y, ci(20), ctypes.c_ubyte(lf.mblim))
            if inner: O.vp8o_loop_filter_simple_bh(y, ci(20), ctypes.c_ubyte(lf.blim))
        else:                               # loopfilter.c:259-280
            if mbv: O.vp8o_loop_filter_mbv(y, du, du, ci(20), ci(24), ctypes.byref(lf))
            if inner: O.vp8o_loop_filter_bv(y, du, du, ci(20), ci(24), ctypes.byref(lf))
            if mbh: O.vp8o_loop_filter_mbh(y, du, du, ci(20), ci(24), ctypes.byref(lf))
            if inner: O.vp8o_loop_filter_bh(y, du, du, ci(20), ci(24), ctypes.byref(lf))
    bad = np.nonzero((got != want).reshape(n, -1).any(axis=1))[0]
    assert bad.size == 0, (bad[:8], par[bad[:4]])
    assert seed == 5 or (want != src).any()              # the filters did something


def test_intra4x4_predictors(L):
    O = oracle()
    rng = np.random.default_rng(7)
    n = 10 * 500 + 3
    mode = (np.arange(n) % 10).astype(np.uint8)
    ctx = rng.integers(0, 256, size=(n, 16)).astype(np.uint8)
    ctx[::7, :8] = ctx[::7, :1]             # flat edges among the random ones
    got = np.zeros((n, 16), np.uint8)
    assert L.vp8hip_lane_intra4x4(vp(mode.ctypes.data), vp(ctx.ctypes.data), vp(got.ctypes.data), n) == 0
    want = np.zeros((n, 16), np.uint8)
    for i in range(n):
        O.vp8o_intra4x4_predict(vp(ctx[i].ctypes.data), vp(ctx[i].ctypes.data + 8), ctypes.c_ubyte(int(ctx[i, 12])), ci(int(mode[i])),
                                vp(want[i].ctypes.data), ci(4))
    assert np.array_equal(got, want), np.nonzero((got != want).any(axis=1))[0][:8]


@pytest.mark.parametrize("big", [False, True])
def test_dequant_idct_add(L, big):
    """the transform on packed pairs, with coefficients up to +-2047 (the (short) truncation of the dequantiser and of the first pass)"""
    O = oracle()
    rng = np.random.default_rng(11 + big)
    n = 64 * 30 + 5
    mag = 2047 if big else 150
    q = rng.integers(-mag, mag + 1, size=(n, 16)).astype(np.int16)            # reference order: q[row * 4 + col]
    q[rng.random((n, 16)) < 0.5] = 0
    dq = np.stack([rng.integers(4, 158, size=n), rng.integers(4, 285, size=n)], axis=1).astype(np.int16)
    pred = rng.integers(0, 256, size=(n, 16)).astype(np.uint8)
    coef_ir = np.ascontiguousarray(q.reshape(n, 4, 4).transpose(0, 2, 1)).reshape(n, 16)   # IR order: coef[col * 4 + row]
    got = np.zeros((n, 16), np.uint8)
    assert L.vp8hip_lane_dequant_idct_add(vp(coef_ir.ctypes.data), vp(dq.ctypes.data), vp(pred.ctypes.data), vp(got.ctypes.data), n) == 0
    want = pred.copy()
    for i in range(n):
        dqv = np.full(16, dq[i, 1], np.int16); dqv[0] = dq[i, 0]
        qi = q[i].copy()
        O.vp8o_dequant_idct_add(vp(qi.ctypes.data), vp(dqv.ctypes.data), vp(want[i].ctypes.data), ci(4))
    assert np.array_equal(got, want), np.nonzero((got != want).any(axis=1))[0][:8]
