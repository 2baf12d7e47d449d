"""CPU: the DEVICE FORM of the IR (include/vp8_ir.h: vp8ir_mbx records + a block stream; what an IR slot holds in HBM and the
kernels read) as the feeder writes it (vp8_parser_decode_mbs_compact) says exactly what the dense arrays of the plain parse say,
for every fixture -- expanded by the form's rule restated in numpy (P.dense_from_compact), and byte for byte what the header's
own vp8ir_compact_mb makes of the dense arrays (P.compact_from_dense restates that); and it is as much smaller as DESIGN.md
says."""
import numpy as np
import pytest

from vp8_testlib import FIXTURES, ivf_path


@pytest.mark.parametrize("name", FIXTURES)
def test_device_form_says_what_the_dense_arrays_say(pkg, name):
    P = pkg
    _, _, frames = P.read_ivf(ivf_path(name))
    dense_p, compact_p = P.Parser(), P.Parser()
    for data in frames[:5]:
        hdr, _, mbs, coef, mvs = P.parse_to_numpy(dense_p, data)
        dense_p.swap(hdr)
        h2, mbx, blocks, mvs2, corrupt = P.parse_to_numpy_compact(compact_p, data)
        compact_p.swap(h2)
        assert bytes(h2) == bytes(hdr) and corrupt == 0 and (mvs2 == mvs).all()
        m2, got = P.dense_from_compact(mbx, blocks)
        assert (m2 == mbs).all()
        skip = (mbs[:, 3] & 1) != 0            # skipped macroblocks: dense contents are undefined
        assert (got[~skip] == coef[~skip]).all()
        # the stream holds the blocks with eob > 1 and nothing else, a row's one after the other
        kind = P.block_kinds(mbs)
        assert blocks.shape[0] == int((kind[:, :24] == 2).sum())
        first = mbx[:, 56:60].copy().view(np.uint32)[:, 0].astype(np.int64)
        cnt = (kind[:, :24] == 2).sum(1)
        cols = hdr.mb_cols
        for r in range(hdr.mb_rows):
            f, c = first[r * cols:(r + 1) * cols], cnt[r * cols:(r + 1) * cols]
            assert (f[1:] == f[:-1] + c[:-1]).all()
        # ... and the record is what vp8ir_compact_mb makes of the dense arrays (skipped macroblocks' coefficients zeroed first)
        c0 = coef.copy(); c0[skip] = 0
        want_mbx, want_blocks = P.compact_from_dense(mbs, c0)
        assert (want_mbx == mbx).all() and (want_blocks == blocks).all()
    dense_p.close(); compact_p.close()


def test_header_helpers_agree_with_the_numpy_restatement(pkg):
    """vp8ir_compact_mb / vp8ir_expand_mb of include/vp8_ir.h (through the host library's vp8ir_compact_frame /
    vp8ir_expand_frame) against P.compact_from_dense / P.dense_from_compact on random IR."""
    import ctypes
    O = pkg.load_host()
    O.vp8ir_compact_frame.restype = ctypes.c_size_t
    O.vp8ir_compact_frame.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    O.vp8ir_expand_frame.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rng = np.random.default_rng(5)
    n = 300
    mbs = np.zeros((n, 64), np.uint8)
    mbs[:, 0] = rng.integers(0, 10, n)                       # y_mode: all of them, B_PRED and SPLITMV included
    mbs[:, 3] = rng.integers(0, 2, n)                        # skip flag
    mbs[:, 8:33] = rng.choice([0, 0, 1, 1, 2, 5, 16], (n, 25))
    has_y2 = (mbs[:, 0] != 4) & (mbs[:, 0] != 9)
    mbs[has_y2, 8:24] = np.maximum(mbs[has_y2, 8:24], 1)     # luma blocks behind a Y2 block start at position 1
    coef = rng.integers(-2047, 2048, (n, 400)).astype(np.int16)
    kind = pkg.block_kinds(mbs)
    c = coef.reshape(n, 25, 16)
    c[kind == 0] = 0
    c[kind == 1, 1:] = 0
    c[:, :16, 0][(kind[:, :16] == 2) & has_y2[:, None]] = 0
    want_mbx, want_blocks = pkg.compact_from_dense(mbs, coef)
    got_mbx = np.zeros((n, 128), np.uint8)
    got_blocks = np.zeros((n * 24, 16), np.int16)
    nb = O.vp8ir_compact_frame(mbs.ctypes.data, coef.ctypes.data, n, got_mbx.ctypes.data, got_blocks.ctypes.data)
    assert nb == want_blocks.shape[0]
    assert (got_mbx == want_mbx).all() and (got_blocks[:nb] == want_blocks).all()
    back_mbs = np.zeros((n, 64), np.uint8)
    back = np.zeros((n, 400), np.int16)
    O.vp8ir_expand_frame(got_mbx.ctypes.data, got_blocks.ctypes.data, n, back_mbs.ctypes.data, back.ctypes.data)
    assert (back_mbs == mbs).all() and (back == coef).all()
    m2, c2 = pkg.dense_from_compact(want_mbx, want_blocks)
    assert (m2 == mbs).all() and (c2 == coef).all()


def test_the_device_form_is_much_smaller_on_the_benchmark_stream(pkg):
    P = pkg
    _, _, frames = P.read_ivf(ivf_path("kf_1920x1080"))
    ps = P.Parser()
    tot = 0
    for data in frames:
        hdr, mbx, blocks, _, _ = P.parse_to_numpy_compact(ps, data)
        ps.swap(hdr)
        tot += mbx.nbytes + blocks.nbytes
    ps.close()
    per_frame = tot / len(frames)
    assert per_frame < 0.45 * (8160 * 864)     # records + block stream vs descriptors + dense coefficients
