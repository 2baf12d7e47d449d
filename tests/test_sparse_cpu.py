"""CPU: the sparse coefficient stream the feeder can emit (include/vp8_ir.h; vp8_parser_decode_mbs_sparse) expands -- by the
rule the device kernel uses, restated here in numpy -- to exactly the dense coefficient array of the plain parse, for every
fixture; and it is as much smaller as DESIGN.md says."""
import numpy as np
import pytest

from vp8_testlib import FIXTURES, ivf_path


def expand(mbs, blocks, dcs, nmb):
    """vp8_ir_expand_kernel / vp8ir_block_kind restated: where a block's coefficients are follows from the descriptor."""
    coef = np.zeros((nmb, 400), np.int16)
    y_mode, flags, eobs = mbs[:, 0], mbs[:, 3], mbs[:, 8:33].astype(int)
    first = mbs[:, 56:60].copy().view(np.uint32)[:, 0]
    dfirst = mbs[:, 60:64].copy().view(np.uint32)[:, 0]
    has_y2 = (y_mode != 4) & (y_mode != 9)
    live = ((flags & 1) == 0)[:, None] & np.concatenate([np.ones((nmb, 24), bool), has_y2[:, None]], axis=1)
    full = (eobs > 1) & live
    dc = (eobs == 1) & live
    dc[:, :16] &= ~has_y2[:, None]
    for kind, src, base in ((full, blocks, first), (dc, None, dfirst)):
        rank = np.cumsum(kind, axis=1) - kind
        mb_i, blk = np.nonzero(kind)
        at = base[mb_i] + rank[mb_i, blk]
        if src is not None:
            coef.reshape(nmb, 25, 16)[mb_i, blk] = src[at]
        else:
            coef.reshape(nmb, 25, 16)[mb_i, blk, 0] = dcs[at]
    return coef, int(full.sum()), int(dc.sum())


@pytest.mark.parametrize("name", FIXTURES)
def test_sparse_streams_expand_to_the_dense_array(pkg, name):
    P = pkg
    _, _, frames = P.read_ivf(ivf_path(name))
    dense_p, sparse_p = P.Parser(), P.Parser()
    for data in frames[:5]:
        hdr, _, mbs, coef, mvs = P.parse_to_numpy(dense_p, data)
        dense_p.swap(hdr)
        h2, _ = sparse_p.begin(data)
        n = h2.mb_cols * h2.mb_rows
        mbs2 = np.zeros((n, 64), np.uint8)
        blocks = np.zeros((n * 25, 16), np.int16)
        dcs = np.zeros(n * 25, np.int16)
        mvs2 = np.zeros((n, 16, 2), np.int16)
        nb, nd, corrupt = sparse_p.decode_mbs_sparse(mbs2.ctypes.data, blocks.ctypes.data, n * 25, dcs.ctypes.data, mvs2.ctypes.data)
        sparse_p.swap(h2)
        assert bytes(h2) == bytes(hdr) and corrupt == 0
        m2 = mbs2.copy(); m2[:, 56:64] = 0
        assert (m2 == mbs).all() and (mvs2 == mvs).all()
        got, nfull, ndc = expand(mbs2, blocks, dcs, n)
        assert (nfull, ndc) == (nb, nd)
        skip = (mbs[:, 3] & 1) != 0            # skipped macroblocks: dense contents are undefined
        assert (got[~skip] == coef[~skip]).all()
    dense_p.close(); sparse_p.close()


def test_sparse_streams_are_much_smaller_on_the_benchmark_stream(pkg):
    P = pkg
    _, _, frames = P.read_ivf(ivf_path("kf_1920x1080"))
    ps = P.Parser()
    tot = 0
    for data in frames:
        hdr, _ = ps.begin(data)
        n = hdr.mb_cols * hdr.mb_rows
        mbs = np.zeros((n, 64), np.uint8); blocks = np.zeros((n * 25, 16), np.int16); dcs = np.zeros(n * 25, np.int16)
        nb, nd, _ = ps.decode_mbs_sparse(mbs.ctypes.data, blocks.ctypes.data, n * 25, dcs.ctypes.data, None)
        ps.swap(hdr)
        tot += nb * 32 + nd * 2
    ps.close()
    per_frame = tot / len(frames) + 8160 * 64
    assert per_frame < 0.40 * (8160 * 864)     # descriptors + sparse streams vs descriptors + dense coefficients
