"""CPU: token partitions decoded on several threads (vp8_parser_set_threads; vp8/decoder/decodframe.c:501-592 says they are
independent streams) give the serial feeder's IR: the dense arrays byte for byte, and sparse streams that expand to the
same coefficients (their entries are ordered by thread, which the per-macroblock indices account for)."""
import numpy as np
import pytest

from test_sparse_cpu import expand
from vp8_testlib import ivf_path

STREAMS = ["kf_8part_1920x1080", "p_split_352x288", "p_prof1_640x360", "p_prof3_640x360", "kf_640x360"]


@pytest.mark.parametrize("name", STREAMS)
@pytest.mark.parametrize("threads", [2, 3, 8])
def test_threaded_parse_equals_serial(pkg, name, threads):
    P = pkg
    _, _, frames = P.read_ivf(ivf_path(name))
    serial, dense_t, sparse_t = P.Parser(), P.Parser(), P.Parser()
    dense_t.set_threads(threads)
    sparse_t.set_threads(threads)
    for data in frames[:6]:
        hdr, _, mbs, coef, mvs = P.parse_to_numpy(serial, data)
        serial.swap(hdr)
        h1, _, mbs1, coef1, mvs1 = P.parse_to_numpy(dense_t, data)
        dense_t.swap(h1)
        assert bytes(h1) == bytes(hdr) and (mbs1 == mbs).all() and (coef1 == coef).all() and (mvs1 == mvs).all()
        h2, _ = sparse_t.begin(data)
        n = h2.mb_cols * h2.mb_rows
        mbs2 = np.zeros((n, 64), np.uint8)
        blocks = np.zeros((n * 25, 16), np.int16)
        dcs = np.zeros(n * 25, np.int16)
        mvs2 = np.zeros((n, 16, 2), np.int16)
        nb, nd, corrupt = sparse_t.decode_mbs_sparse(mbs2.ctypes.data, blocks.ctypes.data, n * 25, dcs.ctypes.data, mvs2.ctypes.data)
        sparse_t.swap(h2)
        assert corrupt == 0
        m2 = mbs2.copy(); m2[:, 56:64] = 0
        assert (m2 == mbs).all() and (mvs2 == mvs).all()
        coef2, nfull, ndc = expand(mbs2, blocks, dcs, n)
        live = (mbs[:, 3] & 1) == 0
        assert nfull == nb and ndc == nd
        assert (coef2[live] == coef[live]).all()
        # the streams are dense: every entry below the counts is used exactly once
        first = mbs2[:, 56:60].copy().view(np.uint32)[:, 0]
        assert first.max() <= nb and np.unique(first[live]).size <= live.sum()
    for p in (serial, dense_t, sparse_t):
        p.close()


def test_a_small_sparse_array_falls_back_to_the_serial_decode(pkg):
    P = pkg
    _, _, frames = P.read_ivf(ivf_path("p_split_352x288"))
    p = P.Parser()
    p.set_threads(4)
    h, _ = p.begin(frames[0])
    n = h.mb_cols * h.mb_rows
    mbs = np.zeros((n, 64), np.uint8)
    cap = n * 24                                  # one short of the worst case the threads need, plenty for this frame
    blocks = np.zeros((cap, 16), np.int16)
    dcs = np.zeros(n * 25, np.int16)
    mvs = np.zeros((n, 16, 2), np.int16)
    nb, nd, corrupt = p.decode_mbs_sparse(mbs.ctypes.data, blocks.ctypes.data, cap, dcs.ctypes.data, mvs.ctypes.data)
    first = mbs[:, 56:60].copy().view(np.uint32)[:, 0]
    assert corrupt == 0 and 0 < nb <= cap and (np.diff(first.astype(np.int64)) >= 0).all()     # serial order: monotonic
    p.close()
