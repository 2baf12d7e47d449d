"""CPU: token partitions decoded on several threads (vp8_parser_set_threads; vp8/decoder/decodframe.c:501-592 says they are
independent streams) give the serial feeder's IR: the dense arrays byte for byte, and a device form (include/vp8_ir.h) that
expands to the same coefficients (its rows stand in the block stream thread by thread, which the form allows: a row's blocks
stay together and are found through its first macroblock's sparse_first)."""
import numpy as np
import pytest

from vp8_testlib import ivf_path

STREAMS = ["kf_8part_1920x1080", "p_split_352x288", "p_prof1_640x360", "p_prof3_640x360", "kf_640x360"]


@pytest.mark.parametrize("name", STREAMS)
@pytest.mark.parametrize("threads", [2, 3, 8])
def test_threaded_parse_equals_serial(pkg, name, threads):
    P = pkg
    _, _, frames = P.read_ivf(ivf_path(name))
    serial, dense_t, compact_t = P.Parser(), P.Parser(), P.Parser()
    dense_t.set_threads(threads)
    compact_t.set_threads(threads)
    for data in frames[:6]:
        hdr, _, mbs, coef, mvs = P.parse_to_numpy(serial, data)
        serial.swap(hdr)
        h1, _, mbs1, coef1, mvs1 = P.parse_to_numpy(dense_t, data)
        dense_t.swap(h1)
        assert bytes(h1) == bytes(hdr) and (mbs1 == mbs).all() and (coef1 == coef).all() and (mvs1 == mvs).all()
        h2, mbx, blocks, mvs2, corrupt = P.parse_to_numpy_compact(compact_t, data)
        compact_t.swap(h2)
        assert corrupt == 0 and (mvs2 == mvs).all()
        m2, coef2 = P.dense_from_compact(mbx, blocks)
        assert (m2 == mbs).all()
        live = (mbs[:, 3] & 1) == 0
        assert (coef2[live] == coef[live]).all()
        # the stream has no gaps: every block below the count belongs to exactly one macroblock, a row's follow each other
        kind = P.block_kinds(mbs)
        cnt = (kind[:, :24] == 2).sum(1)
        first = mbx[:, 56:60].copy().view(np.uint32)[:, 0].astype(np.int64)
        assert blocks.shape[0] == cnt.sum()
        used = np.zeros(blocks.shape[0] + 1, np.int64)
        np.add.at(used, first, 1 * (cnt > 0)); np.add.at(used, first + cnt, -1 * (cnt > 0))
        assert (np.cumsum(used)[:-1] == 1).all()
        cols = hdr.mb_cols
        for r in range(hdr.mb_rows):
            f, c = first[r * cols:(r + 1) * cols], cnt[r * cols:(r + 1) * cols]
            assert (f[1:] == f[:-1] + c[:-1]).all()
    for p in (serial, dense_t, compact_t):
        p.close()


def test_a_small_block_array_falls_back_to_the_serial_decode(pkg):
    P = pkg
    _, _, frames = P.read_ivf(ivf_path("p_split_352x288"))
    p = P.Parser()
    p.set_threads(4)
    h, _ = p.begin(frames[0])
    n = h.mb_cols * h.mb_rows
    mbx = np.zeros((n, 128), np.uint8)
    cap = n * 24 - 1                              # one short of the worst case the threads need, plenty for this frame
    blocks = np.zeros((cap, 16), np.int16)
    mvs = np.zeros((n, 16, 2), np.int16)
    nb, corrupt = p.decode_mbs_compact(mbx.ctypes.data, blocks.ctypes.data, cap, mvs.ctypes.data)
    first = mbx[:, 56:60].copy().view(np.uint32)[:, 0]
    assert corrupt == 0 and 0 < nb <= cap and (np.diff(first.astype(np.int64)) >= 0).all()     # serial order: monotonic
    p.close()
