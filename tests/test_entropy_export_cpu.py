"""CPU: vp8_parser_export_entropy (csrc/host/vp8_parser.h) -- what the host's header parse hands to the device's entropy decoder.
The exported first-partition state is continued here by a restatement of the device's 32-bit bool decoder in Python (three bytes
per refill, vp8_entropy.hip) through the key-frame mode syntax (vp8_kfread_modes, decodemv.c:50-173): the modes that come out are
the host feeder's, on fixtures whose header ends at every kind of window fill."""
import numpy as np
import pytest

from vp8_testlib import ivf_path, load_package


class Bool32:
    def __init__(self, data, ef):
        self.d, self.value, self.bits, self.range = data, ef.first_value, ef.first_bits, ef.first_range
        self.pos, self.end = ef.first_pos, ef.first_end

    def get(self, prob):
        split = 1 + (((self.range - 1) * prob) >> 8)
        if self.bits < 0:
            nxt = 0
            for k in range(3):
                nxt = nxt << 8 | (self.d[self.pos + k] if self.pos + k < self.end else 0)
            self.value |= nxt << -self.bits
            self.bits += 24
            self.pos += 3
        big = split << 24
        bit = self.value >= big
        if bit:
            self.value -= big
            self.range -= split
        else:
            self.range = split
        shift = 8 - self.range.bit_length()
        self.range <<= shift
        self.value = (self.value << shift) & 0xffffffff
        self.bits -= shift
        return int(bit)


def _bmode(b, pr):
    if not b.get(pr[0]): return 0
    if not b.get(pr[1]): return 1
    if not b.get(pr[2]): return 2
    if not b.get(pr[3]):
        if not b.get(pr[4]): return 3
        return 6 if b.get(pr[5]) else 5
    if not b.get(pr[6]): return 4
    if not b.get(pr[7]): return 7
    return 9 if b.get(pr[8]) else 8


@pytest.mark.parametrize("name", ["kf_odd_67x45", "kf_q0_176x144", "kf_640x360"])
def test_exported_state_continues_into_the_modes(name):
    P = load_package()
    H = P.load_host()
    import ctypes
    kfb = (ctypes.c_uint8 * 900).in_dll(H, "vp8t_kf_bmode_probs")
    _, _, frames = P.read_ivf(ivf_path(name))
    ph, pd = P.Parser(), P.Parser()
    for data in frames[:3]:
        hdr, _, mbs, _, _ = P.parse_to_numpy(ph, data)
        ph.swap(hdr)
        h2, _ = pd.begin(data)
        ef = pd.export_entropy()
        pd.swap(h2)
        assert ef is not None and ef.first_range in range(128, 256) and -8 <= ef.first_bits <= 24
        assert ef.first_value & ((1 << max(0, 24 - max(ef.first_bits, 0))) - 1) == 0 or ef.first_bits < 0
        assert ef.num_tok == hdr.num_token_partitions and ef.tok_end[ef.num_tok - 1] == len(data)
        assert bytes(ef.coef_probs) != bytes(1056)
        b = Bool32(data, ef)
        cols, rows = hdr.mb_cols, hdr.mb_rows
        above = [[0] * 4 for _ in range(cols)]
        for r in range(rows):
            left = [0] * 4
            for c in range(cols):
                m = mbs[r * cols + c]
                seg = 0
                if ef.update_mb_segmentation_map:
                    tp = ef.segment_tree_probs
                    seg = 2 + b.get(tp[2]) if b.get(tp[0]) else b.get(tp[1])
                skip = b.get(ef.prob_skip_false) if ef.mb_no_coeff_skip else 0
                if not b.get(145): ym = 4
                elif not b.get(156): ym = 1 if b.get(163) else 0
                else: ym = 3 if b.get(128) else 2
                assert (ym, seg) == (int(m[0]), int(m[4])), (name, r, c)
                if not int(m[3]) & 1:
                    assert skip == 0           # (the IR's flag is also set for coded macroblocks that turned out empty)
                if ym == 4:
                    bm = []
                    for i in range(16):
                        A = above[c][i] if i < 4 else bm[i - 4]
                        L = left[i >> 2] if (i & 3) == 0 else bm[i - 1]
                        bm.append(_bmode(b, kfb[(A * 10 + L) * 9:(A * 10 + L) * 9 + 9]))
                    assert bm == [int(x) for x in m[40:56]], (name, r, c)
                    above[c], left = bm[12:], bm[3::4]
                else:
                    im = {1: 2, 2: 3, 3: 1}.get(ym, 0)
                    above[c], left = [im] * 4, [im] * 4
                if not b.get(142): uv = 0
                elif not b.get(114): uv = 1
                else: uv = 3 if b.get(183) else 2
                assert uv == int(m[1]), (name, r, c)
    ph.close(); pd.close()


def test_export_goes_through_a_stream_and_refuses_what_the_device_does_not_decode():
    P = load_package()
    _, _, frames = P.read_ivf(ivf_path("p_lowrate_640x360"))
    p = P.Parser()
    for i, data in enumerate(frames[:4]):        # key frame, then inter frames: only the headers are read here
        hdr, _ = p.begin(data)
        ef = p.export_entropy()
        assert ef is not None and ef.hdr.frame_type == (0 if i == 0 else 1)
        if i:
            assert ef.prob_intra and bytes(ef.mvc) != bytes(38) and bytes(ef.ymode_prob) != bytes(4)
        p.swap(hdr)
    p.close()
    p = P.Parser()                               # with concealment the frames stay with the host feeder
    p.set_error_concealment(True)
    hdr, _ = p.begin(frames[0])
    assert p.export_entropy() is None
    n = hdr.mb_cols * hdr.mb_rows
    mbs, coef, mvs = np.zeros((n, 64), np.uint8), np.zeros((n, 400), np.int16), np.zeros((n, 16, 2), np.int16)
    p.decode_mbs(mbs.ctypes.data, coef.ctypes.data, mvs.ctypes.data)
    p.close()
