"""CPU: host feeder logic -- headers, error behaviour (mirrors vp8_dx_iface.c / decodframe.c error
paths), reference-buffer bookkeeping."""
import ctypes

import numpy as np
import pytest

from vp8_testlib import ivf_path


def test_peek_and_header(pkg):
    w, h, frames = pkg.read_ivf(ivf_path("kf_640x360"))
    L = pkg.load_host()
    k, ww, hh = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    L.vp8_parser_peek.argtypes = [ctypes.c_char_p, ctypes.c_size_t] + [ctypes.c_void_p] * 3
    assert L.vp8_parser_peek(frames[0], len(frames[0]), ctypes.byref(k), ctypes.byref(ww), ctypes.byref(hh)) == 0
    assert (k.value, ww.value, hh.value) == (1, 640, 360)
    p = pkg.Parser()
    hdr, changed = p.begin(frames[0])
    assert changed and (hdr.width, hdr.height, hdr.mb_cols, hdr.mb_rows) == (640, 360, 40, 23)
    assert hdr.frame_type == 0 and hdr.refresh_golden == 1 and hdr.refresh_alt == 1 and hdr.refresh_last == 1
    p.close()


def test_stream_must_start_with_key_frame(pkg):
    w, h, frames = pkg.read_ivf(ivf_path("p_odd_130x98"))
    p = pkg.Parser()
    with pytest.raises(ValueError) as e:
        p.begin(frames[1])                      # an inter frame first: VPX_CODEC_CORRUPT_FRAME (7)
    assert "error 7" in str(e.value)
    p.close()


def test_truncated_and_garbage(pkg):
    w, h, frames = pkg.read_ivf(ivf_path("kf_odd_67x45"))
    p = pkg.Parser()
    with pytest.raises(ValueError):
        p.begin(frames[0][:2])                  # "Truncated packet"
    bad = bytearray(frames[0]); bad[3] = 0      # break the 9d 01 2a sync code: VPX_CODEC_UNSUP_BITSTREAM (5)
    with pytest.raises(ValueError) as e:
        p.begin(bytes(bad))
    assert "error 5" in str(e.value)
    # A key frame cut in the middle of its token partition parses (zeros are read past the end, as in
    # the reference's bool decoder) but, being the first frame, is rejected like the reference does
    # (decodframe.c:1143-1151 "A stream must start with a complete key frame").
    hdr, changed = p.begin(frames[0][: len(frames[0]) // 2])
    n = hdr.mb_cols * hdr.mb_rows
    mbs = np.zeros((n, 64), np.uint8); coef = np.zeros((n, 400), np.int16)
    with pytest.raises(ValueError) as e:
        p.decode_mbs(mbs.ctypes.data, coef.ctypes.data, None)
    assert "error 7" in str(e.value)
    # after a good key frame, a truncated frame decodes and is only flagged corrupt
    hdr, changed = p.begin(frames[0])
    assert p.decode_mbs(mbs.ctypes.data, coef.ctypes.data, None) == 0
    p.swap(hdr)
    hdr, changed = p.begin(frames[1][: len(frames[1]) // 2])
    assert p.decode_mbs(mbs.ctypes.data, coef.ctypes.data, None) == 1
    p.close()


def test_ref_bookkeeping_matches_reference_sequence(pkg):
    """swap_frame_buffers semantics (onyxd_if.c:261-316) over a stream with golden refreshes."""
    w, h, frames = pkg.read_ivf(ivf_path("p_arf_176x144"))
    p = pkg.Parser()
    for i, data in enumerate(frames):
        hdr, changed, mbs, coef, mvs = pkg.parse_to_numpy(p, data)
        r = p.refs
        assert r.new_idx not in (r.lst_idx, r.gld_idx, r.alt_idx) or i == 0
        p.swap(hdr)
        r = p.refs
        assert sum(r.ref_cnt) == 3 and all(c >= 0 for c in r.ref_cnt)
        assert r.ref_cnt[r.lst_idx] >= 1 and r.ref_cnt[r.gld_idx] >= 1 and r.ref_cnt[r.alt_idx] >= 1
        if hdr.frame_type == 0:
            assert r.lst_idx == r.gld_idx == r.alt_idx
    p.close()
