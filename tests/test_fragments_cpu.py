"""CPU: a frame handed to the feeder in fragments (vp8_parser_begin_frame_fragments -- what VPX_CODEC_USE_INPUT_FRAGMENTS
reaches, vp8/decoder/onyxd_if.c:336-366, decodframe.c:501-592) parses to exactly the IR of the whole frame, however the
partitions are grouped into fragments; a fragment cut short is reported the way the reference reports it."""
import ctypes

import numpy as np
import pytest

from vp8_testlib import ivf_path, load_package


def cuts_for_stream(P, frames):
    """partition cut offsets of every frame of a stream (the partition count comes from the feeder's header parse)."""
    ps = P.Parser()
    out = []
    for data in frames:
        hdr, _, _, _, _ = P.parse_to_numpy(ps, data)
        ps.swap(hdr)
        n = hdr.num_token_partitions
        key = not (data[0] & 1)
        first_len = (data[0] | (data[1] << 8) | (data[2] << 16)) >> 5
        sizes = 3 + (7 if key else 0) + first_len
        cuts = [0, sizes + 3 * (n - 1)]
        for i in range(n - 1):
            cuts.append(cuts[-1] + (data[sizes + 3 * i] | (data[sizes + 3 * i + 1] << 8) | (data[sizes + 3 * i + 2] << 16)))
        cuts.append(len(data))
        out.append(cuts)
    ps.close()
    return out


def _parse_fragments(P, parser, parts):
    L = P.load_host()
    L.vp8_parser_begin_frame_fragments.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    keep = [ctypes.create_string_buffer(p, len(p)) for p in parts]
    ptrs = (ctypes.c_void_p * len(parts))(*[ctypes.addressof(k) for k in keep])
    szs = (ctypes.c_size_t * len(parts))(*[len(p) for p in parts])
    hdr = P.FrameHdr()
    assert L.vp8_refs_get_free(ctypes.byref(parser.refs)) >= 0
    rc = L.vp8_parser_begin_frame_fragments(parser.p, ptrs, szs, len(parts), ctypes.byref(hdr))
    if rc:
        L.vp8_refs_release_new(ctypes.byref(parser.refs))
        return rc, None
    if parser.dims != (hdr.width, hdr.height):
        parser.dims = (hdr.width, hdr.height)
        L.vp8_refs_on_alloc(ctypes.byref(parser.refs))
    n = hdr.mb_cols * hdr.mb_rows
    mbs = np.zeros((n, 64), np.uint8); coef = np.zeros((n, 400), np.int16); mvs = np.zeros((n, 16, 2), np.int16)
    corrupt = parser.decode_mbs(mbs.ctypes.data, coef.ctypes.data, mvs.ctypes.data)
    parser.swap(hdr)
    return 0, (bytes(hdr), mbs, coef, mvs, corrupt)


@pytest.mark.parametrize("name", ["p_prof1_640x360", "p_split_352x288", "kf_640x360"])
@pytest.mark.parametrize("group", ["each", "header+rest", "whole"])
def test_fragments_parse_like_the_whole_frame(pkg, name, group):
    """The groupings the reference's unpacking supports (decodframe.c:523-571: fragment k is partition k unless an earlier
    fragment held several, which then take the following slots): one partition per fragment; header + first partition, then
    all token partitions in one fragment; everything in one fragment."""
    P = pkg
    _, _, frames = P.read_ivf(ivf_path(name))
    frames = frames[:6]
    cuts = cuts_for_stream(P, frames)
    whole, pieces = P.Parser(), P.Parser()
    for data, c in zip(frames, cuts):
        hdr, _, mbs, coef, mvs = P.parse_to_numpy(whole, data)
        whole.swap(hdr)
        parts = [data[a:b] for a, b in zip(c[:-1], c[1:])]
        if group == "header+rest":
            parts = [parts[0], b"".join(parts[1:])]
        elif group == "whole":
            parts = [data]
        rc, got = _parse_fragments(P, pieces, parts)
        assert rc == 0
        assert got[0] == bytes(hdr) and (got[1] == mbs).all() and (got[2] == coef).all() and (got[3] == mvs).all() and got[4] == 0
    whole.close(); pieces.close()


def test_a_short_fragment_is_an_error_not_a_shift(pkg):
    """A token partition that arrives shorter than the size table says: `Truncated packet or corrupt partition` (the
    reference's read_available_partition_size, decodframe.c:456-497) -- the following partitions are NOT read from
    shifted offsets, which is what handing the concatenation to the whole-frame entry point would do."""
    P = pkg
    _, _, frames = P.read_ivf(ivf_path("p_split_352x288"))
    c = cuts_for_stream(P, frames[:1])[0]
    data = frames[0]
    parts = [data[a:b] for a, b in zip(c[:-1], c[1:])]
    assert len(parts) >= 3
    parts[1] = parts[1][:-5]
    ps = P.Parser()
    rc, _ = _parse_fragments(P, ps, parts)
    assert rc == 7          # VPX_CODEC_CORRUPT_FRAME
    assert b"partition" in P.load_host().vp8_parser_error(ps.p)
    ps.close()
