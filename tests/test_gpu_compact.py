"""GPU (-m gpu): the feeder's device-form output (include/vp8_ir.h; vp8_parser_decode_mbs_compact into the pinned staging of a
slot, vp8hip_ir_upload_compact: one copy, nothing in between) gives the frames the dense upload gives: reference MD5s, key and
inter frames, dense and sparse content, tiny frames; and a slot reads back (vp8hip_ir_fetch) as the dense arrays of the plain
parse."""
import numpy as np
import pytest

from vp8_testlib import golden_md5, ivf_path

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["kf_q0_176x144", "kf_odd_67x45", "kf_640x360", "p_split_352x288", "p_prof1_640x360",
                                  "p_odd_130x98", "p_arf_176x144", "p_dense_1920x1080"])
def test_compact_upload_decodes_to_the_reference_frames(pkg, name):
    w, h, frames = pkg.read_ivf(ivf_path(name))
    gold = golden_md5(name)
    parser, ctx = pkg.Parser(), pkg.Vp8Hip(0)
    got, total = [], 0
    try:
        ctx.configure(w, h, 4, 1)
        for data in frames[:30]:
            ctx.sync()                           # (the staging is rewritten below: the copy of the frame before has run)
            hdr, nbytes = ctx.parse_into_slot_compact(parser, data, 0)
            total += nbytes
            r = parser.refs
            ctx.decode([(0, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx))], pkg.STAGE_ALL)
            parser.swap(hdr)
            if hdr.show_frame:
                got.append(pkg.planes_md5(*ctx.download_planes(parser.refs.show_idx)))
            else:
                ctx.sync()
    finally:
        ctx.close()
        parser.close()
    assert got == gold[:len(got)] and len(got) >= 3
    assert total < 30 * ctx.nmb * 800


@pytest.mark.parametrize("name", ["kf_640x360", "p_split_352x288"])
def test_slot_round_trip(pkg, name):
    """dense arrays -> vp8hip_ir_upload (converted on the host) -> device -> vp8hip_ir_fetch: the same arrays (zeros where a block
    has no coefficients); the same through the feeder's own device-form output and vp8hip_ir_copy."""
    w, h, frames = pkg.read_ivf(ivf_path(name))
    p1, p2, ctx = pkg.Parser(), pkg.Parser(), pkg.Vp8Hip(0)
    try:
        ctx.configure(w, h, 1, 3)
        for data in frames[:3]:
            hdr, _, mbs, coef, mvs = pkg.parse_to_numpy(p1, data)
            p1.swap(hdr)
            ctx.fill_slot(0, hdr, mbs, coef, mvs)
            ctx.sync()
            h2, _ = ctx.parse_into_slot_compact(p2, data, 1)
            p2.swap(h2)
            ctx.ir_copy(2, 1)
            want = coef.copy()
            skip = (mbs[:, 3] & 1) != 0
            want[skip] = 0
            kind = pkg.block_kinds(mbs)
            for slot in (0, 1, 2):
                gm, gc = ctx.ir_fetch(slot)
                assert (gm == mbs).all(), slot
                assert (gc == want).all(), slot
            assert kind.any()
    finally:
        ctx.close(); p1.close(); p2.close()
