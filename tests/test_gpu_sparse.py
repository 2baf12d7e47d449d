"""GPU (-m gpu): the sparse coefficient upload (include/vp8_ir.h; vp8hip_ir_map_sparse / vp8hip_ir_upload_sparse: feeder ->
blocks + DCs -> expansion on the device) gives the frames the dense upload gives: reference MD5s, key and inter frames,
dense and sparse content, tiny frames."""
import pytest

from vp8_testlib import golden_md5, ivf_path

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["kf_q0_176x144", "kf_odd_67x45", "kf_640x360", "p_split_352x288", "p_prof1_640x360",
                                  "p_odd_130x98", "p_arf_176x144", "p_dense_1920x1080"])
def test_sparse_upload_decodes_to_the_reference_frames(pkg, name):
    w, h, frames = pkg.read_ivf(ivf_path(name))
    gold = golden_md5(name)
    parser, ctx = pkg.Parser(), pkg.Vp8Hip(0)
    got, total = [], 0
    try:
        ctx.configure(w, h, 4, 1)
        for data in frames[:30]:
            hdr, nbytes = ctx.parse_into_slot_sparse(parser, data, 0)
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
