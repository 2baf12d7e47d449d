"""CPU: the key-frame stream writer (tests/vp8_writer.py) against the two decoders that matter:
 * the host feeder must read back exactly the IR the stream was written from (modes, segments, sub-block modes, eobs,
   coefficients) -- a round trip through bool coder, mode trees, token trees and contexts;
 * the REAL reference decoder (oracle/_ref, built from /root/reference) must decode the stream to the frames the
   oracle produces from that IR.  This pins feeder AND oracle on content no encoder would choose: every mode
   everywhere, all segment / delta features, 1..8 token partitions, coefficients up to +-2047, odd sizes."""
import os
import subprocess

import numpy as np
import pytest

from vp8_testlib import ROOT, load_package, oracle_decode, synth_ir
from vp8_writer import write_ivf, write_key_frame

REF_MD5 = os.path.join(ROOT, "oracle", "_ref", "ref_md5")

CASES = [  # width, height, seed, log2 partitions, filter_type, dense, big coefficients, segmented
    (16, 16, 1, 0, 0, 0.5, False, True), (48, 32, 2, 1, 1, 0.4, False, True), (176, 144, 3, 2, 0, 0.3, True, True),
    (130, 98, 4, 3, 0, 0.3, False, False), (33, 200, 5, 0, 1, 0.2, True, True), (640, 368, 6, 2, 0, 0.08, False, True),
    (320, 16, 7, 3, 0, 0.6, True, False),
]


def _stream(case):
    w, h, seed, lp, ftype, dense, big, seg = case
    hdr, mbs, coef, mvs = synth_ir(w, h, seed, inter=False, filter_type=ftype, dense=dense, big=big, segmented=seg)
    hdr.num_token_partitions = 1 << lp
    return hdr, mbs, coef, mvs, write_key_frame(hdr, mbs, coef, log2_parts=lp)


@pytest.mark.parametrize("case", CASES)
def test_feeder_reads_back_the_ir(pkg, case):
    hdr, mbs, coef, mvs, data = _stream(case)
    parser = pkg.Parser()
    h2, _, m2, c2, _ = pkg.parse_to_numpy(parser, data)
    parser.close()
    for f in ("width", "height", "mb_cols", "mb_rows", "frame_type", "version", "filter_type", "filter_level",
              "sharpness_level", "segmentation_enabled", "mode_ref_lf_delta_enabled", "base_qindex", "y1dc_delta_q",
              "y2dc_delta_q", "y2ac_delta_q", "uvdc_delta_q", "uvac_delta_q", "num_token_partitions"):
        assert getattr(h2, f) == getattr(hdr, f), f
    if hdr.segmentation_enabled:
        assert h2.mb_segment_abs_delta == hdr.mb_segment_abs_delta
        assert list(h2.segment_quant) == list(hdr.segment_quant) and list(h2.segment_lf) == list(hdr.segment_lf)
        assert np.array_equal(m2[:, 4], mbs[:, 4])
    if hdr.mode_ref_lf_delta_enabled:
        assert list(h2.ref_lf_deltas) == list(hdr.ref_lf_deltas) and list(h2.mode_lf_deltas) == list(hdr.mode_lf_deltas)
    assert np.array_equal(m2[:, 0], mbs[:, 0]) and np.array_equal(m2[:, 1], mbs[:, 1])
    assert np.array_equal(m2[:, 3] & 1, mbs[:, 3] & 1)
    bp = mbs[:, 0] == 4
    assert np.array_equal(m2[bp, 40:56], mbs[bp, 40:56])
    live = (mbs[:, 3] & 1) == 0
    assert np.array_equal(m2[live, 8:33], mbs[live, 8:33])
    assert np.array_equal(c2[live], coef[live])


@pytest.mark.skipif(not os.path.exists(REF_MD5), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("case", CASES)
def test_reference_decodes_written_streams_like_the_oracle(pkg, case, tmp_path):
    hdr, mbs, coef, mvs, data = _stream(case)
    ivf, out = tmp_path / "s.ivf", tmp_path / "s.md5"
    write_ivf(ivf, hdr.width, hdr.height, [data, data])
    r = subprocess.run([REF_MD5, str(ivf), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ref = [l.split()[0] for l in open(out)]
    g = pkg.geom(hdr.width, hdr.height)
    buf = np.zeros(g.frame_size, np.uint8)
    oracle_decode(hdr, mbs, coef, mvs, buf, (None, None, None))
    mine = pkg.frame_md5(buf, g, hdr.width, hdr.height)
    assert ref == [mine, mine]
