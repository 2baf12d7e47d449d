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
from vp8_writer import write_inter_frame, write_ivf, write_key_frame

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


# ---- inter frames: every inter mode, all three references with sign biases, split vectors, and the segment map's three lives
# (coded, kept with new data, kept untouched) -- content the reference ENCODER never produces (it codes no map on inter frames)
def inter_sequence(w, h, seed, plan, lp=0, dense=0.25, big=False):
    """A key frame and one inter frame per entry of plan ("keep": segmentation on, nothing coded; "data": new segment data, map
    kept; "map": new map, data kept; "both"; "off": segmentation off).  Returns the frames and, per frame, the IR the stream was
    written from as a decoder is to read it back: (hdr, mbs, coef, mvs, segment map known)."""
    rng = np.random.default_rng(seed)
    hdr, mbs, coef, mvs = synth_ir(w, h, seed, inter=False, dense=dense, big=big, segmented=True)
    hdr.num_token_partitions = 1 << lp
    frames, expect = [write_key_frame(hdr, mbs, coef, log2_parts=lp)], [(hdr, mbs, coef, mvs, True)]
    segmap, absd = mbs[:, 4].copy(), hdr.mb_segment_abs_delta
    segq, seglf = list(hdr.segment_quant), list(hdr.segment_lf)
    known = True
    for i, step in enumerate(plan):
        hdr, mbs, coef, mvs = synth_ir(w, h, seed * 100 + i + 1, inter=True, dense=dense, big=big, segmented=step != "off")
        hdr.num_token_partitions = 1 << lp
        hdr.refresh_last = int(rng.integers(0, 4) > 0)
        hdr.refresh_golden, hdr.refresh_alt = int(rng.integers(0, 3) == 0), int(rng.integers(0, 3) == 0)
        hdr.copy_buffer_to_gf = 0 if hdr.refresh_golden else int(rng.integers(0, 3))
        hdr.copy_buffer_to_arf = 0 if hdr.refresh_alt else int(rng.integers(0, 3))
        hdr.sign_bias_golden, hdr.sign_bias_alt = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        hdr.show_frame = int(rng.integers(0, 5) > 0)
        um = step in ("map", "both")
        ud = step in ("data", "both")
        if step != "off":
            if not um:
                mbs[:, 4] = segmap
            if not ud:
                hdr.mb_segment_abs_delta = absd
                for k in range(4):
                    hdr.segment_quant[k], hdr.segment_lf[k] = segq[k], seglf[k]
        data, mbs2, mvs2 = write_inter_frame(hdr, mbs, coef, mvs, log2_parts=lp, update_map=um, update_data=ud)
        if um:
            segmap, known = mbs[:, 4].copy(), True
        if ud:
            absd, segq, seglf = hdr.mb_segment_abs_delta, list(hdr.segment_quant), list(hdr.segment_lf)
        if step == "off":
            known = False                               # (what a later frame that keeps "the" map sees is the decoder's business)
        frames.append(data)
        expect.append((hdr, mbs2, coef, mvs2, known))
    return frames, expect


INTER_CASES = [  # width, height, seed, plan, log2 partitions, big coefficients
    (64, 48, 11, ("keep", "data", "keep", "map", "keep", "both", "keep"), 0, False),
    (176, 144, 12, ("keep", "keep", "data", "off", "both", "keep"), 1, False),
    (130, 98, 13, ("both", "keep", "map", "data"), 2, True),
    (33, 200, 14, ("off", "off", "both", "keep"), 0, False),
    (320, 192, 15, ("keep", "data", "map"), 3, False),
]


def _feed(pkg, frames):
    """feeder over the frames -> per frame (hdr, mbs, coef, mvs) and the frame-buffer indices before / after"""
    parser = pkg.Parser()
    out = []
    for data in frames:
        hdr, changed, mbs, coef, mvs = pkg.parse_to_numpy(parser, data)
        r = parser.refs
        idx = (r.new_idx, r.lst_idx, r.gld_idx, r.alt_idx)
        parser.swap(hdr)
        out.append((hdr, mbs, coef, mvs, idx, parser.refs.show_idx))
    parser.close()
    return out


@pytest.mark.parametrize("case", INTER_CASES)
def test_feeder_reads_back_inter_frames(pkg, case):
    w, h, seed, plan, lp, big = case
    frames, expect = inter_sequence(w, h, seed, plan, lp, big=big)
    got = _feed(pkg, frames)
    for k, ((hdr, mbs, coef, mvs, known), (h2, m2, c2, v2, _, _)) in enumerate(zip(expect, got)):
        for f in ("frame_type", "show_frame", "filter_type", "filter_level", "sharpness_level", "segmentation_enabled", "base_qindex",
                  "refresh_last", "refresh_golden", "refresh_alt", "copy_buffer_to_gf", "copy_buffer_to_arf", "sign_bias_golden",
                  "sign_bias_alt", "num_token_partitions"):
            if k == 0 and f.startswith(("refresh", "copy", "sign")):
                continue
            assert getattr(h2, f) == getattr(hdr, f), (k, f)
        if hdr.segmentation_enabled:
            assert h2.mb_segment_abs_delta == hdr.mb_segment_abs_delta, k
            assert list(h2.segment_quant) == list(hdr.segment_quant) and list(h2.segment_lf) == list(hdr.segment_lf), k
            if known:
                assert np.array_equal(m2[:, 4], mbs[:, 4]), k
        assert np.array_equal(m2[:, 0], mbs[:, 0]) and np.array_equal(m2[:, 2], mbs[:, 2]), k
        intra = mbs[:, 2] == 0
        assert np.array_equal(m2[intra, 1], mbs[intra, 1]), k
        assert np.array_equal(m2[:, 3] & 3, mbs[:, 3] & 3), k
        split = mbs[:, 0] == 9
        assert np.array_equal(m2[split, 5], mbs[split, 5]), k
        assert np.array_equal(v2, mvs), k
        live = (mbs[:, 3] & 1) == 0
        assert np.array_equal(m2[live, 8:33], mbs[live, 8:33]) and np.array_equal(c2[live], coef[live]), k


@pytest.mark.skipif(not os.path.exists(REF_MD5), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("case", INTER_CASES)
def test_reference_decodes_written_inter_streams_like_the_oracle(pkg, case, tmp_path):
    w, h, seed, plan, lp, big = case
    frames, _ = inter_sequence(w, h, seed, plan, lp, big=big)
    ivf, out = tmp_path / "s.ivf", tmp_path / "s.md5"
    write_ivf(ivf, w, h, frames)
    r = subprocess.run([REF_MD5, str(ivf), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ref = [l.split()[0] for l in open(out)]
    g = pkg.geom(w, h)
    bufs = [np.zeros(g.frame_size, np.uint8) for _ in range(4)]
    mine = []
    for hdr, mbs, coef, mvs, (new, lst, gld, alt), show in _feed(pkg, frames):
        oracle_decode(hdr, mbs, coef, mvs, bufs[new], (bufs[lst], bufs[gld], bufs[alt]))
        if hdr.show_frame:
            mine.append(pkg.frame_md5(bufs[show], g, w, h))
    assert mine == ref


def fuzz_case(seed):
    """a small inter sequence from a seed: size, plan, partitions, coefficient range"""
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(1, 14)) * 16 + int(rng.integers(-15, 1)), int(rng.integers(1, 11)) * 16 + int(rng.integers(-15, 1))
    w, h = max(w, 2), max(h, 2)
    plan = tuple(rng.choice(["keep", "data", "map", "both", "off"], size=int(rng.integers(2, 7))))
    return w, h, seed, plan, int(rng.integers(0, 4)), bool(rng.integers(0, 2))


FUZZ_SEEDS = list(range(100, 124))


@pytest.mark.skipif(not os.path.exists(REF_MD5), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("seed", FUZZ_SEEDS)
def test_reference_decodes_fuzzed_inter_streams_like_the_oracle(pkg, seed, tmp_path):
    test_reference_decodes_written_inter_streams_like_the_oracle(pkg, fuzz_case(seed), tmp_path)


@pytest.mark.parametrize("seed", FUZZ_SEEDS[:8])
def test_feeder_reads_back_fuzzed_inter_frames(pkg, seed):
    test_feeder_reads_back_inter_frames(pkg, fuzz_case(seed))
