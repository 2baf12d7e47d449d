"""CPU: host feeder + oracle reproduce the REAL reference decoder's per-frame MD5s on every fixture.
This is what pins the oracle (and the feeder) to the reference: tests/golden/*.md5 were printed by the
reference's own decoder (oracle/_ref, built from /root/reference) -- see tests/golden/make_fixtures.py."""
import pytest

from vp8_testlib import FIXTURES, golden_md5, oracle_decode_ivf


@pytest.mark.parametrize("name", FIXTURES)
def test_fixture_md5(name):
    assert oracle_decode_ivf(name) == golden_md5(name)


def test_fixture_matrix_covers_the_paths(pkg):
    """The fixture set exercises what SURVEY.md 8c asks for: key/inter, all 4 versions, both filter
    types, filter off, SPLITMV, B_PRED, intra MBs in inter frames, golden/alt refs, odd sizes, 1/4/8 partitions."""
    from vp8_testlib import ivf_path
    seen = {"versions": set(), "filter_types": set(), "lf0": False, "split": False, "bpred": False,
            "intra_in_inter": False, "refs": set(), "parts": set(), "odd": False, "sharp": False, "skip": False}
    for name in FIXTURES:
        w, h, frames = pkg.read_ivf(ivf_path(name))
        if name in ("kf_3840x2160", "kf_1920x1080", "p_1920x1080"):
            frames = frames[:2]
        parser = pkg.Parser()
        for data in frames:
            hdr, _, mbs, coef, mvs = pkg.parse_to_numpy(parser, data)
            parser.swap(hdr)
            seen["versions"].add(hdr.version)
            seen["filter_types"].add(hdr.filter_type)
            seen["lf0"] |= hdr.filter_level == 0
            seen["parts"].add(hdr.num_token_partitions)
            seen["odd"] |= (hdr.width % 16 != 0)
            seen["sharp"] |= hdr.sharpness_level != 0
            seen["split"] |= bool((mbs[:, 0] == 9).any())
            seen["bpred"] |= bool((mbs[:, 0] == 4).any())
            seen["skip"] |= bool((mbs[:, 3] & 1).any())
            if hdr.frame_type:
                seen["intra_in_inter"] |= bool((mbs[:, 2] == 0).any())
                seen["refs"] |= set(mbs[:, 2].tolist())
        parser.close()
    assert seen["versions"] == {0, 1, 2, 3}
    assert seen["filter_types"] == {0, 1}
    assert seen["lf0"] and seen["split"] and seen["bpred"] and seen["intra_in_inter"] and seen["odd"]
    assert seen["sharp"] and seen["skip"]
    assert {1, 2}.issubset(seen["refs"])
    assert {1, 4, 8}.issubset(seen["parts"])


# ---- post-processed output (vp8/common/postproc.c): tests/golden/<stream>.pp_<tag>.md5, printed by the reference decoder
#      with VPX_CODEC_USE_POSTPROC + VP8_SET_POSTPROC (make_fixtures.py --postproc) ----
PP_CONFIGS = {"deblock": (1, 0, 0), "demacro4": (2, 4, 0), "demacro9": (2, 9, 0), "demacro0": (3, 0, 0), "noise3": (4, 0, 3),
              "deblock_noise1": (5, 0, 1), "demacro6_noise2": (6, 6, 2)}
PP_STREAMS = ("p_arf_176x144", "p_lowrate_640x360", "kf_odd_67x45", "p_odd_130x98", "p_sharp_320x240", "kf_640x360")
# VP8_MFQE (1024): alone on any stream; with the deblocking filters the reference only survives 16-aligned sizes
# (tests/golden/make_fixtures.py)
MFQE_CONFIGS = {"mfqe": (1024, 0, 0), "mfqe_noise3": (1028, 0, 3)}
MFQE_STREAMS = PP_STREAMS + ("p_split_352x288", "p_prof1_640x360", "kf_1920x1080")
MFQE_FILTER_CONFIGS = {"default": (1027, 4, 0), "mfqe_deblock": (1025, 0, 0), "mfqe_demacro4": (1026, 4, 0),
                       "mfqe_deblock_noise1": (1029, 0, 1), "mfqe_demacro6_noise2": (1030, 6, 2)}
MFQE_FILTER_STREAMS = ("p_arf_176x144", "p_sharp_320x240", "p_split_352x288")
MFQE_CASES = [(n, t, MFQE_CONFIGS[t]) for n in MFQE_STREAMS for t in MFQE_CONFIGS] + \
             [(n, t, MFQE_FILTER_CONFIGS[t]) for n in MFQE_FILTER_STREAMS for t in MFQE_FILTER_CONFIGS]


def golden_pp_md5(name, tag):
    import os
    from vp8_testlib import GOLDEN
    return [l.split()[0] for l in open(os.path.join(GOLDEN, f"{name}.pp_{tag}.md5"))]


@pytest.mark.parametrize("name", PP_STREAMS)
@pytest.mark.parametrize("tag", PP_CONFIGS)
def test_postproc_fixture_md5(name, tag):
    from vp8_testlib import oracle_postproc_ivf
    assert oracle_postproc_ivf(name, *PP_CONFIGS[tag]) == golden_pp_md5(name, tag)


@pytest.mark.parametrize("name,tag,cfg", MFQE_CASES, ids=[f"{n}-{t}" for n, t, _ in MFQE_CASES])
def test_mfqe_fixture_md5(name, tag, cfg):
    """vp8_multiframe_quality_enhance and vp8_post_proc_frame's use of it (postproc.c:802-900, 929-969), restated in
    oracle/vp8_postproc_oracle.c and tests/vp8_testlib.py, against what the reference decoder showed."""
    from vp8_testlib import oracle_postproc_ivf
    gold = golden_pp_md5(name, tag)
    assert oracle_postproc_ivf(name, *cfg) == gold
    if tag == "mfqe" and name != "kf_odd_67x45":     # the fixtures do exercise the path (all but the one whose quantiser never rises)
        from vp8_testlib import golden_md5
        assert gold != golden_md5(name)
