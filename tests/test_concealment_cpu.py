"""CPU: error concealment of the host feeder (csrc/host/vp8_ec.h, vp8_parser_set_error_concealment) on damaged streams -- frames
that never arrive, frames whose token partitions end early -- with the ORACLE's pixel path behind it, against the listings the
reference decoder configured --enable-error-concealment printed for the same damage (tests/golden/ec_*.md5,
tests/golden/gen_ec_listings.py).  The concealed frames are ordinary inter frames to the pixel path: what is tested here is
the feeder's estimated / interpolated motion vectors, thrown-away residuals and reference-refresh decisions."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "debug"))
from ec_cases import CASES
from ec_cpu import damaged_listing
from vp8_testlib import GOLDEN

SMALL = [n for n, (f, _, _) in CASES.items() if "1080" not in f]


@pytest.mark.parametrize("name", SMALL + ["lost_1080p", "cut_1080p"])
def test_concealed_stream_equals_the_reference_with_error_concealment(name):
    fixture, lose, cut = CASES[name]
    want = open(os.path.join(GOLDEN, f"ec_{name}.md5")).read().splitlines()
    assert damaged_listing(fixture, lose, cut, ec=True) == want


def test_without_the_flag_nothing_is_concealed():
    """the same damage without VPX_CODEC_USE_ERROR_CONCEALMENT: a lost frame is not decoded at all (nothing shown for it), and the
    frames after it differ from the concealed ones"""
    fixture, lose, cut = CASES["sharpness"]
    plain = damaged_listing(fixture, lose, (), ec=False)
    concealed = open(os.path.join(GOLDEN, "ec_sharpness.md5")).read().splitlines()
    assert len(plain) == len(concealed) - len(lose)
    assert plain[:lose[0] - 1] == concealed[:lose[0] - 1]
    assert plain[lose[0] - 1] != concealed[lose[0]]


def test_a_short_first_partition_stays_an_error():
    """(the reference reads behind the buffer there: not followed, vp8_parser.h)"""
    got = damaged_listing("p_lowrate_640x360", (), ((5, 250),), ec=True)
    assert got[4] == "decode-error 0005"
