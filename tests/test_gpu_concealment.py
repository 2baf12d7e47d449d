"""GPU: error concealment end to end through the public API -- bin/decode_damaged (vpx_codec_dec_init with
VPX_CODEC_USE_ERROR_CONCEALMENT; lost frames as vpx_codec_decode(NULL, 0); frames cut short) on the HIP pixel path -- against the
listings of the reference decoder configured --enable-error-concealment (tests/golden/ec_*.md5; where oracle/_ref/ref_md5_ec has
travelled, against a fresh run of it too)."""
import os
import subprocess

import pytest

from ec_cases import CASES, materialize, tool_args
from vp8_testlib import GOLDEN, ROOT, ivf_path

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "libvpx.opencl_amd", "bin", "decode_damaged")
REF = os.path.join(ROOT, "oracle", "_ref", "ref_md5_ec")


@pytest.mark.parametrize("name", list(CASES))
def test_damaged_stream_with_concealment_equals_the_reference(tmp_path, name):
    fixture, lose, cut = CASES[name]
    got = tmp_path / "hip.md5"
    stream = materialize(fixture, GOLDEN, tmp_path)
    r = subprocess.run([BIN] + tool_args(lose, cut) + [stream, str(got)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    listing = open(got).read()
    assert listing == open(os.path.join(GOLDEN, f"ec_{name}.md5")).read()
    if os.path.exists(REF):
        want = tmp_path / "ref.md5"
        subprocess.run([REF, "--damage"] + tool_args(lose, cut) + [stream, str(want)], check=True, capture_output=True)
        assert listing == open(want).read()


def test_lost_frames_without_the_flag_show_nothing(tmp_path):
    """onyxd_if.c:375-407: without concealment a lost frame only marks the last reference corrupt"""
    fixture, lose, _ = CASES["sharpness"]
    got = tmp_path / "hip.md5"
    a = [x for x in tool_args(lose, ()) if x != "--ec"]
    r = subprocess.run([BIN] + a + [ivf_path(fixture), str(got)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = open(got).read().splitlines()
    gold = open(os.path.join(GOLDEN, fixture + ".md5")).read().splitlines()
    assert len(lines) == len(gold) - len(lose)
    assert lines[:lose[0] - 1] == gold[:lose[0] - 1]
