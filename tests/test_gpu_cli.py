"""GPU (-m gpu): the C host decoder behind the public vpx codec API, driven by the two command line
tools exactly as a user of the reference would drive them."""
import filecmp
import os
import subprocess

import pytest

from vp8_testlib import FIXTURES, GOLDEN, ROOT, ivf_path

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "libvpx.opencl_amd", "bin")


@pytest.mark.parametrize("name", FIXTURES)
def test_decode_to_md5_listing_is_byte_identical(name, tmp_path):
    out = tmp_path / "out.md5"
    r = subprocess.run([os.path.join(BIN, "decode_to_md5"), ivf_path(name), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(out, os.path.join(GOLDEN, name + ".md5"), shallow=False)


@pytest.mark.parametrize("name", ["kf_640x360", "p_split_352x288", "p_prof3_640x360", "kf_1920x1080"])
def test_vpxdec_md5(name):
    r = subprocess.run([os.path.join(BIN, "vpxdec"), "--md5", "--i420", "--summary", ivf_path(name)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split()[0] == open(os.path.join(GOLDEN, name + ".vpxdec_md5")).read().strip()
    assert "decoded frames" in r.stderr


def test_vpxdec_writes_i420(tmp_path):
    out = tmp_path / "o.i420"
    r = subprocess.run([os.path.join(BIN, "vpxdec"), "--i420", "-o", str(out), "--limit=2", ivf_path("kf_odd_67x45")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(out) == 2 * (67 * 45 + 2 * 34 * 23)


def test_api_rejects_bad_streams():
    r = subprocess.run([os.path.join(BIN, "decode_to_md5"), os.path.join(GOLDEN, "kf_640x360.md5"), "/dev/null"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "not an IVF" in r.stderr
