"""GPU: frames whose token partition stops short.  The reference decodes on: from the macroblock at which its bool decoder
reports the end of data (vp8dx_bool_error, vp8/decoder/dboolhuff.h:131-153) no tokens are read, the macroblocks keep their
parsed skip flag -- so their inner edges are still loop-filtered -- and get no residual (decode_macroblock,
vp8/decoder/decodframe.c:119-130), and the frame is flagged corrupt.  The product's feeder has to switch over at the same
macroblock: its decode_to_md5 listing of damaged streams equals the one the reference decoder (oracle/_ref/ref_md5, built from
/root/reference) prints for the same file."""
import os
import struct
import subprocess

import pytest

from vp8_testlib import ROOT, ivf_path

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "libvpx.opencl_amd", "bin", "decode_to_md5")
REF = os.path.join(ROOT, "oracle", "_ref", "ref_md5")


def damaged(src, dst, victim, cut):
    """copy an IVF, dropping the last `cut` bytes of frame `victim`"""
    data = open(src, "rb").read()
    out = bytearray(data[:32])
    off, k = 32, 0
    while off + 12 <= len(data):
        size, = struct.unpack_from("<I", data, off)
        frame = data[off + 12: off + 12 + size]
        if k == victim:
            frame = frame[:len(frame) - cut]
        out += struct.pack("<I", len(frame)) + data[off + 4: off + 12] + frame
        off += 12 + size
        k += 1
    open(dst, "wb").write(out)


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref (the reference decoder built from /root/reference) is not here")
@pytest.mark.parametrize("name,victim,cuts", [("kf_640x360", 1, (1, 7, 100, 1000, 9000)), ("p_1920x1080", 3, (3, 50, 250, 2000)),
                                              ("kf_odd_67x45", 0, (2, 30)), ("p_split_352x288", 2, (5, 200))])
def test_truncated_token_partition_like_the_reference(tmp_path, name, victim, cuts):
    for cut in cuts:
        bad = tmp_path / f"{name}_{victim}_{cut}.ivf"
        damaged(ivf_path(name), bad, victim, cut)
        want, got = tmp_path / "ref.md5", tmp_path / "hip.md5"
        r = subprocess.run([REF, str(bad), str(want)], capture_output=True, text=True)
        g = subprocess.run([BIN, str(bad), str(got)], capture_output=True, text=True)
        # a cut that reaches into the first partition is an error in both (decodframe.c:733-736), after the same frames
        assert (g.returncode == 0) == (r.returncode == 0), (name, victim, cut, r.stderr, g.stderr)
        if r.returncode:
            assert "Corrupt frame detected" in r.stderr and "Corrupt frame detected" in g.stderr, (r.stderr, g.stderr)
            assert len(open(want).read().splitlines()) == victim
        assert open(got).read() == open(want).read(), (name, victim, cut)
