"""GPU: streams synthesised by tests/vp8_writer.py (any size, every mode, all segment / delta features, up to 8 token
partitions, coefficients up to +-2047) through the product -- host feeder + HIP pixel path behind the C ABI -- against
the oracle's decode of the IR the stream was written from.  Sizes the committed fixtures do not have: one macroblock,
one macroblock row / column, 8K wide, wider than 4K."""
import os
import subprocess

import numpy as np
import pytest

from vp8_testlib import ROOT, load_package, oracle_decode, synth_ir
from vp8_writer import write_ivf, write_key_frame

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "libvpx.opencl_amd", "bin")

CASES = [  # width, height, seed, log2 partitions, filter_type, dense, big, segmented
    (16, 16, 11, 0, 0, 0.5, True, True), (16, 400, 12, 1, 0, 0.3, False, True), (1200, 16, 13, 3, 1, 0.3, True, True),
    (352, 288, 14, 2, 0, 0.3, True, True), (1920, 1080, 15, 3, 0, 0.03, False, True), (8192, 48, 16, 0, 0, 0.05, True, False),
    (4112, 80, 17, 2, 0, 0.05, False, True),
]


@pytest.fixture(params=["auto", "wave1cu", "lane"], autouse=True)
def kernel_family(request, monkeypatch):
    """The library's own choice for the launch size (small launches: cross-CU wave-per-row kernels), and the other
    kernel variants forced through the tuning knobs (see tests/test_gpu_parity.py)."""
    for k in ("VP8HIP_RECON", "VP8HIP_XCU"):
        monkeypatch.delenv(k, raising=False)
    if request.param == "wave1cu":
        monkeypatch.setenv("VP8HIP_RECON", "wave")
        monkeypatch.setenv("VP8HIP_XCU", "0")
    elif request.param.startswith("lane"):
        monkeypatch.setenv("VP8HIP_RECON", "simt")
    return request.param


def _oracle_md5(pkg, hdr, mbs, coef, mvs):
    g = pkg.geom(hdr.width, hdr.height)
    buf = np.zeros(g.frame_size, np.uint8)
    oracle_decode(hdr, mbs, coef, mvs, buf, (None, None, None))
    return pkg.frame_md5(buf, g, hdr.width, hdr.height)


@pytest.mark.parametrize("case", CASES)
def test_written_streams_through_the_c_abi(pkg, case):
    w, h, seed, lp, ftype, dense, big, seg = case
    hdr, mbs, coef, mvs = synth_ir(w, h, seed, inter=False, filter_type=ftype, dense=dense, big=big, segmented=seg)
    hdr.num_token_partitions = 1 << lp
    data = write_key_frame(hdr, mbs, coef, log2_parts=lp)
    want = _oracle_md5(pkg, hdr, mbs, coef, mvs)
    ctx = pkg.Vp8Hip(0)
    try:
        ctx.configure(w, h, 3, 3)
        parser = pkg.Parser()
        for slot in range(3):                      # one launch of three frames and three of one: both launch shapes
            h2 = ctx.parse_into_slot(parser, data, slot)
            parser.swap(h2)
            ctx.upload(slot)
        parser.close()
        ctx.decode([(s, s, None) for s in range(3)], 7)
        got = [pkg.planes_md5(*ctx.download_planes(s)) for s in range(3)]
        ctx.decode([(1, 0, None)], 7)
        got.append(pkg.planes_md5(*ctx.download_planes(0)))
    finally:
        ctx.close()
    assert got == [want] * 4


def test_written_stream_through_the_tools(pkg, tmp_path):
    """Twelve different synthetic key frames in one IVF: decode_to_md5 (vpx_codec API, one frame per launch) and
    batch_md5 (threaded feeder, batched launches) both list the oracle's digests."""
    w, h = 208, 176
    frames, want = [], []
    for seed in range(30, 42):
        hdr, mbs, coef, mvs = synth_ir(w, h, seed, inter=False, filter_type=seed & 1, dense=0.3, big=bool(seed & 2))
        lp = seed % 4
        hdr.num_token_partitions = 1 << lp
        frames.append(write_key_frame(hdr, mbs, coef, log2_parts=lp))
        want.append(_oracle_md5(pkg, hdr, mbs, coef, mvs))
    ivf = tmp_path / "s.ivf"
    write_ivf(ivf, w, h, frames)
    for tool, extra in (("decode_to_md5", []), ("batch_md5", ["--threads", "3", "--batch", "5"]),
                        ("batch_md5", ["--device-entropy", "--batch", "5"])):
        out = tmp_path / (tool + "".join(extra) + ".md5")
        r = subprocess.run([os.path.join(BIN, tool)] + extra + [str(ivf), str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert [l.split()[0] for l in open(out)] == want, tool
