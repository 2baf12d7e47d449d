"""GPU: the output-side post-processing of vp8/common/postproc.c.
 * vp8hip_postproc's kernels against the oracle's filters on synthetic planes, over thresholds on both sides of every
   comparison and frame sizes that are not multiples of the 256-pixel workgroups;
 * VPX_CODEC_USE_POSTPROC + VP8_SET_POSTPROC through the vpx_codec API against what the REFERENCE decoder showed for the
   same configuration (tests/golden/<stream>.pp_<tag>.md5, tests/golden/make_fixtures.py --postproc);
 * the command line tools with the reference's option names."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from test_gpu_codec_api import VPX_DECODER_ABI_VERSION, _lib, _md5
from test_oracle_golden import PP_CONFIGS, PP_STREAMS, golden_pp_md5
from vp8_testlib import GOLDEN, ROOT, OraclePostproc, coded_area_equal, ivf_path, load_package

pytestmark = pytest.mark.gpu

VPX_CODEC_USE_POSTPROC = 0x10000
VP8_SET_POSTPROC = 3


class PostprocCfg(ctypes.Structure):     # vp8_postproc_cfg_t, include/vpx/vp8.h (vpx/vp8.h:76-81)
    _fields_ = [("post_proc_flag", ctypes.c_int), ("deblocking_level", ctypes.c_int), ("noise_level", ctypes.c_int)]


def _frame(P, g, rng, kind):
    """a frame buffer with extended-looking borders: noise, or 8x8 flat patches with mild texture (what the filters are for)"""
    h = g.frame_size // g.y_stride
    if kind == "noise":
        return rng.integers(0, 256, size=g.frame_size).astype(np.uint8)
    base = rng.integers(0, 256, size=(h // 8 + 2, g.y_stride // 8 + 2)).astype(np.int32)
    img = np.kron(base, np.ones((8, 8), np.int32))[:h, :g.y_stride]
    img = np.clip(img + rng.integers(-3, 4, size=img.shape), 0, 255).astype(np.uint8)
    out = rng.integers(0, 256, size=g.frame_size).astype(np.uint8)
    out[:img.size] = img.reshape(-1)
    return out


@pytest.mark.parametrize("w,h", [(176, 144), (67, 45), (640, 360), (1920, 1080), (272, 16)])
def test_filters_against_the_oracle(w, h):
    P = load_package()
    ctx = P.Vp8Hip()
    ctx.configure(w, h, 3, 1)
    g = ctx.g
    rng = np.random.default_rng(w * 7 + h)
    libc = ctypes.CDLL(None)
    cases = [(1, 0, 0, 7), (1, 0, 0, 0), (1, 0, 0, 63), (2, 4, 0, 20), (2, 9, 0, 63), (2, 0, 0, 8), (3, 6, 0, 40), (4, 0, 3, 12),
             (5, 0, 1, 30), (6, 5, 2, 50), (4, 0, 0, 0)]
    for kind in ("flat", "noise"):
        src = _frame(P, g, rng, kind)
        ctx.upload_frame(0, src)
        for flags, level, noise_level, filter_level in cases:
            if (flags & 4) and g.aligned_w > 2816:
                continue
            seed = int(rng.integers(1, 1 << 30))
            libc.srand(seed)
            ora = OraclePostproc(flags, level, noise_level)
            expect = ora.frame(src, g, filter_level)
            # the same decisions for the device: thresholds from the oracle's policy function, the random phases in the same order
            O = ora_lib()
            q, ppl, ppl_dm, mbl = (ctypes.c_int() for _ in range(4))
            O.vp8o_pp_strengths(filter_level, level, ctypes.byref(q), ctypes.byref(ppl), ctypes.byref(ppl_dm), ctypes.byref(mbl))
            libc.srand(seed)
            rv = (libc.rand() & 63) if flags & 2 else 0
            noise = rows = None
            if flags & 4:
                r = np.array([libc.rand() & 0xff for _ in range(3072)], np.uint8)
                rows = np.array([libc.rand() & 0xff for _ in range(g.aligned_h)], np.uint8)
                noise = ora.noise
            ctx.postproc(0, 1, 2, (2 if flags & 2 else flags & 1) | (flags & 4), ppl_dm.value if flags & 2 else ppl.value, mbl.value, rv,
                         noise, ora.clamp, rows)
            got = ctx.download_full(1)
            assert coded_area_equal(got, expect, g) == [], (kind, flags, level, noise_level, filter_level)
    ctx.close()


def ora_lib():
    from vp8_testlib import oracle
    return oracle()


@pytest.mark.parametrize("name", PP_STREAMS)
@pytest.mark.parametrize("tag", PP_CONFIGS)
def test_codec_api_against_the_reference_decoder(name, tag):
    P = load_package()
    _, _, frames = P.read_ivf(ivf_path(name))
    gold = golden_pp_md5(name, tag)
    L = _lib()
    ctypes.CDLL(None).srand(12345)           # the decoder's phases do not depend on the process's rand() (vp8_postproc_host.h)
    ctx = ctypes.create_string_buffer(256)
    assert L.vpx_codec_dec_init_ver(ctx, L.vpx_codec_vp8_dx(), None, VPX_CODEC_USE_POSTPROC, VPX_DECODER_ABI_VERSION) == 0
    cfg = PostprocCfg(*PP_CONFIGS[tag])
    assert L.vpx_codec_control_(ctx, VP8_SET_POSTPROC, ctypes.byref(cfg)) == 0
    got = []
    for data in frames:
        assert L.vpx_codec_decode(ctx, data, len(data), None, 0) == 0
        it = ctypes.c_void_p()
        img = L.vpx_codec_get_frame(ctx, ctypes.byref(it))
        if img:
            got.append(_md5(img.contents))
    L.vpx_codec_destroy(ctx)
    assert got == gold


def test_postproc_needs_the_init_flag_and_leaves_decoding_alone():
    """Without VPX_CODEC_USE_POSTPROC a VP8_SET_POSTPROC is stored and has no effect (vp8_dx_iface.c:446-449); with it, the
    reference frames stay unfiltered: the md5s of the post-processed output differ from the plain ones, and a second decoder
    without post-processing fed the same stream still matches the plain listing frame by frame."""
    from vp8_testlib import golden_md5
    P = load_package()
    name = "p_lowrate_640x360"
    _, _, frames = P.read_ivf(ivf_path(name))
    plain = golden_md5(name)
    L = _lib()
    for flags in (0, VPX_CODEC_USE_POSTPROC):
        ctx = ctypes.create_string_buffer(256)
        assert L.vpx_codec_dec_init_ver(ctx, L.vpx_codec_vp8_dx(), None, flags, VPX_DECODER_ABI_VERSION) == 0
        cfg = PostprocCfg(1, 0, 0)
        assert L.vpx_codec_control_(ctx, VP8_SET_POSTPROC, ctypes.byref(cfg)) == 0
        assert L.vpx_codec_control_(ctx, VP8_SET_POSTPROC, None) != 0
        got = []
        for data in frames:
            assert L.vpx_codec_decode(ctx, data, len(data), None, 0) == 0
            it = ctypes.c_void_p()
            got.append(_md5(L.vpx_codec_get_frame(ctx, ctypes.byref(it)).contents))
        L.vpx_codec_destroy(ctx)
        if flags:
            assert got == golden_pp_md5(name, "deblock") and got != plain
        else:
            assert got == plain


@pytest.mark.parametrize("tool", ["vpxdec", "vpxdec_ref_on_hip"])
def test_command_line_options(tool):
    """`--deblock`, `--demacroblock-level`, `--noise-level` of the product's vpxdec and of the REFERENCE's vpxdec.c built against
    the product: the digest over all post-processed frames the reference's own binary printed (*.pp_vpxdec_md5)."""
    exe = os.path.join(ROOT, "libvpx.opencl_amd", "bin", tool) if tool == "vpxdec" else os.path.join(ROOT, "oracle", "_ref", tool)
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not built")
    for line in open(os.path.join(GOLDEN, "postproc.pp_vpxdec_md5")):
        name, md5, *args = line.split()
        r = subprocess.run([exe, *args, "--md5", "--i420", ivf_path(name)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert r.stdout.split()[0] == md5, (name, args)
