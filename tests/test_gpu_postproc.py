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
from test_oracle_golden import MFQE_CASES, PP_CONFIGS, PP_STREAMS, golden_pp_md5
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
        return _edge_rows(rng.integers(0, 256, size=g.frame_size).astype(np.uint8), g)
    base = rng.integers(0, 256, size=(h // 8 + 2, g.y_stride // 8 + 2)).astype(np.int32)
    img = np.kron(base, np.ones((8, 8), np.int32))[:h, :g.y_stride]
    img = np.clip(img + rng.integers(-3, 4, size=img.shape), 0, 255).astype(np.uint8)
    out = rng.integers(0, 256, size=g.frame_size).astype(np.uint8)
    out[:img.size] = img.reshape(-1)
    return _edge_rows(out, g)


def _edge_rows(buf, g):
    """the two rows above and below each plane repeat its first / last row, as in every frame the decoder shows
    (vp8_yv12_extend_frame_borders): the reference's filters read them, the product's clamp the row instead"""
    for off, stride, rows, cols in ((g.y_off, g.y_stride, g.aligned_h, g.aligned_w), (g.u_off, g.uv_stride, g.aligned_h // 2, g.aligned_w // 2),
                                    (g.v_off, g.uv_stride, g.aligned_h // 2, g.aligned_w // 2)):
        e = off + (rows - 1) * stride
        for k in (1, 2):
            buf[off - k * stride:off - k * stride + cols] = buf[off:off + cols]
            buf[e + k * stride:e + k * stride + cols] = buf[e:e + cols]
    return buf


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


@pytest.mark.parametrize("w,h", [(176, 144), (67, 45), (640, 360), (1920, 1080)])
def test_mfqe_against_the_oracle(w, h):
    """vp8hip_mfqe against the oracle's vp8_multiframe_quality_enhance on synthetic pairs of pictures: the old one a noisy,
    blurred or shifted version of the new one (so that blocks fall on every side of the threshold), bright and dark (the
    reference's variance wraps for bright 16x16 blocks), key and inter frames, every macroblock class, in place and not."""
    P = load_package()
    ctx = P.Vp8Hip()
    ctx.configure(w, h, 4, 1)
    g = ctx.g
    O = ora_lib()
    rng = np.random.default_rng(w + 3 * h)
    mb_cols, mb_rows = g.aligned_w // 16, g.aligned_h // 16
    n = mb_cols * mb_rows
    G = (ctypes.c_int * 10)(*[getattr(g, f) for f, _ in g._fields_])
    seen = set()
    for trial, (frame_type, qcurr, qprev, bright, amp) in enumerate(
            [(0, 40, 20, 0, 2), (1, 127, 0, 1, 1), (1, 60, 50, 1, 3), (0, 100, 36, 1, 6), (1, 90, 10, 0, 1), (1, 127, 117, 1, 1)]):
        show = _frame(P, g, rng, "flat")
        if bright:
            show = np.maximum(show, 170 + (show >> 2)).astype(np.uint8)
        prev = np.clip(show.astype(np.int32) + rng.integers(-amp, amp + 1, size=show.size), 0, 255).astype(np.uint8)
        wild = rng.integers(0, 256, size=show.size).astype(np.uint8)            # some macroblocks far from the new picture
        sel = np.repeat(rng.random(show.size // 64 + 1) < 0.15, 64)[:show.size]
        prev = np.where(sel, wild, prev).astype(np.uint8)
        mbs = np.zeros((n, 64), np.uint8)
        mbs[:, 0] = rng.choice([0, 1, 3, 4, 5, 7, 8, 9], size=n)                # y_mode: DC/V/TM, B_PRED, inter modes, SPLITMV
        intra = mbs[:, 0] <= 4
        mbs[:, 2] = np.where(intra, 0, rng.integers(1, 4, size=n))              # ref_frame
        mvs = rng.integers(-14, 15, size=(n, 16, 2)).astype(np.int16)
        mvs[rng.random(n) < 0.5] //= 4
        hdr = P.FrameHdr()
        hdr.width, hdr.height, hdr.mb_cols, hdr.mb_rows, hdr.frame_type, hdr.base_qindex = w, h, mb_cols, mb_rows, frame_type, qcurr
        expect = prev.copy()
        O.vp8o_mfqe(ctypes.byref(hdr), G, ctypes.c_void_p(mbs.ctypes.data), ctypes.c_void_p(mvs.ctypes.data),
                    ctypes.c_void_p(show.ctypes.data), ctypes.c_void_p(expect.ctypes.data), ctypes.c_int(qcurr), ctypes.c_int(qprev))
        H = P.load_host()
        cls = np.zeros(n, np.uint8)
        H.vp8_pp_mfqe_classes(ctypes.byref(hdr), ctypes.c_void_p(mbs.ctypes.data), ctypes.c_size_t(64), ctypes.c_void_p(mvs.ctypes.data),
                              ctypes.c_void_p(cls.ctypes.data))
        seen |= set(cls.tolist())
        ctx.upload_frame(0, show)
        for dst in (1, 2):                     # in place, and into a third buffer
            ctx.upload_frame(1, prev)
            ctx.mfqe(0, 1, dst, cls, qcurr, qprev)
            got = ctx.download_full(dst)
            assert coded_area_equal(got, expect, g) == [], (trial, dst)
        # all three outcomes occur: copied, kept / blended, and (inter frames) macroblocks that moved too far
        same_new = coded_area_equal(expect, show, g) == []
        same_old = coded_area_equal(expect, prev, g) == []
        assert not same_new and not same_old, trial
    assert seen == {0, 1, 2}
    ctx.close()


@pytest.mark.parametrize("name,tag,cfg", MFQE_CASES, ids=[f"{n}-{t}" for n, t, _ in MFQE_CASES])
def test_mfqe_through_the_codec_api_against_the_reference_decoder(name, tag, cfg):
    assert _codec_api_listing(name, cfg) == golden_pp_md5(name, tag)


@pytest.mark.parametrize("name", ["kf_640x360", "p_odd_130x98"])
def test_mfqe_with_the_filters_on_sizes_the_reference_dies_on(name):
    """VP8_MFQE together with the deblocking filters on a size that is not a multiple of 16: the reference never gets that far
    (make_fixtures.py), so no listing exists; the product treats such a stream like any other, which is what the restatement
    in tests/vp8_testlib.py does as well.  The default configuration (`vpxdec --postproc`) is this combination."""
    from vp8_testlib import oracle_postproc_ivf
    for cfg in ((1027, 4, 0), (1025, 0, 0)):
        got = _codec_api_listing(name, cfg)
        assert got == oracle_postproc_ivf(name, *cfg)
        assert got != golden_pp_md5(name, "demacro4" if cfg[0] == 1027 else "deblock")      # ... and MFQE did act


def _codec_api_listing(name, cfg):
    P = load_package()
    _, _, frames = P.read_ivf(ivf_path(name))
    L = _lib()
    ctypes.CDLL(None).srand(4321)
    ctx = ctypes.create_string_buffer(256)
    assert L.vpx_codec_dec_init_ver(ctx, L.vpx_codec_vp8_dx(), None, VPX_CODEC_USE_POSTPROC, VPX_DECODER_ABI_VERSION) == 0
    if cfg is not None:
        c = PostprocCfg(*cfg)
        assert L.vpx_codec_control_(ctx, VP8_SET_POSTPROC, ctypes.byref(c)) == 0
    got = []
    for data in frames:
        assert L.vpx_codec_decode(ctx, data, len(data), None, 0) == 0
        it = ctypes.c_void_p()
        img = L.vpx_codec_get_frame(ctx, ctypes.byref(it))
        if img:
            got.append(_md5(img.contents))
    L.vpx_codec_destroy(ctx)
    return got


def test_default_configuration_is_the_references():
    """No VP8_SET_POSTPROC: deblock + demacroblock + MFQE at level 4 (vp8_dx_iface.c:421-431)."""
    assert _codec_api_listing("p_arf_176x144", None) == golden_pp_md5("p_arf_176x144", "default")


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
