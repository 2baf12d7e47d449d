"""CPU: the C-ABI libraries load and export every symbol their headers declare; without a GPU the
pixel path refuses to start (no fallback)."""
import ctypes
import os
import re

import pytest

from vp8_testlib import ROOT


def declared(header, prefix):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"^[ \t]*#[ \t]*define(?:.*\\\n)*.*$", "", src, flags=re.M)     # macros are not symbols
    return sorted(set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", src)))


def test_vp8hip_exports(pkg):
    L = pkg.load_hip()
    names = declared("vp8hip.h", "vp8hip_")
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), f"libvp8hip.so does not export {n}"


def test_no_gpu_no_decode(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError) as e:
        pkg.Vp8Hip()
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_host_library_exports(pkg):
    L = pkg.load_host()
    for n in ("vp8_parser_create", "vp8_parser_begin_frame", "vp8_parser_decode_mbs", "vp8_refs_swap"):
        assert hasattr(L, n)
    vpx = [n for n in declared("vpx/vpx_decoder.h", "vpx_codec_") + declared("vpx/vpx_codec.h", "vpx_codec_")
           + declared("vpx/vpx_image.h", "vpx_img_") + declared("vpx/vp8dx.h", "vpx_codec_")]
    assert "vpx_codec_dec_init_ver" in vpx and "vpx_codec_decode" in vpx and "vpx_codec_get_frame" in vpx
    for n in vpx:
        assert hasattr(L, n), f"libvpx_hip.so does not export {n}"


def test_product_does_not_link_the_oracle():
    import subprocess
    for lib in ("libvp8hip.so", "libvpx_hip.so"):
        path = os.path.join(ROOT, "libvpx.opencl_amd", "lib", lib)
        out = subprocess.run(["ldd", path], capture_output=True, text=True).stdout
        assert "oracle" not in out and "vpxref" not in out
        syms = subprocess.run(["nm", "-D", path], capture_output=True, text=True).stdout
        assert "vp8o_" not in syms


def test_rtcd_table_exports(pkg):
    """include/vp8_rtcd.h: every entry is a function pointer in libvpx_hip.so whose `_hip` specialisation libvp8hip.so
    exports; the per-block part carries the names of the reference's decoder prototypes (vp8/common/rtcd_defs.sh:20-204)."""
    src = open(os.path.join(ROOT, "include", "vp8_rtcd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    blocks = re.findall(r"VP8_RTCD_ENTRY\(\w+, (vp8_\w+),", src)
    frames = re.findall(r"RTCD_EXTERN int \(\*(\w+)\)", src)
    assert len(blocks) == 32 and len(frames) == 4
    host, hip = pkg.load_host(), pkg.load_hip()
    for n in frames:
        ctypes.c_void_p.in_dll(host, n)
        assert hasattr(host, n + "_hip")
    for n in blocks:
        ctypes.c_void_p.in_dll(host, n)
        assert hasattr(hip, n + "_hip"), f"libvp8hip.so does not export {n}_hip"
    reference = """vp8_dequantize_b vp8_dequant_idct_add vp8_dequant_idct_add_y_block vp8_dequant_idct_add_uv_block
        vp8_loop_filter_mbv vp8_loop_filter_bv vp8_loop_filter_mbh vp8_loop_filter_bh vp8_loop_filter_simple_mbv
        vp8_loop_filter_simple_mbh vp8_loop_filter_simple_bv vp8_loop_filter_simple_bh vp8_short_idct4x4llm
        vp8_short_inv_walsh4x4_1 vp8_short_inv_walsh4x4 vp8_dc_only_idct_add vp8_copy_mem16x16 vp8_copy_mem8x8
        vp8_copy_mem8x4 vp8_build_intra_predictors_mby vp8_build_intra_predictors_mby_s vp8_build_intra_predictors_mbuv
        vp8_build_intra_predictors_mbuv_s vp8_intra4x4_predict vp8_sixtap_predict16x16 vp8_sixtap_predict8x8
        vp8_sixtap_predict8x4 vp8_sixtap_predict4x4 vp8_bilinear_predict16x16 vp8_bilinear_predict8x8
        vp8_bilinear_predict8x4 vp8_bilinear_predict4x4""".split()
    ours = [n[:-3] if n.endswith("_px") else n for n in blocks]      # the MACROBLOCKD entries, by fields: `_px`
    assert sorted(ours) == sorted(reference)
    # no frame-granular entry reuses a reference name with another signature
    assert not set(frames) & {"vp8_loop_filter_frame", "vp8_yv12_extend_frame_borders_ptr"}
    host.vpx_rtcd()
    for n in frames + blocks:
        assert ctypes.c_void_p.in_dll(host, n).value, n


def test_postproc_phase_generator_is_the_c_librarys(pkg):
    """The post-processing filters' random phases (vp8_postproc_host.h): the reference draws them from the C library's never
    seeded rand(); the product reproduces that sequence per decoder instead of sharing rand() with the GPU runtime."""
    H = pkg.load_host()
    libc = ctypes.CDLL(None)
    libc.srand(1)
    st = ctypes.create_string_buffer(40000)
    assert all(H.vp8_pp_rand(st) == libc.rand() for _ in range(20000))


def test_mfqe_policy(pkg):
    """vp8_pp_mfqe_step / vp8_pp_mfqe_classes (vp8_postproc_host.h) against vp8_post_proc_frame's bookkeeping (postproc.c:948-986)
    and vp8_multiframe_quality_enhance's per-macroblock choice (:834-843), spelled out here."""
    import numpy as np
    H = pkg.load_host()

    class Cfg(ctypes.Structure):
        _fields_ = [("post_proc_flag", ctypes.c_int), ("deblocking_level", ctypes.c_int), ("noise_level", ctypes.c_int)]

    rng = np.random.default_rng(5)
    for flag in (1024, 1027, 3):
        st = ctypes.create_string_buffer(40000)
        last, shown = 0, 0
        for q in [int(x) for x in rng.integers(0, 128, size=200)]:
            shown += 1
            qprev = ctypes.c_int(-1)
            got = H.vp8_pp_mfqe_step(st, ctypes.byref(Cfg(flag, 4, 0)), q, ctypes.byref(qprev))
            want = bool(flag & 1024) and shown >= 2 and q - last >= 10
            assert bool(got) == want
            if want:
                assert qprev.value == last
                last = (3 * last + q) >> 2
            else:
                last = q
    n = 500
    mbs = np.zeros((n, 64), np.uint8)
    mbs[:, 0] = rng.integers(0, 10, size=n)
    mbs[:, 2] = np.where(mbs[:, 0] <= 4, 0, rng.integers(1, 4, size=n))
    mvs = rng.integers(-13, 14, size=(n, 16, 2)).astype(np.int16)
    for frame_type in (0, 1):
        hdr = pkg.FrameHdr()
        hdr.mb_cols, hdr.mb_rows, hdr.frame_type = 25, 20, frame_type
        cls = np.zeros(n, np.uint8)
        H.vp8_pp_mfqe_classes(ctypes.byref(hdr), ctypes.c_void_p(mbs.ctypes.data), ctypes.c_size_t(64), ctypes.c_void_p(mvs.ctypes.data), ctypes.c_void_p(cls.ctypes.data))
        for i in range(n):
            mv = (0, 0) if mbs[i, 2] == 0 else tuple(int(v) for v in mvs[i, 15])
            still = frame_type == 0 or (abs(mv[0]) <= 10 and abs(mv[1]) <= 10)
            assert cls[i] == (0 if not still else 2 if mbs[i, 0] in (4, 9) else 1)
