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
