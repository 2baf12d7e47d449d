"""GPU (-m gpu): the vpx_codec API of libvpx_hip.so called directly (ctypes), for the parts the command line
tools do not reach: VP8_COPY_REFERENCE / VP8_SET_REFERENCE (vp8/vp8_dx_iface.c:611-651 in the reference)."""
import ctypes
import hashlib
import os

import pytest

from vp8_testlib import ROOT, golden_md5, ivf_path, load_package

pytestmark = pytest.mark.gpu

VPX_DECODER_ABI_VERSION = 2 + 2 + 1
VPX_IMG_FMT_I420 = 0x100 | 2
VP8_SET_REFERENCE, VP8_COPY_REFERENCE = 1, 2
VP8_LAST_FRAME, VP8_GOLD_FRAME, VP8_ALTR_FRAME = 1, 2, 4


class VpxImage(ctypes.Structure):        # include/vpx/vpx_image.h (vpx/vpx_image.h:103-147 in the reference)
    _fields_ = [("fmt", ctypes.c_int), ("w", ctypes.c_uint), ("h", ctypes.c_uint), ("d_w", ctypes.c_uint),
                ("d_h", ctypes.c_uint), ("x_chroma_shift", ctypes.c_uint), ("y_chroma_shift", ctypes.c_uint),
                ("planes", ctypes.POINTER(ctypes.c_ubyte) * 4), ("stride", ctypes.c_int * 4), ("bps", ctypes.c_int),
                ("user_priv", ctypes.c_void_p), ("img_data", ctypes.c_void_p), ("img_data_owner", ctypes.c_int),
                ("self_allocd", ctypes.c_int)]


class VpxRefFrame(ctypes.Structure):     # include/vpx/vp8.h (vpx/vp8.h:94-98)
    _fields_ = [("frame_type", ctypes.c_int), ("img", VpxImage)]


def _lib():
    L = ctypes.CDLL(os.path.join(ROOT, "libvpx.opencl_amd", "lib", "libvpx_hip.so"))
    L.vpx_codec_vp8_dx.restype = ctypes.c_void_p
    L.vpx_codec_dec_init_ver.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int]
    L.vpx_codec_decode.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_long]
    L.vpx_codec_get_frame.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    L.vpx_codec_get_frame.restype = ctypes.POINTER(VpxImage)
    L.vpx_codec_control_.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.vpx_codec_destroy.argtypes = [ctypes.c_void_p]
    L.vpx_img_alloc.argtypes = [ctypes.POINTER(VpxImage), ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint]
    L.vpx_img_alloc.restype = ctypes.POINTER(VpxImage)
    L.vpx_img_free.argtypes = [ctypes.POINTER(VpxImage)]
    return L


def _plane(img, k, w, h):
    st = img.stride[k]
    return b"".join(ctypes.string_at(ctypes.addressof(img.planes[k].contents) + r * st, w) for r in range(h))


def _md5(img):
    w, h = img.d_w, img.d_h
    m = hashlib.md5()
    m.update(_plane(img, 0, w, h)); m.update(_plane(img, 1, (w + 1) // 2, (h + 1) // 2)); m.update(_plane(img, 2, (w + 1) // 2, (h + 1) // 2))
    return m.hexdigest()


def test_copy_and_set_reference():
    P = load_package()
    name = "p_odd_130x98"                    # 130x98 -> frame buffers of 144x112; MVs reach into the borders
    w, h, frames = P.read_ivf(ivf_path(name))
    gold = golden_md5(name)
    L = _lib()
    ctx = ctypes.create_string_buffer(256)
    assert L.vpx_codec_dec_init_ver(ctx, L.vpx_codec_vp8_dx(), None, 0, VPX_DECODER_ABI_VERSION) == 0

    def decode(i):
        assert L.vpx_codec_decode(ctx, frames[i], len(frames[i]), None, 0) == 0
        it = ctypes.c_void_p()
        img = L.vpx_codec_get_frame(ctx, ctypes.byref(it))
        assert img and _md5(img.contents) == gold[i], i
        return img.contents

    for i in range(4):
        shown = decode(i)
    ref = VpxRefFrame()
    ref.frame_type = VP8_LAST_FRAME
    assert L.vpx_img_alloc(ctypes.byref(ref.img), VPX_IMG_FMT_I420, 144, 112, 1)
    # every inter frame of this stream refreshes LAST: the copied reference is the frame just shown
    assert L.vpx_codec_control_(ctx, VP8_COPY_REFERENCE, ctypes.byref(ref)) == 0
    assert _plane(ref.img, 0, w, h) == _plane(shown, 0, w, h)
    assert _plane(ref.img, 1, (w + 1) // 2, (h + 1) // 2) == _plane(shown, 1, (w + 1) // 2, (h + 1) // 2)
    # wrong dimensions are refused like the reference does
    bad = VpxRefFrame()
    bad.frame_type = VP8_GOLD_FRAME
    assert L.vpx_img_alloc(ctypes.byref(bad.img), VPX_IMG_FMT_I420, 130, 98, 1)
    assert L.vpx_codec_control_(ctx, VP8_COPY_REFERENCE, ctypes.byref(bad)) != 0
    assert L.vpx_codec_control_(ctx, VP8_SET_REFERENCE, ctypes.byref(bad)) != 0
    L.vpx_img_free(ctypes.byref(bad.img))
    # setting LAST to its own content (new buffer, host-side border extension) must leave the stream bit-exact
    assert L.vpx_codec_control_(ctx, VP8_SET_REFERENCE, ctypes.byref(ref)) == 0
    for i in range(4, len(frames)):
        decode(i)
    L.vpx_img_free(ctypes.byref(ref.img))
    L.vpx_codec_destroy(ctx)
