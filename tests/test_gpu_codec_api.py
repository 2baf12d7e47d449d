"""GPU (-m gpu): the vpx_codec API of libvpx_hip.so called directly (ctypes), for the parts the command line
tools do not reach: VP8_COPY_REFERENCE / VP8_SET_REFERENCE (vp8/vp8_dx_iface.c:611-651 in the reference),
VP8D_GET_LAST_REF_UPDATES / _USED / VP8D_GET_FRAME_CORRUPTED (:653-720) and VPX_CODEC_USE_INPUT_FRAGMENTS
(:416-417; vp8/decoder/onyxd_if.c:336-366, decodframe.c:501-592)."""
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


VP8D_GET_LAST_REF_UPDATES, VP8D_GET_FRAME_CORRUPTED, VP8D_GET_LAST_REF_USED = 256, 257, 258
VPX_CODEC_USE_INPUT_FRAGMENTS = 0x40000


@pytest.mark.parametrize("name", ["p_arf_176x144", "p_split_352x288"])
def test_last_ref_controls_answer_like_the_reference(name):
    """Packet by packet (hidden alt-ref frames included) the three read-only decoder controls give what the reference
    decoder gave for the same stream: tests/golden/<name>.refctl, recorded from oracle/_ref/libvpxref.so by
    tests/golden/make_fixtures.py --ref-controls."""
    P = load_package()
    _, _, frames = P.read_ivf(ivf_path(name))
    want = [tuple(int(v) for v in l.split()[1:]) for l in open(os.path.join(ROOT, "tests", "golden", name + ".refctl")) if l[0] != "#"]
    assert len(want) == len(frames)
    L = _lib()
    ctx = ctypes.create_string_buffer(256)
    assert L.vpx_codec_dec_init_ver(ctx, L.vpx_codec_vp8_dx(), None, 0, VPX_DECODER_ABI_VERSION) == 0
    for i, fr in enumerate(frames):
        assert L.vpx_codec_decode(ctx, fr, len(fr), None, 0) == 0
        got = []
        for ctl in (VP8D_GET_LAST_REF_UPDATES, VP8D_GET_LAST_REF_USED, VP8D_GET_FRAME_CORRUPTED):
            v = ctypes.c_int(-1)
            assert L.vpx_codec_control_(ctx, ctl, ctypes.byref(v)) == 0
            got.append(v.value)
        assert tuple(got) == want[i], (i, got, want[i])
    L.vpx_codec_destroy(ctx)


@pytest.mark.parametrize("name,how", [("p_prof1_640x360", "each"), ("p_split_352x288", "each"), ("p_split_352x288", "header+rest"),
                                      ("kf_640x360", "each"), ("p_odd_130x98", "whole")])
def test_input_fragments(name, how):
    """VPX_CODEC_USE_INPUT_FRAGMENTS: the frame arrives as several vpx_codec_decode calls (one partition per call; header and first
    partition, then all token partitions in one call, which have to be unpacked; or the whole frame as one fragment) and is decoded by the flushing call
    (NULL, 0): every shown frame equals the reference MD5, and nothing is shown before the flush."""
    P = load_package()
    _, _, frames = P.read_ivf(ivf_path(name))
    gold = golden_md5(name)
    L = _lib()
    ctx = ctypes.create_string_buffer(256)
    assert L.vpx_codec_dec_init_ver(ctx, L.vpx_codec_vp8_dx(), None, VPX_CODEC_USE_INPUT_FRAGMENTS, VPX_DECODER_ABI_VERSION) == 0
    from test_fragments_cpu import cuts_for_stream
    shown = 0
    for fr, c in zip(frames, cuts_for_stream(P, frames)):
        parts = [fr[a:b] for a, b in zip(c[:-1], c[1:])]
        if how == "header+rest":
            parts = [parts[0], b"".join(parts[1:])]
        elif how == "whole":
            parts = [fr]
        keep = [ctypes.create_string_buffer(p, len(p)) for p in parts]       # borrowed by the decoder until the flush
        for k in keep:
            assert L.vpx_codec_decode(ctx, ctypes.cast(k, ctypes.c_char_p), len(k), None, 0) == 0
            it = ctypes.c_void_p()
            assert not L.vpx_codec_get_frame(ctx, ctypes.byref(it))
        assert L.vpx_codec_decode(ctx, None, 0, None, 0) == 0
        it = ctypes.c_void_p()
        img = L.vpx_codec_get_frame(ctx, ctypes.byref(it))
        if img:
            assert _md5(img.contents) == gold[shown], shown
            shown += 1
    assert shown == len(gold)
    L.vpx_codec_destroy(ctx)
