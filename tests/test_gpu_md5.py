"""GPU: the frames' MD5s computed on the device (vp8hip_frames_fetch_async, csrc/hip/vp8_md5.hip: a frame per lane) are the MD5s the
reference's decode_to_md5 prints -- the committed golden listings -- and what hashing the downloaded planes on the host gives."""
import numpy as np
import pytest

from vp8_testlib import golden_md5, ivf_path

pytestmark = pytest.mark.gpu


def _decode_all(P, ctx, name, copies=1):
    """all-key-frame fixture into frame buffers 0 .. copies * n - 1 (frame i of the stream in buffers i, i + n, ...)"""
    w, h, frames = P.read_ivf(ivf_path(name))
    n = len(frames)
    ctx.configure(w, h, copies * n, n)
    parser = P.Parser()
    for i, data in enumerate(frames):
        hdr = ctx.parse_into_slot(parser, data, i)
        assert hdr.frame_type == 0
        parser.swap(hdr)
        ctx.upload(i)
    ctx.decode([(i % n, i, None) for i in range(copies * n)], P.STAGE_ALL)
    parser.close()
    return n


@pytest.mark.parametrize("name,copies", [("kf_640x360", 1), ("kf_1920x1080", 1), ("kf_640x360", 13)])
def test_device_md5_equals_the_reference_listing(pkg, name, copies):
    P = pkg
    ctx = P.Vp8Hip(0)
    try:
        n = _decode_all(P, ctx, name, copies)
        gold = golden_md5(name)
        got = ctx.frames_md5(0, copies * n)          # 130 frames: three waves, the last one partly idle
        assert got == [gold[i % n] for i in range(copies * n)]
        assert got[n - 1] == P.planes_md5(*ctx.download_planes(n - 1))
        part = ctx.frames_md5(3, 2)                  # any run of frame buffers
        assert part == [gold[3 % n], gold[4 % n]]
    finally:
        ctx.close()


def test_frames_and_digests_in_one_call(pkg):
    import ctypes
    P = pkg
    ctx = P.Vp8Hip(0)
    try:
        n = _decode_all(P, ctx, "kf_640x360")
        L = ctx.L
        L.vp8hip_frame_stride.restype = ctypes.c_size_t
        L.vp8hip_frame_stride.argtypes = [ctypes.c_void_p]
        L.vp8hip_frames_fetch_async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.vp8hip_download_wait.argtypes = [ctypes.c_void_p]
        stride = L.vp8hip_frame_stride(ctx.h)
        frames = np.zeros((n, stride), np.uint8)
        dig = np.zeros(16 * n, np.uint8)
        ctx._chk(L.vp8hip_frames_fetch_async(ctx.h, 0, n, frames.ctypes.data, dig.ctypes.data), "fetch")
        ctx._chk(L.vp8hip_download_wait(ctx.h), "wait")
        gold = golden_md5("kf_640x360")
        for i in range(n):
            assert dig[16 * i: 16 * i + 16].tobytes().hex() == gold[i]
            assert P.frame_md5(frames[i], ctx.g, ctx.width, ctx.height) == gold[i]
    finally:
        ctx.close()


def test_widths_that_are_not_whole_md5_blocks_are_refused(pkg):
    P = pkg
    ctx = P.Vp8Hip(0)
    try:
        _decode_all(P, ctx, "kf_odd_67x45")
        with pytest.raises(RuntimeError, match="multiple of 128"):
            ctx.frames_md5(0, 1)
    finally:
        ctx.close()
