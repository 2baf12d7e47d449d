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


@pytest.mark.parametrize("name", ["kf_odd_67x45", "kf_q0_176x144"])
def test_widths_that_are_not_whole_md5_blocks(pkg, name):
    """Any display size: rows that are not whole MD5 blocks (odd widths, odd heights -- chroma planes of (w + 1) / 2 x (h + 1) / 2 --,
    messages that end anywhere in a block, 56..63 bytes into it included) are hashed by vp8_md5_any_kernel."""
    P = pkg
    ctx = P.Vp8Hip(0)
    try:
        n = _decode_all(P, ctx, name)
        assert ctx.frames_md5(0, n) == golden_md5(name)[:n]
    finally:
        ctx.close()


@pytest.mark.parametrize("w,h", [(1, 1), (2, 2), (7, 9), (10, 5), (11, 5), (16, 16), (55, 1), (56, 1), (63, 3), (64, 2), (66, 2), (120, 3), (129, 2)])
def test_message_tails(pkg, w, h):
    """The padding cases of RFC 1321 3.1 by size: w * h + 2 * ((w + 1) / 2) * ((h + 1) / 2) bytes leave 0..63 in the last block --
    55 (the 0x80 and the length just fit), 56 (they do not: one more block), 0 (a block of padding alone)."""
    import hashlib
    P = pkg
    ctx = P.Vp8Hip(0)
    try:
        ctx.configure(w, h, 2, 1)
        g = ctx.g
        rng = np.random.default_rng(w * 1000 + h)
        want = []
        for fb in range(2):
            buf = rng.integers(0, 256, size=g.frame_size).astype(np.uint8)
            ctx.upload_frame(fb, buf)
            want.append(P.frame_md5(buf, g, w, h))
        assert ctx.frames_md5(0, 2) == want
    finally:
        ctx.close()


@pytest.mark.parametrize("pieces", ["1", "2", "4"])
def test_batch_download_in_pieces(pkg, pieces, monkeypatch):
    """A batch download of 64 frames or more goes in VP8HIP_D2H_STREAMS pieces, each on a stream of its own (default 2): the same
    bytes land, and the wait covers all of them.  vp8hip_reserve allocates both frame-buffer pools ahead; nothing else changes."""
    import ctypes
    monkeypatch.setenv("VP8HIP_D2H_STREAMS", pieces)
    P = pkg
    ctx = P.Vp8Hip(0)
    try:
        w, h, frames = P.read_ivf(ivf_path("kf_640x360"))
        n = 13 * len(frames)
        ctx.configure(w, h, n, len(frames))
        L = ctx.L
        L.vp8hip_reserve.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        ctx._chk(L.vp8hip_reserve(ctx.h, 1, 1), "vp8hip_reserve")
        parser = P.Parser()
        for i, data in enumerate(frames):
            hdr = ctx.parse_into_slot(parser, data, i)
            parser.swap(hdr)
            ctx.upload(i)
        parser.close()
        ctx.decode([(i % len(frames), i, None) for i in range(n)], P.STAGE_ALL)
        L.vp8hip_frame_stride.restype = ctypes.c_size_t
        L.vp8hip_frame_stride.argtypes = [ctypes.c_void_p]
        L.vp8hip_frames_fetch_async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.vp8hip_download_wait.argtypes = [ctypes.c_void_p]
        stride = L.vp8hip_frame_stride(ctx.h)
        got = np.zeros((n, stride), np.uint8)
        ctx._chk(L.vp8hip_frames_fetch_async(ctx.h, 0, n, got.ctypes.data, None), "fetch")
        ctx._chk(L.vp8hip_download_wait(ctx.h), "wait")
        gold = golden_md5("kf_640x360")
        assert [P.frame_md5(got[i], ctx.g, w, h) for i in range(n)] == [gold[i % len(frames)] for i in range(n)]
    finally:
        ctx.close()


@pytest.mark.parametrize("name,copies", [("kf_640x360", 1), ("kf_640x360", 60), ("kf_1920x1080", 1), ("kf_q0_176x144", 70)])
def test_frames_as_packed_i420(pkg, name, copies):
    """vp8hip_frames_fetch_i420_async: the frames as `vpxdec --i420` writes them -- luma, then the two chroma planes, no borders, no
    strides --, packed on the device from whichever form the launch left (copies = 1: a small launch, raster; 60 / 70 copies: more
    than 512 frames, tiles; 176x144: a picture narrower than a tile row's eight tiles, an odd number of chroma rows is covered by
    1080: 540), the digests beside them: MD5 of a delivered frame = the digest = the reference's."""
    import ctypes
    import hashlib
    P = pkg
    ctx = P.Vp8Hip(0)
    try:
        n = _decode_all(P, ctx, name, copies)
        N = copies * n
        L = ctx.L
        L.vp8hip_i420_bytes.restype = ctypes.c_size_t
        L.vp8hip_i420_bytes.argtypes = [ctypes.c_void_p]
        L.vp8hip_frames_fetch_i420_async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.vp8hip_download_wait.argtypes = [ctypes.c_void_p]
        fb = L.vp8hip_i420_bytes(ctx.h)
        w, h = ctx.width, ctx.height
        assert fb == w * h + 2 * ((w + 1) // 2) * ((h + 1) // 2)
        got = np.full((N, fb), 0xAA, np.uint8)
        dig = np.zeros(16 * N, np.uint8)
        ctx._chk(L.vp8hip_frames_fetch_i420_async(ctx.h, 0, N, got.ctypes.data, dig.ctypes.data), "fetch_i420")
        ctx._chk(L.vp8hip_download_wait(ctx.h), "wait")
        gold = golden_md5(name)
        for i in range(N):
            assert hashlib.md5(got[i].tobytes()).hexdigest() == gold[i % n], i
            assert dig[16 * i: 16 * i + 16].tobytes().hex() == gold[i % n], i
    finally:
        ctx.close()


def test_packed_staging_is_a_cache(pkg, monkeypatch):
    """A digest-only fetch of a large batch of TILED frames is hashed from a packed copy (vp8_pack_i420_tiles_kernel) whose buffer the
    context keeps -- vp8hip_memory_usage().packed_staging -- until vp8hip_release_staging gives it back (round 6: the library frees it by
    itself when a pool finds no room beside it); with or without it the digests are the reference's."""
    P = pkg
    monkeypatch.setenv("VP8HIP_RECON", "simt")            # (tiles: the packed copy is made of them)
    monkeypatch.setenv("VP8HIP_MD5_PACK_FROM", "8")
    ctx = P.Vp8Hip(0)
    try:
        name, copies = "kf_640x360", 3
        n = _decode_all(P, ctx, name, copies)
        gold = golden_md5(name)
        want = [gold[i % n] for i in range(copies * n)]
        assert ctx.memory_usage()["packed_staging"] == 0
        assert ctx.frames_md5(0, copies * n) == want
        held = ctx.memory_usage()["packed_staging"]
        assert held >= copies * n * 640 * 360 * 3 // 2
        ctx.L.vp8hip_release_staging.argtypes = [__import__("ctypes").c_void_p]
        ctx._chk(ctx.L.vp8hip_release_staging(ctx.h), "vp8hip_release_staging")
        assert ctx.memory_usage()["packed_staging"] == 0
        assert ctx.frames_md5(0, copies * n) == want          # (allocated again)
        assert ctx.memory_usage()["packed_staging"] == held
        assert ctx.frames_md5(2, 4) == want[2:6]              # (a small batch: from the tiles)
    finally:
        ctx.close()
