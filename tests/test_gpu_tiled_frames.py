"""GPU (-m gpu): the two forms of a frame buffer (csrc/hip/vp8hip_launch.hip).  A large launch leaves its frames as
macroblock-window tiles; nothing converts them until something asks for the raster form.  Every consumer must see the reference's
frames whichever form it reads:
 * the MD5 kernel on the tiles (vp8_md5_tiles_kernel) and on the raster form (vp8_md5_kernel);
 * a batch download into page-locked memory (the tiled -> raster pass writes the host buffer itself: vp8_detile_run_kernel) and
   into ordinary memory (raster form first, then a copy);
 * vp8hip_frame_download, whole buffer, borders included, against the oracle (the conversion + vp8_extend_kernel on demand);
 * vp8hip_frame_copy of a frame that only exists as tiles; frames in both forms inside one batch;
 * a launch of inter frames whose reference frames only exist as tiles (test_gpu_inter_launches.py chains such launches)."""
import ctypes

import numpy as np
import pytest

from vp8_testlib import bordered_area_equal, golden_md5, ivf_path, oracle_decode_ivf

pytestmark = pytest.mark.gpu


def _setup(P, ctx, name, n, monkeypatch, lane=True):
    monkeypatch.setenv("VP8HIP_RECON", "simt" if lane else "wave")
    w, h, frames = P.read_ivf(ivf_path(name))
    ctx.configure(w, h, n + 2, n)
    parser = P.Parser()
    for i, data in enumerate(frames[:n]):
        ctx.sync()
        hdr, _ = ctx.parse_into_slot_compact(parser, data, i)
        parser.swap(hdr)
    parser.close()
    for i in range(len(frames), n):
        ctx.ir_copy(i, i % len(frames))
    L = ctx.L
    L.vp8hip_frame_stride.restype = ctypes.c_size_t
    L.vp8hip_frame_stride.argtypes = [ctypes.c_void_p]
    L.vp8hip_frames_fetch_async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.vp8hip_download_wait.argtypes = [ctypes.c_void_p]
    L.vp8hip_host_alloc.restype = ctypes.c_void_p
    L.vp8hip_host_alloc.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    L.vp8hip_host_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    return len(frames), golden_md5(name)


@pytest.mark.parametrize("name,n", [("kf_640x360", 23), ("kf_1920x1080", 12)])
def test_tiles_are_read_where_they_lie(pkg, monkeypatch, name, n):
    P = pkg
    ctx = P.Vp8Hip(0)
    try:
        nsrc, gold = _setup(P, ctx, name, n, monkeypatch)
        L = ctx.L
        ctx.decode([(i, i, None) for i in range(n)], P.STAGE_ALL)
        assert ctx.stats().fused == 1 and ctx.stats().detile_pass == 0
        want = [gold[i % nsrc] for i in range(n)]
        # digests from the tiles
        assert ctx.frames_md5(0, n) == want
        assert ctx.frames_md5(3, 2) == want[3:5]
        # frames + digests in one call, the frames into page-locked memory, direct downloads on: the pass writes host memory
        L.vp8hip_set_direct_download(ctx.h, 1)
        stride = L.vp8hip_frame_stride(ctx.h)
        pinned = L.vp8hip_host_alloc(ctx.h, stride * n)
        assert pinned
        try:
            host = np.ctypeslib.as_array(ctypes.cast(pinned, ctypes.POINTER(ctypes.c_uint8)), shape=(n, stride))
            host[:] = 0x5a
            dig = np.zeros(16 * n, np.uint8)
            ctx._chk(L.vp8hip_frames_fetch_async(ctx.h, 0, n, pinned, dig.ctypes.data), "fetch")
            ctx._chk(L.vp8hip_download_wait(ctx.h), "wait")
            for i in range(n):
                assert dig[16 * i: 16 * i + 16].tobytes().hex() == want[i]
                assert P.frame_md5(host[i], ctx.g, ctx.width, ctx.height) == want[i], i
            assert ctx.stats().detile_pass == 0 and not any(host[0, :ctx.g.y_off - 8] != 0x5a)      # (no raster form, no borders)
        finally:
            L.vp8hip_host_free(ctx.h, pinned)
            L.vp8hip_set_direct_download(ctx.h, 0)
        # ... into ordinary memory: raster form on the device first, then a copy (and now both forms hold the frames)
        frames = np.zeros((n, stride), np.uint8)
        ctx._chk(L.vp8hip_frames_fetch_async(ctx.h, 0, n, frames.ctypes.data, None), "fetch")
        ctx._chk(L.vp8hip_download_wait(ctx.h), "wait")
        for i in range(n):
            assert P.frame_md5(frames[i], ctx.g, ctx.width, ctx.height) == want[i], i
        assert ctx.frames_md5(0, n) == want
        for i in (0, n // 2, n - 1):
            assert P.planes_md5(*ctx.download_planes(i)) == want[i]
    finally:
        ctx.close()


def test_whole_buffer_with_borders_after_the_lazy_pass(pkg, monkeypatch):
    P = pkg
    name, n = "kf_640x360", 10
    _, kept = oracle_decode_ivf(name, keep_frames=True)
    ctx = P.Vp8Hip(0)
    try:
        nsrc, gold = _setup(P, ctx, name, n, monkeypatch)
        ctx.decode([(i, i, None) for i in range(n)], P.STAGE_ALL)
        # a copy of a frame that only exists as tiles, then the whole buffers of the copy and of the originals
        ctx._chk(ctx.L.vp8hip_frame_copy(ctx.h, n, 3), "frame_copy")
        for i in list(range(n)) + [n]:
            src = 3 if i == n else i
            d = bordered_area_equal(ctx.download_full(i), kept[src][4], ctx.g)
            assert not d, (i, d)
    finally:
        ctx.close()


def test_both_forms_in_one_batch(pkg, monkeypatch):
    """Frame buffers 0..n-1 from a large launch (tiles), one of them then overwritten by an upload (raster): digests and downloads
    over the whole run take whichever form holds each frame."""
    P = pkg
    name, n = "kf_640x360", 12
    ctx = P.Vp8Hip(0)
    try:
        nsrc, gold = _setup(P, ctx, name, n, monkeypatch)
        ctx.decode([(i, i, None) for i in range(n)], P.STAGE_ALL)
        want = [gold[i % nsrc] for i in range(n)]
        # frame 5 <- frame 2 in raster form, by way of the host
        buf = ctx.download_full(2)
        ctx.upload_frame(5, buf)
        want[5] = want[2]
        assert ctx.frames_md5(0, n) == want
        for i in range(n):
            assert P.planes_md5(*ctx.download_planes(i)) == want[i], i
    finally:
        ctx.close()


def test_eager_raster_knob(pkg, monkeypatch):
    P = pkg
    monkeypatch.setenv("VP8HIP_EAGER_RASTER", "1")
    ctx = P.Vp8Hip(0)
    try:
        nsrc, gold = _setup(P, ctx, "kf_640x360", 10, monkeypatch)
        ctx.decode([(i, i, None) for i in range(10)], P.STAGE_ALL)
        st = ctx.stats()
        assert st.fused == 1 and st.detile_pass == 1 and st.extend_ms > 0
        assert ctx.frames_md5(0, 10) == [gold[i % nsrc] for i in range(10)]
    finally:
        ctx.close()


def test_the_pools_come_with_their_first_user(pkg):
    """vp8hip_memory_usage: a context holds its IR slots from vp8hip_configure on; the tiled forms come with the first large launch,
    the raster forms with the first reader that needs one -- hashing the frames and taking them out as packed I420 do not (both read
    tiles), vp8hip_frame_download does --, and a small launch of key frames in a context that has only ever held tiles stays with
    the kernels that write tiles."""
    P = pkg
    w, h, frames = P.read_ivf(ivf_path("kf_640x360"))
    n = 600
    ctx = P.Vp8Hip(0)
    try:
        ctx.configure(w, h, n, len(frames))
        m = ctx.memory_usage()
        assert m["slots"] > 0 and m["raster_pool"] == 0 and m["tile_pool"] == 0 and m["block_pool"] == 0
        parser = P.Parser()
        for i, data in enumerate(frames):
            hdr = ctx.parse_into_slot(parser, data, i)
            parser.swap(hdr)
            ctx.upload(i)
        parser.close()
        gold = golden_md5("kf_640x360")
        ctx.decode([(i % len(frames), i, None) for i in range(n)], P.STAGE_ALL)          # a large launch: tiles
        m = ctx.memory_usage()
        assert m["tile_pool"] > 0 and m["raster_pool"] == 0
        assert ctx.frames_md5(0, n) == [gold[i % len(frames)] for i in range(n)]         # hashed as tiles
        import hashlib
        assert [hashlib.md5(f.tobytes()).hexdigest() for f in ctx.frames_i420(0, 70)] == [gold[i % len(frames)] for i in range(70)]
        m = ctx.memory_usage()
        assert m["raster_pool"] == 0 and m["packed_staging"] > 0
        ctx.decode([(i, i, None) for i in range(3)], P.STAGE_ALL)                        # a small launch: stays tiled here
        assert ctx.memory_usage()["raster_pool"] == 0
        assert ctx.frames_md5(0, 3) == gold[:3]
        assert P.planes_md5(*ctx.download_planes(1)) == gold[1]                          # a reader of the raster form
        m = ctx.memory_usage()
        assert m["raster_pool"] >= n * ctx.g.frame_size
        ctx.decode([(i, i, None) for i in range(3)], P.STAGE_ALL)                        # small launches write raster from now on
        assert [P.planes_md5(*ctx.download_planes(i)) for i in range(3)] == gold[:3]
    finally:
        ctx.close()


@pytest.mark.parametrize("n,pack_from", [(200, 100), (129, 65), (64, 64), (333, 70)])
def test_large_batches_are_hashed_from_a_packed_copy(pkg, monkeypatch, n, pack_from):
    """A digest-only batch of VP8HIP_MD5_PACK_FROM tiled frames and more (default 12,288: where the tiles' lines no longer fit the
    Infinity Cache) is hashed from a packed I420 copy (vp8_pack_i420_tiles_kernel + vp8_md5_kernel over a geometry without borders),
    smaller ones and sub-ranges below the threshold straight from the tiles: every digest is the reference decoder's either way
    (tests/test_gpu_bench_shapes.py has the 16,384-frame case with the default threshold)."""
    P = pkg
    monkeypatch.setenv("VP8HIP_MD5_PACK_FROM", str(pack_from))
    ctx = P.Vp8Hip(0)
    try:
        nsrc, gold = _setup(P, ctx, "kf_640x360", n, monkeypatch)
        ctx.decode([(i, i, None) for i in range(n)], P.STAGE_ALL)
        assert ctx.stats().fused == 1
        want = [gold[i % nsrc] for i in range(n)]
        for rep in range(2):
            assert ctx.frames_md5(0, n) == want
        assert ctx.frames_md5(5, n - 5) == want[5:]
        assert ctx.memory_usage()["raster_pool"] == 0
    finally:
        ctx.close()
