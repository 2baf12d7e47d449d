"""GPU (-m gpu): the HIP pixel path, called through the C ABI (include/vp8hip.h), against
 (1) the reference decoder's per-frame MD5s on every fixture,
 (2) the oracle, stage by stage and whole-buffer (borders included), on fixtures and on seeded random IR,
 (3) size-independent properties at the benchmark's full batch sizes."""
import ctypes

import numpy as np
import pytest

from vp8_testlib import (FIXTURES, bordered_area_equal, coded_area_equal, golden_md5, ivf_path, oracle_decode,
                         random_frame, synth_ir)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Vp8Hip(0)
    yield c
    c.close()


@pytest.fixture(params=["wave", "wave1cu", "wave_split", "wave1cu_split", "lane"], autouse=True)
def kernel_family(request, monkeypatch):
    """Every test runs with all kernel variants: "wave" = one wave per MB row with a frame pair spread over several
    CUs where the launch is small enough (granule hand-over through global memory), "wave1cu" = the same kernels with
    a frame pair on one CU (hand-over through LDS; what larger launches use), "lane" = one MB row per lane (what large launches
    run): launches with both stages run vp8_keyframe_kernel, or vp8_inter_pred_kernel + vp8_interframe_kernel when inter frames
    are among them (reconstruction + loop filter in one pass into macroblock-window tiles, then the tiled -> raster pass);
    single-stage launches run the wave-per-row kernels whatever the knob says.  "..._split": launches with inter frames
    always run vp8_inter_mb_kernel (every inter macroblock on its own) before the row-ordered kernel does the intra macroblocks
    -- by default only launches of up to 384 frames do; "wave1cu" never does.  VP8HIP_RECON / VP8HIP_XCU / VP8HIP_INTER_SPLIT are
    the library's tuning knobs that override the automatic choice (libvpx.opencl_amd/csrc/hip/vp8hip_launch.hip: launch_regime)."""
    monkeypatch.setenv("VP8HIP_RECON", "simt" if request.param.startswith("lane") else "wave")
    if request.param.startswith("wave1cu"):
        monkeypatch.setenv("VP8HIP_XCU", "0")
    else:
        monkeypatch.delenv("VP8HIP_XCU", raising=False)
    if request.param.endswith("_split"):
        monkeypatch.setenv("VP8HIP_INTER_SPLIT", "100000")
    elif request.param == "wave1cu":
        monkeypatch.setenv("VP8HIP_INTER_SPLIT", "0")
    else:
        monkeypatch.delenv("VP8HIP_INTER_SPLIT", raising=False)
    return request.param


@pytest.mark.parametrize("name", FIXTURES)
def test_fixture_md5_frame_by_frame(pkg, name):
    """decode_to_md5 parity: every shown frame equals the reference decoder's MD5."""
    assert pkg.decode_ivf_gpu(ivf_path(name), device=0) == golden_md5(name)


@pytest.mark.parametrize("name", ["kf_640x360", "p_split_352x288", "p_prof1_640x360", "p_prof3_640x360",
                                  "p_sharp_320x240", "p_odd_130x98", "kf_q0_176x144"])
def test_stages_against_oracle(pkg, ctx, name):
    """recon / recon+LF on the coded area, full pipeline on the whole buffer incl. borders.  References
    are re-seeded from the oracle every frame so a mismatch is attributed to the frame it occurs in."""
    w, h, frames = pkg.read_ivf(ivf_path(name))
    parser = pkg.Parser()
    obufs = None
    for data in frames[:6]:
        hdr, changed, mbs, coef, mvs = pkg.parse_to_numpy(parser, data)
        if changed:
            ctx.configure(hdr.width, hdr.height, 4, 1)
            g = ctx.g
            obufs = [np.zeros(g.frame_size, np.uint8) for _ in range(4)]
        r = parser.refs
        refs = (r.lst_idx, r.gld_idx, r.alt_idx)
        for idx in set(refs) - {r.new_idx}:
            ctx.upload_frame(idx, obufs[idx])
        ctx.fill_slot(0, hdr, mbs, coef, mvs)
        final = None
        for stages in (1, 3, 7):
            o = np.zeros(g.frame_size, np.uint8)
            oracle_decode(hdr, mbs, coef, mvs, o, tuple(obufs[i] for i in refs), stages)
            ctx.decode([(0, r.new_idx, refs)], stages)
            got = ctx.download_full(r.new_idx)
            d = coded_area_equal(got, o, g) if stages != 7 else bordered_area_equal(got, o, g)
            assert not d, (name, stages, d)
            final = o
        obufs[r.new_idx][:] = final
        parser.swap(hdr)
    parser.close()


@pytest.mark.parametrize("w,h", [(16, 16), (48, 32), (176, 144), (640, 368), (1000, 40), (33, 600)])
@pytest.mark.parametrize("inter,version,ftype", [(False, 0, 0), (True, 0, 0), (True, 1, 1), (True, 2, 0), (True, 3, 1)])
def test_random_ir_against_oracle(pkg, ctx, w, h, inter, version, ftype):
    """Seeded random IR (random modes, dense and sparse coefficients up to +-2047, random MVs incl. ones
    that need the UMV clamp, random segment / loop-filter parameters): whole buffer must match the oracle."""
    ctx.configure(w, h, 4, 1)
    g = ctx.g
    for seed in range(4):
        hdr, mbs, coef, mvs = synth_ir(w, h, seed * 7 + w + 3 * version, inter=inter, version=version,
                                       filter_type=ftype, dense=(0.15, 0.5, 0.9, 0.3)[seed], big=seed == 2)
        refs_np = [random_frame(g, 100 + seed * 3 + k) for k in range(3)]
        for k in range(3):
            ctx.upload_frame(1 + k, refs_np[k])
        ctx.fill_slot(0, hdr, mbs, coef, mvs)
        o = np.zeros(g.frame_size, np.uint8)
        oracle_decode(hdr, mbs, coef, mvs, o, tuple(refs_np), 7)
        ctx.decode([(0, 0, (1, 2, 3))], 7)
        d = bordered_area_equal(ctx.download_full(0), o, g)
        assert not d, (seed, d)


@pytest.mark.parametrize("w,h,n", [(176, 144, 40), (48, 80, 150), (640, 368, 12)])
def test_random_ir_batches_against_oracle(pkg, ctx, w, h, n):
    """Many DIFFERENT random key frames in one launch (different modes, quantisers, filter types and levels per job):
    the frames of a launch share waves -- lane halves in one family, strands of lanes in the other -- and must not
    influence each other."""
    ctx.configure(w, h, n, n)
    g = ctx.g
    irs = []
    for i in range(n):
        hdr, mbs, coef, mvs = synth_ir(w, h, 1000 + 13 * i + w, inter=False, version=i % 4, filter_type=i % 2,
                                       dense=(0.1, 0.5, 0.9)[i % 3], big=i % 5 == 0, segmented=i % 3 != 0)
        ctx.fill_slot(i, hdr, mbs, coef, mvs)
        irs.append((hdr, mbs, coef, mvs))
    ctx.decode([(i, i, None) for i in range(n)], 7)
    for i in range(n):
        o = np.zeros(g.frame_size, np.uint8)
        oracle_decode(*irs[i], o, (o, o, o), 7)
        d = bordered_area_equal(ctx.download_full(i), o, g)
        assert not d, (i, d)


def _batch(pkg, ctx, name, nframes):
    w, h, frames = pkg.read_ivf(ivf_path(name))
    gold = golden_md5(name)
    nsrc = len(frames)
    ctx.configure(w, h, nframes, nframes)
    parser = pkg.Parser()
    for i, data in enumerate(frames[:nframes]):
        hdr = ctx.parse_into_slot(parser, data, i)
        assert hdr.frame_type == 0
        parser.swap(hdr)
        ctx.upload(i)
    parser.close()
    for i in range(nsrc, nframes):
        ctx.ir_copy(i, i % nsrc)
    ctx.decode([(i, i, None) for i in range(nframes)], 7)
    ctx.sync()
    return gold, nsrc


@pytest.mark.parametrize("name,nframes", [("kf_odd_67x45", 700), ("kf_q0_176x144", 300), ("kf_640x360", 530)])
def test_batched_key_frames(pkg, ctx, name, nframes):
    """Many independent key frames in ONE launch (more jobs than workgroups: persistent loop, line-slot
    reuse across frames, frames with fewer MB rows than waves): every frame equals its reference MD5."""
    gold, nsrc = _batch(pkg, ctx, name, nframes)
    for i in list(range(0, nframes, 37)) + [nframes - 1, 255, 256, 257]:
        assert pkg.planes_md5(*ctx.download_planes(i)) == gold[i % nsrc], i


def test_full_size_1080p_batch_properties(pkg, ctx):
    """At the benchmark's size (1920x1080 x 512 frames per launch): replication invariance -- frame i and
    frame i+10 were decoded from copies of the same IR by different workgroups/waves and must be
    bit-identical over the WHOLE buffer; a sample is also checked against the reference MD5s."""
    n = 512
    gold, nsrc = _batch(pkg, ctx, "kf_1920x1080", n)
    for i in (0, 1, 9, 250, 256, 511):
        assert pkg.planes_md5(*ctx.download_planes(i)) == gold[i % nsrc], i
    a = ctx.download_full(3)
    for j in (13, 263, 503):
        assert np.array_equal(a, ctx.download_full(j)), j
    # idempotence: decoding the same jobs again into the same buffers changes nothing
    ctx.decode([(i, i, None) for i in range(n)], 7)
    assert np.array_equal(a, ctx.download_full(3))


def test_automatic_kernel_choice(pkg, ctx, kernel_family, monkeypatch):
    """Without the override a launch of more than 2 key frames per CU takes the lane-per-row kernels, a smaller one the
    wave-per-row kernels; both sides of the threshold produce the reference's frames."""
    if kernel_family != "wave":
        pytest.skip("one run is enough")
    monkeypatch.delenv("VP8HIP_RECON")
    for nframes in (40, 1100):
        gold, nsrc = _batch(pkg, ctx, "kf_q0_176x144", nframes)
        for i in (0, 1, nframes // 2, nframes - 1):
            assert pkg.planes_md5(*ctx.download_planes(i)) == gold[i % nsrc], (nframes, i)
    gold, nsrc = _batch(pkg, ctx, "kf_640x360", 900)
    for i in (0, 7, 450, 899):
        assert pkg.planes_md5(*ctx.download_planes(i)) == gold[i % nsrc], i


def test_cross_cu_small_launches(pkg, ctx, kernel_family):
    """Launches of up to 64 frame pairs spread every pair over S workgroups on different CUs (S shrinks as the launch
    grows: 17 for one 1080p pair, 32 / ceil(pairs / 8) at most, with eight instead of four waves per workgroup once
    there are fewer waves than rows); the rows hand their bottom lines over through tagged
    granules in global memory.  Odd frame counts, one pair per XCD, several pairs per XCD, and the launch sizes
    either side of the limit all give the reference's frames."""
    if kernel_family != "wave":
        pytest.skip("cross-CU variant only")
    # (fixture, frames per launch, waves per workgroup expected: 4 / 8 = cross-CU, 0 = do not care)
    # (a frame of nine macroblock rows gets no more waves from three CUs than from one: the library keeps it on one)
    for name, nframes, waves in (("kf_q0_176x144", 1, 0), ("kf_q0_176x144", 3, 0), ("kf_q0_176x144", 130, 0),
                                 ("kf_640x360", 1, 4), ("kf_640x360", 2, 4), ("kf_640x360", 17, 4), ("kf_640x360", 66, 4),
                                 ("kf_640x360", 200, 8), ("kf_640x360", 300, 0), ("kf_odd_67x45", 5, 0)):
        gold, nsrc = _batch(pkg, ctx, name, nframes)
        st = ctx.stats()
        if waves:
            assert st.recon_waves == waves and st.lf_waves == waves and st.workgroups % 8 == 0, (name, nframes, st.recon_waves)
        for i in range(nframes):
            assert pkg.planes_md5(*ctx.download_planes(i)) == gold[i % nsrc], (name, nframes, i)


def test_back_to_back_launches_and_stats_ring(pkg, ctx, kernel_family):
    """Launches are asynchronous and pipeline on the device: three launches without a host sync in between, then downloads
    -- which ask for the raster form of frames the lane family left as tiles -- must see the reference's frames; the per-launch
    kernel times of all three are still readable afterwards (vp8hip_get_stats_at)."""
    n = 64
    w, h, frames = pkg.read_ivf(ivf_path("kf_q0_176x144"))
    gold = golden_md5("kf_q0_176x144")
    ctx.configure(w, h, n, n)
    parser = pkg.Parser()
    for i, data in enumerate(frames):
        hdr = ctx.parse_into_slot(parser, data, i)
        parser.swap(hdr)
        ctx.upload(i)
    parser.close()
    for i in range(len(frames), n):
        ctx.ir_copy(i, i % len(frames))
    for rot in range(3):          # launch k decodes slot (i + k) into buffer i
        ctx.decode([((i + rot) % n, i, None) for i in range(n)], 7)
    for i in (0, 1, 17, n - 1):
        assert pkg.planes_md5(*ctx.download_planes(i)) == gold[((i + 2) % n) % len(frames)], i
    for back in range(3):
        st = ctx.stats(back)
        assert st.recon_ms > 0 and st.extend_ms >= 0
        assert st.recon_waves == (1 if kernel_family.startswith("lane") else st.recon_waves)
    ctx.join()
    ctx.sync()


def test_full_size_4k(pkg, ctx):
    gold, nsrc = _batch(pkg, ctx, "kf_3840x2160", 96)
    for i in (0, 1, 2, 47, 95):
        assert pkg.planes_md5(*ctx.download_planes(i)) == gold[i % nsrc], i


def test_abi_error_paths(pkg, ctx):
    ctx.configure(64, 64, 2, 1)
    with pytest.raises(RuntimeError):
        ctx.decode([(5, 0, None)], 7)              # slot out of range
    with pytest.raises(RuntimeError):
        ctx.decode([(0, 9, None)], 7)              # frame buffer out of range
    hdr, mbs, coef, mvs = synth_ir(64, 64, 1, inter=True)
    ctx.fill_slot(0, hdr, mbs, coef, mvs)
    with pytest.raises(RuntimeError):
        ctx.decode([(0, 0, (-1, -1, -1))], 7)      # inter frame without references
    with pytest.raises(RuntimeError):
        ctx.decode([(0, 1, (1, 1, 1))], 7)         # decoding into its own reference


@pytest.mark.parametrize("name,seed", [("p_lowrate_640x360", 1), ("p_split_352x288", 2), ("p_odd_130x98", 3), ("p_prof1_640x360", 4),
                                       ("p_arf_176x144", 5)])
def test_damaged_streams_still_match_the_oracle(pkg, ctx, name, seed):
    """Inter frames with random byte damage behind their frame header: whatever modes, MVs and coefficients the feeder
    reads out of them (or none: frames it rejects are dropped), the HIP path must neither fault nor hang and must
    still produce the oracle's frame from the same IR -- arbitrary in-range MVs incl. far-out clamped ones, dense
    garbage coefficients.  A clean launch afterwards shows the context is healthy."""
    rng = np.random.default_rng(seed)
    w, h, frames = pkg.read_ivf(ivf_path(name))
    parser = pkg.Parser()
    obufs, ndamaged, ndecoded = None, 0, 0
    for k, data in enumerate(frames[:24]):
        if k > 0 and rng.random() < 0.7 and len(data) > 40:
            bad = bytearray(data)
            for _ in range(int(rng.integers(1, 9))):
                bad[int(rng.integers(10, len(bad)))] = int(rng.integers(0, 256))
            data = bytes(bad)
            ndamaged += 1
        try:
            hdr, changed, mbs, coef, mvs = pkg.parse_to_numpy(parser, data)
        except ValueError:
            continue                                    # the feeder rejected the frame (and gave its buffer back)
        assert (hdr.width, hdr.height) == (w, h)        # bytes 0..9 (frame tag, key-frame size) are never damaged
        if changed or obufs is None:
            ctx.configure(hdr.width, hdr.height, 4, 1)
            g = ctx.g
            obufs = [np.zeros(g.frame_size, np.uint8) for _ in range(4)]
        r = parser.refs
        refs = (r.lst_idx, r.gld_idx, r.alt_idx)
        for idx in set(refs) - {r.new_idx}:
            ctx.upload_frame(idx, obufs[idx])
        ctx.fill_slot(0, hdr, mbs, coef, mvs)
        o = np.zeros(g.frame_size, np.uint8)
        oracle_decode(hdr, mbs, coef, mvs, o, tuple(obufs[i] for i in refs), 7)
        ctx.decode([(0, r.new_idx, refs if hdr.frame_type else None)], 7)
        d = bordered_area_equal(ctx.download_full(r.new_idx), o, g)
        assert not d, (name, k, d)
        obufs[r.new_idx][:] = o
        parser.swap(hdr)
        ndecoded += 1
    parser.close()
    assert ndamaged >= 5 and ndecoded >= 5
    gold, nsrc = _batch(pkg, ctx, "kf_odd_67x45", 3)
    for i in range(3):
        assert pkg.planes_md5(*ctx.download_planes(i)) == gold[i % nsrc]
