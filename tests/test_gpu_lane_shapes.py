"""GPU (-m gpu): every launch shape of the lane-per-row kernel family, through the C ABI.

The host picks the lanes per strand G = 2 .. 64 from the launch size (libvpx.opencl_amd/csrc/hip/vp8hip_launch.hip), so
small test launches always ran G = 64 (one strand per wave) while the benchmark's 8192-frame launch runs G = 8:
eight strands per wave, where the DPP `wave_shr:1` hand-over crosses strand boundaries and is gated off for the
first lane of every strand, strands take jobs q, q + nstrands, ..., and the three tile sets rotate.  These tests
force every G (VP8HIP_SIMT_LGG) and few waves (VP8HIP_SIMT_WAVES) so that strands carry several jobs, on content that
differs per strand:

 * seeded random IR against the oracle, whole buffer incl. borders (decode_mb_row order, vp8/decoder/decodframe.c:334-436;
   loop filter order, vp8/common/loopfilter.c:265-299), sizes on both sides of cols = 2G+2, filtered and unfiltered
   frames mixed in one launch;
 * the key-frame fixtures at frame counts that fill several waves, against the reference decoder's MD5s;
 * the benchmark's own shape at full size (8192 x 1080p, or what fits): replication invariance over >= 64 frames
   spread over strands / waves / scratch sets, three launches back to back.
"""
import numpy as np
import pytest

from vp8_testlib import bordered_area_equal, golden_md5, ivf_path, oracle_decode, synth_ir

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Vp8Hip(0)
    yield c
    c.close()


@pytest.fixture(params=[1, 2, 3, 4, 5, 6])
def lgG(request):
    return request.param


@pytest.fixture
def lane_shape(lgG, monkeypatch):
    """Reconstruction and loop filter in one kernel (vp8_keyframe_kernel), what large all-key-frame launches run, at every G."""
    monkeypatch.setenv("VP8HIP_RECON", "simt")
    monkeypatch.setenv("VP8HIP_SIMT_LGG", str(lgG))
    return lgG, "fused"


def _waves_for(n, lgG, jobs_per_strand):
    spw = 64 >> lgG
    return max(1, n // (spw * jobs_per_strand))


# cols = 3 / 11 / 40: below 2G+2 for most G, around it, above it for G <= 16
@pytest.mark.parametrize("w,h,n", [(48, 80, 160), (176, 144, 96), (640, 368, 24)])
def test_random_ir_batches_every_shape(pkg, ctx, lane_shape, monkeypatch, w, h, n):
    lg, _ = lane_shape
    # strands carry about three jobs each (at least one wave; G = 64 and small n: one strand takes them all)
    monkeypatch.setenv("VP8HIP_SIMT_WAVES", str(_waves_for(n, lg, 3)))
    ctx.configure(w, h, n, n)
    g = ctx.g
    irs = []
    for i in range(n):
        hdr, mbs, coef, mvs = synth_ir(w, h, 5000 + 17 * i + w + lg, inter=False, version=i % 4, filter_type=(i // 2) % 2,
                                       dense=(0.1, 0.5, 0.9)[i % 3], big=i % 7 == 0, segmented=i % 3 != 0)
        if i % 4 == 1:
            hdr.filter_level = 0           # unfiltered frames among filtered ones
        ctx.fill_slot(i, hdr, mbs, coef, mvs)
        irs.append((hdr, mbs, coef, mvs))
    for rot in range(3):                   # three launches back to back: the scratch sets / job tables rotate
        ctx.decode([((i + rot) % n, i, None) for i in range(n)], 7)
    for i in range(n):
        o = np.zeros(g.frame_size, np.uint8)
        oracle_decode(*irs[(i + 2) % n], o, (o, o, o), 7)
        d = bordered_area_equal(ctx.download_full(i), o, g)
        assert not d, (lane_shape, i, d)


@pytest.mark.parametrize("name,n", [("kf_odd_67x45", 260), ("kf_q0_176x144", 130), ("kf_640x360", 70)])
def test_fixture_batches_every_shape(pkg, ctx, lane_shape, monkeypatch, name, n):
    lg, _ = lane_shape
    monkeypatch.setenv("VP8HIP_SIMT_WAVES", str(_waves_for(n, lg, 2)))
    w, h, frames = pkg.read_ivf(ivf_path(name))
    gold = golden_md5(name)
    nsrc = len(frames)
    ctx.configure(w, h, n, n)
    parser = pkg.Parser()
    for i, data in enumerate(frames[:n]):
        hdr = ctx.parse_into_slot(parser, data, i)
        parser.swap(hdr)
        ctx.upload(i)
    parser.close()
    for i in range(nsrc, n):
        ctx.ir_copy(i, i % nsrc)
    ctx.decode([(i, i, None) for i in range(n)], 7)
    for i in range(n):
        assert pkg.planes_md5(*ctx.download_planes(i)) == gold[i % nsrc], (lane_shape, i)


def test_benchmark_shape_full_size(pkg, monkeypatch):
    """The launch bench.py times: 1080p all-key-frame stream, as many frames as it uses (8192) or as fit, automatic
    kernel and shape choice.  Frames i and i + 10k decode copies of the same IR on different strands, waves and scratch
    sets: >= 64 of them, spread over the launch, must equal the reference MD5 and each other byte for byte (borders
    included); three launches back to back rotate the scratch sets and job tables."""
    for k in ("VP8HIP_RECON", "VP8HIP_SIMT_LGG", "VP8HIP_SIMT_WAVES"):
        monkeypatch.delenv(k, raising=False)
    import torch
    free, _total = torch.cuda.mem_get_info(0)
    name = "kf_1920x1080"
    w, h, frames = pkg.read_ivf(ivf_path(name))
    gold = golden_md5(name)
    nsrc = len(frames)
    per_frame = 3_428_352 + 7_700_000 + 3 * 3_423_000 + 4096
    n = 8192
    while n > 1024 and n * per_frame > free * 0.9:
        n //= 2
    ctx = pkg.Vp8Hip(0)
    try:
        ctx.configure(w, h, n, n)
        parser = pkg.Parser()
        for i, data in enumerate(frames):
            hdr = ctx.parse_into_slot(parser, data, i)
            parser.swap(hdr)
            ctx.upload(i)
        parser.close()
        for i in range(nsrc, n):
            ctx.ir_copy(i, i % nsrc)
        jobs = (pkg.Job * n)()
        for i in range(n):
            jobs[i].ir_slot, jobs[i].dst_fb = i, i
        # launches of this size try the loop filter as luma + chroma kernels side by side (launches 0, 1) and as one kernel
        # (launch 2) and keep the faster: every one of these is checked, the later ones are whatever won
        for trial in range(3):
            ctx.decode_array(jobs, n, 7)
            ctx.sync()
            for i in (0, 9, n // 2 + 3, n - 1):
                assert pkg.planes_md5(*ctx.download_planes(i)) == gold[i % nsrc], (trial, i)
        for _ in range(3):
            ctx.decode_array(jobs, n, 7)
        ctx.sync()
        rng = np.random.default_rng(11)
        sample = sorted(set([0, 1, 7, 8, 9, 63, 64, 65, n // 2 - 1, n // 2, n - 2, n - 1]
                            + [int(v) for v in rng.integers(0, n, size=72)]))
        assert len(sample) >= 64
        first = {}
        for i in sample:
            buf = ctx.download_full(i)
            assert pkg.frame_md5(buf, ctx.g, w, h) == gold[i % nsrc], i
            k = i % nsrc
            if k in first:
                assert np.array_equal(buf, first[k][1]), (i, first[k][0])
            else:
                first[k] = (i, buf)
    finally:
        ctx.close()
