"""GPU: the exact launches bench.py times, under pytest -m gpu (round 5; through round 4 these shapes were only checked by bench.py's
own gate).  1080p: 16,384 key frames per launch -- two frames per strand of 8 lanes, 17 rounds -- from slots in the device form the
host feeder uploaded (vp8_parser_decode_mbs_compact + one copy; the rest device-to-device copies), automatic kernel and shape
choice; 3840x2160: 4096 frames per launch at 32 lanes per strand.  EVERY frame's MD5 is computed on the device from the tiles the
launch left (vp8hip_frames_fetch_async -> vp8_md5_tiles_kernel) and compared with the reference decoder's listing
(tests/golden/*.md5, printed by oracle/_ref/ref_md5), and >= 64 frames spread over strands and waves are packed on the device,
downloaded and hashed on the HOST (hashlib), so that the device's hash is not the only witness.  Half the frames where the
device's memory is short (another process on it)."""
import hashlib

import numpy as np
import pytest

from vp8_testlib import golden_md5, ivf_path

pytestmark = pytest.mark.gpu


def _bench_launch(pkg, monkeypatch, name, n, per_frame_bytes, floor):
    for k in ("VP8HIP_RECON", "VP8HIP_SIMT_LGG", "VP8HIP_SIMT_WAVES"):
        monkeypatch.delenv(k, raising=False)
    import torch
    free, _total = torch.cuda.mem_get_info(0)
    while n > floor and n * per_frame_bytes > free * 0.92:
        n //= 2
    w, h, frames = pkg.read_ivf(ivf_path(name))
    gold = golden_md5(name)
    nsrc = len(frames)
    ctx = pkg.Vp8Hip(0)
    try:
        ctx.configure(w, h, n, n)
        parser = pkg.Parser()
        for i, data in enumerate(frames):
            hdr, _ = ctx.parse_into_slot_compact(parser, data, i)
            assert hdr.frame_type == 0
            parser.swap(hdr)
        parser.close()
        for i in range(nsrc, n):
            ctx.ir_copy(i, i % nsrc)
        jobs = (pkg.Job * n)()
        for i in range(n):
            jobs[i].ir_slot, jobs[i].dst_fb = i, i
            for k in range(4):
                jobs[i].ref_fb[k] = -1
        for _ in range(2):                                   # (back to back: the job tables rotate)
            ctx.decode_array(jobs, n, pkg.STAGE_ALL)
        ctx.sync()
        st = ctx.stats()
        assert st.fused == 1 and st.recon_waves == 1
        assert ctx.memory_usage()["raster_pool"] == 0        # the frames are tiles and nothing asked for more
        digests = ctx.frames_md5(0, n)
        bad = [i for i in range(n) if digests[i] != gold[i % nsrc]]
        assert not bad, (len(bad), bad[:8])
        rng = np.random.default_rng(5)
        sample = sorted(set([0, 1, 7, 8, 9, 63, 64, 65, n // 2 - 1, n // 2, n - 2, n - 1] + [int(v) for v in rng.integers(0, n, size=72)]))
        assert len(sample) >= 64
        for i in sample:
            assert hashlib.md5(ctx.frames_i420(i, 1)[0].tobytes()).hexdigest() == gold[i % nsrc], i
        return n
    finally:
        ctx.close()


def test_benchmark_launch_1080p_every_digest(pkg, monkeypatch):
    # slot 7.8 MB + tiles 3.42 MB per frame
    n = _bench_launch(pkg, monkeypatch, "kf_1920x1080", 16384, 7_840_000 + 3_430_000, 2048)
    assert n >= 2048


def test_benchmark_launch_4k_every_digest(pkg, monkeypatch):
    # slot 31.1 MB + tiles 13.6 MB per frame
    n = _bench_launch(pkg, monkeypatch, "kf_3840x2160", 4096, 31_200_000 + 13_600_000, 512)
    assert n >= 512


def _lockstep_streams(pkg, monkeypatch, name, n, floor):
    """n copies of an inter stream decoded in lock step, the way bench.py's inter probes and bin/batch_md5 --streams run them: a
    launch per frame position, every stream with its own IR slot and its own four frame buffers, every inter launch predicting from
    the TILES the launch before left (no raster form is ever made).  What it stands for in the reference: the frame lifecycle of
    vp8dx_receive_compressed_data (vp8/decoder/onyxd_if.c:318-706 -- decode, swap_frame_buffers, the reference counts) run for n
    decoders side by side.  Every shown frame of every stream is hashed on the device and compared with the reference decoder's
    listing; >= 64 frames spread over positions and streams are packed, downloaded and hashed on the host as well."""
    for k in ("VP8HIP_RECON", "VP8HIP_SIMT_LGG", "VP8HIP_SIMT_WAVES", "VP8HIP_PRED_TILES"):
        monkeypatch.delenv(k, raising=False)
    import torch
    free, _total = torch.cuda.mem_get_info(0)
    while n > floor and n * (7_840_000 + 4 * 3_430_000) > free * 0.9:
        n //= 2
    w, h, frames = pkg.read_ivf(ivf_path(name))
    gold = golden_md5(name)
    ctx = pkg.Vp8Hip(0)
    rng = np.random.default_rng(11)
    hosted = 0
    try:
        ctx.configure(w, h, 4 * n, n)                  # frame buffer b of stream i: b * n + i (a position's frames are neighbours)
        parser = pkg.Parser()
        shown = 0
        for f, data in enumerate(frames):
            hdr = ctx.parse_into_slot(parser, data, 0)
            ctx.upload(0)
            for i in range(1, n):
                ctx.ir_copy(i, 0)
            r = parser.refs
            jobs = (pkg.Job * n)()
            for i in range(n):
                jobs[i].ir_slot, jobs[i].dst_fb = i, r.new_idx * n + i
                for q, ref in enumerate((r.lst_idx, r.gld_idx, r.alt_idx)):
                    jobs[i].ref_fb[1 + q] = ref * n + i if hdr.frame_type else -1
            ctx.decode_array(jobs, n, pkg.STAGE_ALL)
            st = ctx.stats()
            assert st.fused == 1 and st.pred_tiles == (1 if hdr.frame_type else 0), (f, st.fused, st.pred_tiles)
            new = r.new_idx
            parser.swap(hdr)
            if not hdr.show_frame:
                continue
            digests = ctx.frames_md5(new * n, n)
            bad = [i for i in range(n) if digests[i] != gold[shown]]
            assert not bad, (f, len(bad), bad[:8])
            for i in sorted(set([0, n - 1] + [int(v) for v in rng.integers(0, n, size=max(6, 70 // len(frames) + 1))])):
                assert hashlib.md5(ctx.frames_i420(new * n + i, 1)[0].tobytes()).hexdigest() == gold[shown], (f, i)
                hosted += 1
            shown += 1
        assert shown == len(gold)
        assert hosted >= 64
        assert ctx.memory_usage()["raster_pool"] == 0       # chained through tiles: nothing ever asked for a raster form
        parser.close()
        return n
    finally:
        ctx.close()


def test_lockstep_streams_1080p_every_digest(pkg, monkeypatch):
    """SURVEY 8(d)'s config-3 input: p_1920x1080 (a key frame and nine P frames, mostly skipped macroblocks with sub-pixel vectors)."""
    assert _lockstep_streams(pkg, monkeypatch, "p_1920x1080", 4096, 512) >= 512


def test_lockstep_streams_dense_1080p_every_digest(pkg, monkeypatch):
    """The inter probe's stream: p_dense_1920x1080 (thirteen blocks with more than a first coefficient per macroblock, none skipped)."""
    assert _lockstep_streams(pkg, monkeypatch, "p_dense_1920x1080", 2048, 512) >= 512
