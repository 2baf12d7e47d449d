"""GPU: launches with inter frames at the sizes where vp8hip_decode changes its kernels by itself (no knobs set): up to 384
frames -- inter macroblocks by vp8_inter_mb_kernel first --, 385..512 -- the row-ordered kernels alone --, more than two frames per
CU -- all inter predictions by vp8_inter_pred_kernel, then residual + loop filter one macroblock row per lane.  Real 1080p P frames (dense fixture), every job decoding
the same frame from the same references into its own buffer: all outputs equal the reference decoder's MD5 and each other,
whole buffers (borders included) are the same on every path."""
import numpy as np
import pytest

from vp8_testlib import golden_md5, ivf_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def decoded(pkg):
    """context with frames 0..1 of the dense P stream decoded; frame 2 (inter) parsed into slot 1"""
    P = pkg
    name, k = "p_dense_1920x1080", 2
    w, h, frames = P.read_ivf(ivf_path(name))
    ctx = P.Vp8Hip(0)
    ctx.configure(w, h, 4 + 700, 2)
    parser = P.Parser()
    for data in frames[:k]:
        hdr = ctx.parse_into_slot(parser, data, 0)
        ctx.upload(0)
        r = parser.refs
        ctx.decode([(0, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx) if hdr.frame_type else None)], P.STAGE_ALL)
        ctx.sync()
        parser.swap(hdr)
    hdr = ctx.parse_into_slot(parser, frames[k], 1)
    assert hdr.frame_type == 1
    ctx.upload(1)
    r = parser.refs
    yield P, ctx, (r.lst_idx, r.gld_idx, r.alt_idx), golden_md5(name)[k]
    parser.close()
    ctx.close()


def _launch(P, ctx, refs, n):
    jobs = (P.Job * n)()
    for i in range(n):
        jobs[i].ir_slot, jobs[i].dst_fb = 1, 4 + i
        jobs[i].ref_fb[1], jobs[i].ref_fb[2], jobs[i].ref_fb[3] = refs
    ctx.decode_array(jobs, n, P.STAGE_ALL)
    ctx.sync()
    return ctx.stats()


def test_every_launch_size_regime_gives_the_reference_frame(decoded, monkeypatch):
    P, ctx, refs, gold = decoded
    for v in ("VP8HIP_INTER_SPLIT", "VP8HIP_RECON", "VP8HIP_XCU"):
        monkeypatch.delenv(v, raising=False)
    whole = None
    for n, lane_lf in ((1, False), (3, False), (384, False), (500, False), (513, True), (700, True)):
        for i in range(n):                                   # nothing left over from the previous launch
            ctx.upload_frame(4 + i, np.zeros(ctx.g.frame_size, np.uint8)) if i in (0, n // 2, n - 1) else None
        st = _launch(P, ctx, refs, n)
        assert (st.lf_waves == 1) == lane_lf, (n, st.lf_waves)       # more than two frames per CU: one macroblock row per lane ...
        assert st.fused == int(lane_lf), (n, st.fused)               # ... prediction kernel + vp8_interframe_kernel
        for i in sorted({0, n // 2, n - 1}):
            assert P.planes_md5(*ctx.download_planes(4 + i)) == gold, (n, i)
        full = ctx.download_full(4 + n - 1)
        if whole is None:
            whole = full
        assert np.array_equal(full, whole), n                # borders included, bit for bit the same on every path


@pytest.mark.parametrize("knobs", [{"VP8HIP_RECON": "simt"}])                                   # the lane-per-row kernels, their pass on stream 2
def test_launches_chained_without_sync_wait_for_the_raster_pass(pkg, monkeypatch, knobs):
    """A launch whose tiled -> raster pass still runs (or has not even been launched) on the second stream, followed at once --
    no sync, no download -- by launches of inter frames that read those frame buffers as references: the library has to join
    the pass first (vp8hip_launch.hip: `reads_pending`).  Eight streams side by side, three frames each."""
    P = pkg
    for k in ("VP8HIP_RECON", "VP8HIP_INTER_SPLIT"):
        monkeypatch.delenv(k, raising=False)
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    name, n = "p_dense_1920x1080", 8
    w, h, frames = P.read_ivf(ivf_path(name))
    gold = golden_md5(name)
    ctx = P.Vp8Hip(0)
    try:
        ctx.configure(w, h, 4 * n, 3)
        parser = P.Parser()
        launches = []
        for f in range(3):
            hdr = ctx.parse_into_slot(parser, frames[f], f)
            ctx.upload(f)
            r = parser.refs
            launches.append((f, hdr.frame_type, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx)))
            parser.swap(hdr)
        for f, ftype, new_idx, refs in launches:                # back to back: nothing waits in between
            jobs = (P.Job * n)()
            for i in range(n):
                jobs[i].ir_slot, jobs[i].dst_fb = f, 4 * i + new_idx
                for q in range(3):
                    jobs[i].ref_fb[1 + q] = 4 * i + refs[q] if ftype else -1
            ctx.decode_array(jobs, n, P.STAGE_ALL)
        ctx.sync()
        last_new = launches[-1][2]
        for i in range(n):
            assert P.planes_md5(*ctx.download_planes(4 * i + last_new)) == gold[2], (knobs, i)
        parser.close()
    finally:
        ctx.close()


def test_independent_launches_run_past_the_raster_pass_dependent_ones_wait(pkg, monkeypatch):
    """Launches with inter frames by the prediction kernel + vp8_interframe_kernel (forced at this size), back to back without a
    sync: frame 2 of eight streams into their decoder buffers, frame 2 AGAIN into spare buffers -- reads only frames decoded
    long ago: the library lets it start while the first launch's tiled -> raster pass still runs --, then frame 3, which
    predicts from what the first launch wrote and has to wait for that pass.  Everything equals the reference decoder's frames."""
    P = pkg
    for k in ("VP8HIP_INTER_SPLIT",):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("VP8HIP_RECON", "simt")
    name, n = "p_dense_1920x1080", 8
    w, h, frames = P.read_ivf(ivf_path(name))
    gold = golden_md5(name)
    ctx = P.Vp8Hip(0)
    try:
        ctx.configure(w, h, 5 * n, 4)
        parser = P.Parser()
        plan = []
        for f in range(4):
            hdr = ctx.parse_into_slot(parser, frames[f], f)
            ctx.upload(f)
            r = parser.refs
            plan.append((f, hdr.frame_type, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx)))
            parser.swap(hdr)

        def launch(f, spare=False):
            _, ftype, new_idx, refs = plan[f]
            jobs = (P.Job * n)()
            for i in range(n):
                jobs[i].ir_slot, jobs[i].dst_fb = f, (4 * n + i) if spare else 4 * i + new_idx
                for q in range(3):
                    jobs[i].ref_fb[1 + q] = 4 * i + refs[q] if ftype else -1
            ctx.decode_array(jobs, n, P.STAGE_ALL)
            return ctx.stats()

        launch(0); launch(1)
        ctx.sync()
        assert launch(2).fused == 1
        launch(2, spare=True)
        launch(3)
        ctx.sync()
        for i in range(n):
            assert P.planes_md5(*ctx.download_planes(4 * n + i)) == gold[2], i
            assert P.planes_md5(*ctx.download_planes(4 * i + plan[3][2])) == gold[3], i
            assert P.planes_md5(*ctx.download_planes(4 * i + plan[2][2])) == gold[2], i
        parser.close()
    finally:
        ctx.close()


@pytest.mark.parametrize("sizes", [(520, 3, 700, 520), (700, 700, 1, 600), (2, 640, 640, 640), (600, 8, 8, 520)])
def test_mixed_launch_sizes_chained_without_sync(pkg, monkeypatch, sizes):
    """Four frames of the dense P stream, each decoded for as many of the streams as `sizes` says (the first sizes[f] streams), back
    to back without a sync, with no knob set: every launch picks its kernels by its own size -- wave-per-row with and without the
    order-free inter kernel, prediction kernel + vp8_interframe_kernel with the raster pass held back -- and every one reads what
    some launch before it wrote.  A stream's frame f is only launched where its frame f - 1 was; all of them must be the
    reference decoder's."""
    P = pkg
    for k in ("VP8HIP_RECON", "VP8HIP_INTER_SPLIT"):
        monkeypatch.delenv(k, raising=False)
    name = "p_dense_1920x1080"
    w, h, frames = P.read_ivf(ivf_path(name))
    gold = golden_md5(name)
    n = max(sizes)
    ctx = P.Vp8Hip(0)
    try:
        ctx.configure(w, h, 4 * n, 4)
        parser = P.Parser()
        plan = []
        for f in range(4):
            hdr = ctx.parse_into_slot(parser, frames[f], f)
            ctx.upload(f)
            r = parser.refs
            plan.append((hdr.frame_type, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx)))
            parser.swap(hdr)
        count = n
        done = []
        for f, (ftype, new_idx, refs) in enumerate(plan):
            count = min(count, sizes[f])
            jobs = (P.Job * count)()
            for i in range(count):
                jobs[i].ir_slot, jobs[i].dst_fb = f, 4 * i + new_idx
                for q in range(3):
                    jobs[i].ref_fb[1 + q] = 4 * i + refs[q] if ftype else -1
            ctx.decode_array(jobs, count, P.STAGE_ALL)
            done.append(count)
        ctx.sync()
        for i in sorted({0, done[-1] // 2, done[-1] - 1}):
            assert P.planes_md5(*ctx.download_planes(4 * i + plan[3][1])) == gold[3], (sizes, i)
        if done[2] > done[3]:            # streams that stopped after frame 2
            i = done[2] - 1
            assert P.planes_md5(*ctx.download_planes(4 * i + plan[2][1])) == gold[2], (sizes, i)
        parser.close()
    finally:
        ctx.close()


def test_key_and_inter_frames_in_one_launch(pkg, monkeypatch):
    """One launch of the prediction kernel + vp8_interframe_kernel with key frames AND inter frames among its jobs (streams at
    different points: some start over with their key frame while the others decode a P frame): the key frames go the key-frame way
    inside the inter-frame kernel, nobody's tiles or references get mixed up."""
    P = pkg
    for k in ("VP8HIP_INTER_SPLIT",):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("VP8HIP_RECON", "simt")
    name, n = "p_dense_1920x1080", 12
    w, h, frames = P.read_ivf(ivf_path(name))
    gold = golden_md5(name)
    ctx = P.Vp8Hip(0)
    try:
        ctx.configure(w, h, 4 * n, 3)
        parser = P.Parser()
        plan = []
        for f in range(3):
            hdr = ctx.parse_into_slot(parser, frames[f], f)
            ctx.upload(f)
            r = parser.refs
            plan.append((hdr.frame_type, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx)))
            parser.swap(hdr)

        def job(j, i, f):
            ftype, new_idx, refs = plan[f]
            j.ir_slot, j.dst_fb = f, 4 * i + new_idx
            for q in range(3):
                j.ref_fb[1 + q] = 4 * i + refs[q] if ftype else -1

        for f in (0, 1):                                      # all streams up to frame 1
            jobs = (P.Job * n)()
            for i in range(n):
                job(jobs[i], i, f)
            ctx.decode_array(jobs, n, P.STAGE_ALL)
        jobs = (P.Job * n)()                                  # odd streams: frame 2 (inter); even streams: frame 0 again (key)
        for i in range(n):
            job(jobs[i], i, 2 if i & 1 else 0)
        ctx.decode_array(jobs, n, P.STAGE_ALL)
        assert ctx.stats().fused == 1
        ctx.sync()
        for i in range(n):
            f = 2 if i & 1 else 0
            assert P.planes_md5(*ctx.download_planes(4 * i + plan[f][1])) == gold[f], i
        parser.close()
    finally:
        ctx.close()
