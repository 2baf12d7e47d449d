"""GPU: inter prediction from reference frames that are there as TILES (vp8_inter_pred_tiles_kernel, round 5).

What it stands for in the reference: vp8_build_inter_predictors_mb (vp8/common/reconinter.c:560-606) reading xd->pre -- the previous
frame buffer WITH its extended borders (vp8_extend_mb_row, vp8_yv12_extend_frame_borders).  A large launch leaves its frames as
macroblock-window tiles without borders; the launch after it predicts from those directly: two loads per source row, the window
shift decided row by row, rows and columns beyond the frame replicated by address clamps and a fix-up of the loaded bytes.  Checked
here: seeded random IR (every inter mode incl. SPLITMV, motion vectors up to 66 pixels outside the frame on every side, six-tap /
bilinear / full-pixel, both loop filters) whose three references are tile-only frames, whole buffers (borders included) against
the oracle, and the same launches with the raster-reading kernel (VP8HIP_PRED_TILES=0) bit for bit."""
import numpy as np
import pytest

from vp8_testlib import bordered_area_equal, golden_md5, ivf_path, oracle_decode, synth_ir

pytestmark = pytest.mark.gpu


def _tile_only_refs(P, ctx, w, h, seed):
    """Three random key frames decoded by the lane-per-row kernel into frame buffers 1..3: tiles, no raster form.  Returns the
    oracle's frame buffers (borders extended) for them."""
    g = ctx.g
    outs = []
    for k in range(3):
        hdr, mbs, coef, mvs = synth_ir(w, h, 500 + 11 * seed + k, inter=False, dense=0.4)
        o = np.zeros(g.frame_size, np.uint8)
        oracle_decode(hdr, mbs, coef, mvs, o, (None, None, None), 7)
        ctx.fill_slot(1 + k, hdr, mbs, coef, mvs)
        outs.append(o)
    ctx.decode([(1 + k, 1 + k, None) for k in range(3)], 7)
    st = ctx.stats()
    assert st.fused == 1
    return outs


@pytest.mark.parametrize("w,h", [(16, 16), (32, 48), (48, 32), (176, 144), (640, 368), (1000, 40), (33, 600)])
@pytest.mark.parametrize("version,ftype", [(0, 0), (1, 1), (2, 0), (3, 1)])
def test_random_inter_ir_from_tiled_references(pkg, monkeypatch, w, h, version, ftype):
    P = pkg
    monkeypatch.setenv("VP8HIP_RECON", "simt")             # (the library reads its knobs when a context is configured)
    ctx = P.Vp8Hip(0)
    try:
        ctx.configure(w, h, 5, 4)
        g = ctx.g
        for seed in range(3):
            refs = _tile_only_refs(P, ctx, w, h, seed)
            hdr, mbs, coef, mvs = synth_ir(w, h, seed * 5 + w + 7 * version, inter=True, version=version, filter_type=ftype,
                                           dense=(0.2, 0.6, 0.35)[seed], big=seed == 1)
            o = np.zeros(g.frame_size, np.uint8)
            oracle_decode(hdr, mbs, coef, mvs, o, tuple(refs), 7)
            ctx.fill_slot(0, hdr, mbs, coef, mvs)
            ctx._chk(ctx.L.vp8hip_set_pred_tiles(ctx.h, 1), "vp8hip_set_pred_tiles")
            ctx.decode([(0, 0, (1, 2, 3))], 7)
            st = ctx.stats()
            # (frames one macroblock wide stay with the raster reader: a chroma strip reaches past both vertical edges there)
            assert st.fused == 1 and st.pred_tiles == (1 if w > 16 else 0), (st.fused, st.pred_tiles)
            got = ctx.download_full(0)
            d = bordered_area_equal(got, o, g)
            assert not d, (seed, "tiles", d)
            # the same launch from the references' raster form (the library converts them first): the other kernel, the same bytes
            ctx._chk(ctx.L.vp8hip_set_pred_tiles(ctx.h, 0), "vp8hip_set_pred_tiles")
            ctx.decode([(0, 4, (1, 2, 3))], 7)
            st = ctx.stats()
            assert st.fused == 1 and st.pred_tiles == 0
            assert np.array_equal(ctx.download_full(4), got), seed
    finally:
        ctx.close()


@pytest.mark.parametrize("w,h", [(176, 144), (640, 368)])
@pytest.mark.parametrize("mode", [1, 2])
def test_references_in_mixed_forms(pkg, monkeypatch, w, h, mode):
    """A launch whose `last` frame exists only as tiles while its golden frame was UPLOADED (raster form only: VP8_SET_REFERENCE, or
    a frame a small launch decoded): the tile reader runs, the uploaded frame is given its tiled form once (vp8_retile_kernel) and
    keeps its raster bytes; both modes that read tiles, against the oracle and against the raster reader."""
    P = pkg
    monkeypatch.setenv("VP8HIP_RECON", "simt")
    ctx = P.Vp8Hip(0)
    try:
        ctx.configure(w, h, 6, 4)
        g = ctx.g
        refs = _tile_only_refs(P, ctx, w, h, 4)
        # the golden frame comes from outside: its own pixels, borders extended as vp8_yv12_extend_frame_borders leaves them
        hdr_g, mbs_g, coef_g, mvs_g = synth_ir(w, h, 4242, inter=False, dense=0.5)
        gold = np.zeros(g.frame_size, np.uint8)
        oracle_decode(hdr_g, mbs_g, coef_g, mvs_g, gold, (None, None, None), 7)
        ctx.upload_frame(2, gold)
        refs[1] = gold
        hdr, mbs, coef, mvs = synth_ir(w, h, 77 + w, inter=True, dense=0.4, big=True)
        o = np.zeros(g.frame_size, np.uint8)
        oracle_decode(hdr, mbs, coef, mvs, o, tuple(refs), 7)
        ctx.fill_slot(0, hdr, mbs, coef, mvs)
        ctx._chk(ctx.L.vp8hip_set_pred_tiles(ctx.h, mode), "vp8hip_set_pred_tiles")
        for rep in range(2):            # (the second launch finds the golden frame in both forms)
            ctx.decode([(0, 4 + rep, (1, 2, 3))], 7)
            st = ctx.stats()
            assert st.fused == 1 and st.pred_tiles == 1, (st.fused, st.pred_tiles)
            d = bordered_area_equal(ctx.download_full(4 + rep), o, g)
            assert not d, (rep, d)
        assert np.array_equal(ctx.download_full(2), gold)          # the raster form was kept
        ctx._chk(ctx.L.vp8hip_set_pred_tiles(ctx.h, 0), "vp8hip_set_pred_tiles")
        ctx.decode([(0, 0, (1, 2, 3))], 7)
        assert ctx.stats().pred_tiles == 0
        assert np.array_equal(ctx.download_full(0), ctx.download_full(4))
    finally:
        ctx.close()


def test_chained_streams_never_make_a_raster_form(pkg, monkeypatch):
    """Eight copies of the 1080p inter stream in lock step through the lane-per-row kernels, a launch per position, nothing downloaded
    in between: every inter launch predicts from the tiles the launch before left (stats.pred_tiles), the raster pool is never
    allocated, and the frames -- hashed from their tiles on the device -- are the reference decoder's."""
    P = pkg
    monkeypatch.setenv("VP8HIP_RECON", "simt")
    monkeypatch.delenv("VP8HIP_PRED_TILES", raising=False)
    name, n = "p_dense_1920x1080", 8
    w, h, frames = P.read_ivf(ivf_path(name))
    gold = golden_md5(name)
    ctx = P.Vp8Hip(0)
    try:
        ctx.configure(w, h, 4 * n, len(frames))
        parser = P.Parser()
        for f, data in enumerate(frames):
            hdr = ctx.parse_into_slot(parser, data, f)
            ctx.upload(f)
            r = parser.refs
            jobs = (P.Job * n)()
            for i in range(n):
                jobs[i].ir_slot, jobs[i].dst_fb = f, 4 * i + r.new_idx
                for q, ref in enumerate((r.lst_idx, r.gld_idx, r.alt_idx)):
                    jobs[i].ref_fb[1 + q] = 4 * i + ref if hdr.frame_type else -1
            ctx.decode_array(jobs, n, P.STAGE_ALL)
            st = ctx.stats()
            assert st.fused == 1 and st.pred_tiles == (1 if hdr.frame_type else 0), (f, st.fused, st.pred_tiles)
            new = r.new_idx
            parser.swap(hdr)
            if hdr.show_frame:
                for i in (0, n - 1):
                    assert ctx.frames_md5(4 * i + new, 1)[0] == gold[f], (f, i)
        assert ctx.memory_usage()["raster_pool"] == 0
        parser.close()
    finally:
        ctx.close()


@pytest.mark.parametrize("name", ["p_split_352x288", "p_prof1_640x360", "p_prof2_640x360", "p_prof3_640x360", "p_odd_130x98",
                                  "p_arf_176x144", "p_sharp_320x240"])
def test_fixture_streams_through_tiled_references(pkg, monkeypatch, name):
    """Whole fixture streams (SPLITMV, bilinear and full-pixel profiles, odd sizes, golden / alt-ref references with hidden frames)
    with every frame decoded by the lane-per-row kernels and every inter frame predicted from tiles: the per-frame listing is the
    reference decoder's."""
    monkeypatch.setenv("VP8HIP_RECON", "simt")
    monkeypatch.setenv("VP8HIP_PRED_TILES", "2")        # (the frames are downloaded one by one, so they have a raster form as well: prefer the tiles)
    assert pkg.decode_ivf_gpu(ivf_path(name), device=0) == golden_md5(name)
