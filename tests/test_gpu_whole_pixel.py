"""GPU: macroblocks whose motion vector has no fraction -- the reference's copy branch (vp8_build_inter16x16_predictors_mb,
vp8/common/reconinter.c:402-417: `if (mv.as_int & 0x00070007) sixtap else vp8_copy_mem16x16`; :419-441 for the chroma vector
derived from it; vp8_copy_mem16x16 / vp8_copy_mem8x8 at :22-110).  Round 6: the prediction kernels sort such macroblocks into lists
of their own (luma and chroma separately: a whole luma vector halves to a chroma vector that may end on a half pixel) and copy --
no halo rows, no filter passes.

Streams written by tests/vp8_writer.py in which well over half of the macroblocks stand still or move by whole pixels (ZEROMV,
NEWMV with vectors that are multiples of eight, and whatever NEARESTMV / NEARMV inherit from those), a key frame and three inter
frames, decoded by the lane-per-row kernels with the prediction from raster references and from tiled ones: every frame buffer,
borders included, against the oracle's, and the listing against the reference decoder's where its binary is there."""
import os
import subprocess

import numpy as np
import pytest

from vp8_testlib import ROOT, bordered_area_equal, oracle_decode, synth_ir
from vp8_writer import write_inter_frame, write_ivf, write_key_frame

pytestmark = pytest.mark.gpu
REF_MD5 = os.path.join(ROOT, "oracle", "_ref", "ref_md5")


def whole_pixel_sequence(w, h, seed, share=0.8, version=0, keep_split=True):
    rng = np.random.default_rng(seed)
    hdr, mbs, coef, mvs = synth_ir(w, h, seed, inter=False, dense=0.3, version=version)
    frames = [write_key_frame(hdr, mbs, coef)]
    whole = total = 0
    cols = (w + 15) // 16
    for i in range(3):
        hdr, mbs, coef, mvs = synth_ir(w, h, seed * 10 + i + 1, inter=True, dense=(0.3, 0.05, 0.5)[i], version=version)
        hdr.refresh_last = 1
        hdr.show_frame = 1
        for k in range(mbs.shape[0]):
            if mbs[k, 2] != 0 and mbs[k, 0] == 9 and not keep_split:
                mbs[k, 0] = 8               # (tools/whole_pixel_time.py: SPLITMV macroblocks go a 4x4 block per lane and would be all one measures)
                mvs[k, :] = mvs[k, 0]
            if mbs[k, 2] == 0 or mbs[k, 0] == 9 or rng.random() > share:
                continue
            if rng.random() < 0.4:
                mbs[k, 0], mv = 7, (0, 0)                                   # ZEROMV
            else:
                mbs[k, 0] = 8                                               # NEWMV, whole pixels, up to 24 away (and so past the edges)
                mv = (int(rng.integers(-24, 25)) * 8, int(rng.integers(-24, 25)) * 8)
            mvs[k, :] = mv
        data, mbs2, mvs2 = write_inter_frame(hdr, mbs, coef, mvs)
        one = (mbs2[:, 2] != 0) & (mbs2[:, 0] != 9)
        whole += int(np.sum(one & ((mvs2[:, 0, 0] & 7) == 0) & ((mvs2[:, 0, 1] & 7) == 0)))
        total += mbs2.shape[0]
        frames.append(data)
    assert whole >= 0.5 * total, (whole, total)
    return frames


@pytest.mark.parametrize("w,h,seed,version", [(176, 144, 3, 0), (640, 368, 4, 0), (130, 98, 5, 0), (352, 288, 6, 1), (320, 192, 7, 3)])
@pytest.mark.parametrize("pred_tiles", [0, 2])
def test_whole_pixel_vectors_are_copied_bit_exact(pkg, monkeypatch, tmp_path, w, h, seed, version, pred_tiles):
    P = pkg
    monkeypatch.setenv("VP8HIP_RECON", "simt")
    monkeypatch.setenv("VP8HIP_PRED_TILES", str(pred_tiles))
    frames = whole_pixel_sequence(w, h, seed, version=version)
    g = P.geom(w, h)
    bufs = [np.zeros(g.frame_size, np.uint8) for _ in range(4)]
    ctx = P.Vp8Hip(0)
    mine = []
    try:
        ctx.configure(w, h, 4, 1)
        parser, oparser = P.Parser(), P.Parser()
        for f, data in enumerate(frames):
            hdr_o, _changed, mbs, coef, mvs = P.parse_to_numpy(oparser, data)
            ro = oparser.refs
            oracle_decode(hdr_o, mbs, coef, mvs, bufs[ro.new_idx], (bufs[ro.lst_idx], bufs[ro.gld_idx], bufs[ro.alt_idx]))
            hdr = ctx.parse_into_slot(parser, data, 0)
            ctx.upload(0)
            r = parser.refs
            ctx.decode([(0, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx) if hdr.frame_type else None)], P.STAGE_ALL)
            st = ctx.stats()
            assert st.fused == 1 and st.pred_tiles == (1 if hdr.frame_type and pred_tiles else 0), (f, st.fused, st.pred_tiles)
            if pred_tiles == 0 or f == len(frames) - 1:            # (a download gives the frame its raster form as well)
                d = bordered_area_equal(ctx.download_full(r.new_idx), bufs[ro.new_idx], g)
                assert not d, (f, d)
            else:
                assert ctx.frames_md5(r.new_idx, 1)[0] == P.frame_md5(bufs[ro.new_idx], g, w, h), f
            mine.append(P.frame_md5(bufs[ro.new_idx], g, w, h))
            parser.swap(hdr); oparser.swap(hdr_o)
        parser.close(); oparser.close()
    finally:
        ctx.close()
    if os.path.exists(REF_MD5):
        ivf, out = tmp_path / "s.ivf", tmp_path / "s.md5"
        write_ivf(ivf, w, h, frames)
        r = subprocess.run([REF_MD5, str(ivf), str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert [l.split()[0] for l in open(out)] == mine
