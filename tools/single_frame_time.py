"""Dev aid (GPU): one frame per launch -- the single-stream case of vpx_codec_decode --, kernel times by frame type.
   [VP8HIP_LIB=...] python3 tools/single_frame_time.py [fixture] [loops]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from vp8_testlib import load_package, ivf_path, golden_md5
P = load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "p_1920x1080"
loops = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w, h, frames = P.read_ivf(ivf_path(name))
gold = golden_md5(name)
ctx = P.Vp8Hip(0); ctx.configure(w, h, 4, 1)
acc = {0: [0, 0.0, 0.0, 0.0], 1: [0, 0.0, 0.0, 0.0]}
ok = True
for it in range(loops):
    parser = P.Parser(); shown = 0
    for data in frames:
        hdr = ctx.parse_into_slot(parser, data, 0); ctx.upload(0)
        r = parser.refs
        ctx.sync(); t = time.perf_counter()
        ctx.decode([(0, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx) if hdr.frame_type else None)], P.STAGE_ALL); ctx.sync()
        dt = (time.perf_counter() - t) * 1e3
        st = ctx.stats()
        a = acc[1 if hdr.frame_type else 0]; a[0] += 1; a[1] += st.recon_ms; a[2] += st.lf_ms; a[3] += dt
        new = r.new_idx; parser.swap(hdr)
        if hdr.show_frame:
            ok &= P.planes_md5(*ctx.download_planes(new)) == gold[shown]; shown += 1
    parser.close()
for k, nm in ((0, "key"), (1, "inter")):
    n, rc, lf, wall = acc[k]
    if n: print(f"{os.environ.get('VP8HIP_LIB', 'product')}: {name} {nm} frames ({n}): recon {rc / n:.3f} ms, loop filter {lf / n:.3f} ms, launch to sync {wall / n:.3f} ms; md5 {'ok' if ok else 'DIFFERS'}")
ctx.close()
