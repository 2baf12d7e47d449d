import os, sys, time
sys.path.insert(0, "tests")
from vp8_testlib import load_package, ivf_path
P = load_package()
for name in ("kf_1920x1080", "p_1920x1080", "kf_640x360"):
    w, h, frames = P.read_ivf(ivf_path(name))
    ctx = P.Vp8Hip(0); ctx.configure(w, h, 4, 1)
    parser = P.Parser()
    for i, data in enumerate(frames[:6]):
        hdr = ctx.parse_into_slot(parser, data, 0); ctx.upload(0)
        r = parser.refs; refs = (r.lst_idx, r.gld_idx, r.alt_idx)
        job = [(0, r.new_idx, refs if hdr.frame_type else None)]
        ctx.decode(job, 7); ctx.sync()
        t = time.perf_counter()
        for _ in range(10): ctx.decode(job, 7)
        ctx.sync(); dt = (time.perf_counter() - t) / 10 * 1e3
        st = ctx.stats()
        if i in (0, 5): print(f"{name} frame {i} type {hdr.frame_type}: {dt:.2f} ms per launch; recon {st.recon_ms:.2f} lf {st.lf_ms:.2f} ext {st.extend_ms:.2f}")
        parser.swap(hdr)
    ctx.close()
