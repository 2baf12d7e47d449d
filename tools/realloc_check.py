import os, sys
sys.path.insert(0, "tests")
from vp8_testlib import load_package, ivf_path
P = load_package()
n = 8192
w, h, frames = P.read_ivf(ivf_path("kf_1920x1080"))
for trial in range(6):
    ctx = P.Vp8Hip(0); ctx.configure(w, h, n, n)
    parser = P.Parser()
    for i, data in enumerate(frames):
        hdr = ctx.parse_into_slot(parser, data, i); parser.swap(hdr); ctx.upload(i)
    for i in range(len(frames), n): ctx.ir_copy(i, i % len(frames))
    jobs = (P.Job * n)()
    for i in range(n): jobs[i].ir_slot, jobs[i].dst_fb = i, i
    ctx.decode_array(jobs, n, 7); ctx.sync()
    out = []
    for _ in range(3):
        ctx.decode_array(jobs, n, 7); ctx.sync(); st = ctx.stats(); out.append(f"{st.recon_ms:.1f}/{st.lf_ms:.1f}")
    print("trial", trial, " ".join(out), flush=True)
    parser.close(); ctx.close()
