"""Dev aid (GPU): does the slow mode of a process follow its allocations?  One process, several (re)configurations."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from vp8_testlib import load_package, ivf_path
P = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
w, h, frames = P.read_ivf(ivf_path("kf_1920x1080"))
ctx = P.Vp8Hip(0)
for rnd in range(k):
    ctx.configure(w, h, n, n)
    parser = P.Parser()
    for i, data in enumerate(frames):
        hdr = ctx.parse_into_slot(parser, data, i); parser.swap(hdr); ctx.upload(i)
    for i in range(len(frames), n): ctx.ir_copy(i, i % len(frames))
    jobs = (P.Job * n)()
    for i in range(n): jobs[i].ir_slot, jobs[i].dst_fb = i, i
    ctx.decode_array(jobs, n, 7); ctx.sync()
    for _ in range(3): ctx.decode_array(jobs, n, 7)
    ctx.sync()
    print("configure", rnd, "recon_ms", " ".join(f"{ctx.stats(b).recon_ms:.2f}" for b in range(3)), flush=True)
ctx.close()
