"""Dev aid (GPU): the time of the benchmark's launch, for A/B runs of library variants (VP8HIP_LIB=... python3 tools/kf_time.py [frames] [reps])."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vp8_testlib import load_package, ivf_path
P = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
w, h, frames = P.read_ivf(ivf_path("kf_1920x1080"))
ctx = P.Vp8Hip(0)
ctx.configure(w, h, n, n)
parser = P.Parser()
for i, d in enumerate(frames):
    hdr, _ = ctx.parse_into_slot_compact(parser, d, i); parser.swap(hdr)
for i in range(len(frames), n):
    ctx.ir_copy(i, i % len(frames))
jobs = (P.Job * n)()
for i in range(n):
    jobs[i].ir_slot, jobs[i].dst_fb = i, i
    for k in range(4): jobs[i].ref_fb[k] = -1
ts = []
for r in range(reps + 2):
    ctx.sync(); t = time.perf_counter()
    ctx.decode_array(jobs, n, P.STAGE_ALL); ctx.sync()
    if r >= 2: ts.append((time.perf_counter() - t) * 1e3)
ts.sort()
print(f"{os.environ.get('VP8HIP_LIB', 'product')}: {n} frames: min {ts[0]:.2f} median {ts[len(ts)//2]:.2f} max {ts[-1]:.2f} ms")
ctx.close()
