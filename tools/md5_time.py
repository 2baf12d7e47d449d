"""Dev aid (GPU): the time of the MD5 kernel over n frames left as tiles by one large launch (VP8HIP_LIB=... python3 tools/md5_time.py [frames ...])."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from vp8_testlib import load_package, ivf_path, golden_md5
P = load_package()
sizes = [int(a) for a in sys.argv[1:]] or [16384, 64]
n = max(sizes)
w, h, frames = P.read_ivf(ivf_path("kf_1920x1080"))
gold = golden_md5("kf_1920x1080")
ctx = P.Vp8Hip(0)
ctx.configure(w, h, max(n, 600), len(frames))
parser = P.Parser()
for i, d in enumerate(frames):
    hdr, _ = ctx.parse_into_slot_compact(parser, d, i); parser.swap(hdr)
N = max(n, 600)
jobs = (P.Job * N)()
for i in range(N):
    jobs[i].ir_slot, jobs[i].dst_fb = i % len(frames), i
    for k in range(4): jobs[i].ref_fb[k] = -1
ctx.decode_array(jobs, N, P.STAGE_ALL); ctx.sync()
for m in sizes:
    ts = []
    for r in range(4):
        t = time.perf_counter(); got = ctx.frames_md5(0, m); ts.append((time.perf_counter() - t) * 1e3)
    ok = all(got[i] == gold[i % len(frames)] for i in range(m))
    print(f"{os.environ.get('VP8HIP_LIB', 'product')}: md5 of {m} frames: {min(ts):.2f} ms (digests {'ok' if ok else 'DIFFER'})")
ctx.close()
