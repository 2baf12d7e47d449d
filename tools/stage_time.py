"""Dev aid (GPU): kernel times of the 1080p all-key-frame launch for the current library / env knobs.
   python3 tools/stage_time.py [frames=8192] [stages=7] [reps=3] [fixture=kf_1920x1080]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from vp8_testlib import load_package, ivf_path, golden_md5
P = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
stages = int(sys.argv[2]) if len(sys.argv) > 2 else 7
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
name = sys.argv[4] if len(sys.argv) > 4 else "kf_1920x1080"
w, h, frames = P.read_ivf(ivf_path(name))
gold = golden_md5(name)
ctx = P.Vp8Hip(0); ctx.configure(w, h, n, n)
parser = P.Parser()
for i, data in enumerate(frames):
    hdr = ctx.parse_into_slot(parser, data, i); parser.swap(hdr); ctx.upload(i)
for i in range(len(frames), n): ctx.ir_copy(i, i % len(frames))
jobs = (P.Job * n)()
for i in range(n): jobs[i].ir_slot, jobs[i].dst_fb = i, i
ctx.decode_array(jobs, n, stages); ctx.sync()
ok = all(P.planes_md5(*ctx.download_planes(i)) == gold[i % len(frames)] for i in (0, 1, 9, n // 2 + 3, n - 1)) if stages == 7 else None
r = l = e = 0.0
for _ in range(reps):
    ctx.decode_array(jobs, n, stages); ctx.sync(); st = ctx.stats(); r += st.recon_ms; l += st.lf_ms; e += st.extend_ms
print(f"{name} frames={n} stages={stages} md5_ok={ok} recon {r/reps:.3f} ms  lf {l/reps:.3f} ms  extend {e/reps:.3f} ms  workgroups {st.workgroups}  env " +
      " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("VP8HIP")))
ctx.close()
