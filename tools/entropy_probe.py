"""Dev aid: rate of the device entropy decoder (vp8hip_entropy_decode) on N copies of a key-frame fixture's frames, by lanes per
wave.   python3 tools/entropy_probe.py [frames] [fixture] [lanes ...]   (run on the GPU box; no oracle involved)"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from vp8_testlib import load_package, ivf_path
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
name = sys.argv[2] if len(sys.argv) > 2 else "kf_1920x1080"
lanes = [int(x) for x in sys.argv[3:]] or [0]
P = load_package()
w, h, frames = P.read_ivf(ivf_path(name))
parser = P.Parser()
efs = []
for d in frames:
    hdr, _ = parser.begin(d); efs.append(parser.export_entropy()); parser.swap(hdr)
reps = (n + len(frames) - 1) // len(frames)
F = (frames * reps)[:n]; E = (efs * reps)[:n]
arr = (P.EntropyFrame * n)()
off = 0
for i in range(n):
    ctypes.memmove(ctypes.byref(arr[i]), ctypes.byref(E[i]), ctypes.sizeof(P.EntropyFrame)); arr[i].data_off = off; off += len(F[i])
blob = b"".join(F)
for lp in lanes:
    if lp: os.environ["VP8HIP_ENTROPY_LANES"] = str(lp)
    ctx = P.Vp8Hip(); ctx.configure(w, h, 1, n)
    ts = []
    for rep in range(3):
        t = time.perf_counter()
        ctx._chk(ctx.L.vp8hip_entropy_decode(ctx.h, 0, n, ctypes.byref(arr), blob, len(blob)), "entropy"); ctx.sync()
        ts.append(time.perf_counter() - t)
    print(f"{name} frames {n} lanes/wave {lp or 'auto'}: {min(ts)*1e3:.1f} ms per launch (incl. {len(blob)/1e6:.0f} MB H2D from pageable memory) "
          f"= {n/min(ts):.0f} frames/s = {n/min(ts)*w*h/1e9:.2f} Gpix/s", flush=True)
    ctx.close()
