"""Dev aid: decisions per frame of the device entropy decoder (a library built with -DENT_STATS: tools/variant.sh entstats
-DENT_STATS; VP8HIP_LIB=...) and the time of a launch: cycles per decision of the slowest lane."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from vp8_testlib import load_package, ivf_path
name = sys.argv[1] if len(sys.argv) > 1 else "kf_1920x1080"
P = load_package()
w, h, frames = P.read_ivf(ivf_path(name))
parser = P.Parser(); efs = []
for d in frames:
    hdr, _ = parser.begin(d); efs.append(parser.export_entropy()); parser.swap(hdr)
ctx = P.Vp8Hip(); ctx.configure(w, h, 1, len(frames))
for rep in range(2):
    t = time.perf_counter(); st = ctx.entropy_decode(0, efs, frames); dt = time.perf_counter() - t
for i, s in enumerate(st):
    modes, toks = (int(s) & 0xffff) << 4, (int(s) >> 16) << 8
    print(f"frame {i}: {len(frames[i])} bytes, ~{modes} mode decisions, ~{toks} token decisions ({(modes+toks)/len(frames[i])/8:.2f} per bit)")
tot = max(((int(s) & 0xffff) << 4) + ((int(s) >> 16) << 8) for s in st)
print(f"launch {dt*1e3:.1f} ms -> {dt/tot*1e9:.1f} ns per decision of the longest frame")
