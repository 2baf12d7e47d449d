import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from vp8_testlib import load_package, ivf_path
P = load_package()
stage = int(sys.argv[1]); n = int(sys.argv[2])
w, h, frames = P.read_ivf(ivf_path("kf_1920x1080"))
ctx = P.Vp8Hip(0); ctx.configure(w, h, n, n)
parser = P.Parser()
for i, data in enumerate(frames):
    hdr = ctx.parse_into_slot(parser, data, i); parser.swap(hdr); ctx.upload(i)
for i in range(len(frames), n): ctx.ir_copy(i, i % len(frames))
ctx.decode([(i, i, None) for i in range(n)], stage); ctx.sync()
print("done", stage, n)
