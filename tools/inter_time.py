"""Dev aid (GPU): the inter-frame probe of bench.py alone, for A/B runs of library variants (VP8HIP_LIB=... python3 tools/inter_time.py [jobs])."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from vp8_testlib import load_package
r = bench.inter_frame_probe(load_package(), 0, n=int(sys.argv[1]) if len(sys.argv) > 1 else 8192)
print(os.environ.get("VP8HIP_LIB", "product"), r["md5_ok"], r["ms_per_launch"], r["chained"]["ms_per_launch"], r["roofline"]["frac"])
