"""Dev aid (GPU): the inter-frame probes of bench.py alone, for A/B runs of library variants
   [VP8HIP_LIB=...] python3 tools/inter_time.py [jobs] [fixture] [frame] [--json]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from vp8_testlib import load_package
a = [v for v in sys.argv[1:] if not v.startswith("--")]
n = int(a[0]) if a else 8192
name = a[1] if len(a) > 1 else "p_dense_1920x1080"
k = int(a[2]) if len(a) > 2 else 2
r = bench.inter_frame_probe(load_package(), 0, n=n, name=name, k=k)
if "--json" in sys.argv:
    print(json.dumps(r))
print(os.environ.get("VP8HIP_LIB", "product"), name, k, n, "md5", r["md5_ok"], r["chained"]["md5_ok"], "ms", r["ms_per_launch"], r["chained"]["ms_per_launch"],
      "frac", r["roofline"]["frac"], r["chained"]["roofline_frac"], r["references_read_as"][:6], r["chained"]["references_read_as"][:6])
