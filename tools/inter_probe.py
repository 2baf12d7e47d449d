#!/usr/bin/env python3
"""Dev tool: bench.py's inter-frame probe on its own.  usage: inter_probe.py [jobs per launch] [fixture] [frame]"""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from vp8_testlib import load_package
P = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
name = sys.argv[2] if len(sys.argv) > 2 else "p_dense_1920x1080"
k = int(sys.argv[3]) if len(sys.argv) > 3 else 2
print(json.dumps(bench.inter_frame_probe(P, 0, n=n, name=name, k=k)))
