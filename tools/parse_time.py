"""Dev aid: host feeder time per frame (entropy decode into the device form of the IR) by number of token-partition threads.
   python3 tools/parse_time.py [fixture ...]      (CPU only; run it where the cores are)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from vp8_testlib import load_package, ivf_path
P = load_package()
for name in sys.argv[1:] or ["kf_8part_1920x1080", "kf_1920x1080"]:
    _, _, frames = P.read_ivf(ivf_path(name))
    for th in (1, 2, 4, 8):
        p = P.Parser(); p.set_threads(th); per = [0.0] * len(frames); parts = 0
        for rep in range(3):
            for i, data in enumerate(frames):
                h, _ = p.begin(data); parts = h.num_token_partitions
                n = h.mb_cols * h.mb_rows
                if rep == 0 and i == 0:
                    mbx = np.zeros((n, 128), np.uint8); blocks = np.ones((n * 24, 16), np.int16)
                    mvs = np.zeros((n, 16, 2), np.int16)
                t = time.perf_counter()
                p.decode_mbs_compact(mbx.ctypes.data, blocks.ctypes.data, n * 24, mvs.ctypes.data); p.swap(h)
                per[i] += (time.perf_counter() - t) / 3 * 1e3
        print(f"{name} ({parts} partitions) threads={th}: " + " ".join(f"{x:.2f}" for x in per[:4]) + " ms per frame")
        p.close()
