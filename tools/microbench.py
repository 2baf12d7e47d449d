#!/usr/bin/env python3
"""Kernel-cost breakdown on synthetic IR (GPU): how much does each MB type cost?  Dev tool."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from vp8_testlib import load_package, synth_ir
P = load_package()
W, H = (int(v) for v in os.environ.get("MB_SIZE", "1920x1088").split("x"))
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = P.Vp8Hip(0); ctx.configure(W, H, F, F)
n = ctx.nmb

def make(kind):
    hdr, mbs, coef, mvs = synth_ir(W, H, 5, dense=0.5)
    hdr.filter_level = 20; hdr.segmentation_enabled = 0; hdr.mode_ref_lf_delta_enabled = 0; hdr.sharpness_level = 0
    hdr.frame_type = 0
    mbs[:] = 0; coef[:] = 0
    if kind.startswith("dc"):
        mbs[:, 0] = 0
    elif kind.startswith("tm"):
        mbs[:, 0] = 3; mbs[:, 1] = 3
    elif kind.startswith("bpred"):
        mbs[:, 0] = 4
        rng = np.random.default_rng(1); mbs[:, 40:56] = rng.integers(0, 10, size=(n, 16))
    if kind.endswith("skip"):
        mbs[:, 3] = 1
    else:
        rng = np.random.default_rng(2)
        coef[:] = rng.integers(-30, 31, size=coef.shape)
        mbs[:, 8:33] = 15
        if mbs[0, 0] != 4:
            coef[:, 0:256:16] = 0  # Y DC comes from Y2 (column-major index 0 is DC)
        else:
            mbs[:, 32] = 0; coef[:, 384:] = 0
    return hdr, mbs, coef, mvs

jobs = (P.Job * F)()
for i in range(F):
    jobs[i].ir_slot, jobs[i].dst_fb = i, i
def make_inter(split):
    hdr, mbs, coef, mvs = synth_ir(W, H, 7, inter=True, dense=0.3, segmented=False)
    hdr.filter_level = 20; hdr.mode_ref_lf_delta_enabled = 0; hdr.sharpness_level = 0
    if not split:
        sel = mbs[:, 0] == 9
        mbs[sel, 0] = 8
        mvs[sel, :] = mvs[sel, :1]
    return hdr, mbs, coef, mvs

KINDS = ("dc_skip", "dc_dense", "tm_dense", "bpred_skip", "bpred_dense", "inter_nosplit", "inter_mixed")
if len(sys.argv) > 2:
    KINDS = tuple(sys.argv[2].split(","))
from vp8_testlib import random_frame
for k in range(3):
    ctx.upload_frame(F - 1 - k, random_frame(ctx.g, 50 + k)) if False else None
for kind in KINDS:
    inter = kind.startswith("inter")
    hdr, mbs, coef, mvs = make_inter(kind == "inter_mixed") if inter else make(kind)
    if inter:   # every job reads the same three reference buffers (the last three of the pool), decodes elsewhere
        for i in range(F):
            jobs[i].ref_fb[1], jobs[i].ref_fb[2], jobs[i].ref_fb[3] = F - 1, F - 2, F - 3
        njobs = F - 3
    else:
        njobs = F
    ctx.fill_slot(0, hdr, mbs, coef, mvs)
    for i in range(1, F): ctx.ir_copy(i, 0)
    ctx.decode_array(jobs, njobs, 7); ctx.sync()
    r = l = 0.0
    for _ in range(3):
        ctx.decode_array(jobs, njobs, 7); st = ctx.stats(); r += st.recon_ms; l += st.lf_ms
    r /= 3; l /= 3
    per_mb_ns = r * 1e6 / (F * n)
    print(f"{kind:12s} recon {r:8.3f} ms  lf {l:8.3f} ms   recon {per_mb_ns*256/1:8.1f} ns per MB per CU   ({F*n*1217/r/1e6:7.1f} GB/s alg)")
