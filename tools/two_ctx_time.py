"""Dev aid (GPU): do the two kernels of launches with inter frames overlap when TWO contexts (a stream each) share the device?
n jobs of p_dense_1920x1080 frame 2 in all: one context with n jobs against two contexts with n / 2 jobs each, their launches
issued alternately; unchained (every launch the same jobs) and chained (every launch predicts from what the launch before wrote).
   [VP8HIP_LIB=...] python3 tools/two_ctx_time.py [jobs] [reps] [contexts]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from vp8_testlib import load_package, ivf_path, golden_md5
P = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nctx = int(sys.argv[3]) if len(sys.argv) > 3 else 2
name, k = "p_dense_1920x1080", 2
w, h, frames = P.read_ivf(ivf_path(name))
gold = golden_md5(name)


def make(m):
    ctx = P.Vp8Hip(0)
    ctx.configure(w, h, 4 + 2 * m, 2 + m)
    parser = P.Parser()
    for data in frames[:k]:
        hdr = ctx.parse_into_slot(parser, data, 0); ctx.upload(0)
        r = parser.refs
        ctx.decode([(0, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx) if hdr.frame_type else None)], P.STAGE_ALL); ctx.sync()
        parser.swap(hdr)
    ctx.parse_into_slot(parser, frames[k], 1); ctx.upload(1)
    r = parser.refs
    fwd, back = (P.Job * m)(), (P.Job * m)()
    for i in range(m):
        ctx.ir_copy(2 + i, 1)
        ctx.L.vp8hip_frame_copy(ctx.h, 4 + 2 * i, r.lst_idx)
        fwd[i].ir_slot, fwd[i].dst_fb = 2 + i, 5 + 2 * i
        fwd[i].ref_fb[1], fwd[i].ref_fb[2], fwd[i].ref_fb[3] = 4 + 2 * i, r.gld_idx, r.alt_idx
        back[i].ir_slot, back[i].dst_fb = 2 + i, 4 + 2 * i
        back[i].ref_fb[1], back[i].ref_fb[2], back[i].ref_fb[3] = 5 + 2 * i, r.gld_idx, r.alt_idx
    ctx.decode_array(fwd, m, P.STAGE_ALL); ctx.decode_array(fwd, m, P.STAGE_ALL); ctx.sync()
    ok = ctx.frames_md5(5 + 2 * (m // 2), 1)[0] == gold[k]
    parser.close()
    return ctx, fwd, back, ok


def run(cs, m):
    for c, fwd, back, ok in cs: c.decode_array(fwd, m, P.STAGE_ALL)
    for c, *_ in cs: c.sync()
    t = time.perf_counter()
    for _ in range(reps):
        for c, fwd, back, ok in cs: c.decode_array(fwd, m, P.STAGE_ALL)
    for c, *_ in cs: c.sync()
    un = (time.perf_counter() - t) / reps * 1e3
    for c, fwd, back, ok in cs: c.decode_array(back, m, P.STAGE_ALL); c.decode_array(fwd, m, P.STAGE_ALL)
    for c, *_ in cs: c.sync()
    t = time.perf_counter()
    for _ in range(reps):
        for c, fwd, back, ok in cs: c.decode_array(back, m, P.STAGE_ALL)
        for c, fwd, back, ok in cs: c.decode_array(fwd, m, P.STAGE_ALL)
    for c, *_ in cs: c.sync()
    ch = (time.perf_counter() - t) / (2 * reps) * 1e3
    return un, ch


tag = os.environ.get("VP8HIP_LIB", "product")
if nctx == 1:
    cs = [make(n)]
    un, ch = run(cs, n)
    print(f"{tag}: ONE context, {n} jobs: {un:.2f} ms per launch unchained, {ch:.2f} chained (md5 {'ok' if cs[0][3] else 'DIFFERS'})")
else:
    m = n // nctx
    os.environ.setdefault("VP8HIP_SIMT_WAVES", str(1024 // nctx))
    os.environ.setdefault("VP8HIP_SIMT_LGG", "3")
    cs = [make(m) for _ in range(nctx)]
    un, ch = run(cs, m)
    print(f"{tag}: {nctx} contexts x {m} jobs (SIMT_WAVES {os.environ['VP8HIP_SIMT_WAVES']}, LGG {os.environ['VP8HIP_SIMT_LGG']}): "
          f"{un:.2f} ms per round of launches unchained, {ch:.2f} chained (md5 {'ok' if all(c[3] for c in cs) else 'DIFFERS'})")
for c, *_ in cs: c.close()
