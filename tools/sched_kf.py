"""Dev aid (GPU, diagnostic library built with -DVP8_STAMPS): where the waves of vp8_keyframe_kernel ran and in which role."""
import ctypes, os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from vp8_testlib import load_package, ivf_path
P = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
w, h, frames = P.read_ivf(ivf_path("kf_1920x1080"))
ctx = P.Vp8Hip(0); ctx.configure(w, h, n, n)
parser = P.Parser()
for i, data in enumerate(frames):
    hdr = ctx.parse_into_slot(parser, data, i); parser.swap(hdr); ctx.upload(i)
for i in range(len(frames), n): ctx.ir_copy(i, i % len(frames))
jobs = (P.Job * n)()
for i in range(n): jobs[i].ir_slot, jobs[i].dst_fb = i, i
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    ctx.decode_array(jobs, n, 7); ctx.sync(); st = ctx.stats()
    NW = 16 + 16384 + 4 * 4096
    buf = (ctypes.c_uint * NW)()
    ctx.L.vp8hip_debug_sched.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    rc = ctx.L.vp8hip_debug_sched(ctx.h, buf, NW)
    nw = st.workgroups * 2
    log = [(buf[16 + 16384 + 4 * b], buf[16 + 16384 + 4 * b + 1], buf[16 + 16384 + 4 * b + 2] & 0xff, buf[16 + 16384 + 4 * b + 2] >> 8, buf[16 + 16384 + 4 * b + 3]) for b in range(nw)]
    by_simd = collections.defaultdict(list)
    items = collections.defaultdict(dict)
    fields = collections.defaultdict(set)
    for hw, xcc, role, seen, item in log:
        key = ((hw >> 4) & 3) | (((hw >> 8) & 0xff) << 2) | ((xcc & 15) << 10)
        by_simd[key].append(role)
        items[key][role] = item
        for name, lo, nb in (("wave", 0, 4), ("simd", 4, 2), ("pipe", 6, 2), ("cu", 8, 4), ("sh", 12, 1), ("se", 13, 3), ("tg", 16, 4), ("vm", 20, 4), ("queue", 24, 3), ("state", 27, 3), ("me", 30, 2)):
            fields[name].add((hw >> lo) & ((1 << nb) - 1))
        fields["xcc"].add(xcc)
    combos = collections.Counter(tuple(sorted(v)) for v in by_simd.values())
    print(f"launch {rep}: rc={rc} kernel {st.recon_ms:.2f} ms, waves {nw}, distinct SIMD keys {len(by_simd)}, role combinations per SIMD: {dict(combos)}")
    same = sum(1 for v in items.values() if len(v) == 2 and v[0] == v[1])
    near = sum(1 for v in items.values() if len(v) == 2 and abs(v[0] - v[1]) <= 8)
    xcd_of = collections.defaultdict(set)
    for key, v in items.items():
        for role, it in v.items(): xcd_of[key >> 10].add((role, it % 8))
    print(f"   SIMDs whose luma and chroma wave work on the same strands: {same}, within 8 items: {near}; (role, item % 8) per XCD: { {k: len(v) for k, v in xcd_of.items()} }")
    if rep > 2: continue
    print("   work counters", buf[0], buf[1], " field values:", {k: sorted(v) for k, v in fields.items()})
ctx.close()
