"""Dev aid: N copies of an inter-frame stream decoded side by side -- position t of all of them in one launch -- with the entropy
decoder on the device (vp8hip_entropy_decode: only the frame headers are read on the host, once per position) or with the host
feeder (one parse per position, the IR copied to the other streams' slots on the device: what N streams would cost a host is N
times the printed feeder time).  Every shown frame of every stream is hashed on the device and compared with the reference
decoder's listing.   python3 tools/streams_probe.py [streams] [fixture] [host|device]   (run on the GPU box)"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from vp8_testlib import load_package, ivf_path, golden_md5
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
name = sys.argv[2] if len(sys.argv) > 2 else "p_1920x1080"
mode = sys.argv[3] if len(sys.argv) > 3 else "device"
P = load_package()
w, h, frames = P.read_ivf(ivf_path(name))
gold = golden_md5(name)
ctx = P.Vp8Hip(); ctx.configure(w, h, 4 * n, n)
parser = P.Parser()
shown = bad = 0
t_feed = t_all = 0.0
arr = (P.EntropyFrame * n)()
for rep in range(2):                                   # (the first pass pays the allocations)
    parser.close(); parser = P.Parser(); shown = 0
    ctx.sync(); t0 = time.perf_counter(); t_feed = 0.0
    for data in frames:
        tf = time.perf_counter()
        if mode == "device":
            hdr, _ = parser.begin(data)
            ef = parser.export_entropy()
            sz = ctypes.sizeof(P.EntropyFrame)
            rec = np.tile(np.frombuffer(bytes(ef), np.uint8), n).reshape(n, sz)          # the same description n times ...
            rec[:, 64:72] = (np.arange(n, dtype=np.uint64) * len(data)).view(np.uint8).reshape(n, 8)   # ... each with its own data_off
            ctypes.memmove(arr, rec.ctypes.data, n * sz)
            blob = data * n
            t_feed += time.perf_counter() - tf
            ctx._chk(ctx.L.vp8hip_entropy_decode(ctx.h, 0, n, ctypes.byref(arr), blob, len(blob)), "entropy")
        else:
            hdr = ctx.parse_into_slot(parser, data, 0)
            t_feed += time.perf_counter() - tf
            ctx.upload(0)
            for i in range(1, n): ctx.ir_copy(i, 0)
        r = parser.refs
        ctx.decode([(i, r.new_idx * n + i, (r.lst_idx * n + i, r.gld_idx * n + i, r.alt_idx * n + i)) for i in range(n)], P.STAGE_ALL)
        parser.swap(hdr)
        if hdr.show_frame:
            digs = ctx.frames_md5(parser.refs.show_idx * n, n)
            bad += sum(d != gold[shown] for d in digs)
            shown += 1
    ctx.sync(); t_all = time.perf_counter() - t0
print(f"{name} x {n} streams, {len(frames)} frames each, entropy decode on the {mode}: {t_all:.3f} s = {n*len(frames)/t_all:.0f} frames/s = "
      f"{n*len(frames)/t_all*w*h/1e9:.2f} Gpix/s; host time in the feeder {t_feed*1e3:.1f} ms per pass; digests differing from the reference's: {bad}")
