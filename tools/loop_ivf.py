#!/usr/bin/env python3
"""Dev aid: write an IVF that repeats the frames of a fixture n times (every fixture starts with a key frame, so the
concatenation is a valid stream).  tools/loop_ivf.py in.ivf out.ivf n"""
import struct, sys
src, dst, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
d = open(src, "rb").read()
hdr = bytearray(d[:32])
frames, off = [], 32
while off + 12 <= len(d):
    sz = struct.unpack_from("<I", d, off)[0]
    frames.append(d[off + 12:off + 12 + sz]); off += 12 + sz
struct.pack_into("<I", hdr, 24, len(frames) * n)
with open(dst, "wb") as f:
    f.write(hdr)
    pts = 0
    for _ in range(n):
        for fr in frames:
            f.write(struct.pack("<IQ", len(fr), pts)); f.write(fr); pts += 1
