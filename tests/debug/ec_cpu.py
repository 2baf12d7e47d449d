#!/usr/bin/env python3
"""Dev aid (CPU only, dev container: needs oracle/_ref/ref_md5_ec): the feeder's error concealment + the ORACLE's pixel path on a
damaged stream against the reference decoder built with error concealment.
    python3 tests/debug/ec_cpu.py <fixture> [--lose 3,4] [--cut 5:700,...] [--no-ec]"""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from vp8_testlib import ROOT, load_package, ivf_path, oracle_decode


def _stream(name, tmp):
    from ec_cases import materialize
    from vp8_testlib import GOLDEN
    return materialize(name, GOLDEN, tmp)


def damaged_listing(name, lose=(), cut=(), ec=True):
    P = load_package()
    with tempfile.TemporaryDirectory() as tmp:
        w, h, frames = P.read_ivf(_stream(name, tmp))
    parser = P.Parser()
    if ec:
        parser.set_error_concealment(True)
    cut = dict(cut)
    out, bufs, g = [], None, None
    for k, data in enumerate(frames, 1):
        if k in lose:
            if not parser.L.vp8_parser_conceals(parser.p):
                continue              # (the product's iface: last reference marked corrupt, nothing shown)
            data = b""
        elif k in cut:
            data = data[:cut[k]]
        try:
            hdr, changed = parser.begin(data)
        except ValueError as e:
            out.append(f"decode-error {k:04d}")
            continue
        n = hdr.mb_cols * hdr.mb_rows
        mbs = np.zeros((n, 64), np.uint8); coef = np.zeros((n, 400), np.int16); mvs = np.zeros((n, 16, 2), np.int16)
        try:
            parser.decode_mbs(mbs.ctypes.data, coef.ctypes.data, mvs.ctypes.data)
        except ValueError:
            out.append(f"decode-error {k:04d}")
            continue
        parser.final_hdr(hdr)
        if changed:
            g = P.geom(hdr.width, hdr.height)
            bufs = [np.zeros(g.frame_size, np.uint8) for _ in range(4)]
        r = parser.refs
        oracle_decode(hdr, mbs, coef, mvs, bufs[r.new_idx], (bufs[r.lst_idx], bufs[r.gld_idx], bufs[r.alt_idx]), 7)
        parser.swap(hdr)
        if hdr.show_frame:
            out.append(f"{P.frame_md5(bufs[parser.refs.show_idx], g, hdr.width, hdr.height)}  img-{hdr.width}x{hdr.height}-{k:04d}.i420")
    parser.close()
    return out


def reference_listing(name, lose=(), cut=(), ec=True):
    tool = os.path.join(ROOT, "oracle", "_ref", "ref_md5_ec")
    with tempfile.TemporaryDirectory() as d:
        o = os.path.join(d, "o.md5")
        args = [tool, "--damage"] + (["--ec"] if ec else [])
        if lose: args += ["--lose", ",".join(map(str, lose))]
        if cut: args += ["--cut", ",".join(f"{a}:{b}" for a, b in cut)]
        r = subprocess.run(args + [_stream(name, d), o], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return open(o).read().splitlines()


if __name__ == "__main__":
    a = sys.argv[1:]
    name = a[0]
    lose = tuple(int(x) for x in a[a.index("--lose") + 1].split(",")) if "--lose" in a else ()
    cut = tuple(tuple(int(v) for v in x.split(":")) for x in a[a.index("--cut") + 1].split(",")) if "--cut" in a else ()
    ec = "--no-ec" not in a
    want = reference_listing(name, lose, cut, ec)
    got = damaged_listing(name, lose, cut, ec)
    ok = 0
    for i in range(max(len(want), len(got))):
        x = want[i] if i < len(want) else "-"
        y = got[i] if i < len(got) else "-"
        ok += x == y
        print("  " if x == y else "!!", x, "|", y if x != y else "")
    print(f"{ok}/{max(len(want), len(got))} lines equal")
