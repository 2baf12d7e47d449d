"""Shared helpers for the test-suite: package loader, the ORACLE binding (tests only!), fixtures."""
import ctypes
import importlib.util
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_LIB = os.path.join(ROOT, "oracle", "libvp8oracle.so")
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libvpxref.so")

FIXTURES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".ivf"))


def load_package():
    """Import libvpx.opencl_amd/ (dot in the directory name -> load by path)."""
    name = "libvpx_opencl_amd"
    if name in sys.modules:
        return sys.modules[name]
    pkg = os.path.join(ROOT, "libvpx.opencl_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(pkg, "__init__.py"),
                                                  submodule_search_locations=[pkg])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def golden_md5(name):
    return [l.split()[0] for l in open(os.path.join(GOLDEN, name + ".md5"))]


def ivf_path(name):
    return os.path.join(GOLDEN, name + ".ivf")


_oracle = None


def oracle():
    """ctypes handle of oracle/libvp8oracle.so -- the CPU checker.  Tests only."""
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_LIB):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"])
        L = ctypes.CDLL(ORACLE_LIB)
        vp = ctypes.c_void_p
        L.vp8o_decode_frame.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_int]
        _oracle = L
    return _oracle


def oracle_decode(hdr, mbs, coef, mvs, dst, refs, stages=7):
    """refs: (last, golden, alt) numpy frame buffers or None."""
    rp = (ctypes.c_void_p * 4)(None, *[(r.ctypes.data if r is not None else None) for r in refs])
    oracle().vp8o_decode_frame(ctypes.byref(hdr), mbs.ctypes.data, coef.ctypes.data, mvs.ctypes.data,
                               dst.ctypes.data, rp, stages)


def oracle_decode_ivf(name, stages=7, keep_frames=False):
    """Feeder + oracle over a whole fixture: per-shown-frame MD5s (and optionally every frame buffer)."""
    P = load_package()
    w, h, frames = P.read_ivf(ivf_path(name))
    parser = P.Parser()
    out, kept, bufs, g = [], [], None, None
    for data in frames:
        hdr, changed, mbs, coef, mvs = P.parse_to_numpy(parser, data)
        if changed:
            g = P.geom(hdr.width, hdr.height)
            bufs = [np.zeros(g.frame_size, np.uint8) for _ in range(4)]
        r = parser.refs
        oracle_decode(hdr, mbs, coef, mvs, bufs[r.new_idx], (bufs[r.lst_idx], bufs[r.gld_idx], bufs[r.alt_idx]), stages)
        new = r.new_idx
        parser.swap(hdr)
        if keep_frames:
            kept.append((hdr, mbs, coef, mvs, bufs[new].copy()))
        if hdr.show_frame:
            out.append(P.frame_md5(bufs[parser.refs.show_idx], g, hdr.width, hdr.height))
    parser.close()
    return (out, kept) if keep_frames else out


def coded_area_equal(a, b, g):
    """Compare two frame buffers over the coded (16-aligned) area of all three planes."""
    diffs = []
    for name, off, stride, w, h in (("Y", g.y_off, g.y_stride, g.aligned_w, g.aligned_h),
                                    ("U", g.u_off, g.uv_stride, g.aligned_w // 2, g.aligned_h // 2),
                                    ("V", g.v_off, g.uv_stride, g.aligned_w // 2, g.aligned_h // 2)):
        pa = np.lib.stride_tricks.as_strided(a[off:], shape=(h, w), strides=(stride, 1))
        pb = np.lib.stride_tricks.as_strided(b[off:], shape=(h, w), strides=(stride, 1))
        d = pa != pb
        if d.any():
            ys, xs = np.nonzero(d)
            diffs.append((name, int(d.sum()), int(ys[0]), int(xs[0])))
    return diffs


def bordered_area_equal(a, b, g):
    """Compare including the 32/16-pixel borders (everything vp8_yv12_extend_frame_borders defines)."""
    diffs = []
    for name, off, stride, w, h, bd in (("Y", g.y_off, g.y_stride, g.aligned_w, g.aligned_h, 32),
                                        ("U", g.u_off, g.uv_stride, g.aligned_w // 2, g.aligned_h // 2, 16),
                                        ("V", g.v_off, g.uv_stride, g.aligned_w // 2, g.aligned_h // 2, 16)):
        o = off - bd * stride - bd
        pa = np.lib.stride_tricks.as_strided(a[o:], shape=(h + 2 * bd, w + 2 * bd), strides=(stride, 1))
        pb = np.lib.stride_tricks.as_strided(b[o:], shape=(h + 2 * bd, w + 2 * bd), strides=(stride, 1))
        d = pa != pb
        if d.any():
            ys, xs = np.nonzero(d)
            diffs.append((name, int(d.sum()), int(ys[0]) - bd, int(xs[0]) - bd))
    return diffs
