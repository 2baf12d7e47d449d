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


# ------------------------------------------------------------------------------------------
# synthetic, well-formed IR (no bitstream involved): random modes / coefficients / MVs
# ------------------------------------------------------------------------------------------
ZIGZAG_RASTER = [0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15]
ZIGZAG_COLMAJOR = [(z % 4) * 4 + z // 4 for z in ZIGZAG_RASTER]


def synth_ir(width, height, seed, inter=False, version=0, filter_type=0, dense=0.3, big=False, segmented=True):
    """One frame of random IR.  Coefficients and eobs are mutually consistent the way the feeder
    produces them (eob = last coded zig-zag position + 1; Y blocks of an MB with a Y2 block start at 1)."""
    P = load_package()
    rng = np.random.default_rng(seed)
    cols, rows = (width + 15) // 16, (height + 15) // 16
    n = cols * rows
    hdr = P.FrameHdr()
    hdr.width, hdr.height, hdr.mb_cols, hdr.mb_rows = width, height, cols, rows
    hdr.frame_type = 1 if inter else 0
    hdr.version = version
    hdr.show_frame = 1
    hdr.filter_type = filter_type
    hdr.filter_level = int(rng.integers(0, 64)) if seed % 5 else 0
    hdr.sharpness_level = int(rng.integers(0, 8))
    hdr.segmentation_enabled = 1 if segmented else 0
    hdr.mb_segment_abs_delta = int(rng.integers(0, 2))
    for i in range(4):
        hdr.segment_quant[i] = int(rng.integers(0, 128)) if hdr.mb_segment_abs_delta else int(rng.integers(-40, 41))
        hdr.segment_lf[i] = int(rng.integers(0, 64)) if hdr.mb_segment_abs_delta else int(rng.integers(-30, 31))
        hdr.ref_lf_deltas[i] = int(rng.integers(-20, 21))
        hdr.mode_lf_deltas[i] = int(rng.integers(-20, 21))
    hdr.mode_ref_lf_delta_enabled = int(rng.integers(0, 2))
    hdr.base_qindex = int(rng.integers(0, 128))
    for f in ("y1dc_delta_q", "y2dc_delta_q", "y2ac_delta_q", "uvdc_delta_q", "uvac_delta_q"):
        setattr(hdr, f, int(rng.integers(-15, 16)))
    mbs = np.zeros((n, 64), np.uint8)
    coef = np.zeros((n, 400), np.int16)
    mvs = np.zeros((n, 16, 2), np.int16)
    zz = np.array(ZIGZAG_COLMAJOR)
    for i in range(n):
        r, c = divmod(i, cols)
        is_inter = inter and rng.random() < 0.8
        if is_inter:
            y_mode = int(rng.choice([5, 6, 7, 8, 9]))
            mbs[i, 2] = int(rng.integers(1, 4))
            mbs[i, 1] = 0
        else:
            y_mode = int(rng.choice([0, 1, 2, 3, 4], p=[0.15, 0.15, 0.15, 0.15, 0.4]))
            mbs[i, 1] = int(rng.integers(0, 4))
            mbs[i, 40:56] = rng.integers(0, 10, size=16)
        mbs[i, 0] = y_mode
        mbs[i, 4] = int(rng.integers(0, 4)) if segmented else 0
        has_y2 = y_mode not in (4, 9)
        skip = rng.random() < 0.15
        if is_inter:
            # MVs in 1/8 pel units (even for luma, as the bitstream stores quarter pel << 1), kept within
            # the range the feeder's bounds logic allows without clamping, or flagged for clamping
            clampflag = rng.random() < 0.3
            lim_l, lim_r = -(c * 16 + 16) * 8, ((cols - 1 - c) * 16 + 16) * 8
            lim_t, lim_b = -(r * 16 + 16) * 8, ((rows - 1 - r) * 16 + 16) * 8
            if clampflag:
                lim_l -= 400; lim_r += 400; lim_t -= 400; lim_b += 400
                mbs[i, 3] |= 2
            def rmv():
                return (int(rng.integers(lim_t // 2, lim_b // 2 + 1)) * 2, int(rng.integers(lim_l // 2, lim_r // 2 + 1)) * 2)
            if y_mode == 9:
                part = int(rng.integers(0, 4))
                mbs[i, 5] = part
                groups = {0: lambda b: b >> 3, 1: lambda b: (b >> 1) & 1, 2: lambda b: ((b >> 3) << 1) | ((b >> 1) & 1),
                          3: lambda b: b}[part]
                table = {}
                for b in range(16):
                    gidx = groups(b)
                    if gidx not in table:
                        table[gidx] = rmv() if rng.random() < 0.8 else (0, 0)
                    mvs[i, b] = table[gidx]
            else:
                mv = (0, 0) if y_mode == 7 else rmv()
                mvs[i, :] = mv
        if skip:
            mbs[i, 3] |= 1
            continue
        mag = 2047 if big else 60
        total = 0
        for b in range(25):
            if b == 24 and not has_y2:
                continue
            first = 1 if (has_y2 and b < 16) else 0
            if rng.random() < dense:
                last = int(rng.integers(first, 16))
                vals = rng.integers(-mag, mag + 1, size=16).astype(np.int16)
                vals[rng.random(16) < 0.5] = 0
                vals[:first] = 0
                vals[last + 1:] = 0
                if vals[last] == 0:
                    vals[last] = 1
                coef[i, b * 16 + zz] = vals
                eob = 15 if last == 15 else last + 1
            else:
                eob = first
            mbs[i, 8 + b] = eob
            total += eob
        if has_y2:
            total -= 16
        if total == 0:
            mbs[i, 3] |= 1
            mbs[i, 8:33] = 0
            coef[i] = 0
    return hdr, mbs, coef, mvs


def random_frame(g, seed):
    """A fully defined random reference frame buffer (smooth + noise so filters do interesting things)."""
    rng = np.random.default_rng(seed)
    buf = rng.integers(0, 256, size=g.frame_size).astype(np.uint8)
    return buf


# ------------------------------------------------------------------------------------------
# output-side post-processing: vp8_post_proc_frame (vp8/common/postproc.c:903-1000) over the oracle's filters
# ------------------------------------------------------------------------------------------
PP_DEBLOCK, PP_DEMACROBLOCK, PP_ADDNOISE, PP_MFQE = 1, 2, 4, 1024


class OraclePostproc:
    """One decoder's post-processing state (vp8_post_proc_frame, postproc.c:903-1000).  rand() is the C library's, drawn in the
    reference's order: once per demacroblocked frame (postproc.c:286), 3072 times per noise table (:456) and once per noisy row
    (:499).  Call libc.srand(1) first to be in the state a fresh process (the reference's vpxdec) is in.  With PP_MFQE the
    caller passes the frame's header and IR as well: the buffer that was shown before is blended into frames whose quantiser
    index is 10 or more above the running one (:948-969)."""

    def __init__(self, flags, deblocking_level, noise_level):
        self.flags, self.level, self.noise_level = flags, deblocking_level, noise_level
        self.last_q = self.last_noise = 0
        self.noise = np.zeros(3072, np.int8)
        self.clamp = 0
        self.libc = ctypes.CDLL(None)
        self.shown = 0                   # cm->current_video_frame when the frame is post-processed (onyxd_if.c:646-647)
        self.last_base_qindex = 0        # postproc_state.last_base_qindex
        self.post = None                 # post_proc_buffer
        self.mfqe_frames = 0

    def _filters(self, src, post, g, ppl, ppl_dm, mbl):
        O = oracle()
        ci, vp = ctypes.c_int, ctypes.c_void_p
        planes = _pp_planes(g)
        lim = ppl_dm if self.flags & PP_DEMACROBLOCK else ppl
        for off, stride, rows, cols in planes:
            O.vp8o_post_proc_down_and_across(vp(src.ctypes.data + off), vp(post.ctypes.data + off), ci(stride), ci(stride),
                                             ci(rows), ci(cols), ci(lim))
        if self.flags & PP_DEMACROBLOCK:
            off, stride, rows, cols = planes[0]
            tmp = post.copy()
            O.vp8o_mbpost_proc_across(vp(post.ctypes.data + off), vp(tmp.ctypes.data + off), ci(stride), ci(rows), ci(cols), ci(mbl))
            rv = self.libc.rand() & 63
            O.vp8o_mbpost_proc_down(vp(tmp.ctypes.data + off), vp(post.ctypes.data + off), ci(stride), ci(rows), ci(cols),
                                    ci(mbl), ci(rv))

    def frame(self, buf, g, filter_level, hdr=None, mbs=None, mvs=None):
        O = oracle()
        ci, vp = ctypes.c_int, ctypes.c_void_p
        q, ppl, ppl_dm, mbl = (ctypes.c_int() for _ in range(4))
        O.vp8o_pp_strengths(ci(filter_level), ci(self.level), ctypes.byref(q), ctypes.byref(ppl), ctypes.byref(ppl_dm), ctypes.byref(mbl))
        planes = _pp_planes(g)
        self.shown += 1
        if self.post is None or self.post.size != buf.size:
            self.post = np.zeros(buf.size, np.uint8)
        post = self.post
        base_q = hdr.base_qindex if hdr is not None else 0
        filtering = self.flags & (PP_DEBLOCK | PP_DEMACROBLOCK)
        if (self.flags & PP_MFQE) and hdr is not None and self.shown >= 2 and base_q - self.last_base_qindex >= 10:
            G = (ctypes.c_int * 10)(*[getattr(g, n) for n, _ in g._fields_])
            O.vp8o_mfqe(ctypes.byref(hdr), G, vp(mbs.ctypes.data), vp(mvs.ctypes.data if mvs is not None else None),
                        vp(buf.ctypes.data), vp(post.ctypes.data), ci(base_q), ci(self.last_base_qindex))
            self.mfqe_frames += 1
            if filtering:                # post_proc_buffer -> post_proc_buffer_int (with borders, yv12extend.c:224-260) -> filters
                mid = post.copy()
                for off, stride, rows, cols in planes:
                    for k in (1, 2):
                        mid[off - k * stride:off - k * stride + cols] = mid[off:off + cols]
                        e = off + (rows - 1) * stride
                        mid[e + k * stride:e + k * stride + cols] = mid[e:e + cols]
                self._filters(mid, post, g, ppl.value, ppl_dm.value, mbl.value)
            self.last_base_qindex = (3 * self.last_base_qindex + base_q) >> 2
        else:
            if filtering:
                self._filters(buf, post, g, ppl.value, ppl_dm.value, mbl.value)
            else:
                post[:] = buf
            self.last_base_qindex = base_q
        if self.flags & PP_ADDNOISE:
            if self.last_q != q.value or self.last_noise != self.noise_level:      # :988-993; fillrd stores ITS q = 63 - q
                r = np.array([self.libc.rand() & 0xff for _ in range(3072)], np.uint8)
                c = ctypes.c_int()
                O.vp8o_pp_noise_table(ci(63 - q.value), ci(self.noise_level), vp(r.ctypes.data), vp(self.noise.ctypes.data), ctypes.byref(c))
                self.clamp = c.value
                self.last_q, self.last_noise = 63 - q.value, self.noise_level
            off, stride, rows, cols = planes[0]
            offs = np.array([self.libc.rand() & 0xff for _ in range(rows)], np.uint8)
            O.vp8o_plane_add_noise(vp(post.ctypes.data + off), vp(self.noise.ctypes.data), ci(self.clamp), ci(cols), ci(rows), ci(stride),
                                   vp(offs.ctypes.data))
        return post.copy()


def _pp_planes(g):
    return ((g.y_off, g.y_stride, g.aligned_h, g.aligned_w), (g.u_off, g.uv_stride, g.aligned_h // 2, g.aligned_w // 2),
            (g.v_off, g.uv_stride, g.aligned_h // 2, g.aligned_w // 2))


def oracle_postproc_ivf(name, flags, level, noise):
    """Feeder + oracle + oracle post-processing over a fixture: per-shown-frame MD5s of the post-processed output."""
    P = load_package()
    _, kept = oracle_decode_ivf(name, keep_frames=True)
    ctypes.CDLL(None).srand(1)
    pp = OraclePostproc(flags, level, noise)
    out = []
    for hdr, mbs, coef, mvs, frame in kept:
        if hdr.show_frame:
            g = P.geom(hdr.width, hdr.height)
            out.append(P.frame_md5(pp.frame(frame, g, hdr.filter_level, hdr, mbs, mvs), g, hdr.width, hdr.height))
    return out
