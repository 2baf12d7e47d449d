"""libvpx.opencl_amd -- MI355X-native VP8 decode pixel path (host-side Python plumbing).

The product is the two native libraries built from ``csrc/``:

* ``lib/libvp8hip.so``  -- hand-written HIP kernels for gfx950 + the C-ABI shim (``include/vp8hip.h``)
* ``lib/libvpx_hip.so`` -- the C host side: bitstream feeder (``vp8_parser``), decoder core and the
  ``vpx_codec`` / ``vp8_dx`` interface (``include/vpx/*.h``), linked against ``libvp8hip.so``

This module only binds them with ctypes for the tests and ``bench.py`` (the reference is C; the
drop-in boundary is the C ABI, Python is plumbing).  It deliberately has NO CPU fallback:
``Vp8Hip()`` raises if the HIP library or a gfx950 device is missing.

The directory name contains a dot, so it is loaded by path (see ``__graft_entry__.load_package``).
"""
import ctypes
import hashlib
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIBDIR = os.path.join(HERE, "lib")
HIP_LIB = os.environ.get("VP8HIP_LIB") or os.path.join(LIBDIR, "libvp8hip.so")      # (VP8HIP_LIB: a diagnostic variant, tools/variant.sh)
HOST_LIB = os.path.join(LIBDIR, "libvpx_hip.so")

c_void_p, c_int, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t

STAGE_RECON, STAGE_LF, STAGE_EXTEND, STAGE_ALL = 1, 2, 4, 7


def build(verbose=False):
    """Compile every native piece in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    out = subprocess.run(["make", "-C", os.path.join(HERE, "csrc"), "all"], capture_output=True, text=True)
    if out.returncode:
        raise RuntimeError("native build failed:\n" + out.stdout + out.stderr)
    if verbose:
        print(out.stdout)


# ------------------------------------------------------------------------------------------
# IR structures (include/vp8_ir.h)
# ------------------------------------------------------------------------------------------
class FrameHdr(ctypes.Structure):
    _fields_ = [("width", ctypes.c_uint16), ("height", ctypes.c_uint16), ("mb_cols", ctypes.c_uint16),
                ("mb_rows", ctypes.c_uint16), ("frame_type", ctypes.c_uint8), ("version", ctypes.c_uint8),
                ("show_frame", ctypes.c_uint8), ("filter_type", ctypes.c_uint8), ("filter_level", ctypes.c_uint8),
                ("sharpness_level", ctypes.c_uint8), ("segmentation_enabled", ctypes.c_uint8),
                ("mb_segment_abs_delta", ctypes.c_uint8), ("segment_quant", ctypes.c_int8 * 4),
                ("segment_lf", ctypes.c_int8 * 4), ("mode_ref_lf_delta_enabled", ctypes.c_uint8),
                ("ref_lf_deltas", ctypes.c_int8 * 4), ("mode_lf_deltas", ctypes.c_int8 * 4),
                ("base_qindex", ctypes.c_uint8), ("y1dc_delta_q", ctypes.c_int8), ("y2dc_delta_q", ctypes.c_int8),
                ("y2ac_delta_q", ctypes.c_int8), ("uvdc_delta_q", ctypes.c_int8), ("uvac_delta_q", ctypes.c_int8),
                ("refresh_last", ctypes.c_uint8), ("refresh_golden", ctypes.c_uint8), ("refresh_alt", ctypes.c_uint8),
                ("copy_buffer_to_gf", ctypes.c_uint8), ("copy_buffer_to_arf", ctypes.c_uint8),
                ("sign_bias_golden", ctypes.c_uint8), ("sign_bias_alt", ctypes.c_uint8),
                ("color_space", ctypes.c_uint8), ("clamping_type", ctypes.c_uint8),
                ("num_token_partitions", ctypes.c_uint8), ("lf_key_frame", ctypes.c_uint8), ("rsv", ctypes.c_uint8 * 14)]


assert ctypes.sizeof(FrameHdr) == 64


class EntropyFrame(ctypes.Structure):
    """vp8hip_entropy_frame (include/vp8hip.h): what the host's header parse hands to the device's entropy decoder."""
    _fields_ = [("hdr", FrameHdr), ("data_off", ctypes.c_uint64), ("first_pos", ctypes.c_uint32), ("first_end", ctypes.c_uint32),
                ("first_value", ctypes.c_uint32), ("first_bits", ctypes.c_int32), ("first_range", ctypes.c_uint32),
                ("num_tok", ctypes.c_uint32), ("tok_pos", ctypes.c_uint32 * 8), ("tok_end", ctypes.c_uint32 * 8),
                ("update_mb_segmentation_map", ctypes.c_uint8), ("mb_no_coeff_skip", ctypes.c_uint8),
                ("prob_skip_false", ctypes.c_uint8), ("segmap_keep", ctypes.c_uint8), ("segment_tree_probs", ctypes.c_uint8 * 3),
                ("rsv1", ctypes.c_uint8), ("coef_probs", ctypes.c_uint8 * 1056),
                ("prob_intra", ctypes.c_uint8), ("prob_last", ctypes.c_uint8), ("prob_gf", ctypes.c_uint8), ("rsv2", ctypes.c_uint8),
                ("ymode_prob", ctypes.c_uint8 * 4), ("uvmode_prob", ctypes.c_uint8 * 3), ("rsv3", ctypes.c_uint8),
                ("mvc", ctypes.c_uint8 * 38), ("rsv4", ctypes.c_uint8 * 2), ("rsv5", ctypes.c_uint8 * 4)]


class Geom(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("aligned_w", "aligned_h", "y_stride", "uv_stride", "y_plane_size",
                                     "uv_plane_size", "frame_size", "y_off", "u_off", "v_off")]


def geom(width, height):
    """vp8ir_geom_init (include/vp8_ir.h) restated for numpy-side indexing."""
    g = Geom()
    aw, ah = (width + 15) & ~15, (height + 15) & ~15
    g.aligned_w, g.aligned_h = aw, ah
    g.y_stride = (aw + 64 + 31) & ~31
    g.uv_stride = g.y_stride >> 1
    g.y_plane_size = (ah + 64) * g.y_stride
    g.uv_plane_size = (ah // 2 + 32) * g.uv_stride
    g.frame_size = g.y_plane_size + 2 * g.uv_plane_size
    g.y_off = 32 * g.y_stride + 32
    g.u_off = g.y_plane_size + 16 * g.uv_stride + 16
    g.v_off = g.y_plane_size + g.uv_plane_size + 16 * g.uv_stride + 16
    return g


class Refs(ctypes.Structure):
    _fields_ = [("new_idx", c_int), ("lst_idx", c_int), ("gld_idx", c_int), ("alt_idx", c_int),
                ("ref_cnt", c_int * 4), ("show_idx", c_int)]


class Job(ctypes.Structure):
    _fields_ = [("ir_slot", ctypes.c_int32), ("dst_fb", ctypes.c_int32), ("ref_fb", ctypes.c_int32 * 4)]


class Stats(ctypes.Structure):
    _fields_ = [("recon_ms", ctypes.c_float), ("lf_ms", ctypes.c_float), ("extend_ms", ctypes.c_float),
                ("recon_waves", c_int), ("lf_waves", c_int), ("workgroups", c_int), ("detile_pass", c_int),
                ("lf_kernels", c_int), ("fused", c_int), ("pred_tiles", c_int)]


# ------------------------------------------------------------------------------------------
# IVF container + MD5 of a decoded frame (vpxdec.c:386-441 / examples/decode_to_md5.txt:28-47)
# ------------------------------------------------------------------------------------------
def read_ivf(path):
    data = open(path, "rb").read()
    if len(data) < 32 or data[:4] != b"DKIF":
        raise ValueError(f"{path}: not an IVF file")
    w, h = int.from_bytes(data[12:14], "little"), int.from_bytes(data[14:16], "little")
    frames, pos = [], 32
    while pos + 12 <= len(data):
        sz = int.from_bytes(data[pos:pos + 4], "little")
        pos += 12
        if pos + sz > len(data):
            break
        frames.append(data[pos:pos + sz])
        pos += sz
    return w, h, frames


def frame_md5(buf, g, width, height):
    """MD5 over the visible Y, U, V rows of a whole frame buffer (numpy uint8, vp8ir_geom layout)."""
    m = hashlib.md5()
    cw, ch = (width + 1) // 2, (height + 1) // 2
    for off, stride, w, h in ((g.y_off, g.y_stride, width, height), (g.u_off, g.uv_stride, cw, ch),
                              (g.v_off, g.uv_stride, cw, ch)):
        plane = np.lib.stride_tricks.as_strided(buf[off:], shape=(h, w), strides=(stride, 1))
        m.update(np.ascontiguousarray(plane).tobytes())
    return m.hexdigest()


def planes_md5(y, u, v):
    m = hashlib.md5()
    for p in (y, u, v):
        m.update(np.ascontiguousarray(p).tobytes())
    return m.hexdigest()


# ------------------------------------------------------------------------------------------
# host feeder (vp8_parser.h), exported by libvpx_hip.so
# ------------------------------------------------------------------------------------------
_host = None


def load_host():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise RuntimeError(f"{HOST_LIB} missing: run __graft_entry__.build()")
        # libvpx_hip.so links libvp8hip.so (rpath $ORIGIN)
        L = ctypes.CDLL(HOST_LIB)
        L.vp8_parser_create.restype = c_void_p
        L.vp8_parser_destroy.argtypes = [c_void_p]
        L.vp8_parser_set_threads.argtypes = [c_void_p, c_int]
        L.vp8_parser_set_device_segmap.argtypes = [c_void_p, c_int]
        L.vp8_parser_set_error_concealment.argtypes = [c_void_p, c_int]
        L.vp8_parser_conceals.argtypes = [c_void_p]
        L.vp8_parser_frame_hdr.argtypes = [c_void_p, c_void_p]
        L.vp8_parser_begin_frame.argtypes = [c_void_p, ctypes.c_char_p, c_size_t, c_void_p]
        L.vp8_parser_decode_mbs.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        L.vp8_parser_decode_mbs_compact.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t, ctypes.POINTER(c_size_t), c_void_p, c_void_p]
        L.vp8_parser_export_entropy.argtypes = [c_void_p, c_void_p]
        L.vp8_parser_error.argtypes = [c_void_p]
        L.vp8_parser_error.restype = ctypes.c_char_p
        for f in ("vp8_refs_init", "vp8_refs_on_alloc", "vp8_refs_release_new"):
            getattr(L, f).argtypes = [c_void_p]
        L.vp8_refs_get_free.argtypes = [c_void_p]
        L.vp8_refs_swap.argtypes = [c_void_p, c_void_p]
        _host = L
    return _host


class Parser:
    """Bitstream -> IR.  Owns the reference-buffer bookkeeping too (vp8_refs)."""

    def __init__(self):
        self.L = load_host()
        self.p = c_void_p(self.L.vp8_parser_create())
        self.refs = Refs()
        self.L.vp8_refs_init(ctypes.byref(self.refs))
        self.dims = None

    def close(self):
        if self.p:
            self.L.vp8_parser_destroy(self.p)
            self.p = None

    def set_threads(self, n):
        """token partitions of a frame on up to n threads (vp8_parser_set_threads)"""
        self.L.vp8_parser_set_threads(self.p, n)

    def set_device_segmap(self, on=True):
        """the device keeps this stream's segment map, in the IR slot its frames are decoded into (vp8_parser_set_device_segmap)"""
        self.L.vp8_parser_set_device_segmap(self.p, int(on))

    def final_hdr(self, hdr):
        """after decode_mbs: the header as the pixel path is to see it (vp8_parser_frame_hdr)"""
        self.L.vp8_parser_frame_hdr(self.p, ctypes.byref(hdr))
        return hdr

    def set_error_concealment(self, on=True):
        """before the first frame: conceal lost frames and lost residuals (vp8_parser_set_error_concealment)"""
        self.L.vp8_parser_set_error_concealment(self.p, int(on))

    def begin(self, data):
        """-> (hdr, dims_changed).  Acquires refs.new_idx like the reference's get_free_fb."""
        hdr = FrameHdr()
        if self.L.vp8_refs_get_free(ctypes.byref(self.refs)) < 0:
            raise RuntimeError("no free frame buffer")
        self._frame_data = data          # the parser borrows the compressed frame until decode_mbs has run (vp8_parser.h)
        rc = self.L.vp8_parser_begin_frame(self.p, data, len(data), ctypes.byref(hdr))
        if rc:
            self.L.vp8_refs_release_new(ctypes.byref(self.refs))
            raise ValueError(f"vp8 header error {rc}: {self.L.vp8_parser_error(self.p).decode()}")
        changed = self.dims != (hdr.width, hdr.height)
        if changed:
            self.dims = (hdr.width, hdr.height)
            self.L.vp8_refs_on_alloc(ctypes.byref(self.refs))
        return hdr, changed

    def export_entropy(self):
        """After begin() on a key frame: the frame's vp8hip_entropy_frame (offsets relative to the frame's first byte); the
        parser is done with the frame.  None when the frame is not one the device decodes (it stays open for decode_mbs)."""
        out = EntropyFrame()
        rc = self.L.vp8_parser_export_entropy(self.p, ctypes.byref(out))
        if rc == 5:
            return None
        if rc:
            self.L.vp8_refs_release_new(ctypes.byref(self.refs))
            raise ValueError(f"vp8 header error {rc}: {self.L.vp8_parser_error(self.p).decode()}")
        return out

    def decode_mbs(self, mbs_ptr, coef_ptr, mvs_ptr):
        corrupt = c_int(0)
        rc = self.L.vp8_parser_decode_mbs(self.p, mbs_ptr, coef_ptr, mvs_ptr, ctypes.byref(corrupt))
        if rc:
            self.L.vp8_refs_release_new(ctypes.byref(self.refs))
            raise ValueError(f"vp8 macroblock data error {rc}: {self.L.vp8_parser_error(self.p).decode()}")
        return corrupt.value

    def decode_mbs_compact(self, mbx_ptr, blocks_ptr, cap_blocks, mvs_ptr):
        """The macroblocks in the DEVICE FORM of include/vp8_ir.h (records + block stream) -> (blocks written, corrupt flag)"""
        corrupt, nb = c_int(0), c_size_t(0)
        rc = self.L.vp8_parser_decode_mbs_compact(self.p, mbx_ptr, blocks_ptr, cap_blocks, ctypes.byref(nb), mvs_ptr, ctypes.byref(corrupt))
        if rc:
            self.L.vp8_refs_release_new(ctypes.byref(self.refs))
            raise ValueError(f"vp8 macroblock data error {rc}: {self.L.vp8_parser_error(self.p).decode()}")
        return nb.value, corrupt.value

    def swap(self, hdr):
        self.L.vp8_refs_swap(ctypes.byref(self.refs), ctypes.byref(hdr))


def parse_to_numpy_compact(parser, data):
    """One frame in the device form -> (hdr, mbx uint8[n,128], blocks int16[nb,16], mvs int16[n,16,2], corrupt)."""
    hdr, changed = parser.begin(data)
    n = hdr.mb_cols * hdr.mb_rows
    mbx = np.zeros((n, 128), np.uint8)
    blocks = np.zeros((n * 24, 16), np.int16)
    mvs = np.zeros((n, 16, 2), np.int16)
    nb, corrupt = parser.decode_mbs_compact(mbx.ctypes.data, blocks.ctypes.data, n * 24, mvs.ctypes.data)
    return hdr, mbx, blocks[:nb].copy(), mvs, corrupt


def block_kinds(mbs):
    """vp8ir_block_kind for every block of every macroblock: uint8[n, 25] of 0 (nothing), 1 (a lone first coefficient), 2 (more).
    mbs: uint8[n, >=64] descriptors."""
    ymode, flags, eobs = mbs[:, 0], mbs[:, 3], mbs[:, 8:33]
    has_y2 = (ymode != 4) & (ymode != 9)
    kind = np.zeros(eobs.shape, np.uint8)
    kind[eobs == 1] = 1
    kind[eobs > 1] = 2
    kind[:, :16][(eobs[:, :16] == 1) & has_y2[:, None]] = 0
    kind[~has_y2, 24] = 0
    kind[(flags & 1) != 0] = 0
    return kind


def compact_from_dense(mbs, coef):
    """The device form of include/vp8_ir.h (vp8ir_compact_mb restated with numpy): (mbx uint8[n,128], blocks int16[nb,16])."""
    n = mbs.shape[0]
    kind = block_kinds(mbs)
    c = coef.reshape(n, 25, 16)
    mbx = np.zeros((n, 128), np.uint8)
    mbx[:, :64] = mbs[:, :64]
    full = kind[:, :24] == 2
    first = np.concatenate(([0], np.cumsum(full.sum(1))[:-1])).astype(np.uint32)
    mbx[:, 56:60] = first.view(np.uint8).reshape(n, 4)
    mbx[:, 60:64] = 0
    aux = np.zeros((n, 32), np.int16)
    has_y2 = (mbs[:, 0] != 4) & (mbs[:, 0] != 9)
    skip = (mbs[:, 3] & 1) != 0
    y2rows = has_y2 & ~skip & (mbs[:, 8 + 24] != 0)
    aux[y2rows, :16] = c[y2rows, 24, :]
    lone = kind[:, :24] == 1
    dc = c[:, :24, 0]
    aux[:, :16] = np.where(lone[:, :16], dc[:, :16], aux[:, :16])
    aux[:, 16:24] = np.where(lone[:, 16:24], dc[:, 16:24], 0)
    mbx[:, 64:128] = aux.view(np.uint8).reshape(n, 64)
    return mbx, np.ascontiguousarray(c[:, :24][full])


def dense_from_compact(mbx, blocks):
    """vp8ir_expand_mb restated with numpy: (mbs uint8[n,64] with sparse_first cleared, coef int16[n,400])."""
    n = mbx.shape[0]
    mbs = mbx[:, :64].copy()
    first = mbs[:, 56:60].copy().view(np.uint32).reshape(n)
    mbs[:, 56:64] = 0
    kind = block_kinds(mbs)
    aux = mbx[:, 64:128].copy().view(np.int16).reshape(n, 32)
    c = np.zeros((n, 25, 16), np.int16)
    has_y2 = (mbs[:, 0] != 4) & (mbs[:, 0] != 9)
    skip = (mbs[:, 3] & 1) != 0
    y2rows = has_y2 & ~skip & (mbs[:, 8 + 24] != 0)
    c[y2rows, 24, :] = aux[y2rows, :16]
    lone = kind[:, :24] == 1
    c[:, :16, 0] = np.where(lone[:, :16], aux[:, :16], 0)
    c[:, 16:24, 0] = np.where(lone[:, 16:24], aux[:, 16:24], 0)
    full = kind[:, :24] == 2
    rank = np.cumsum(full, 1) - full
    idx = first[:, None] + rank
    c[:, :24][full] = blocks[idx[full]]
    return mbs, c.reshape(n, 400)


def parse_to_numpy(parser, data):
    """One frame -> (hdr, mbs uint8[n,64], coef int16[n,400], mvs int16[n,16,2]) in numpy arrays."""
    hdr, changed = parser.begin(data)
    n = hdr.mb_cols * hdr.mb_rows
    mbs = np.zeros((n, 64), np.uint8)
    coef = np.zeros((n, 400), np.int16)
    mvs = np.zeros((n, 16, 2), np.int16)
    parser.decode_mbs(mbs.ctypes.data, coef.ctypes.data, mvs.ctypes.data)
    return hdr, changed, mbs, coef, mvs


# ------------------------------------------------------------------------------------------
# HIP pixel path (vp8hip.h)
# ------------------------------------------------------------------------------------------
_hip = None


class PostprocParams(ctypes.Structure):     # vp8hip_pp, include/vp8hip.h
    _fields_ = [("flags", ctypes.c_int32), ("flimit", ctypes.c_int32), ("mb_flimit", ctypes.c_int32), ("rv_offset", ctypes.c_int32),
                ("noise_clamp", ctypes.c_int32), ("rv", c_void_p), ("noise", c_void_p), ("noise_rows", c_void_p)]


PP_DEBLOCK, PP_DEMACROBLOCK, PP_ADDNOISE = 1, 2, 4


def load_hip():
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise RuntimeError(f"{HIP_LIB} missing: run __graft_entry__.build(); there is no CPU fallback")
        L = ctypes.CDLL(HIP_LIB, mode=ctypes.RTLD_GLOBAL)
        L.vp8hip_create.argtypes = [c_int, ctypes.POINTER(c_void_p)]
        L.vp8hip_destroy.argtypes = [c_void_p]
        L.vp8hip_last_error.argtypes = [c_void_p]
        L.vp8hip_last_error.restype = ctypes.c_char_p
        L.vp8hip_configure.argtypes = [c_void_p, c_int, c_int, c_int, c_int]
        L.vp8hip_geometry.argtypes = [c_void_p, c_void_p]
        L.vp8hip_ir_map.argtypes = [c_void_p, c_int] + [ctypes.POINTER(c_void_p)] * 4
        L.vp8hip_ir_upload.argtypes = [c_void_p, c_int]
        L.vp8hip_ir_map_compact.argtypes = [c_void_p, c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p),
                                            ctypes.POINTER(c_size_t), ctypes.POINTER(c_void_p)]
        L.vp8hip_ir_upload_compact.argtypes = [c_void_p, c_int, c_size_t]
        L.vp8hip_ir_copy.argtypes = [c_void_p, c_int, c_int]
        L.vp8hip_decode.argtypes = [c_void_p, c_void_p, c_int, c_int]
        L.vp8hip_frame_download.argtypes = [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int]
        L.vp8hip_frame_upload.argtypes = [c_void_p, c_int, c_void_p]
        L.vp8hip_frame_copy.argtypes = [c_void_p, c_int, c_int]
        L.vp8hip_frames_to_raster.argtypes = [c_void_p, c_int, c_int]
        L.vp8hip_set_direct_download.argtypes = [c_void_p, c_int]
        L.vp8hip_set_pred_tiles.argtypes = [c_void_p, c_int]
        L.vp8hip_sync.argtypes = [c_void_p]
        L.vp8hip_join.argtypes = [c_void_p]
        L.vp8hip_get_stats_at.argtypes = [c_void_p, c_int, ctypes.POINTER(Stats)]
        L.vp8hip_get_stats.argtypes = [c_void_p, c_void_p]
        L.vp8hip_stream.argtypes = [c_void_p]
        L.vp8hip_stream.restype = c_void_p
        L.vp8hip_postproc.argtypes = [c_void_p, c_int, c_int, c_int, ctypes.POINTER(PostprocParams)]
        L.vp8hip_mfqe.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int]
        L.vp8hip_entropy_decode.argtypes = [c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t]
        L.vp8hip_entropy_status.argtypes = [c_void_p, c_int, c_void_p]
        L.vp8hip_ir_fetch.argtypes = [c_void_p, c_int, c_void_p, c_void_p]
        L.vp8hip_ir_fetch_mvs.argtypes = [c_void_p, c_int, c_void_p]
        _hip = L
    return _hip


class Vp8Hip:
    """One HIP context = one GPU's frame-buffer pool + IR slots + stream."""

    def __init__(self, device=-1):
        self.L = load_hip()
        h = c_void_p()
        if self.L.vp8hip_create(device, ctypes.byref(h)):
            raise RuntimeError("vp8hip_create: " + self.L.vp8hip_last_error(None).decode())
        self.h = h
        self.width = self.height = 0
        self.g = None

    def _chk(self, rc, what):
        if rc:
            raise RuntimeError(f"{what}: {self.L.vp8hip_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.L.vp8hip_destroy(self.h)
            self.h = None

    def configure(self, width, height, num_fb, num_slots):
        self._chk(self.L.vp8hip_configure(self.h, width, height, num_fb, num_slots), "vp8hip_configure")
        self.width, self.height = width, height
        self.g = geom(width, height)
        self.nmb = (self.g.aligned_w // 16) * (self.g.aligned_h // 16)
        self.num_fb, self.num_slots = num_fb, num_slots

    def configure_pooled(self, width, height, num_fb, num_slots, pool_bytes):
        """vp8hip_configure_pooled: slots without block streams of their own + one pool the device's entropy decoder fills"""
        self.L.vp8hip_configure_pooled.argtypes = [c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t]
        self._chk(self.L.vp8hip_configure_pooled(self.h, width, height, num_fb, num_slots, pool_bytes), "vp8hip_configure_pooled")
        self.width, self.height = width, height
        self.g = geom(width, height)
        self.nmb = (self.g.aligned_w // 16) * (self.g.aligned_h // 16)
        self.num_fb, self.num_slots = num_fb, num_slots

    def memory_usage(self):
        """vp8hip_memory_usage -> dict of bytes the context holds on the device: raster_pool, tile_pool, slots, block_pool,
        entropy_input, packed_staging"""
        out = (ctypes.c_size_t * 6)()
        self.L.vp8hip_memory_usage.argtypes = [c_void_p, c_void_p]
        self._chk(self.L.vp8hip_memory_usage(self.h, out), "vp8hip_memory_usage")
        return dict(zip(("raster_pool", "tile_pool", "slots", "block_pool", "entropy_input", "packed_staging"), [int(v) for v in out]))

    def pool_reset(self):
        self.L.vp8hip_pool_reset.argtypes = [c_void_p]
        self._chk(self.L.vp8hip_pool_reset(self.h), "vp8hip_pool_reset")

    def pool_usage(self):
        """-> (bytes taken since the last reset, bytes the pool holds)"""
        used, size = ctypes.c_size_t(), ctypes.c_size_t()
        self.L.vp8hip_pool_usage.argtypes = [c_void_p, c_void_p, c_void_p]
        self._chk(self.L.vp8hip_pool_usage(self.h, ctypes.byref(used), ctypes.byref(size)), "vp8hip_pool_usage")
        return used.value, size.value

    def ir_map(self, slot):
        ptrs = [c_void_p() for _ in range(4)]
        self._chk(self.L.vp8hip_ir_map(self.h, slot, *[ctypes.byref(p) for p in ptrs]), "vp8hip_ir_map")
        return [p.value for p in ptrs]   # hdr, mbs, coef, mvs (pinned host addresses)

    def fill_slot(self, slot, hdr, mbs, coef, mvs):
        """Copy numpy IR arrays into a slot's pinned staging and upload it."""
        ph, pm, pc, pv = self.ir_map(slot)
        ctypes.memmove(ph, ctypes.byref(hdr), 64)
        ctypes.memmove(pm, mbs.ctypes.data, mbs.nbytes)
        ctypes.memmove(pc, coef.ctypes.data, coef.nbytes)
        if hdr.frame_type != 0:
            ctypes.memmove(pv, mvs.ctypes.data, mvs.nbytes)
        self.upload(slot)

    def parse_into_slot(self, parser, data, slot):
        """Feeder writes straight into the pinned staging of `slot`; returns hdr (not yet uploaded)."""
        hdr, changed = parser.begin(data)
        if (hdr.width, hdr.height) != (self.width, self.height):
            raise RuntimeError("dimension change: reconfigure the context first")
        ph, pm, pc, pv = self.ir_map(slot)
        parser.decode_mbs(pm, pc, pv)
        ctypes.memmove(ph, ctypes.byref(hdr), 64)
        return hdr

    def ir_map_compact(self, slot):
        """Pinned staging of `slot` in the device form: (hdr, mbx, blocks, mvs addresses, blocks the stream may take)."""
        ph, pm, pb, pv, cap = c_void_p(), c_void_p(), c_void_p(), c_void_p(), c_size_t()
        self._chk(self.L.vp8hip_ir_map_compact(self.h, slot, ctypes.byref(ph), ctypes.byref(pm), ctypes.byref(pb), ctypes.byref(cap),
                                               ctypes.byref(pv)), "vp8hip_ir_map_compact")
        return ph.value, pm.value, pb.value, pv.value, cap.value

    def parse_into_slot_compact(self, parser, data, slot):
        """Feeder writes the frame in the DEVICE FORM (include/vp8_ir.h) into the pinned staging of `slot` and queues the upload
        (one copy; nothing on the device touches the slot before the pixel kernels read it); returns (hdr, bytes uploaded)."""
        hdr, changed = parser.begin(data)
        if (hdr.width, hdr.height) != (self.width, self.height):
            raise RuntimeError("dimension change: reconfigure the context first")
        ph, pm, pb, pv, cap = self.ir_map_compact(slot)
        nb, _ = parser.decode_mbs_compact(pm, pb, cap, pv)
        parser.final_hdr(hdr)
        ctypes.memmove(ph, ctypes.byref(hdr), 64)
        self._chk(self.L.vp8hip_ir_upload_compact(self.h, slot, nb), "vp8hip_ir_upload_compact")
        return hdr, self.nmb * 128 + nb * 32

    def fill_slot_compact(self, slot, hdr, mbx, blocks, mvs):
        """numpy arrays in the device form (compact_from_dense) into a slot's pinned staging, and up."""
        ph, pm, pb, pv, cap = self.ir_map_compact(slot)
        assert blocks.shape[0] <= cap
        ctypes.memmove(ph, ctypes.byref(hdr), 64)
        ctypes.memmove(pm, mbx.ctypes.data, mbx.nbytes)
        if blocks.nbytes:
            ctypes.memmove(pb, blocks.ctypes.data, blocks.nbytes)
        if hdr.frame_type != 0:
            ctypes.memmove(pv, mvs.ctypes.data, mvs.nbytes)
        self._chk(self.L.vp8hip_ir_upload_compact(self.h, slot, blocks.shape[0]), "vp8hip_ir_upload_compact")

    def upload(self, slot):
        self._chk(self.L.vp8hip_ir_upload(self.h, slot), "vp8hip_ir_upload")

    def ir_copy(self, dst, src):
        self._chk(self.L.vp8hip_ir_copy(self.h, dst, src), "vp8hip_ir_copy")

    def decode(self, jobs, stages=STAGE_ALL):
        """jobs: list of (ir_slot, dst_fb, (last, golden, alt))"""
        arr = (Job * len(jobs))()
        for i, (slot, dst, refs) in enumerate(jobs):
            arr[i].ir_slot, arr[i].dst_fb = slot, dst
            arr[i].ref_fb[0] = -1
            for k in range(3):
                arr[i].ref_fb[k + 1] = refs[k] if refs is not None else -1
        self._jobs_keepalive = arr
        self._chk(self.L.vp8hip_decode(self.h, arr, len(jobs), stages), "vp8hip_decode")

    def decode_array(self, job_array, n, stages=STAGE_ALL):
        self._chk(self.L.vp8hip_decode(self.h, job_array, n, stages), "vp8hip_decode")

    def sync(self):
        self._chk(self.L.vp8hip_sync(self.h), "vp8hip_sync")

    def join(self):
        """Order the context's main stream behind a tiled->raster pass still running on the internal stream."""
        self._chk(self.L.vp8hip_join(self.h), "vp8hip_join")

    def stats(self, back=0):
        """Kernel times of the last launch (back=0) or of an earlier one (back <= 31); waits for that launch only."""
        s = Stats()
        self._chk(self.L.vp8hip_get_stats_at(self.h, back, ctypes.byref(s)), "vp8hip_get_stats_at")
        return s

    def stream(self):
        return self.L.vp8hip_stream(self.h)

    def download_full(self, fb):
        buf = np.empty(self.g.frame_size, np.uint8)
        self._chk(self.L.vp8hip_frame_download(self.h, fb, 1, buf.ctypes.data, None, None, 0, 0), "download")
        return buf

    def download_planes(self, fb):
        w, h = self.width, self.height
        cw, ch = (w + 1) // 2, (h + 1) // 2
        y, u, v = np.empty((h, w), np.uint8), np.empty((ch, cw), np.uint8), np.empty((ch, cw), np.uint8)
        self._chk(self.L.vp8hip_frame_download(self.h, fb, 0, y.ctypes.data, u.ctypes.data, v.ctypes.data, w, cw),
                  "download")
        return y, u, v

    def frames_md5(self, first_fb, count):
        """MD5s of `count` consecutive frame buffers computed on the device (vp8hip_frames_fetch_async): list of hex digests."""
        out = np.zeros(16 * count, np.uint8)
        self.L.vp8hip_frames_fetch_async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        self.L.vp8hip_download_wait.argtypes = [ctypes.c_void_p]
        self._chk(self.L.vp8hip_frames_fetch_async(self.h, first_fb, count, None, out.ctypes.data), "vp8hip_frames_fetch_async")
        self._chk(self.L.vp8hip_download_wait(self.h), "vp8hip_download_wait")
        return [out[16 * i: 16 * i + 16].tobytes().hex() for i in range(count)]

    def frames_i420(self, first_fb, count):
        """`count` consecutive frame buffers as packed I420 (vp8hip_frames_fetch_i420_async: packed on the device from whichever form
        they are in -- tiles as they are --, no raster pool needed): uint8 array [count, w * h + 2 * (w / 2) * ((h + 1) / 2)]."""
        self.L.vp8hip_i420_bytes.restype = ctypes.c_size_t
        self.L.vp8hip_i420_bytes.argtypes = [ctypes.c_void_p]
        self.L.vp8hip_frames_fetch_i420_async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        self.L.vp8hip_download_wait.argtypes = [ctypes.c_void_p]
        nb = self.L.vp8hip_i420_bytes(self.h)
        # (the destination of a batch fetch is page-locked memory, include/vp8hip.h: into pageable memory the asynchronous copies of
        # the fetch's streams would be staged by the runtime one after the other)
        self.L.vp8hip_host_alloc.restype = ctypes.c_void_p
        self.L.vp8hip_host_alloc.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        self.L.vp8hip_host_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        host = self.L.vp8hip_host_alloc(self.h, nb * count)
        if not host:
            raise RuntimeError("vp8hip_host_alloc: " + self.L.vp8hip_last_error(self.h).decode())
        try:
            self._chk(self.L.vp8hip_frames_fetch_i420_async(self.h, first_fb, count, host, None), "vp8hip_frames_fetch_i420_async")
            self._chk(self.L.vp8hip_download_wait(self.h), "vp8hip_download_wait")
            out = np.ctypeslib.as_array(ctypes.cast(host, ctypes.POINTER(ctypes.c_uint8)), shape=(count, nb)).copy()
        finally:
            self.L.vp8hip_host_free(self.h, host)
        return out

    def frames_to_raster(self, first_fb, count):
        """Ask for the raster form of frame buffers a large launch left as tiles (vp8hip_frames_to_raster; asynchronous)."""
        self._chk(self.L.vp8hip_frames_to_raster(self.h, first_fb, count), "vp8hip_frames_to_raster")

    def upload_frame(self, fb, buf):
        assert buf.nbytes == self.g.frame_size
        self._chk(self.L.vp8hip_frame_upload(self.h, fb, buf.ctypes.data), "upload")

    def postproc(self, src_fb, dst_fb, tmp_fb, flags, flimit=0, mb_flimit=0, rv_offset=0, noise=None, noise_clamp=0, noise_rows=None):
        """Output-side filters of vp8/common/postproc.c from frame buffer src_fb into dst_fb (include/vp8hip.h).  noise: int8
        array of 3072, noise_rows: uint8 array, one phase per row of the aligned height."""
        pp = PostprocParams(flags, flimit, mb_flimit, rv_offset, noise_clamp,
                            ctypes.cast(load_host().vp8t_pp_rv, c_void_p) if flags & PP_DEMACROBLOCK else None,
                            noise.ctypes.data if noise is not None else None,
                            noise_rows.ctypes.data if noise_rows is not None else None)
        self._chk(self.L.vp8hip_postproc(self.h, src_fb, dst_fb, tmp_fb, ctypes.byref(pp)), "postproc")

    def entropy_decode(self, first_slot, frames, datas):
        """vp8hip_entropy_decode: frames = EntropyFrame list (from Parser.export_entropy), datas = the frames' bytes; slot
        first_slot + i receives frame i's IR.  Returns the per-frame status words (synchronises)."""
        n = len(frames)
        arr = (EntropyFrame * n)()
        off = 0
        for i, (f, d) in enumerate(zip(frames, datas)):
            ctypes.memmove(ctypes.byref(arr[i]), ctypes.byref(f), ctypes.sizeof(EntropyFrame))
            arr[i].data_off = off
            off += len(d)
        blob = b"".join(datas)
        self._chk(self.L.vp8hip_entropy_decode(self.h, first_slot, n, ctypes.byref(arr), blob, len(blob)), "entropy_decode")
        st = np.zeros(n, np.uint32)
        self._chk(self.L.vp8hip_entropy_status(self.h, n, st.ctypes.data), "entropy_status")
        return st

    def mvs_fetch(self, slot):
        """The slot's motion vectors as they stand on the device: int16[n * 16, 2] (row, col)."""
        mv = np.zeros((self.g_mbs() * 16, 2), np.int16)
        self._chk(self.L.vp8hip_ir_fetch_mvs(self.h, slot, mv.ctypes.data), "ir_fetch_mvs")
        return mv

    def ir_fetch(self, slot):
        """The slot's IR as it stands on the device: (mbs uint8[n,64], coef int16[n,400])."""
        n = self.g_mbs()
        mbs = np.zeros((n, 64), np.uint8)
        coef = np.zeros((n, 400), np.int16)
        self._chk(self.L.vp8hip_ir_fetch(self.h, slot, mbs.ctypes.data, coef.ctypes.data), "ir_fetch")
        return mbs, coef

    def mfqe(self, show_fb, prev_fb, dst_fb, mb_class, qcurr, qprev):
        """vp8_multiframe_quality_enhance (postproc.c:802-900; include/vp8hip.h): mb_class a uint8 array, a byte per macroblock."""
        mb_class = np.ascontiguousarray(mb_class, np.uint8)
        assert mb_class.size == self.g_mbs()
        self._chk(self.L.vp8hip_mfqe(self.h, show_fb, prev_fb, dst_fb, mb_class.ctypes.data, qcurr, qprev), "mfqe")

    def g_mbs(self):
        return (self.g.aligned_w // 16) * (self.g.aligned_h // 16)


def decode_ivf_gpu(path, device=-1, stages=STAGE_ALL):
    """Decode a whole IVF on the GPU, frame by frame (the latency path): list of per-shown-frame MD5s."""
    w, h, frames = read_ivf(path)
    parser, ctx = Parser(), Vp8Hip(device)
    out = []
    try:
        for data in frames:
            hdr, changed = parser.begin(data)
            if changed:
                ctx.configure(hdr.width, hdr.height, 4, 1)
            _, pm, pc, pv = ctx.ir_map(0)
            ph = ctx.ir_map(0)[0]
            parser.decode_mbs(pm, pc, pv)
            ctypes.memmove(ph, ctypes.byref(hdr), 64)
            ctx.upload(0)
            r = parser.refs
            ctx.decode([(0, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx))], stages)
            parser.swap(hdr)
            if hdr.show_frame:
                out.append(planes_md5(*ctx.download_planes(parser.refs.show_idx)))
            else:
                ctx.sync()
    finally:
        ctx.close()
        parser.close()
    return out
