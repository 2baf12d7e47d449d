"""TEST INFRASTRUCTURE: a VP8 key-frame *stream writer* -- frame-batched IR (include/vp8_ir.h) -> RFC 6386 bitstream -> IVF.

SURVEY.md 8(f)2: with it the tests (and the GPU box, where the reference encoder does not exist) can synthesise
key-frame streams of any size, seed and feature mix -- every 16x16 / chroma / 4x4 mode anywhere, all four segments
with absolute or delta quantiser / filter data, loop-filter deltas, 1..8 token partitions, coefficients up to the
DCT_CAT6 range -- instead of depending on what the reference encoder happens to choose.  It is the inverse of the
host feeder (libvpx.opencl_amd/csrc/host/vp8_parser.c) and of the reference's own vp8_decode_frame /
vp8_kfread_modes / vp8_decode_mb_tokens (vp8/decoder/decodframe.c:690-1077, decodemv.c:70-170,
detokenize.c:183-405); the probability tables are the feeder's (exported by libvpx_hip.so), so a table error cannot
hide: the REAL reference decoder (oracle/_ref) reads the same streams in tests/test_writer_cpu.py.

Written from RFC 6386 (sections 7, 9, 11, 13, 19.2/19.3), not from the reference's encoder.
"""
import ctypes
import struct

import numpy as np

from vp8_testlib import ZIGZAG_COLMAJOR, load_package

_tables = None


def tables():
    global _tables
    if _tables is None:
        L = load_package().load_host()

        def arr(name, n):
            return list((ctypes.c_uint8 * n).in_dll(L, name))
        _tables = {"coef_update": arr("vp8t_coef_update_probs", 1056), "coef": arr("vp8t_default_coef_probs", 1056),
                   "kf_bmode": arr("vp8t_kf_bmode_probs", 900)}
    return _tables


KF_YMODE_PROB = (145, 156, 163, 128)          # RFC 6386 11.2
KF_UV_MODE_PROB = (142, 114, 183)
COEF_BANDS = (0, 1, 2, 3, 6, 4, 5, 6, 6, 6, 6, 6, 6, 6, 6, 7)
CAT_BASE = (5, 7, 11, 19, 35, 67)
CAT_PROBS = ((159,), (165, 145), (173, 148, 140), (176, 155, 140, 135), (180, 157, 141, 134, 130),
             (254, 254, 243, 230, 196, 177, 153, 140, 133, 130, 129))
# 16x16 mode -> the sub-block mode it implies for the contexts of neighbouring B_PRED blocks (RFC 6386 11.3)
IMPLIED_BMODE = {0: 0, 1: 2, 2: 3, 3: 1}      # DC -> B_DC, V -> B_VE, H -> B_HE, TM -> B_TM


class BoolEncoder:
    """RFC 6386 section 7.3."""

    def __init__(self):
        self.out = bytearray()
        self.range = 255
        self.bottom = 0
        self.bit_count = 24

    def _carry(self):
        i = len(self.out) - 1
        while i >= 0 and self.out[i] == 255:
            self.out[i] = 0
            i -= 1
        self.out[i] += 1

    def put(self, bit, prob):
        split = 1 + (((self.range - 1) * prob) >> 8)
        if bit:
            self.bottom += split
            self.range -= split
        else:
            self.range = split
        while self.range < 128:
            self.range <<= 1
            if self.bottom & (1 << 31):
                self._carry()
            self.bottom = (self.bottom << 1) & 0xFFFFFFFF
            self.bit_count -= 1
            if self.bit_count == 0:
                self.out.append(self.bottom >> 24)
                self.bottom &= (1 << 24) - 1
                self.bit_count = 8

    def literal(self, value, bits):
        for i in range(bits - 1, -1, -1):
            self.put((value >> i) & 1, 128)

    def flag_value(self, value, bits):
        """"flag, magnitude, sign" fields of the frame header; the flag is always set"""
        self.put(1, 128)
        self.literal(abs(int(value)), bits)
        self.put(1 if value < 0 else 0, 128)

    def finish(self):
        c, v = self.bit_count, self.bottom
        if v & (1 << (32 - c)):
            self._carry()
        v = (v << (c & 7)) & 0xFFFFFFFF
        c >>= 3
        while c > 0:
            v = (v << 8) & 0xFFFFFFFF
            c -= 1
        for _ in range(4):
            self.out.append(v >> 24)
            v = (v << 8) & 0xFFFFFFFF
        return bytes(self.out)


def _put_ymode(e, m):
    p = KF_YMODE_PROB                  # vp8_kf_ymode_tree: B_PRED = "0", DC "100", V "101", H "110", TM "111"
    if m == 4:
        e.put(0, p[0])
        return
    e.put(1, p[0])
    e.put(m >> 1, p[1])
    e.put(m & 1, p[2] if m < 2 else p[3])


def _put_uvmode(e, m):
    p = KF_UV_MODE_PROB                # DC "0", V "10", H "110", TM "111"
    e.put(1 if m else 0, p[0])
    if m:
        e.put(1 if m > 1 else 0, p[1])
        if m > 1:
            e.put(m - 2, p[2])


_BMODE_CODE = {                        # vp8_bmode_tree (RFC 6386 11.2): (bit, index of its probability) pairs
    0: ((0, 0),), 1: ((1, 0), (0, 1)), 2: ((1, 0), (1, 1), (0, 2)),
    3: ((1, 0), (1, 1), (1, 2), (0, 3), (0, 4)),
    5: ((1, 0), (1, 1), (1, 2), (0, 3), (1, 4), (0, 5)), 6: ((1, 0), (1, 1), (1, 2), (0, 3), (1, 4), (1, 5)),
    4: ((1, 0), (1, 1), (1, 2), (1, 3), (0, 6)),
    7: ((1, 0), (1, 1), (1, 2), (1, 3), (1, 6), (0, 7)),
    8: ((1, 0), (1, 1), (1, 2), (1, 3), (1, 6), (1, 7), (0, 8)), 9: ((1, 0), (1, 1), (1, 2), (1, 3), (1, 6), (1, 7), (1, 8)),
}


def _put_token(e, v, p, after_zero):
    """One coefficient (not EOB) with the 11 probabilities p of its position; returns the context it leaves (0, 1, 2)."""
    a = abs(v)
    if not after_zero:
        e.put(1, p[0])                 # not EOB
    if a == 0:
        e.put(0, p[1])
        return 0
    e.put(1, p[1])
    if a == 1:
        e.put(0, p[2])
    else:
        e.put(1, p[2])
        if a <= 4:
            e.put(0, p[3])
            if a == 2:
                e.put(0, p[4])
            else:
                e.put(1, p[4])
                e.put(a - 3, p[5])
        else:
            e.put(1, p[3])
            cat = 0 if a < 7 else 1 if a < 11 else 2 if a < 19 else 3 if a < 35 else 4 if a < 67 else 5
            if cat < 2:
                e.put(0, p[6])
                e.put(cat, p[7])
            else:
                e.put(1, p[6])
                e.put(0 if cat < 4 else 1, p[8])
                e.put((cat - 2) & 1, p[9] if cat < 4 else p[10])
            extra, probs = a - CAT_BASE[cat], CAT_PROBS[cat]
            assert 0 <= extra < (1 << len(probs)), "coefficient out of range for DCT_CAT6"
            for i, pb in enumerate(probs):
                e.put((extra >> (len(probs) - 1 - i)) & 1, pb)
    e.put(1 if v < 0 else 0, 128)
    return 1 if a == 1 else 2


def write_key_frame(hdr, mbs, coef, log2_parts=0, prob_skip_false=200, segment_tree_probs=(120, 90, 200), rng=None):
    """hdr: FrameHdr (frame_type 0); mbs uint8[n,64], coef int16[n,400] as produced by vp8_testlib.synth_ir or the feeder.
    Returns the compressed frame.  Every MB with the skip flag is coded as skipped (mb_no_coeff_skip = 1)."""
    T = tables()
    cols, rows = hdr.mb_cols, hdr.mb_rows
    assert hdr.frame_type == 0
    zz = np.array(ZIGZAG_COLMAJOR)
    e = BoolEncoder()
    e.literal(hdr.color_space, 1)
    e.literal(hdr.clamping_type, 1)
    e.literal(hdr.segmentation_enabled, 1)
    if hdr.segmentation_enabled:
        e.literal(1, 1)                                  # update_mb_segmentation_map
        e.literal(1, 1)                                  # update_segment_feature_data
        e.literal(hdr.mb_segment_abs_delta, 1)
        for i in range(4):
            e.flag_value(hdr.segment_quant[i], 7)
        for i in range(4):
            e.flag_value(hdr.segment_lf[i], 6)
        for p in segment_tree_probs:
            e.literal(1, 1)
            e.literal(p, 8)
    e.literal(hdr.filter_type, 1)
    e.literal(hdr.filter_level, 6)
    e.literal(hdr.sharpness_level, 3)
    e.literal(hdr.mode_ref_lf_delta_enabled, 1)
    if hdr.mode_ref_lf_delta_enabled:
        e.literal(1, 1)                                  # mode_ref_lf_delta_update
        for i in range(4):
            e.flag_value(hdr.ref_lf_deltas[i], 6)
        for i in range(4):
            e.flag_value(hdr.mode_lf_deltas[i], 6)
    e.literal(log2_parts, 2)
    e.literal(hdr.base_qindex, 7)
    for d in (hdr.y1dc_delta_q, hdr.y2dc_delta_q, hdr.y2ac_delta_q, hdr.uvdc_delta_q, hdr.uvac_delta_q):
        if d:
            e.flag_value(d, 4)
        else:
            e.put(0, 128)
    e.literal(1, 1)                                      # refresh_entropy_probs
    for p in T["coef_update"]:                           # no coefficient probability updates: the defaults stay
        e.put(0, p)
    e.literal(1, 1)                                      # mb_no_coeff_skip
    e.literal(prob_skip_false, 8)

    # ---- per-MB modes (vp8_kfread_modes, decodemv.c:70-170)
    above_b = [0] * (cols * 4)                           # B_DC_PRED outside the frame
    for r in range(rows):
        left_b = [0] * 4
        for c in range(cols):
            m = mbs[r * cols + c]
            if hdr.segmentation_enabled:
                s, sp = int(m[4]), segment_tree_probs
                e.put(s >> 1, sp[0])
                e.put(s & 1, sp[1] if s < 2 else sp[2])
            e.put(int(m[3]) & 1, prob_skip_false)
            ym = int(m[0])
            _put_ymode(e, ym)
            if ym == 4:
                bm = [int(x) for x in m[40:56]]
                for b in range(16):
                    A = above_b[c * 4 + (b & 3)] if b < 4 else bm[b - 4]
                    Lm = left_b[b >> 2] if (b & 3) == 0 else bm[b - 1]
                    pr = T["kf_bmode"][(A * 10 + Lm) * 9:(A * 10 + Lm) * 9 + 9]
                    for bit, pi in _BMODE_CODE[bm[b]]:
                        e.put(bit, pr[pi])
                for i in range(4):
                    above_b[c * 4 + i] = bm[12 + i]
                    left_b[i] = bm[4 * i + 3]
            else:
                for i in range(4):
                    above_b[c * 4 + i] = left_b[i] = IMPLIED_BMODE[ym]
            _put_uvmode(e, int(m[1]))
    first = e.finish()

    # ---- coefficient tokens (vp8_decode_mb_tokens, detokenize.c:183-405), partition = MB row modulo their number
    nparts = 1 << log2_parts
    encs = [BoolEncoder() for _ in range(nparts)]
    cp = T["coef"]
    aY, aU, aV, aY2 = [0] * (cols * 4), [0] * (cols * 2), [0] * (cols * 2), [0] * cols
    for r in range(rows):
        te = encs[r % nparts]
        lY, lU, lV, lY2 = [0] * 4, [0] * 2, [0] * 2, 0

        def block(vals, btype, first_c, ctx):
            """vals: 16 coefficients in zig-zag order.  Returns the block's "has coefficients" context flag."""
            last = -1
            for i in range(15, first_c - 1, -1):
                if vals[i]:
                    last = i
                    break
            after_zero = False
            for i in range(first_c, 16):
                p0 = ((btype * 8 + COEF_BANDS[i]) * 3 + ctx) * 11
                p = cp[p0:p0 + 11]
                if i > last:
                    assert not after_zero
                    te.put(0, p[0])                      # EOB
                    break
                ctx = _put_token(te, int(vals[i]), p, after_zero)
                after_zero = vals[i] == 0
            return 1 if last >= first_c else 0

        for c in range(cols):
            m = mbs[r * cols + c]
            ym = int(m[0])
            has_y2 = ym != 4
            if int(m[3]) & 1:                            # skipped: contexts cleared, Y2's only if the MB has a Y2
                for i in range(4):
                    aY[c * 4 + i] = 0
                    lY[i] = 0
                for i in range(2):
                    aU[c * 2 + i] = aV[c * 2 + i] = 0
                    lU[i] = lV[i] = 0
                if has_y2:
                    aY2[c] = 0
                    lY2 = 0
                continue
            q = coef[r * cols + c]
            if has_y2:
                f = block(q[384 + zz], 1, 0, aY2[c] + lY2)
                aY2[c] = lY2 = f
            for b in range(16):
                bx, by = b & 3, b >> 2
                f = block(q[b * 16 + zz], 0 if has_y2 else 3, 1 if has_y2 else 0, aY[c * 4 + bx] + lY[by])
                aY[c * 4 + bx] = lY[by] = f
            for b in range(4):
                bx, by = b & 1, b >> 1
                f = block(q[256 + b * 16 + zz], 2, 0, aU[c * 2 + bx] + lU[by])
                aU[c * 2 + bx] = lU[by] = f
            for b in range(4):
                bx, by = b & 1, b >> 1
                f = block(q[320 + b * 16 + zz], 2, 0, aV[c * 2 + bx] + lV[by])
                aV[c * 2 + bx] = lV[by] = f
    parts = [x.finish() for x in encs]

    # ---- frame tag (RFC 6386 9.1), key-frame start code and dimensions, partition sizes
    tag = 0 | (hdr.version << 1) | (1 << 4) | (len(first) << 5)
    out = bytearray(struct.pack("<I", tag)[:3])
    out += b"\x9d\x01\x2a" + struct.pack("<HH", hdr.width & 0x3FFF, hdr.height & 0x3FFF)
    out += first
    for p in parts[:-1]:
        out += struct.pack("<I", len(p))[:3]
    for p in parts:
        out += p
    return bytes(out)


def write_ivf(path, width, height, frames):
    with open(path, "wb") as f:
        f.write(b"DKIF" + struct.pack("<HHIHHIIII", 0, 32, 0x30385056, width, height, 30, 1, len(frames), 0))
        for i, fr in enumerate(frames):
            f.write(struct.pack("<IQ", len(fr), i))
            f.write(fr)
