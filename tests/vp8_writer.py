"""TEST INFRASTRUCTURE: a VP8 *stream writer* (key frames, and since round 4 inter frames) -- frame-batched IR
(include/vp8_ir.h) -> RFC 6386 bitstream -> IVF.

SURVEY.md 8(f)2: with it the tests (and the GPU box, where the reference encoder does not exist) can synthesise
key-frame streams of any size, seed and feature mix -- every 16x16 / chroma / 4x4 mode anywhere, all four segments
with absolute or delta quantiser / filter data, loop-filter deltas, 1..8 token partitions, coefficients up to the
DCT_CAT6 range -- instead of depending on what the reference encoder happens to choose.  It is the inverse of the
host feeder (libvpx.opencl_amd/csrc/host/vp8_parser.c) and of the reference's own vp8_decode_frame /
vp8_kfread_modes / vp8_decode_mb_tokens (vp8/decoder/decodframe.c:690-1077, decodemv.c:70-170,
detokenize.c:183-405); the probability tables are the feeder's (exported by libvpx_hip.so), so a table error cannot
hide: the REAL reference decoder (oracle/_ref) reads the same streams in tests/test_writer_cpu.py.

Inter frames (write_inter_frame): every reference frame, ZEROMV / NEARESTMV / NEARMV / NEWMV / SPLITMV with all four
partitionings and all four sub-vector codes, intra macroblocks among them, golden / alt-ref sign bias and buffer copies, and the
segmentation cases no encoder at hand produces -- a segment map that is KEPT from frame to frame while the per-segment data
change, a map updated by an inter frame.  The writer walks the macroblocks as the DECODER does (vp8_find_near_mvs and the mode
contexts, vp8/decoder/decodemv.c:323-569, restated below from RFC 6386 section 18.3) and codes what the IR asks for where the
bitstream can say it -- a NEARESTMV macroblock gets the vector the neighbours give it, whatever the IR had --, so the IR a stream
really carries is what the host feeder reads back, and the REAL reference decoder arbitrates (tests/test_writer_cpu.py).

Written from RFC 6386 (sections 7, 9, 11, 13, 16, 18, 19.2/19.3), not from the reference's encoder.
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
                   "kf_bmode": arr("vp8t_kf_bmode_probs", 900), "mvc": arr("vp8t_default_mv_context", 38),
                   "mv_update": arr("vp8t_mv_update_probs", 38),
                   "mode_contexts": list((ctypes.c_int32 * 24).in_dll(L, "vp8t_mode_contexts"))}
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


def _write_tokens(hdr, mbs, coef, log2_parts):
    """The token partitions of a frame (vp8_decode_mb_tokens, detokenize.c:183-405), partition = MB row modulo their number."""
    T = tables()
    cols, rows = hdr.mb_cols, hdr.mb_rows
    zz = np.array(ZIGZAG_COLMAJOR)
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
            has_y2 = ym not in (4, 9)
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

    return parts


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

    parts = _write_tokens(hdr, mbs, coef, log2_parts)

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


# ---- inter frames -------------------------------------------------------------------------------------------------------------
YMODE_PROB = (112, 86, 140, 37)                # RFC 6386 16.2: the defaults a key frame restores
UV_MODE_PROB = (162, 101, 204)
BMODE_PROB = (120, 90, 79, 133, 87, 85, 80, 111, 151)      # sub-block modes of inter frames: no contexts (RFC 6386 16.2)
SPLIT_PROB = (110, 111, 150)                   # vp8_mbsplit_probs: 16 parts "0", quarters "10", 16x8 "110", 8x16 "111"
SUBMV_PROB = ((147, 136, 18), (223, 1, 34), (106, 145, 1), (208, 1, 1), (179, 121, 1), (223, 1, 34), (179, 121, 1), (208, 1, 1))


def _put_mv_component(e, v, pr):
    """RFC 6386 section 17: v in quarter-pel units (|v| <= 1023), pr = the component's 19 probabilities."""
    x = abs(v)
    if x < 8:
        e.put(0, pr[0])
        e.put(x >> 2, pr[2])                                   # vp8_small_mvtree
        if x < 4:
            e.put((x >> 1) & 1, pr[3])
            e.put(x & 1, pr[4] if x < 2 else pr[5])
        else:
            e.put((x >> 1) & 1, pr[6])
            e.put(x & 1, pr[7] if x < 6 else pr[8])
    else:
        e.put(1, pr[0])
        for i in range(3):
            e.put((x >> i) & 1, pr[9 + i])
        for i in range(9, 3, -1):
            e.put((x >> i) & 1, pr[9 + i])
        if x & 0xfff0:                                         # (x >= 16: bit 3 is coded; below that it is implied)
            e.put((x >> 3) & 1, pr[9 + 3])
    if x:
        e.put(1 if v < 0 else 0, pr[1])


def _clamp_mv(mv, edges):
    (r, c), (left, right, top, bottom) = mv, edges
    return (min(max(r, top), bottom), min(max(c, left), right))


def write_inter_frame(hdr, mbs, coef, mvs, log2_parts=0, prob_skip_false=200, segment_tree_probs=(120, 90, 200), update_map=None,
                      update_data=None, prob_intra=150, prob_last=140, prob_gf=100, lf_delta_update=True):
    """hdr: FrameHdr (frame_type 1); mbs, coef, mvs as vp8_testlib.synth_ir(inter=True) makes them.  update_map / update_data:
    whether the frame codes the macroblocks' segment ids / the segments' data (default: both when segmentation is on; False:
    the decoder keeps what it has -- the caller's mbs[:, 4] and hdr.segment_* then have to say what that is, if they are to be
    compared with anything).  Returns (compressed frame, mbs as the decoder will read them, mvs likewise): modes and vectors
    follow from the neighbours wherever the bitstream derives them (NEARESTMV / NEARMV vectors, clamping flags)."""
    T = tables()
    cols, rows = hdr.mb_cols, hdr.mb_rows
    assert hdr.frame_type == 1
    seg = bool(hdr.segmentation_enabled)
    update_map = seg if update_map is None else (bool(update_map) and seg)
    update_data = seg if update_data is None else (bool(update_data) and seg)
    e = BoolEncoder()
    e.literal(1 if seg else 0, 1)
    if seg:
        e.literal(1 if update_map else 0, 1)
        e.literal(1 if update_data else 0, 1)
        if update_data:
            e.literal(hdr.mb_segment_abs_delta, 1)
            for i in range(4):
                e.flag_value(hdr.segment_quant[i], 7)
            for i in range(4):
                e.flag_value(hdr.segment_lf[i], 6)
        if update_map:
            for p in segment_tree_probs:
                e.literal(1, 1)
                e.literal(p, 8)
    e.literal(hdr.filter_type, 1)
    e.literal(hdr.filter_level, 6)
    e.literal(hdr.sharpness_level, 3)
    e.literal(hdr.mode_ref_lf_delta_enabled, 1)
    if hdr.mode_ref_lf_delta_enabled:
        e.literal(1 if lf_delta_update else 0, 1)
        if lf_delta_update:
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
    # reference-buffer bookkeeping of inter frames (RFC 6386 9.7; decodframe.c:949-1018)
    e.literal(hdr.refresh_golden, 1)
    e.literal(hdr.refresh_alt, 1)
    if not hdr.refresh_golden:
        e.literal(hdr.copy_buffer_to_gf, 2)
    if not hdr.refresh_alt:
        e.literal(hdr.copy_buffer_to_arf, 2)
    e.literal(hdr.sign_bias_golden, 1)
    e.literal(hdr.sign_bias_alt, 1)
    e.literal(1, 1)                                      # refresh_entropy_probs
    e.literal(hdr.refresh_last, 1)
    for p in T["coef_update"]:                           # no coefficient probability updates
        e.put(0, p)
    e.literal(1, 1)                                      # mb_no_coeff_skip
    e.literal(prob_skip_false, 8)
    e.literal(prob_intra, 8)
    e.literal(prob_last, 8)
    e.literal(prob_gf, 8)
    e.literal(0, 1)                                      # intra_16x16_prob_update_flag
    e.literal(0, 1)                                      # intra_chroma_prob_update_flag
    for p in T["mv_update"]:                             # no motion-vector probability updates
        e.put(0, p)
    mvc = (T["mvc"][:19], T["mvc"][19:])                 # row, column
    MC = T["mode_contexts"]
    bias = {1: 0, 2: int(hdr.sign_bias_golden), 3: int(hdr.sign_bias_alt)}

    out_mbs = mbs.copy()
    out_mvs = np.zeros_like(mvs)
    # what a macroblock leaves for its neighbours: vector (SPLITMV: the last block's; intra: 0), reference frame, mode, sixteen vectors
    blank = {"mv": (0, 0), "ref": 0, "ymode": 0, "b": [(0, 0)] * 16}
    above_row = [dict(blank) for _ in range(cols + 1)]   # [c + 1]; [0] = left of the frame
    for r in range(rows):
        new_row = [dict(blank)]
        left, aboveleft = dict(blank), dict(blank)
        for c in range(cols):
            n = r * cols + c
            m = mbs[n]
            above = above_row[c + 1]
            if update_map:
                sid, sp = int(m[4]), segment_tree_probs
                e.put(sid >> 1, sp[0])
                e.put(sid & 1, sp[1] if sid < 2 else sp[2])
            e.put(int(m[3]) & 1, prob_skip_false)
            ref = int(m[2])
            ym = int(m[0])
            me = {"mv": (0, 0), "ref": ref, "ymode": ym, "b": [(0, 0)] * 16}
            out_mbs[n, 3] &= 0xfd
            if ref == 0:
                e.put(0, prob_intra)
                assert ym <= 4
                yp = YMODE_PROB                           # vp8_ymode_tree: DC "0", V "100", H "101", TM "110", B_PRED "111"
                if ym == 0:
                    e.put(0, yp[0])
                else:
                    e.put(1, yp[0])
                    if ym in (1, 2):
                        e.put(0, yp[1]); e.put(ym - 1, yp[2])
                    else:
                        e.put(1, yp[1]); e.put(1 if ym == 4 else 0, yp[3])
                if ym == 4:
                    for b in range(16):
                        for bit, pi in _BMODE_CODE[int(m[40 + b])]:
                            e.put(bit, BMODE_PROB[pi])
                _put_uvmode_with(e, int(m[1]), UV_MODE_PROB)
            else:
                e.put(1, prob_intra)
                e.put(0 if ref == 1 else 1, prob_last)
                if ref != 1:
                    e.put(ref - 2, prob_gf)
                # vp8_find_near_mvs as the decoder inlines it (decodemv.c:340-420; RFC 6386 18.3)
                near, cnt, k = [(0, 0)] * 4, [0, 0, 0, 0], 0
                for nb, weight, first in ((above, 2, True), (left, 2, False), (aboveleft, 1, False)):
                    if nb["ref"] == 0:
                        continue
                    if nb["mv"] != (0, 0):
                        t = nb["mv"]
                        if bias[nb["ref"]] != bias[ref]:
                            t = (-t[0], -t[1])
                        if first or t != near[k]:
                            k += 1
                            near[k] = t
                        cnt[k] += weight
                    else:
                        cnt[0] += weight
                edges = (-((c * 16) << 3) - 128, (((cols - 1 - c) * 16) << 3) + 128, -((r * 16) << 3) - 128, (((rows - 1 - r) * 16) << 3) + 128)
                outside = lambda v: v[1] < edges[0] or v[1] > edges[1] or v[0] < edges[2] or v[0] > edges[3]
                if ym == 7:
                    e.put(0, MC[cnt[0] * 4 + 0])
                    mv = (0, 0)
                    me["b"] = [mv] * 16
                else:
                    e.put(1, MC[cnt[0] * 4 + 0])
                    if cnt[3] and near[k] == near[1]:
                        cnt[1] += 1
                    cnt[3] = ((above["ymode"] == 9) + (left["ymode"] == 9)) * 2 + (aboveleft["ymode"] == 9)
                    if cnt[2] > cnt[1]:
                        cnt[1], cnt[2] = cnt[2], cnt[1]
                        near[1], near[2] = near[2], near[1]
                    if ym == 5:
                        e.put(0, MC[cnt[1] * 4 + 1])
                        mv = _clamp_mv(near[1], edges)
                        me["b"] = [mv] * 16
                    else:
                        e.put(1, MC[cnt[1] * 4 + 1])
                        if ym == 6:
                            e.put(0, MC[cnt[2] * 4 + 2])
                            mv = _clamp_mv(near[2], edges)
                            me["b"] = [mv] * 16
                        else:
                            e.put(1, MC[cnt[2] * 4 + 2])
                            if cnt[1] >= cnt[0]:
                                near[0] = near[1]
                            best = _clamp_mv(near[0], edges)

                            def put_new(target):
                                """target - best as a coded vector (quarter-pel magnitudes up to 1023); returns what the decoder gets"""
                                d = [int(np.clip((int(target[i]) - best[i]) // 2, -1023, 1023)) for i in range(2)]
                                _put_mv_component(e, d[0], mvc[0])
                                _put_mv_component(e, d[1], mvc[1])
                                return (best[0] + 2 * d[0], best[1] + 2 * d[1])
                            if ym == 8:
                                e.put(0, MC[cnt[3] * 4 + 3])
                                mv = put_new(tuple(int(x) for x in mvs[n, 0]))
                                if outside(mv):
                                    out_mbs[n, 3] |= 2
                                me["b"] = [mv] * 16
                            else:
                                assert ym == 9
                                e.put(1, MC[cnt[3] * 4 + 3])
                                sp = int(m[5])                          # 0 = 16x8, 1 = 8x16, 2 = quarters, 3 = sixteen
                                if sp == 3:
                                    e.put(0, SPLIT_PROB[0])
                                else:
                                    e.put(1, SPLIT_PROB[0])
                                    if sp == 2:
                                        e.put(0, SPLIT_PROB[1])
                                    else:
                                        e.put(1, SPLIT_PROB[1]); e.put(sp, SPLIT_PROB[2])
                                nparts = (2, 2, 4, 16)[sp]
                                bvs = [(0, 0)] * 16
                                part_of = lambda bb: (bb >> 3, (bb >> 1) & 1, ((bb >> 3) << 1) | ((bb >> 1) & 1), bb)[sp]
                                for j in range(nparts):
                                    kb = (8 * j, 2 * j, (j & 1) * 2 + (j >> 1) * 8, j)[sp]
                                    leftmv = bvs[kb - 1] if kb & 3 else left["b"][kb + 3]
                                    abovemv = bvs[kb - 4] if kb >= 4 else above["b"][kb + 12]
                                    pr = SUBMV_PROB[((abovemv == (0, 0)) << 2) | ((leftmv == (0, 0)) << 1) | (leftmv == abovemv)]
                                    target = tuple(int(x) for x in mvs[n, kb])
                                    if target == leftmv:
                                        e.put(0, pr[0]); v = leftmv
                                    elif target == abovemv:
                                        e.put(1, pr[0]); e.put(0, pr[1]); v = abovemv
                                    elif target == (0, 0):
                                        e.put(1, pr[0]); e.put(1, pr[1]); e.put(0, pr[2]); v = (0, 0)
                                    else:
                                        e.put(1, pr[0]); e.put(1, pr[1]); e.put(1, pr[2]); v = put_new(target)
                                    if outside(v):
                                        out_mbs[n, 3] |= 2
                                    for bb in range(16):
                                        if part_of(bb) == j:
                                            bvs[bb] = v
                                mv = bvs[15]
                                me["b"] = bvs
                me["mv"] = mv
                for bb in range(16):
                    out_mvs[n, bb] = me["b"][bb]
            new_row.append(me)
            aboveleft = above
            left = me
        above_row = new_row
    first = e.finish()
    parts = _write_tokens(hdr, mbs, coef, log2_parts)
    tag = 1 | (hdr.version << 1) | (int(hdr.show_frame) << 4) | (len(first) << 5)
    out = bytearray(struct.pack("<I", tag)[:3])
    out += first
    for p in parts[:-1]:
        out += struct.pack("<I", len(p))[:3]
    for p in parts:
        out += p
    return bytes(out), out_mbs, out_mvs


def _put_uvmode_with(e, m, p):          # DC "0", V "10", H "110", TM "111"
    e.put(1 if m else 0, p[0])
    if m:
        e.put(1 if m > 1 else 0, p[1])
        if m > 1:
            e.put(m - 2, p[2])


def write_ivf(path, width, height, frames):
    with open(path, "wb") as f:
        f.write(b"DKIF" + struct.pack("<HHIHHIIII", 0, 32, 0x30385056, width, height, 30, 1, len(frames), 0))
        for i, fr in enumerate(frames):
            f.write(struct.pack("<IQ", len(fr), i))
            f.write(fr)
