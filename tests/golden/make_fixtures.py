#!/usr/bin/env python3
"""Generate the golden IVF + per-frame MD5 fixtures with the REAL reference.

Runs only in the dev container: needs oracle/_ref/{vpxenc_ref,ref_md5} (built by
`make -C oracle ref` from /root/reference; see oracle/Makefile).  The outputs
(tests/golden/*.ivf, *.md5) are data: compressed VP8 streams produced by the
reference encoder from numpy-synthesised (seeded) I420 input, and the MD5 of every
frame the reference *decoder* (generic-C path) shows for them, in the reference's
`decode_to_md5` line format (examples/decode_to_md5.txt:28-47).

Fixture matrix follows SURVEY.md §8c.  Everything is deterministic (single thread,
fixed seeds); re-running reproduces the files byte for byte.

    python tests/golden/make_fixtures.py [name ...]
"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
VPXENC = os.path.join(REFDIR, "vpxenc_ref")
REFMD5 = os.path.join(REFDIR, "ref_md5")
VPXDEC = os.path.join(REFDIR, "vpxdec_ref")
ROIENC = os.path.join(REFDIR, "roi_enc")        # oracle/roi_enc.c: the reference ENCODER with a region-of-interest map installed


def synth_i420(w, h, frames, seed, noise=6, speed=(1.375, 0.625)):
    """Moving smooth texture + 32-px checker + moving blocks + uniform noise.

    Sub-pel global motion (so inter MVs hit all eight 1/8-pel chroma phases and
    the odd and even quarter-pel luma phases), locally moving squares (SPLITMV /
    intra-in-P candidates), flat regions (DC/V/H/TM) and texture (B_PRED).
    """
    rng = np.random.default_rng(seed)
    pad = 64
    tw, th = w + 2 * pad + int(abs(speed[0]) * frames) + 8, h + 2 * pad + int(abs(speed[1]) * frames) + 8
    # smooth texture: sum of a few random sinusoids + coarse noise upsampled
    yy, xx = np.mgrid[0:th, 0:tw].astype(np.float32)
    tex = 128 + 40 * np.sin(xx / 37.0 + yy / 91.0) + 30 * np.cos(xx / 11.0 - yy / 23.0)
    coarse = rng.uniform(-35, 35, size=(th // 8 + 2, tw // 8 + 2)).astype(np.float32)
    tex += np.kron(coarse, np.ones((8, 8), np.float32))[:th, :tw]
    tex += ((xx.astype(np.int32) // 32 + yy.astype(np.int32) // 32) % 2) * 24 - 12
    cu = 128 + 50 * np.sin(xx / 53.0) * np.cos(yy / 41.0)
    cv = 128 + 50 * np.cos(xx / 29.0 + yy / 67.0)
    out = bytearray()
    nsq = 6
    sq = rng.uniform(0, 1, size=(nsq, 4))
    for t in range(frames):
        ox, oy = pad + speed[0] * t, pad + speed[1] * t
        ix, iy = int(np.floor(ox)), int(np.floor(oy))
        fx, fy = ox - ix, oy - iy

        def samp(p, sx, sy, ww, hh, fx=fx, fy=fy):
            a = p[sy:sy + hh + 1, sx:sx + ww + 1]
            return ((1 - fy) * ((1 - fx) * a[:-1, :-1] + fx * a[:-1, 1:]) + fy * ((1 - fx) * a[1:, :-1] + fx * a[1:, 1:]))

        y = samp(tex, ix, iy, w, h).copy()
        u = samp(cu, ix, iy, w, h)[::2, ::2].copy()
        v = samp(cv, ix, iy, w, h)[::2, ::2].copy()
        # flat band (favours 16x16 DC/V/H prediction, low-activity loop-filter paths)
        y[: h // 6, : w // 3] = 90 + 0.1 * np.arange(w // 3)[None, :]
        for k in range(nsq):
            s = 24 + int(sq[k, 2] * 40)
            px = int((sq[k, 0] * (w - s) + (k - 2.5) * 3.25 * t) % max(1, w - s))
            py = int((sq[k, 1] * (h - s) + (2.5 - k) * 1.75 * t) % max(1, h - s))
            y[py:py + s, px:px + s] = 40 + 30 * k + 20 * np.sin(np.arange(s) / (2.0 + k))[None, :]
        if noise:
            y += rng.integers(-noise, noise + 1, size=y.shape)
        out += np.clip(np.rint(y), 0, 255).astype(np.uint8).tobytes()
        ch, cw = (h + 1) // 2, (w + 1) // 2
        out += np.clip(np.rint(u[:ch, :cw]), 0, 255).astype(np.uint8).tobytes()
        out += np.clip(np.rint(v[:ch, :cw]), 0, 255).astype(np.uint8).tobytes()
    return bytes(out)


COMMON = ["--ivf", "--i420", "-p", "1", "--lag-in-frames=0", "-t", "1"]
ALLKEY = ["--kf-max-dist=0", "--kf-min-dist=0"]
INTER = ["--kf-max-dist=9999", "--auto-alt-ref=0"]

# name: (w, h, frames, seed, noise, encoder args[, global motion in px/frame])
FIXTURES = {
    # (1) config-1 plumbing stream: 640x360, 10 key frames, profile 0, normal loop filter
    "kf_640x360": (640, 360, 10, 11, 10, ALLKEY + ["--good", "--cpu-used=4", "--end-usage=cq", "--cq-level=14",
                                                    "--target-bitrate=6000"]),
    # (2) headline stream: 1920x1080, 10 key frames, profile 0, loop filter on
    "kf_1920x1080": (1920, 1080, 10, 7, 6, ALLKEY + ["--good", "--cpu-used=5", "--end-usage=cq", "--cq-level=20",
                                                      "--target-bitrate=20000"]),
    # (2b) SURVEY 8(d)'s "noisy" set: uniform +-16 noise at a quantiser where nearly every block keeps coefficients behind its first
    # (the byte model of the roofline is a DENSE-coefficient model; the headline stream codes 13.7 of a macroblock's 25 blocks):
    # two key frames, loop filter on -- bench.py's config.dense_content
    "kf_dense_1920x1080": (1920, 1080, 2, 13, 16, ALLKEY + ["--good", "--cpu-used=5", "--end-usage=cq", "--cq-level=10",
                                                            "--min-q=8", "--max-q=16", "--target-bitrate=400000"]),
    # (3) loop filter OFF (q=0 -> filter_level 0) and very dense coefficients
    "kf_q0_176x144": (176, 144, 4, 3, 12, ALLKEY + ["--good", "--cpu-used=2", "--min-q=0", "--max-q=0",
                                                     "--target-bitrate=40000"]),
    # (4) 1080p key + 9 inter frames, 6-tap MC, normal LF
    "p_1920x1080": (1920, 1080, 10, 7, 6, INTER + ["--good", "--cpu-used=5", "--end-usage=cq", "--cq-level=20",
                                                    "--target-bitrate=8000"]),
    # (4b) the same kind of stream at a quality where the P frames carry real residuals (a third or more of their blocks with
    # more than a DC coefficient, most macroblocks inter with fractional MVs): the throughput probe of BASELINE configs[2]
    "p_dense_1920x1080": (1920, 1080, 4, 9, 10, INTER + ["--good", "--cpu-used=4", "--end-usage=cq", "--cq-level=28",
                                                          "--min-q=20", "--max-q=50", "--target-bitrate=200000"]),
    # (5) profiles 1-3 (bilinear MC / simple LF / full-pixel), SPLITMV, 4 token partitions
    "p_prof1_640x360": (640, 360, 10, 21, 8, INTER + ["--good", "--cpu-used=0", "--profile=1", "--token-parts=2",
                                                       "--target-bitrate=1500"]),
    "p_prof2_640x360": (640, 360, 10, 22, 8, INTER + ["--good", "--cpu-used=0", "--profile=2", "--token-parts=2",
                                                       "--target-bitrate=1500"]),
    "p_prof3_640x360": (640, 360, 10, 23, 8, INTER + ["--good", "--cpu-used=0", "--profile=3", "--token-parts=2",
                                                       "--target-bitrate=1500"]),
    # profile 0 with best-quality search: SPLITMV 16/8x8/4x4, intra MBs in P frames, 8 partitions
    "p_split_352x288": (352, 288, 12, 31, 8, INTER + ["--best", "--cpu-used=0", "--token-parts=3",
                                                       "--target-bitrate=900"]),
    # (6) sizes that are not multiples of 16 (crop) incl. odd width/height
    "p_odd_130x98": (130, 98, 8, 41, 8, INTER + ["--good", "--cpu-used=1", "--target-bitrate=300"]),
    "kf_odd_67x45": (67, 45, 3, 42, 8, ALLKEY + ["--good", "--cpu-used=1", "--target-bitrate=300"]),
    # (7) 4K key frames (config 5)
    "kf_3840x2160": (3840, 2160, 3, 5, 4, ALLKEY + ["--good", "--cpu-used=6", "--end-usage=cq", "--cq-level=24",
                                                     "--target-bitrate=30000"]),
    # golden / alt-ref with hidden (show_frame=0) frames: two-pass, lagged, slow motion so that the encoder's
    # ARF decision fires (5 hidden frames among 95 packets)
    "p_arf_176x144": (176, 144, 90, 52, 2, ["--ivf", "--i420", "-p", "2", "-t", "1", "--kf-max-dist=9999",
                                           "--auto-alt-ref=1", "--lag-in-frames=16", "--good", "--cpu-used=0",
                                           "--target-bitrate=200", "--arnr-maxframes=5", "--arnr-strength=3"],
                      (0.25, 0.125)),
    # sharpness != 0 (loop-filter limit tables), error-resilient stream
    "p_sharp_320x240": (320, 240, 8, 61, 10, INTER + ["--good", "--cpu-used=1", "--sharpness=5",
                                                       "--error-resilient=1", "--target-bitrate=500"]),
    # 1080p key frames coded with 8 token partitions (the reference encoder's --token-parts=3): what the threaded token
    # decode of the feeder is for
    "kf_8part_1920x1080": (1920, 1080, 3, 81, 6, ALLKEY + ["--good", "--cpu-used=5", "--end-usage=cq", "--cq-level=20",
                                                             "--target-bitrate=20000", "--token-parts=3"]),
    # segmentation in inter frames: the real-time encoder's cyclic refresh (error-resilient mode) codes a segment map with every
    # frame -- per-segment quantisers and loop-filter levels, the map's tree in the first partition of inter frames
    "p_seg_176x144": (176, 144, 30, 77, 4, INTER + ["--rt", "--error-resilient=1", "--cpu-used=-6", "--target-bitrate=150"]),
    # low bitrate: high filter levels, many skipped MBs (mb_skip_coeff / skip_lf paths)
    "p_lowrate_640x360": (640, 360, 10, 71, 4, INTER + ["--good", "--cpu-used=3", "--target-bitrate=150"]),
    # inter frames with segmentation ON that code no map (they keep the one they have: decodemv.c:594-606): the reference
    # encoder with a region-of-interest map installed before frames 3 and 8 (oracle/roi_enc.c: VP8E_SET_ROI_MAP; this encoder
    # never gets to code the map or its data with an inter frame, see there).  ("roi": arguments of roi_enc, not vpxenc's)
    "p_roi_640x360": (640, 360, 12, 91, 5, ["roi", "3", "8"]),
}


def run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stdout + r.stderr)
        raise SystemExit("FAILED: " + " ".join(cmd))


def stream_md5(name):
    """`vpxdec --md5 --i420` of the REFERENCE: one digest over all shown frames (vpxdec.c:322-383)."""
    ivf = os.path.join(HERE, name + ".ivf")
    r = subprocess.run([VPXDEC, "--md5", "--i420", ivf], capture_output=True, text=True, check=True)
    with open(os.path.join(HERE, name + ".vpxdec_md5"), "w") as f:
        f.write(r.stdout.split()[0] + "\n")


def ref_controls(name):
    """What the REFERENCE decoder answers, packet by packet, to VP8D_GET_LAST_REF_UPDATES, VP8D_GET_LAST_REF_USED and
    VP8D_GET_FRAME_CORRUPTED (vp8/vp8_dx_iface.c:653-720), through its public API in oracle/_ref/libvpxref.so."""
    import ctypes
    L = ctypes.CDLL(os.path.join(REFDIR, "libvpxref.so"))
    L.vpx_codec_vp8_dx.restype = ctypes.c_void_p
    L.vpx_codec_dec_init_ver.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int]
    L.vpx_codec_decode.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_long]
    L.vpx_codec_control_.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    data = open(os.path.join(HERE, name + ".ivf"), "rb").read()
    pos, frames = 32, []
    while pos + 12 <= len(data):
        sz = int.from_bytes(data[pos:pos + 4], "little")
        frames.append(data[pos + 12:pos + 12 + sz])
        pos += 12 + sz
    ctx = ctypes.create_string_buffer(256)
    assert L.vpx_codec_dec_init_ver(ctx, L.vpx_codec_vp8_dx(), None, 0, 2 + 2 + 1) == 0
    with open(os.path.join(HERE, name + ".refctl"), "w") as f:
        f.write("# packet  VP8D_GET_LAST_REF_UPDATES  VP8D_GET_LAST_REF_USED  VP8D_GET_FRAME_CORRUPTED  (reference decoder)\n")
        for i, fr in enumerate(frames):
            assert L.vpx_codec_decode(ctx, fr, len(fr), None, 0) == 0
            v = [ctypes.c_int(-1) for _ in range(3)]
            for k, ctl in enumerate((256, 258, 257)):     # VP8D_GET_LAST_REF_UPDATES, VP8D_GET_LAST_REF_USED, VP8D_GET_FRAME_CORRUPTED (vpx/vp8dx.h:53-61)
                assert L.vpx_codec_control_(ctx, ctl, ctypes.byref(v[k])) == 0
            f.write(f"{i} {v[0].value} {v[1].value} {v[2].value}\n")
    L.vpx_codec_destroy(ctx)


# Post-processed output (vp8/common/postproc.c behind VPX_CODEC_USE_POSTPROC / VP8_SET_POSTPROC): tag -> the
# vp8_postproc_cfg_t {post_proc_flag, deblocking_level, noise_level} handed to the REFERENCE decoder by oracle/ref_md5 --pp
# (flags: VP8_DEBLOCK 1, VP8_DEMACROBLOCK 2, VP8_ADDNOISE 4, vpx/vp8.h:55-66).  The dither and noise phases come from the C
# library's unseeded rand(), which every run of the reference binary sees in the same state.
# VP8_MFQE (1024; also part of the reference's default configuration, vp8_dx_iface.c:421-431): alone it works on every
# stream; together with VP8_DEBLOCK / VP8_DEMACROBLOCK the reference dies with SIGSEGV on the first shown frame of every stream
# whose width or height is not a multiple of 16 (the intermediate buffer is allocated with the display size, which
# vp8_yv12_alloc_frame_buffer refuses, and then cleared through its null pointer: postproc.c:929-941), `vpxdec --postproc`
# included -- so those combinations are recorded for the 16-aligned streams only (MFQE_FILTER_STREAMS).
PP_CONFIGS = {
    "deblock": (1, 0, 0),
    "demacro4": (2, 4, 0),
    "demacro9": (2, 9, 0),
    "demacro0": (3, 0, 0),
    "noise3": (4, 0, 3),
    "deblock_noise1": (5, 0, 1),
    "demacro6_noise2": (6, 6, 2),
}
PP_STREAMS = ("p_arf_176x144", "p_lowrate_640x360", "kf_odd_67x45", "p_odd_130x98", "p_sharp_320x240", "kf_640x360")
MFQE_CONFIGS = {"mfqe": (1024, 0, 0), "mfqe_noise3": (1028, 0, 3)}
MFQE_STREAMS = PP_STREAMS + ("p_split_352x288", "p_prof1_640x360", "kf_1920x1080")
MFQE_FILTER_CONFIGS = {"default": (1027, 4, 0), "mfqe_deblock": (1025, 0, 0), "mfqe_demacro4": (1026, 4, 0),
                       "mfqe_deblock_noise1": (1029, 0, 1), "mfqe_demacro6_noise2": (1030, 6, 2)}
MFQE_FILTER_STREAMS = ("p_arf_176x144", "p_sharp_320x240", "p_split_352x288")


def postproc_md5(name, configs=PP_CONFIGS):
    for tag, (flags, level, noise) in configs.items():
        out = os.path.join(HERE, f"{name}.pp_{tag}.md5")
        run([REFMD5, "--pp", str(flags), str(level), str(noise), os.path.join(HERE, name + ".ivf"), out])


PP_CLI = (("p_arf_176x144", ["--deblock"]), ("p_arf_176x144", ["--demacroblock-level=6", "--noise-level=2"]),
          ("p_lowrate_640x360", ["--demacroblock-level=3"]), ("p_odd_130x98", ["--noise-level=4", "--deblock"]),
          ("p_arf_176x144", ["--postproc"]), ("p_split_352x288", ["--mfqe", "--deblock"]), ("p_odd_130x98", ["--mfqe"]),
          ("p_lowrate_640x360", ["--mfqe", "--noise-level=2"]))


def webm_fixture():
    """The same encode written twice by the reference's vpxenc, as WebM (its default container, libmkv) and as IVF: the WebM
    reader of the product's tools must hand out the IVF's frames; the digest is the reference vpxdec's over the WebM file."""
    w, h, frames, seed = 176, 144, 12, 91
    args = ["--i420", "-p", "1", "-t", "1", "--kf-max-dist=5", "--good", "--cpu-used=2", "--target-bitrate=300", "--lag-in-frames=0"]
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        yuv = os.path.join(td, "in.yuv")
        with open(yuv, "wb") as f:
            f.write(synth_i420(w, h, frames, seed, 5))
        run([VPXENC, *args, "-w", str(w), "-h", str(h), "-o", os.path.join(HERE, "container_176x144.webm"), yuv])
        run([VPXENC, *args, "--ivf", "-w", str(w), "-h", str(h), "-o", os.path.join(HERE, "container_176x144.ivf_twin"), yuv])
    r = subprocess.run([VPXDEC, "--md5", "--i420", os.path.join(HERE, "container_176x144.webm")], capture_output=True, text=True, check=True)
    with open(os.path.join(HERE, "container_176x144.webm.vpxdec_md5"), "w") as f:
        f.write(r.stdout.split()[0] + "\n")


def main():
    if "--webm" in sys.argv:
        webm_fixture()
        return
    if "--postproc" in sys.argv:
        for name in PP_STREAMS:
            postproc_md5(name)
        for name in MFQE_STREAMS:
            postproc_md5(name, MFQE_CONFIGS)
        for name in MFQE_FILTER_STREAMS:
            postproc_md5(name, MFQE_FILTER_CONFIGS)
        # the reference's vpxdec with its own option names: one digest over all post-processed frames
        with open(os.path.join(HERE, "postproc.pp_vpxdec_md5"), "w") as f:
            for name, args in PP_CLI:
                r = subprocess.run([VPXDEC, *args, "--md5", "--i420", os.path.join(HERE, name + ".ivf")], capture_output=True, text=True, check=True)
                f.write(f"{name} {r.stdout.split()[0]} {' '.join(args)}\n")
        return
    if "--ref-controls" in sys.argv:
        for name in ("p_arf_176x144", "p_split_352x288"):
            ref_controls(name)
        return
    if "--stream-md5-only" in sys.argv:
        for name in FIXTURES:
            stream_md5(name)
        return
    names = sys.argv[1:] or list(FIXTURES)
    for exe in (VPXENC, REFMD5):
        if not os.path.exists(exe):
            raise SystemExit(f"{exe} missing: run `make -C oracle ref` (dev container only)")
    for name in names:
        w, h, frames, seed, noise, args = FIXTURES[name][:6]
        speed = FIXTURES[name][6] if len(FIXTURES[name]) > 6 else (1.375, 0.625)
        ivf = os.path.join(HERE, name + ".ivf")
        md5 = os.path.join(HERE, name + ".md5")
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            yuv = os.path.join(td, "in.yuv")
            with open(yuv, "wb") as f:
                f.write(synth_i420(w, h, frames, seed, noise, speed))
            if args and args[0] == "roi":
                run([ROIENC, str(w), str(h), yuv, ivf, *args[1:]])
            else:
                base = args if "-p" in args else COMMON + args
                run([VPXENC, *base, "-w", str(w), "-h", str(h), "-o", ivf, yuv])
        run([REFMD5, ivf, md5])
        stream_md5(name)
        nshown = sum(1 for _ in open(md5))
        print(f"{name:20s} {os.path.getsize(ivf):8d} B  {nshown:3d} shown frames  "
              f"listing-md5 {hashlib.md5(open(md5,'rb').read()).hexdigest()[:12]}")


if __name__ == "__main__":
    main()
