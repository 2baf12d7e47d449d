"""Damaged-stream cases for error concealment (VPX_CODEC_USE_ERROR_CONCEALMENT): name -> (fixture, frames that never arrive,
frames cut to their first N bytes -- the cuts leave the first partition whole, see vp8_parser.h).  The expected listings,
tests/golden/ec_<name>.md5, were printed by the reference decoder configured --enable-error-concealment
(oracle/_ref/ref_md5_ec --damage --ec ...; tests/golden/gen_ec_listings.py)."""
CASES = {
    "lost_1080p": ("p_1920x1080", (4,), ()),
    "lost_run_1080p": ("p_1920x1080", (3, 4, 7), ()),
    "cut_1080p": ("p_1920x1080", (), ((4, 1200), (9, 8000))),
    "cut_dense_1080p": ("p_dense_1920x1080", (), ((2, 20000), (3, 25000), (4, 30000))),
    "lost_cut_dense_1080p": ("p_dense_1920x1080", (3,), ((4, 100000),)),
    "split": ("p_split_352x288", (3, 5), ((8, 2000), (11, 900))),
    "bilinear": ("p_prof1_640x360", (4,), ((7, 1500),)),
    "bilinear_normal_lf": ("p_prof2_640x360", (6,), ((8, 2400),)),
    "fullpixel": ("p_prof3_640x360", (5,), ((7, 1500),)),
    "odd_size": ("p_odd_130x98", (3,), ((6, 300),)),
    "sharpness": ("p_sharp_320x240", (4,), ((7, 500),)),
    "lowrate": ("p_lowrate_640x360", (3,), ((5, 300),)),
    "golden_altref": ("p_arf_176x144", (5, 9, 20), ((6, 260), (12, 400), (30, 300))),
    "first_inter_frame_lost": ("p_1920x1080", (2,), ()),        # concealment is not at work yet: nothing shown for it
    # damaged KEY frames with concealment at work (a key frame after inter frames): their intra macroblocks behind the damage become
    # inter macroblocks with interpolated vectors, the loop filter keeps a key frame's thresholds (vp8ir_frame_hdr::lf_key_frame)
    "key_frame": ("p_prof3_640x360", (3,), ((5, 5000), (8, 2000))),
    "key_frame_late_cut": ("p_prof3_640x360", (), ((5, 13000),)),
    "key_frame_sixtap": ("2x:p_sharp_320x240", (4,), ((9, 9000), (13, 600))),       # the fixture twice: frame 9 is a key frame
    "key_frame_split": ("2x:p_split_352x288", (), ((13, 12000), (14, 1000))),
}


def materialize(fixture, golden_dir, tmp_dir):
    """path of the stream a case names: a committed fixture, or ("2x:name") that fixture's frames twice in one IVF, written to tmp_dir"""
    import os
    import struct
    if not fixture.startswith("2x:"):
        return os.path.join(golden_dir, fixture + ".ivf")
    name = fixture[3:]
    data = open(os.path.join(golden_dir, name + ".ivf"), "rb").read()
    n, = struct.unpack_from("<I", data, 24)
    out = bytearray(data[:32])
    struct.pack_into("<I", out, 24, 2 * n)
    body = data[32:]
    frames, off = [], 0
    while off + 12 <= len(body):
        sz, = struct.unpack_from("<I", body, off)
        frames.append(body[off + 12: off + 12 + sz])
        off += 12 + sz
    for k, f in enumerate(frames + frames):
        out += struct.pack("<IQ", len(f), k) + f
    path = os.path.join(str(tmp_dir), f"twice_{name}.ivf")
    open(path, "wb").write(out)
    return path


def tool_args(lose, cut):
    a = ["--ec"]
    if lose:
        a += ["--lose", ",".join(str(x) for x in lose)]
    if cut:
        a += ["--cut", ",".join(f"{f}:{n}" for f, n in cut)]
    return a
