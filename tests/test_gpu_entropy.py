"""GPU: entropy decoding of key frames on the device (vp8hip_entropy_decode, csrc/hip/vp8_entropy.hip) against the host feeder:
 * the IR the kernel leaves in a frame's slot -- the device form of include/vp8_ir.h, read back and expanded to descriptors and
   dense coefficients (vp8hip_ir_fetch) -- says what vp8_parser_decode_mbs writes (itself pinned to the reference decoder through
   the oracle and the MD5 listings; zeros where a block has no coefficients), on the
   key-frame fixtures (odd sizes, q = 0, eight token partitions, 4K) and on streams from the test suite's own writer that turn on
   what the fixtures lack (segment map, no skip flag, skipped macroblocks among coded ones);
 * the frames decoded from that IR: the reference decoder's MD5s;
 * damaged input: a frame cut short reports what the host feeder reports, and nothing is read outside the data."""
import ctypes

import numpy as np
import pytest

from vp8_testlib import golden_md5, ivf_path, load_package

pytestmark = pytest.mark.gpu

KEY_STREAMS = ["kf_odd_67x45", "kf_q0_176x144", "kf_640x360", "kf_1920x1080", "kf_8part_1920x1080", "kf_3840x2160"]


def _host_ir(P, frames):
    parser = P.Parser()
    out = []
    for data in frames:
        hdr, _, mbs, coef, _ = P.parse_to_numpy(parser, data)
        parser.swap(hdr)
        out.append((hdr, mbs, coef))
    parser.close()
    return out


def _export(P, frames):
    parser = P.Parser()
    out = []
    for data in frames:
        hdr, _ = parser.begin(data)
        ef = parser.export_entropy()
        assert ef is not None
        parser.swap(hdr)
        out.append(ef)
    parser.close()
    return out


def _configure(ctx, w, h, nfb, nslots, pooled):
    """pooled: slots without block streams of their own + a pool that holds every slot's worst case (vp8hip_configure_pooled)"""
    if pooled:
        cols, nmb = (w + 15) // 16, ((w + 15) // 16) * ((h + 15) // 16)
        ctx.configure_pooled(w, h, nfb, nslots, nslots * nmb * 24 * 32 + (nslots + 3) * 4 * cols * 24 * 32)
    else:
        ctx.configure(w, h, nfb, nslots)


def _compare(ctx, slot, mbs, coef, what):
    dm, dc = ctx.ir_fetch(slot)
    bad = np.nonzero((dm != mbs).any(axis=1))[0]
    assert bad.size == 0, (what, "descriptor", int(bad[0]), dm[bad[0]].tolist(), mbs[bad[0]].tolist())
    coded = (mbs[:, 3] & 1) == 0                        # a skipped macroblock's coefficients are nobody's business
    badc = np.nonzero((dc[coded] != coef[coded]).any(axis=1))[0]
    assert badc.size == 0, (what, "coefficients", int(np.nonzero(coded)[0][badc[0]]))


@pytest.mark.parametrize("name", KEY_STREAMS)
def test_device_ir_is_the_host_feeders_and_decodes_to_the_references_md5(name):
    P = load_package()
    w, h, frames = P.read_ivf(ivf_path(name))
    host = _host_ir(P, frames)
    efs = _export(P, frames)
    n = len(frames)
    ctx = P.Vp8Hip()
    ctx.configure(w, h, n, n)
    st = ctx.entropy_decode(0, efs, frames)
    assert not st.any()
    for i, (hdr, mbs, coef) in enumerate(host):
        _compare(ctx, i, mbs, coef, (name, i))
    ctx.decode([(i, i, (-1, -1, -1)) for i in range(n)], P.STAGE_ALL)
    got = [P.planes_md5(*ctx.download_planes(i)) for i in range(n)]
    assert got == golden_md5(name)
    ctx.close()


@pytest.mark.parametrize("parts", ["0", "1"])
def test_a_partition_per_lane_or_a_frame_per_lane(parts, monkeypatch):
    """Frames coded with several token partitions, all with the same number, are decoded a PARTITION per lane, rows a macroblock
    behind each other (vp8_entropy_parts_kernel); VP8HIP_ENTROPY_PARTS=0 keeps the frame-per-lane kernel.  Same IR either way;
    a launch that mixes partition counts takes the frame-per-lane kernel by itself."""
    monkeypatch.setenv("VP8HIP_ENTROPY_PARTS", parts)
    P = load_package()
    w, h, frames = P.read_ivf(ivf_path("kf_8part_1920x1080"))
    frames = (frames * 4)[:11]                              # (11 frames of 8 partitions: a wave and a half)
    host = _host_ir(P, frames)
    efs = _export(P, frames)
    assert all(e.num_tok == 8 for e in efs)
    ctx = P.Vp8Hip()
    ctx.configure(w, h, 1, len(frames))
    assert not ctx.entropy_decode(0, efs, frames).any()
    for i, (hdr, mbs, coef) in enumerate(host):
        _compare(ctx, i, mbs, coef, (parts, i))
    ctx.close()


@pytest.mark.parametrize("lanes", [1, 5, 64])
def test_lanes_per_wave(lanes, monkeypatch):
    """The launch shape is a tuning knob (VP8HIP_ENTROPY_LANES): any number of frames per wave gives the same IR."""
    monkeypatch.setenv("VP8HIP_ENTROPY_LANES", str(lanes))
    P = load_package()
    w, h, frames = P.read_ivf(ivf_path("kf_640x360"))
    frames = (frames * 8)[:70]
    host = _host_ir(P, frames)
    efs = _export(P, frames)
    ctx = P.Vp8Hip()
    ctx.configure(w, h, 1, len(frames))
    assert not ctx.entropy_decode(0, efs, frames).any()
    for i, (hdr, mbs, coef) in enumerate(host):
        _compare(ctx, i, mbs, coef, (lanes, i))
    ctx.close()


def test_written_streams():
    """Frames from the suite's own writer (tests/vp8_writer.py) turn on what the fixtures lack: a segment map, skipped macroblocks
    between coded ones at several skip probabilities, 2 / 4 / 8 partitions at small sizes, coefficients up to DCT_VAL_CATEGORY6."""
    from test_gpu_writer import CASES
    from vp8_testlib import synth_ir
    from vp8_writer import write_key_frame
    P = load_package()
    for case in CASES:
        w, h, seed, lp, ftype, dense, big, seg = case
        hdr, mbs, coef, mvs = synth_ir(w, h, seed, inter=False, filter_type=ftype, dense=dense, big=big, segmented=seg)
        hdr.num_token_partitions = 1 << lp
        frames = [write_key_frame(hdr, mbs, coef, log2_parts=lp, prob_skip_false=p) for p in (200, 1, 255, 128)]
        host = _host_ir(P, frames)
        efs = _export(P, frames)
        ctx = P.Vp8Hip()
        ctx.configure(w, h, 1, len(frames))
        assert not ctx.entropy_decode(0, efs, frames).any()
        for i, (_, hm, hc) in enumerate(host):
            _compare(ctx, i, hm, hc, (case, i))
        ctx.close()


@pytest.mark.parametrize("name,pooled", [("kf_640x360", False), ("kf_8part_1920x1080", False), ("kf_640x360", True), ("kf_8part_1920x1080", True)])
def test_frames_cut_short(name, pooled):
    """A frame that ends early -- in the last token partition, in an earlier one, in the first partition -- gives the IR and the
    corrupt flag the host feeder gives (zeros are read past the end, as the reference does: dboolhuff.c:44-60; no tokens once a
    partition has run out: decodframe.c:119-130), next to whole frames in the same launch; a frame whose header does not fit is
    refused by the host before anything is launched."""
    P = load_package()
    w, h, frames = P.read_ivf(ivf_path(name))
    data = frames[1]
    n = len(data)
    cuts = [n - 1, n - 7, n - n // 50, n - n // 3, n // 2, n // 4, n // 10]
    variants = [data] + [data[:c] for c in cuts] + [data]
    host, efs, kept = [], [], []
    for v in variants:
        ph, pd = P.Parser(), P.Parser()
        try:
            hdr, _ = pd.begin(v)
            ef = pd.export_entropy()
        except ValueError:
            continue                     # (the partition table itself is cut off: both refuse the frame)
        finally:
            pass
        h0, _, _, _, _ = P.parse_to_numpy(ph, frames[0])       # (a decoder that has seen a complete key frame goes on after a damaged one)
        ph.swap(h0)
        hh, _ = ph.begin(v)
        nmb = hh.mb_cols * hh.mb_rows
        mbs, coef, mvs = np.zeros((nmb, 64), np.uint8), np.zeros((nmb, 400), np.int16), np.zeros((nmb, 16, 2), np.int16)
        corrupt = ph.decode_mbs(mbs.ctypes.data, coef.ctypes.data, mvs.ctypes.data)
        host.append((mbs, coef, corrupt)); efs.append(ef); kept.append(v)
        ph.close(); pd.close()
    assert len(kept) >= 4           # (with several partitions only cuts inside the last one pass the partition table)
    ctx = P.Vp8Hip()
    _configure(ctx, w, h, 1, len(kept), pooled)
    st = ctx.entropy_decode(0, efs, kept)
    assert [int(s) & 1 for s in st] == [int(bool(c)) for _, _, c in host]
    assert st[0] == 0 and st[-1] == 0 and st.any()
    for i, (mbs, coef, _) in enumerate(host):
        _compare(ctx, i, mbs, coef, (name, i))
    ctx.close()


INTER_STREAMS = ["p_seg_176x144", "p_roi_640x360", "p_lowrate_640x360", "p_odd_130x98", "p_sharp_320x240", "p_split_352x288", "p_arf_176x144", "p_prof1_640x360", "p_prof3_640x360",
                 "p_1920x1080", "p_dense_1920x1080"]


@pytest.mark.parametrize("name", INTER_STREAMS)
def test_inter_frames(name):
    """Inter frames: reference frame, near / nearest / new / split vectors with their above, left and above-left candidates
    (decodemv.c:323-569), intra macroblocks among them, golden / alt-ref with sign bias, bilinear and full-pixel versions -- a
    whole stream with only the frame headers read on the host: every frame's IR (descriptors, coefficients, sixteen vectors per
    macroblock) is the host feeder's, and the frames decode to the reference's MD5s.  The stream's frames go through ONE slot, so
    the segment map a frame keeps is the one the frame before left there (vp8_parser_set_device_segmap; p_roi_640x360)."""
    P = load_package()
    w, h, frames = P.read_ivf(ivf_path(name))
    assert _one_slot_stream(P, w, h, frames[:24], golden_md5(name), name) > 0


def _one_slot_stream(P, w, h, frames, gold, name, pooled=False):
    """-> inter frames seen.  pooled: the slot's blocks out of a block pool (vp8hip_configure_pooled), emptied before every frame"""
    ph, pd = P.Parser(), P.Parser()
    pd.set_device_segmap(True)
    ctx = P.Vp8Hip()
    if pooled:
        nmb = ((w + 15) // 16) * ((h + 15) // 16)
        ctx.configure_pooled(w, h, 4, 1, nmb * 24 * 32 + 3 * 4 * ((w + 15) // 16) * 24 * 32)
    else:
        ctx.configure(w, h, 4, 1)
    shown = 0
    n_inter = 0
    for i, data in enumerate(frames):
        hdr, _, mbs, coef, mvs = P.parse_to_numpy(ph, data)
        ph.swap(hdr)
        h2, _ = pd.begin(data)
        ef = pd.export_entropy()
        assert ef is not None
        if pooled:
            ctx.pool_reset()
        assert not ctx.entropy_decode(0, [ef], [data]).any()
        dm, dc = ctx.ir_fetch(0)
        bad = np.nonzero((dm != mbs).any(axis=1))[0]
        assert bad.size == 0, (name, i, "descriptor", int(bad[0]), dm[bad[0]].tolist(), mbs[bad[0]].tolist())
        coded = (mbs[:, 3] & 1) == 0
        assert (dc[coded] == coef[coded]).all(), (name, i, "coefficients")
        r = pd.refs
        if hdr.frame_type:
            n_inter += 1
            dv = ctx.mvs_fetch(0)
            badv = np.nonzero((dv != mvs.reshape(dv.shape)).any(axis=1))[0]
            assert badv.size == 0, (name, i, "vectors", int(badv[0]), dv[badv[0]].tolist(), mvs.reshape(dv.shape)[badv[0]].tolist())
        ctx.decode([(0, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx))], P.STAGE_ALL)
        pd.swap(h2)
        if hdr.show_frame:
            assert P.planes_md5(*ctx.download_planes(pd.refs.show_idx)) == gold[shown], (name, i)
            shown += 1
    ph.close(); pd.close(); ctx.close()
    return n_inter


def oracle_listing(P, w, h, frames):
    """feeder + oracle over a written stream: the MD5 of every shown frame (pinned to the reference decoder by
    tests/test_writer_cpu.py, where /root/reference is)"""
    from vp8_testlib import oracle_decode
    parser = P.Parser()
    g = P.geom(w, h)
    bufs = [np.zeros(g.frame_size, np.uint8) for _ in range(4)]
    out = []
    for data in frames:
        hdr, _, mbs, coef, mvs = P.parse_to_numpy(parser, data)
        r = parser.refs
        oracle_decode(hdr, mbs, coef, mvs, bufs[r.new_idx], (bufs[r.lst_idx], bufs[r.gld_idx], bufs[r.alt_idx]))
        parser.swap(hdr)
        if hdr.show_frame:
            out.append(P.frame_md5(bufs[parser.refs.show_idx], g, w, h))
    parser.close()
    return out


def test_written_inter_streams():
    """Inter frames from the suite's own writer: every mode on every macroblock, three references with either sign bias, split
    vectors of all four shapes, 1..8 partitions -- and what no encoder here produces: inter frames that code a new segment map,
    that keep the map under new segment data, that keep both (the map then lives in the stream's IR slot on the device), frames
    with segmentation off in between."""
    from test_writer_cpu import INTER_CASES, inter_sequence
    P = load_package()
    for w, h, seed, plan, lp, big in INTER_CASES:
        frames, _ = inter_sequence(w, h, seed, plan, lp, big=big)
        gold = oracle_listing(P, w, h, frames)
        assert _one_slot_stream(P, w, h, frames, gold, (w, h, seed)) == len(plan)
        assert _one_slot_stream(P, w, h, frames, gold, (w, h, seed, "pooled"), pooled=True) == len(plan)


@pytest.mark.parametrize("name", ["kf_odd_67x45", "kf_640x360", "kf_1920x1080", "kf_8part_1920x1080", "kf_3840x2160"])
def test_block_pool(name):
    """vp8hip_configure_pooled: the slots' block streams out of one pool, a chunk (four macroblock rows' worst case) at a time.
    Same IR as the host feeder's (vp8hip_ir_fetch follows the records into the pool), same frames (the reference's MD5s); the pool
    holds what the frames need, not their worst case; emptied, it serves the next launch; vp8hip_ir_copy'd slots share their
    source's blocks; a pool too small flags the frames it could not serve (status bit 1) and serves the others; the host-side
    producers are refused."""
    P = load_package()
    w, h, frames = P.read_ivf(ivf_path(name))
    host = _host_ir(P, frames)
    efs = _export(P, frames)
    n = len(frames)
    cols, nmb = (w + 15) // 16, ((w + 15) // 16) * ((h + 15) // 16)
    chunk = 4 * cols * 24 * 32
    worst = n * nmb * 24 * 32
    ctx = P.Vp8Hip()
    ctx.configure_pooled(w, h, n + 1, n + 1, worst + (n + 2) * chunk)
    for rnd in range(2):                                   # (second round: the pool emptied, the frames the other way round)
        order = list(range(n)) if rnd == 0 else list(range(n - 1, -1, -1))
        ctx.pool_reset()
        assert not ctx.entropy_decode(0, [efs[i] for i in order], [frames[i] for i in order]).any()
        used, size = ctx.pool_usage()
        assert 0 < used <= size and used % chunk == 0
        if nmb > 1000:
            assert used < worst                            # (what the frames need, not their worst case)
        for k, i in enumerate(order):
            _compare(ctx, k, host[i][1], host[i][2], (name, rnd, i))
        ctx.ir_copy(n, 0)
        _compare(ctx, n, host[order[0]][1], host[order[0]][2], (name, rnd, "copy"))
        ctx.decode([(k, k, (-1, -1, -1)) for k in range(n + 1)], P.STAGE_ALL)
        got = [P.planes_md5(*ctx.download_planes(k)) for k in range(n + 1)]
        assert got == [golden_md5(name)[i] for i in order + [order[0]]]
    with pytest.raises(RuntimeError, match="block pool"):
        ctx.ir_map_compact(0)
    # a pool of two chunks (+ the one behind them): the frames that find it empty say so, the others are whole
    ctx.configure_pooled(w, h, n + 1, n, 3 * chunk)
    st = ctx.entropy_decode(0, efs, frames)
    if nmb > 1000:
        assert (st & 2).any()
    for i in range(n):
        if not st[i] & 2:
            _compare(ctx, i, host[i][1], host[i][2], (name, "small pool", i))
    used, size = ctx.pool_usage()
    assert size == 2 * chunk and (used > size) == bool((st & 2).any())
    # the pipelines queue vp8hip_decode for a whole launch before its status words are back: decoding the starved frames too is
    # in bounds (include/vp8hip.h: their blocks sit in the spare chunk behind the pool) -- the frames that were served are the
    # reference's, a frame buffer no job names keeps its bytes, and nothing faults
    guard = np.random.default_rng(3).integers(0, 256, size=ctx.g.frame_size).astype(np.uint8)
    ctx.upload_frame(n, guard)
    ctx.decode([(k, k, (-1, -1, -1)) for k in range(n)], P.STAGE_ALL)
    ctx.sync()
    for i in range(n):
        if not st[i] & 2:
            assert P.planes_md5(*ctx.download_planes(i)) == golden_md5(name)[i], (name, "small pool", i)
    assert np.array_equal(ctx.download_full(n), guard)
    ctx.close()


@pytest.mark.parametrize("name,mode", [("p_lowrate_640x360", "device"), ("p_arf_176x144", "device"), ("p_lowrate_640x360", "host")])
def test_streams_side_by_side(name, mode):
    """tools/streams_probe.py: several copies of an inter-frame stream, a launch per position (golden / alt-ref bookkeeping shared,
    four frame buffers per stream), every shown frame of every stream against the reference's listing."""
    import os, subprocess, sys
    from vp8_testlib import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "streams_probe.py"), "5", name, mode], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "digests differing from the reference's: 0" in r.stdout, r.stdout


@pytest.mark.parametrize("name,index,pooled", [("kf_640x360", 1, False), ("p_lowrate_640x360", 3, False), ("p_split_352x288", 5, False),
                                               ("kf_640x360", 1, True), ("p_lowrate_640x360", 3, True)])
def test_damaged_payloads(name, index, pooled):
    """Random damage behind the frame header (bytes overwritten, bytes flipped, the tail cut): whatever the bits then say -- modes,
    vectors, runs of large coefficients, partitions that end early -- the device reads what the host feeder reads: same IR, same
    vectors, same corrupt flag, all frames of the launch side by side."""
    P = load_package()
    w, h, frames = P.read_ivf(ivf_path(name))
    rng = np.random.default_rng(index * 17 + len(name))
    good = frames[index]
    variants = []
    for v in range(24):
        d = bytearray(good)
        lo = 64 + int(rng.integers(0, 64))                      # (past the frame header and the start of the first partition)
        kind = v % 3
        if kind == 0:
            for _ in range(1 + v // 3):
                at = int(rng.integers(lo, len(d)))
                d[at] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            at = int(rng.integers(lo, len(d) - 16))
            d[at:at + 16] = bytes(rng.integers(0, 256, size=16).astype(np.uint8))
        else:
            d = d[:int(rng.integers(len(d) // 2, len(d)))]
        variants.append(bytes(d))
    host, efs, kept = [], [], []
    for v in variants:
        ph, pd = P.Parser(), P.Parser()
        try:
            for data in frames[:index]:                         # the frames before, as they are (state of the header parse)
                hh, _, _, _, _ = P.parse_to_numpy(ph, data); ph.swap(hh)
                hd, _ = pd.begin(data); assert pd.export_entropy() is not None; pd.swap(hd)
            try:
                hd, _ = pd.begin(v)
                ef = pd.export_entropy()
            except ValueError:
                continue                                        # the header itself refuses the frame: nothing to compare
            if ef is None:
                continue
            hh, _ = ph.begin(v)
            n = hh.mb_cols * hh.mb_rows
            mbs, coef, mvs = np.zeros((n, 64), np.uint8), np.zeros((n, 400), np.int16), np.zeros((n, 16, 2), np.int16)
            try:
                corrupt = ph.decode_mbs(mbs.ctypes.data, coef.ctypes.data, mvs.ctypes.data)
            except ValueError:
                continue
            host.append((hh, mbs, coef, mvs, corrupt)); efs.append(ef); kept.append(v)
        finally:
            ph.close(); pd.close()
    assert len(kept) >= 12
    ctx = P.Vp8Hip()
    _configure(ctx, w, h, 1, len(kept), pooled)
    st = ctx.entropy_decode(0, efs, kept)
    assert [int(s) & 1 for s in st] == [int(bool(c)) for *_, c in host]
    for i, (hh, mbs, coef, mvs, _) in enumerate(host):
        _compare(ctx, i, mbs, coef, (name, i))
        if hh.frame_type:
            assert (ctx.mvs_fetch(i) == mvs.reshape(-1, 2)).all(), (name, i)
    ctx.close()


def test_input_staged_ahead_of_the_launch():
    """vp8hip_entropy_stage + vp8hip_entropy_decode(..., NULL, NULL, 0): the launch in two steps gives the IR of the launch in one;
    a second stage before the launch, a launch of another count and a one-step launch over staged input are refused."""
    import ctypes
    P = load_package()
    w, h, frames = P.read_ivf(ivf_path("kf_640x360"))
    host = _host_ir(P, frames)
    efs = _export(P, frames)
    n = len(frames)
    arr = (P.EntropyFrame * n)()
    off = 0
    for i, (f, d) in enumerate(zip(efs, frames)):
        ctypes.memmove(ctypes.byref(arr[i]), ctypes.byref(f), ctypes.sizeof(P.EntropyFrame))
        arr[i].data_off = off
        off += len(d)
    blob = b"".join(frames)
    ctx = P.Vp8Hip()
    ctx.configure(w, h, 1, n)
    L = ctx.L
    L.vp8hip_entropy_stage.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    ctx._chk(L.vp8hip_entropy_stage(ctx.h, n, ctypes.byref(arr), blob, len(blob)), "stage")
    assert L.vp8hip_entropy_stage(ctx.h, n, ctypes.byref(arr), blob, len(blob)) != 0            # one staged input at a time
    assert L.vp8hip_entropy_decode(ctx.h, 0, n - 1, None, None, 0) != 0                         # ... of n frames
    assert L.vp8hip_entropy_decode(ctx.h, 0, n, ctypes.byref(arr), blob, len(blob)) != 0        # ... which is to be launched first
    ctx._chk(L.vp8hip_entropy_decode(ctx.h, 0, n, None, None, 0), "launch")
    st = np.zeros(n, np.uint32)
    ctx._chk(L.vp8hip_entropy_status(ctx.h, n, st.ctypes.data), "status")
    assert not st.any()
    for i, (_, mbs, coef) in enumerate(host):
        _compare(ctx, i, mbs, coef, ("staged", i))
    assert L.vp8hip_entropy_decode(ctx.h, 0, n, None, None, 0) != 0                             # nothing staged any more
    ctx.close()


def test_fuzzed_inter_streams():
    """Twenty-four seeded inter sequences of odd small sizes (tests/test_writer_cpu.py: fuzz_case -- random plans of coded / kept /
    re-dataed segment maps and segmentation switched off, 1..8 partitions, coefficients up to +-2047), each through ONE slot with the
    device's entropy decoder: the IR the host feeder reads, the frames feeder + oracle decode (which the CPU suite pins to the
    reference decoder on the same seeds)."""
    from test_writer_cpu import FUZZ_SEEDS, fuzz_case, inter_sequence
    P = load_package()
    for seed in FUZZ_SEEDS:
        w, h, _, plan, lp, big = fuzz_case(seed)
        frames, _ = inter_sequence(w, h, seed, plan, lp, big=big)
        gold = oracle_listing(P, w, h, frames)
        assert _one_slot_stream(P, w, h, frames, gold, ("fuzz", seed), pooled=bool(seed & 1)) == len(plan)
