"""GPU (-m gpu): the C host decoder behind the public vpx codec API, driven by the two command line
tools exactly as a user of the reference would drive them."""
import filecmp
import os
import subprocess

import pytest

from vp8_testlib import FIXTURES, GOLDEN, ROOT, golden_md5, ivf_path

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "libvpx.opencl_amd", "bin")


@pytest.mark.parametrize("name", FIXTURES)
def test_decode_to_md5_listing_is_byte_identical(name, tmp_path):
    out = tmp_path / "out.md5"
    r = subprocess.run([os.path.join(BIN, "decode_to_md5"), ivf_path(name), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(out, os.path.join(GOLDEN, name + ".md5"), shallow=False)


@pytest.mark.parametrize("name", ["kf_640x360", "p_split_352x288", "p_prof3_640x360", "kf_1920x1080"])
def test_vpxdec_md5(name):
    r = subprocess.run([os.path.join(BIN, "vpxdec"), "--md5", "--i420", "--summary", ivf_path(name)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split()[0] == open(os.path.join(GOLDEN, name + ".vpxdec_md5")).read().strip()
    assert "decoded frames" in r.stderr


def test_vpxdec_writes_i420(tmp_path):
    out = tmp_path / "o.i420"
    r = subprocess.run([os.path.join(BIN, "vpxdec"), "--i420", "-o", str(out), "--limit=2", ivf_path("kf_odd_67x45")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(out) == 2 * (67 * 45 + 2 * 34 * 23)


def test_api_rejects_bad_streams():
    r = subprocess.run([os.path.join(BIN, "decode_to_md5"), os.path.join(GOLDEN, "kf_640x360.md5"), "/dev/null"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "not an IVF" in r.stderr


@pytest.mark.parametrize("name,args", [("kf_640x360", ["--threads", "4", "--batch", "3"]), ("kf_odd_67x45", ["--batch", "64"]),
                                       ("kf_640x360", ["--host-md5", "--batch", "7"]),
                                       ("kf_q0_176x144", ["--threads", "2", "--batch", "4", "--loop", "5"]),
                                       ("kf_1920x1080", []),
                                       # the entropy decoder on the device (vp8hip_entropy_decode): batches that end unevenly, a stream
                                       # shorter than the batch, eight token partitions, the host hashing, frames left on the device
                                       ("kf_640x360", ["--device-entropy", "--batch", "7", "--loop", "3"]),
                                       ("kf_odd_67x45", ["--device-entropy", "--batch", "64", "--loop", "30"]),
                                       ("kf_8part_1920x1080", ["--device-entropy", "--batch", "4", "--loop", "4", "--no-download"]),
                                       ("kf_640x360", ["--device-entropy", "--host-md5", "--batch", "16", "--loop", "4"]),
                                       ("kf_1920x1080", ["--device-entropy", "--batch", "32", "--loop", "7"]),
                                       # ... in launches of more frames than the pixel path takes at a time (the IR in its sparse form in between)
                                       ("kf_640x360", ["--device-entropy", "--batch", "4", "--entropy-batch", "12", "--loop", "5"]),
                                       ("kf_8part_1920x1080", ["--device-entropy", "--batch", "3", "--entropy-batch", "6", "--loop", "5", "--no-download"]),
                                       ("kf_1920x1080", ["--device-entropy", "--batch", "16", "--entropy-batch", "64", "--loop", "13"]),
                                       # ... with IR slots for the whole entropy launch (no sparse form in between)
                                       ("kf_640x360", ["--device-entropy", "--batch", "4", "--entropy-batch", "12", "--entropy-dense", "--loop", "5"]),
                                       ("kf_1920x1080", ["--device-entropy", "--batch", "16", "--entropy-batch", "48", "--entropy-dense", "--loop", "11", "--no-download"])])
def test_batch_md5_listing_equals_decode_to_md5(name, args, tmp_path):
    """The threaded feeder + batched launches (batch_md5) write decode_to_md5's listing, line for line; looped, the
    digests repeat with continuing frame numbers."""
    out = tmp_path / "out.md5"
    r = subprocess.run([os.path.join(BIN, "batch_md5")] + args + [ivf_path(name), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "frames/s" in r.stderr
    # frames are hashed on the device, a frame per lane (vp8_md5.hip), whatever their size (kf_odd_67x45: blocks that straddle rows)
    assert ("MD5 on the device" in r.stderr) == ("--host-md5" not in args), r.stderr
    assert ("entropy decode on the device" in r.stderr) == ("--device-entropy" in args)
    loop = int(args[args.index("--loop") + 1]) if "--loop" in args else 1
    gold = open(os.path.join(GOLDEN, name + ".md5")).read().splitlines()
    got = open(out).read().splitlines()
    assert len(got) == loop * len(gold)
    if loop == 1:
        assert got == gold
    for i, line in enumerate(got):
        digest, label = line.split()
        assert digest == gold[i % len(gold)].split()[0]
        assert label.endswith("-%04d.i420" % (i + 1))


@pytest.mark.parametrize("inputs,streams", [(["p_lowrate_640x360"], 3), (["p_prof1_640x360", "p_lowrate_640x360", "kf_640x360"], 7),
                                            (["p_roi_640x360"], 5), (["p_1920x1080"], 2)])
def test_batch_md5_streams_side_by_side(inputs, streams, tmp_path):
    """--streams S: S streams of any frame types decoded side by side -- position t of all of them in one launch, only the frame
    headers read on the host, the macroblocks' modes, vectors and tokens on the device (inter frames too), a stream's frames through
    one IR slot (p_roi_640x360: inter frames with segmentation on that keep their segment map, which therefore lives on the
    device), every shown frame hashed on the device: stream s's listing is the reference decoder's for input s mod inputs."""
    out = tmp_path / "out.md5"
    r = subprocess.run([os.path.join(BIN, "batch_md5"), "--streams", str(streams)] + [ivf_path(n) for n in inputs] + [str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "streams of" in r.stderr and "0 corrupt" in r.stderr
    golds = [[ln.split()[0] for ln in open(os.path.join(GOLDEN, n + ".md5")).read().splitlines()] for n in inputs]
    T = min(len(g) for g in golds)                # (every frame of these fixtures is shown)
    got = open(out).read().splitlines()
    assert len(got) == streams * T
    for s in range(streams):
        for k in range(T):
            digest, label = got[s * T + k].split()
            assert digest == golds[s % len(inputs)][k], (s, k)
            assert label.startswith("stream%d/" % s) and label.endswith("-%04d.i420" % (k + 1))


@pytest.mark.parametrize("args,name", [(["--loop", "13"], "kf_640x360"),
                                       (["--device-entropy", "--batch", "16", "--loop", "9"], "kf_1920x1080")])
def test_batch_md5_workers_per_device(args, name, tmp_path, monkeypatch):
    """--gpus G: G worker processes over contiguous shares of the looped stream, one device each (here: both on the box's one
    GPU, VP8BATCH_SINGLE_DEVICE=1), listings merged in frame order: byte for byte the single-process listing."""
    monkeypatch.setenv("VP8BATCH_SINGLE_DEVICE", "1")
    one, two = tmp_path / "one.md5", tmp_path / "two.md5"
    for out, extra in ((one, []), (two, ["--gpus", "2"])):
        r = subprocess.run([os.path.join(BIN, "batch_md5")] + args + extra + [ivf_path(name), str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    assert "on 2 GPUs" in r.stderr
    assert open(one).read() == open(two).read()
    gold = open(os.path.join(GOLDEN, name + ".md5")).read().splitlines()
    got = open(two).read().splitlines()
    assert [g.split()[0] for g in got] == [gold[i % len(gold)].split()[0] for i in range(len(got))]


def test_batch_md5_streams_over_two_workers(tmp_path, monkeypatch):
    monkeypatch.setenv("VP8BATCH_SINGLE_DEVICE", "1")
    out = tmp_path / "out.md5"
    r = subprocess.run([os.path.join(BIN, "batch_md5"), "--streams", "5", "--gpus", "2", ivf_path("p_lowrate_640x360"), str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    gold = [ln.split()[0] for ln in open(os.path.join(GOLDEN, "p_lowrate_640x360.md5")).read().splitlines()]
    got = open(out).read().splitlines()
    assert [g.split()[0] for g in got] == gold * 5
    assert [g.split()[1].split("/")[0] for g in got] == ["stream%d" % (i // len(gold)) for i in range(len(got))]


def test_batch_md5_refuses_inter_frames(tmp_path):
    r = subprocess.run([os.path.join(BIN, "batch_md5"), ivf_path("p_lowrate_640x360"), str(tmp_path / "o")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "not a key frame" in r.stderr


REF_VPXDEC_ON_HIP = os.path.join(ROOT, "oracle", "_ref", "vpxdec_ref_on_hip")


@pytest.mark.skipif(not os.path.exists(REF_VPXDEC_ON_HIP), reason="oracle/_ref/vpxdec_ref_on_hip not built (make -C oracle ref)")
@pytest.mark.parametrize("name", FIXTURES)
def test_the_references_own_vpxdec_runs_on_the_product(name):
    """The drop-in claim itself: the REFERENCE's vpxdec.c, compiled unchanged against include/vpx/*.h and linked with
    libvpx_hip.so instead of the reference's libvpx (oracle/Makefile, rule vpxdec_ref_on_hip), prints the reference
    decoder's `--md5 --i420` digest for every fixture (vpxdec.c:322-383)."""
    r = subprocess.run([REF_VPXDEC_ON_HIP, "--md5", "--i420", ivf_path(name)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split()[0] == open(os.path.join(GOLDEN, name + ".vpxdec_md5")).read().strip()


@pytest.mark.parametrize("exe", [os.path.join(BIN, "vpxdec"), REF_VPXDEC_ON_HIP])
def test_webm_input(exe):
    """WebM in, the reference vpxdec's digest out: the product's vpxdec with its own reader (webm.c), and the reference's
    vpxdec.c (nestegg) on the product."""
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not built")
    webm = os.path.join(GOLDEN, "container_176x144.webm")
    r = subprocess.run([exe, "--md5", "--i420", webm], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split()[0] == open(webm + ".vpxdec_md5").read().strip()


def test_vpxdec_threads_option():
    """-t N = vpx_codec_dec_cfg_t::threads: token partitions of a frame on several host threads (vp8_parser_set_threads); the
    8-partition 1080p stream decodes to the reference's digest whatever N."""
    name = "kf_8part_1920x1080"
    for t in ("1", "3", "8"):
        r = subprocess.run([os.path.join(BIN, "vpxdec"), "-t", t, "--md5", "--i420", ivf_path(name)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert r.stdout.split()[0] == open(os.path.join(GOLDEN, name + ".vpxdec_md5")).read().strip(), t


def test_written_inter_streams_through_the_tools(tmp_path):
    """Streams from the suite's own writer (inter frames that code, keep or re-data their segment map; every mode, reference and
    split shape): decode_to_md5 -- the host decoder behind the vpx codec API -- and batch_md5 --streams -- modes and tokens read
    on the device, the kept map in the stream's IR slot -- both list the MD5s feeder + oracle give (which tests/test_writer_cpu.py
    pins to the reference decoder where /root/reference is).  Hidden frames are in there too (show_frame = 0)."""
    from test_gpu_entropy import oracle_listing
    from test_writer_cpu import INTER_CASES, inter_sequence
    from vp8_testlib import load_package
    from vp8_writer import write_ivf
    P = load_package()
    for n, (w, h, seed, plan, lp, big) in enumerate(INTER_CASES):
        frames, _ = inter_sequence(w, h, seed, plan, lp, big=big)
        gold = oracle_listing(P, w, h, frames)
        ivf, out = tmp_path / ("w%d.ivf" % n), tmp_path / ("w%d.md5" % n)
        write_ivf(ivf, w, h, frames)
        r = subprocess.run([os.path.join(BIN, "decode_to_md5"), str(ivf), str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert [ln.split()[0] for ln in open(out)] == gold, (n, "decode_to_md5")
        r = subprocess.run([os.path.join(BIN, "batch_md5"), "--streams", "3", str(ivf), str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert [ln.split()[0] for ln in open(out)] == gold * 3, (n, "batch_md5 --streams")


def test_batch_md5_block_pool_too_small(tmp_path):
    """--entropy-batch: the blocks of a launch come out of one pool (vp8hip_configure_pooled); a pool that cannot hold them is reported
    by the frames that found it empty, nothing decoded from half an IR is hashed, and the work is run again with twice the pool --
    three times at most: a pool of 1 MB is still too small at 8 MB, and the tool gives up with an error."""
    r = subprocess.run([os.path.join(BIN, "batch_md5"), "--device-entropy", "--batch", "4", "--entropy-batch", "12", "--pool-mb", "1", "--loop", "3",
                        ivf_path("kf_1920x1080"), str(tmp_path / "o.md5")], capture_output=True, text=True)
    assert r.returncode != 0 and "found the block pool (1 MB) empty" in r.stderr, r.stderr
    assert "found the block pool (8 MB) empty" in r.stderr and "still too small at 8 times" in r.stderr, r.stderr
    assert not os.path.exists(tmp_path / "o.md5")


def test_batch_md5_block_pool_doubled_until_it_holds(tmp_path):
    """... and a pool that is too small by less than that: the first attempt's launch finds it empty, a later attempt has room, and
    the listing is the reference decoder's (twelve 1080p key frames of the fixture need about 30 MB of blocks)."""
    out = tmp_path / "o.md5"
    r = subprocess.run([os.path.join(BIN, "batch_md5"), "--device-entropy", "--batch", "4", "--entropy-batch", "12", "--pool-mb", "12", "--loop", "3",
                        ivf_path("kf_1920x1080"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "found the block pool (12 MB) empty" in r.stderr and "decoding again with a pool" in r.stderr, r.stderr
    gold = golden_md5("kf_1920x1080")
    assert [ln.split()[0] for ln in open(out)] == gold * 3
