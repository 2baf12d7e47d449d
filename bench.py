#!/usr/bin/env python3
"""bench.py -- VP8 decode pixel path on MI355X: Mpix/s on a 1080p all-key-frame stream.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--workload 1080p|4k]

One "step" = one pass of the whole pixel path (dequant + IDCT / WHT, intra reconstruction, in-loop deblocking filter: the
kernels behind include/vp8hip.h) over a batch of F independent key frames.  What is resident in HBM when the timed region starts
is the frames' IR AS A PRODUCER LEFT IT: every slot holds the device form of include/vp8_ir.h -- the form the host feeder
(vp8_parser_decode_mbs_compact) writes into a slot's pinned staging and one copy uploads, and the form the device's entropy
decoder writes; no kernel conditions it before the pixel kernels read it (rounds 1-3 ran a packing pass over fresh slots
outside the timed region).  The batch is the committed fixture tests/golden/kf_1920x1080.ivf (10 key frames, entropy-decoded
once on the host, outside the timed region) looped F/10 times -- legal because every key frame is independently decodable
(reference: vp8/decoder/decodframe.c:610-639).  What a step leaves: the decoded frames in the TILED form of a large launch
(macroblock-window tiles, include/vp8hip.h "two forms"), which the consumers of such a pipeline read without a further pass --
the device's MD5 kernel (vpxdec --md5 / decode_to_md5), a batch download into page-locked memory (the tiled -> raster pass IS
the download), the border-extended raster form for inter prediction on demand.  `config.consumers` times the step with each of
them behind it; `config.with_raster_form` is round 3's step (decode, then the raster form of every frame in HBM).  Before
timing, decoded frames are checked bit-exactly against the reference decoder's per-frame MD5s (tests/golden/*.md5).

N > 1: one rank per GPU -- under torch.distributed.run, or spawned by bench.py itself when no launcher set
WORLD_SIZE -- decoding ONE looped stream of N * F frames in contiguous blocks of F (rank r: frames [r*F, (r+1)*F),
libvpx.opencl_amd/sharding.py; no pixel exchange: "scaling": "weak").  RCCL (backend "nccl") carries the start/stop
barriers, the per-rank times, the verification flag and the gathered MD5 listing of a sharded prefix stream,
which must equal the 1-GPU listing.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field definitions):
  value      whole-job Mpix/s (display pixels) = N * F * K * w * h / seconds
  roofline   dominant kernel: algorithmic bytes per launch (SURVEY.md 8d byte model) / mean launch
             time measured with HIP events on the launch stream, vs 8 TB/s HBM peak; `traffic` from the counter passes
             recorded in profiles/traffic_per_mb.json (tools/profile_round.sh writes it)
  cpu_baseline  the REAL reference decoder (oracle/_ref, generic-C build of /root/reference) timed on
             this host on the same stream, 1 core; falls back to the repo's C restatement ("port")
"""
import argparse
import hashlib
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

# SURVEY.md 8(d) algorithmic bytes per macroblock, dense-coefficient model
B_RECON = 833 + 384      # coefficients+eobs+params read, pixels written (key frames)
B_LF = 770               # pixels read + written, params
B_EXTEND = 36
B_DETILE = 384 + 384 + 36  # the tiled -> raster pass: tiles read, raster frame + borders written
B_INTER_FULL = 833 + 768 + 770 + 36   # full path on P frames
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def load_traffic():
    """HBM traffic per macroblock of the kernels of a step, from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (separate
    passes; FETCH_SIZE doubled: the gfx950 correction for 16-byte-per-lane loads): profiles/traffic_per_mb.json, written by
    tools/profile_round.sh from the passes whose CSVs lie beside it.  The counter passes do not survive under torch, so bench.py
    scales these per-macroblock figures by the macroblocks of a launch instead of counting live; None if the file is missing."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic_per_mb.json")))
    except Exception:  # noqa: BLE001
        return None


WORKLOADS = {
    "1080p": ("kf_1920x1080", 1920, 1080),
    "4k": ("kf_3840x2160", 3840, 2160),
}


def usable_cpus():
    """The CPUs this process may really use: the affinity mask, cut down to the cgroup's CPU quota where there is one (a box of the
    pool shows 256 logical CPUs and grants 16: cpu.max "1600000 100000").  Returns (count, how it was found)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    how = "sched_getaffinity"
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], int(txt[1])
            else:
                quota, period = txt[0], int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                q = max(1, int(int(quota) / period + 0.999))
                if q < n:
                    n, how = q, f"cgroup quota ({path}: {' '.join(txt)})"
            break
        except Exception:  # noqa: BLE001
            continue
    return max(1, n), how


def cpu_baseline(fixture, budget_s=12.0):
    """Time the reference decoder (or the port) on this host, single thread."""
    ivf = os.path.join(ROOT, "tests", "golden", fixture + ".ivf")
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_md5")
    cores = 1
    if os.path.exists(ref):
        try:
            t0 = time.time()
            out = subprocess.run([ref, "--time", "1", ivf], capture_output=True, text=True, timeout=120)
            one = max(time.time() - t0, 1e-3)
            reps = max(1, min(400, int(budget_s / one)))
            out = subprocess.run([ref, "--time", str(reps), ivf], capture_output=True, text=True, timeout=600)
            frames, pixels, secs = out.stdout.split()
            res = {"value": round(float(pixels) / float(secs) / 1e6, 2), "unit": "Mpix/s", "cores": cores,
                   "kind": "reference",
                   "sample": f"{fixture}.ivf x{reps} passes ({frames} frames) through oracle/_ref (reference "
                             f"generic-C decoder, gcc -O3, 1 thread), time inside vpx_codec_decode only"}
            # the same decoder frame-parallel on every host core (all-key-frame streams shard by frame): one process
            # per core, each decoding the whole sample; aggregate = sum of the per-process rates
            ncpu, how = usable_cpus()
            if ncpu > 1:
                r2 = max(1, reps // 6)
                procs = [subprocess.Popen([ref, "--time", str(r2), ivf], stdout=subprocess.PIPE, text=True)
                         for _ in range(ncpu)]
                agg = 0.0
                for pr in procs:
                    o, _ = pr.communicate(timeout=900)
                    _, px, sc = o.split()
                    agg += float(px) / float(sc) / 1e6
                res["all_cores"] = {"value": round(agg, 1), "unit": "Mpix/s", "cores": ncpu, "cores_from": how,
                                    "logical_cpus_of_the_box": os.cpu_count(),
                                    "sample": f"{ncpu} processes x {r2} passes of the same stream"}
            return res
        except Exception as e:  # noqa: BLE001 - fall through to the port
            sys.stderr.write(f"[bench] reference baseline failed ({e}); using the port\n")
    # port: host feeder + oracle pixel path (checker code, timed here only as a baseline)
    from vp8_testlib import load_package, oracle_decode_ivf
    t0 = time.time()
    n = 0
    P = load_package()
    w, h, frames = P.read_ivf(ivf)
    while time.time() - t0 < budget_s:
        oracle_decode_ivf(fixture)
        n += len(frames)
    secs = time.time() - t0
    return {"value": round(n * w * h / secs / 1e6, 2), "unit": "Mpix/s", "cores": cores, "kind": "port",
            "sample": f"{fixture}.ivf, {n} frames through the host feeder + oracle/ C restatement, 1 thread, "
                      f"whole-loop wall time (includes entropy decode and MD5)"}



def inter_frame_probe(P, device, n=8192, name="p_dense_1920x1080", k=2, reps=3):
    """BASELINE configs[2] beside the headline: six-tap motion compensation + IDCT + loop filter on REAL inter frames, n jobs a launch.
      * `ms_per_launch`: the stream is decoded the normal way up to frame k-1; then n jobs decode frame k, every one from its OWN copy
        of the IR (vp8hip_ir_copy) and its OWN copy of the reference frame (vp8hip_frame_copy: raster form) into its own frame buffer
        -- nothing shared in cache, nothing chained: the launches are repeated over the same references;
      * `chained`: n copies of the stream decoded in LOCK STEP, every stream with its own IR slots and its own four frame buffers (the
        frame lifecycle of vp8dx_receive_compressed_data, onyxd_if.c:318-706, for n decoders side by side), a launch per position up
        to frame k-1; then frames k and k+1 one after the other -- launch k+1 predicts from the frames launch k just wrote, as a
        decoder's launches do, read as the tiles they were left as --, the streams taken back to frame k-1 (untimed) and the pair
        repeated.
    EVERY job's frame is hashed on the device after the timed launches and compared with the reference decoder's listing.  Roofline by
    SURVEY 8(d): 2407 B/MB for the full inter path (833 residual + 768 prediction + 770 loop filter + 36), a model of dense content.
    8192 jobs: a frame per strand of 8 lanes (68 macroblock rows: 8.5 rounds, nine run)."""
    from vp8_testlib import ivf_path, golden_md5
    w, h, frames = P.read_ivf(ivf_path(name))
    gold = golden_md5(name)
    assert k + 1 < len(frames)
    # ---------------- frame k alone, from private raster references
    ctx = P.Vp8Hip(device)
    ctx.configure(w, h, 4 + 2 * n, 2 + n)          # 4 decoder buffers, then n reference copies, then n destinations
    parser = P.Parser()
    shown_k = 0
    for data in frames[:k]:
        hdr = ctx.parse_into_slot(parser, data, 0)
        ctx.upload(0)
        r = parser.refs
        ctx.decode([(0, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx) if hdr.frame_type else None)], P.STAGE_ALL)
        ctx.sync()
        shown_k += 1 if hdr.show_frame else 0
        parser.swap(hdr)
    hdr = ctx.parse_into_slot(parser, frames[k], 1)
    assert hdr.frame_type == 1 and hdr.show_frame
    ctx.upload(1)
    r = parser.refs
    distinct = len({r.lst_idx, r.gld_idx, r.alt_idx})
    jobs = (P.Job * n)()
    for i in range(n):
        ctx.ir_copy(2 + i, 1)
        # (the fixtures' golden and alt-ref are the key frame and not referenced by frame k's macroblocks: one private copy of `last`
        # per job is every byte the job reads; the other two indices still point at valid buffers)
        ctx.L.vp8hip_frame_copy(ctx.h, 4 + i, r.lst_idx)
        jobs[i].ir_slot, jobs[i].dst_fb = 2 + i, 4 + n + i
        jobs[i].ref_fb[1], jobs[i].ref_fb[2], jobs[i].ref_fb[3] = 4 + i, r.gld_idx, r.alt_idx
    for _ in range(2):              # (the first launch allocates the tiled forms)
        ctx.decode_array(jobs, n, P.STAGE_ALL)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.decode_array(jobs, n, P.STAGE_ALL)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 5
    st = ctx.stats()
    unchained_tiles = bool(st.pred_tiles)
    bad = sum(1 for v in ctx.frames_md5(4 + n, n) if v != gold[shown_k])
    nmb = ctx.nmb
    parser.close()
    ctx.close()
    # ---------------- frames k, k+1 chained: n streams in lock step.  Frame buffer b of stream i: b * n + i; slots: set A = 0 .. n-1,
    # set B = n .. 2n-1, then one master per frame 0 .. k+1
    ctx = P.Vp8Hip(device)
    ctx.configure(w, h, 4 * n, 2 * n + k + 2)
    parser = P.Parser()
    M = 2 * n
    info = []                   # per frame: (frame_type, new, last, golden, alt-ref, shown index or -1)
    shown = 0
    for f in range(k + 2):
        hdr = ctx.parse_into_slot(parser, frames[f], M + f)
        ctx.upload(M + f)
        r = parser.refs
        info.append((hdr.frame_type, r.new_idx, r.lst_idx, r.gld_idx, r.alt_idx, shown if hdr.show_frame else -1))
        shown += 1 if hdr.show_frame else 0
        parser.swap(hdr)
    parser.close()
    assert info[0][0] == 0 and info[k][0] == 1 and info[k + 1][0] == 1 and info[k][5] >= 0 and info[k + 1][5] >= 0

    def fill(base, f):          # slot set <- frame f's IR (device-to-device copies of the master)
        for i in range(n):
            ctx.ir_copy(base + i, M + f)

    def jobs_of(base, f):
        ft, new, lst, gld, alt, _ = info[f]
        jj = (P.Job * n)()
        for i in range(n):
            jj[i].ir_slot, jj[i].dst_fb = base + i, new * n + i
            for q, ref in enumerate((lst, gld, alt)):
                jj[i].ref_fb[1 + q] = ref * n + i if ft else -1
        return jj

    def mismatches(f):
        return sum(1 for v in ctx.frames_md5(info[f][1] * n, n) if v != gold[info[f][5]])

    for f in range(k):          # the streams up to frame k-1
        fill(0, f)
        ctx.decode_array(jobs_of(0, f), n, P.STAGE_ALL)
    fill(0, k); fill(n, k + 1)
    jk, jk1 = jobs_of(0, k), jobs_of(n, k + 1)
    # the pair can be repeated when the key frame's buffer outlives it (the streams are then decoded again from frame 1)
    if any(info[f][1] == info[0][1] for f in range(1, k + 2)):
        reps = 1
    t_chain, bad_chain, chained_tiles = 0.0, 0, False
    for rep in range(reps):
        if rep:
            for f in range(1, k):
                fill(0, f)
                ctx.decode_array(jobs_of(0, f), n, P.STAGE_ALL)
            fill(0, k)
        ctx.sync()
        t0 = time.perf_counter()
        ctx.decode_array(jk, n, P.STAGE_ALL)
        ctx.decode_array(jk1, n, P.STAGE_ALL)
        ctx.sync()
        t_chain += time.perf_counter() - t0
        chained_tiles = bool(ctx.stats().pred_tiles)
        bad_chain += mismatches(k) + mismatches(k + 1)
    dt_chained = t_chain / (2 * reps)
    raster_pool = ctx.memory_usage()["raster_pool"]
    ctx.close()
    gbps = B_INTER_FULL * nmb * n / dt / 1e9
    return {"workload": f"{name}.ivf frame {k} (inter: {distinct} distinct references, six-tap, normal loop filter) x {n} jobs per "
                        f"launch, each with its own IR slot, reference buffer and destination",
            "md5_ok": bad == 0, "md5_checked": n, "md5_mismatches": bad,
            "Mpix_s": round(n * w * h / dt / 1e6, 1), "ms_per_launch": round(dt * 1e3, 3),
            "references_read_as": "tiles" if unchained_tiles else "raster (private copies, borders extended)",
            "chained": {"ms_per_launch": round(dt_chained * 1e3, 3), "Mpix_s": round(n * w * h / dt_chained / 1e6, 1),
                        "roofline_frac": round(B_INTER_FULL * nmb * n / dt_chained / 1e9 / HBM_PEAK_GBPS, 5),
                        "launches": f"{n} copies of the stream in lock step (own IR slots, own four frame buffers) up to frame {k - 1}; timed: "
                                    f"frames {k} and {k + 1} of every stream, {reps} time(s) (the streams decoded again from frame 1 in between, untimed)",
                        "md5_checked": 2 * n * reps, "md5_mismatches": bad_chain, "md5_ok": bad_chain == 0,
                        "references_read_as": "tiles (vp8_inter_pred_tiles_kernel: no tiled -> raster pass)" if chained_tiles
                                              else "raster (vp8_detile_kf_kernel + vp8_extend_kernel in front of every launch)",
                        "raster_pool_bytes": raster_pool,
                        "note": "every launch reads the frames the launch before it wrote: n streams in lock step, what a decoder runs"},
            "kernel_ms": {"recon": round(st.recon_ms, 3), "loopfilter": round(st.lf_ms, 3), "extend": round(st.extend_ms, 3)},
            "kernel_family": ("vp8_inter_pred_kernel / vp8_inter_pred_tiles_kernel (every inter macroblock's six-tap prediction, order-free, "
                              "into the macroblock's tile) + vp8_interframe_kernel (residual + loop filter, one macroblock row per lane, luma "
                              "and chroma waves paired on every SIMD: kernel_ms.recon is both); the frames are left as tiles"
                              if st.fused else "one wave per macroblock row"),
            "roofline": {"bound": "hbm", "achieved": round(gbps, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(gbps / HBM_PEAK_GBPS, 5),
                         "note": "SURVEY 8(d) full inter path 2407 B/MB (a model of dense content) x macroblocks per launch / wall time per launch"}}


def batch_md5_probe(fixture, device, loops=205, extra=()):
    """The same end-to-end pipeline with the C host side (bin/batch_md5: feeder threads, batched launches, MD5 on the device where the
    frame size allows): the fixture looped, listing checked line by line against the reference decoder's digests."""
    import tempfile
    tool = os.path.join(ROOT, "libvpx.opencl_amd", "bin", "batch_md5")
    ivf = os.path.join(ROOT, "tests", "golden", fixture + ".ivf")
    gold = [l.split()[0] for l in open(os.path.join(ROOT, "tests", "golden", fixture + ".md5")).read().splitlines()]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "o.md5")
        env = dict(os.environ, VP8HIP_DEVICE=str(device))
        r = subprocess.run([tool, *extra, "--loop", str(loops), ivf, out], capture_output=True, text=True, env=env, timeout=300)
        if r.returncode:
            return {"error": r.stderr[-300:]}
        got = [l.split()[0] for l in open(out).read().splitlines()]
    bad = sum(1 for i, g in enumerate(got) if g != gold[i % len(gold)]) + abs(len(got) - loops * len(gold))
    import re
    m = re.search(r"(\d+) frames in ([0-9.]+) s: ([0-9.]+) frames/s, ([0-9.]+) Mpix/s \((\d+) feeder threads, (\d+) frames per launch, "
                  r"entropy decode on the (\w+), MD5 on the (\w+)(, frames not downloaded)?(?:; (\d+) frames per entropy launch)?(?:; (\d+) corrupt)?"
                  r"(; CORRUPT FRAMES, see above)?\)", r.stderr)
    if not m:
        return {"error": "unparsed: " + r.stderr[-200:]}
    return {"tool": " ".join(["bin/batch_md5", *extra, "--loop", str(loops)]), "frames": int(m.group(1)), "frames_per_s": float(m.group(3)),
            "Mpix_s": float(m.group(4)), "host_threads": int(m.group(5)), "frames_per_launch": int(m.group(6)), "entropy_decode_on": m.group(7),
            "md5_on": m.group(8), "frames_downloaded": m.group(9) is None, "frames_per_entropy_launch": int(m.group(10) or m.group(6)),
            "md5_mismatches": bad, "corrupt_frames": int(m.group(11)) if m.group(11) else (1 if m.group(12) else 0)}


def streams_probe(device, streams=4096, fixture="p_1920x1080"):
    """Inter-frame streams side by side with the entropy decoder on the device (bin/batch_md5 --streams): `streams` copies of the
    fixture (a key frame and nine P frames), a launch per position, only the frame headers read on the host, a stream's frames
    through one IR slot and four frame buffers; every shown frame of every stream hashed on the device and compared with the
    reference decoder's listing."""
    import re
    import tempfile
    tool = os.path.join(ROOT, "libvpx.opencl_amd", "bin", "batch_md5")
    ivf = os.path.join(ROOT, "tests", "golden", fixture + ".ivf")
    gold = [l.split()[0] for l in open(os.path.join(ROOT, "tests", "golden", fixture + ".md5")).read().splitlines()]
    env = dict(os.environ, VP8HIP_DEVICE=str(device))
    pat = (r"(\d+) frames in ([0-9.]+) s: ([0-9.]+) frames/s, ([0-9.]+) Mpix/s \((\d+) streams of (\d+) frames side by side.*; (\d+) corrupt\)")
    first = None
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "o.md5")
        # (the streams are ten frames long: a run is two seconds, of which the allocator takes one to three tenths on a device that
        # has just done the same and up to three seconds on a cold one -- both runs are reported, the second is the figure)
        for attempt in range(2):
            r = subprocess.run([tool, "--streams", str(streams), ivf, out], capture_output=True, text=True, env=env, timeout=300)
            if r.returncode:
                return {"error": r.stderr[-300:]}
            if attempt == 0:
                m0 = re.search(pat, r.stderr)
                first = float(m0.group(3)) if m0 else None
        got = [l.split()[0] for l in open(out).read().splitlines()]
    bad = sum(1 for i, g in enumerate(got) if g != gold[i % len(gold)]) + abs(len(got) - streams * len(gold))
    m = re.search(pat, r.stderr)
    if not m:
        return {"error": "unparsed: " + r.stderr[-200:]}
    return {"tool": "bin/batch_md5 --streams %d %s.ivf" % (streams, fixture), "streams": int(m.group(5)), "frames_per_stream": int(m.group(6)),
            "seconds": float(m.group(2)), "frames_per_s": float(m.group(3)), "Mpix_s": float(m.group(4)), "md5_mismatches": bad,
            "corrupt_frames": int(m.group(7)), "frames_per_s_of_the_run_before_on_a_cold_device": first}


def small_run(P, device, fixture, W, H, n, steps=3):
    """One context of its own with n frames of the looped fixture resident, `steps` timed launches of all n (after one untimed):
    ms per launch, Mpix/s, what the context holds on the device, which kernels ran, and every frame's MD5 (computed on the device
    from whichever form the launch left) against the reference listing."""
    from vp8_testlib import golden_md5
    gold = golden_md5(fixture)
    ctx = P.Vp8Hip(device)
    try:
        ctx.configure(W, H, n, n)
        nsrc, _ = load_stream(P, ctx, fixture, n, 0)
        jobs = (P.Job * n)()
        for i in range(n):
            jobs[i].ir_slot, jobs[i].dst_fb = i, i
            for k in range(4):
                jobs[i].ref_fb[k] = -1
        ctx.decode_array(jobs, n, P.STAGE_ALL); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.decode_array(jobs, n, P.STAGE_ALL)
        ctx.sync()
        dt = (time.perf_counter() - t0) / steps
        st = ctx.stats()
        mem = sum(ctx.memory_usage().values())
        dig = ctx.frames_md5(0, n)
        bad = sum(1 for i in range(n) if dig[i] != gold[i % nsrc])
        return {"frames_per_launch": n, "ms_per_launch": round(dt * 1e3, 3), "Mpix_s": round(n * W * H / dt / 1e6, 1),
                "device_GB": round(mem / 1e9, 2), "kernels": "lane-per-row (tiles)" if st.fused else "wave-per-row (raster)",
                "md5_mismatches": bad}
    finally:
        ctx.close()


def single_inter_latency(P, device, name="p_1920x1080"):
    """One INTER frame per launch -- a single stream through vpx_codec_decode: vp8_inter_mb_kernel (every inter macroblock on its own) +
    the row-ordered kernels for the intra macroblocks and the loop filter --, launch to synchronisation, averaged over the fixture's P
    frames; every shown frame's MD5 checked."""
    from vp8_testlib import ivf_path, golden_md5
    w, h, frames = P.read_ivf(ivf_path(name))
    gold = golden_md5(name)
    ctx = P.Vp8Hip(device)
    try:
        ctx.configure(w, h, 4, 1)
        n, total, ok = 0, 0.0, True
        for it in range(3):
            parser = P.Parser()
            shown = 0
            for data in frames:
                hdr = ctx.parse_into_slot(parser, data, 0)
                ctx.upload(0)
                r = parser.refs
                ctx.sync()
                t = time.perf_counter()
                ctx.decode([(0, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx) if hdr.frame_type else None)], P.STAGE_ALL)
                ctx.sync()
                if hdr.frame_type and it:
                    total += time.perf_counter() - t
                    n += 1
                new = r.new_idx
                parser.swap(hdr)
                if hdr.show_frame:
                    ok = ok and P.planes_md5(*ctx.download_planes(new)) == gold[shown]
                    shown += 1
            parser.close()
        return {"ms": round(total / n * 1e3, 3), "frames": n, "md5_ok": bool(ok), "stream": name}
    finally:
        ctx.close()


def load_stream(P, ctx, fixture, F, lo):
    """Slots 0 .. F-1 of `ctx` <- frames lo .. lo+F-1 of the looped stream (frame i of the stream is source frame i mod nsrc;
    every key frame is independently decodable).  Host feeder once per source frame, device-to-device copies for the rest."""
    from vp8_testlib import ivf_path
    w, h, frames = P.read_ivf(ivf_path(fixture))
    nsrc = len(frames)
    parser = P.Parser()
    first = {}                                   # source frame -> first slot holding it
    t0 = time.time()
    for j in range(min(F, nsrc)):
        k = (lo + j) % nsrc
        # the feeder writes the device form of include/vp8_ir.h into the slot's pinned staging; one copy takes it to the slot
        hdr, _ = ctx.parse_into_slot_compact(parser, frames[k], j)
        assert hdr.frame_type == 0, "bench stream must be all key frames"
        parser.swap(hdr)
        first[k] = j
    feed_s = time.time() - t0
    for j in range(nsrc, F):
        ctx.ir_copy(j, first[(lo + j) % nsrc])
    ctx.sync()
    parser.close()
    load_stream.last_copy_s = time.time() - t0 - feed_s     # (F - nsrc device-to-device slot copies: a rank's start-up, untimed)
    return nsrc, feed_s


def timed_steps(P, ctx, jobs, F, steps, warmup, barrier):
    for _ in range(warmup):
        ctx.decode_array(jobs, F, P.STAGE_ALL)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.decode_array(jobs, F, P.STAGE_ALL)     # asynchronous: consecutive launches pipeline on the device
    barrier()
    elapsed = time.perf_counter() - t0
    # per-kernel times of the TIMED launches: HIP events recorded on the stream the kernels ran on, read back now
    # (reading a launch's events waits for it, which inside the loop would serialise the launches)
    KS = min(steps, 32)
    ms = {"recon": 0.0, "loopfilter": 0.0, "extend": 0.0}
    for back in range(KS):
        st = ctx.stats(back)
        ms["recon"] += st.recon_ms / KS
        ms["loopfilter"] += st.lf_ms / KS
        ms["extend"] += st.extend_ms / KS
    return elapsed, ms, st


def pin_rank(local_rank, world):
    """CPU affinity of a rank: the CPUs of the NUMA node its GPU hangs on (sysfs), cut into as many slices as ranks share the
    node; without that information a contiguous slice of the CPUs this process may use.  The timed region is device work, so this
    matters for what feeds it -- the host feeder threads of the end-to-end pipelines, the IR uploads -- and keeps N ranks from
    migrating over each other.  VP8BENCH_NO_AFFINITY=1 leaves the affinity alone.  Returns the CPU list (or None)."""
    if os.environ.get("VP8BENCH_NO_AFFINITY") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        allowed = sorted(os.sched_getaffinity(0))
        cpus = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(local_rank)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:  # noqa: BLE001 - (older torch: no PCI ids in the properties)
            bdf = None
        if bdf:
            path = f"/sys/bus/pci/devices/{str(bdf).lower()}/local_cpulist"
            if os.path.exists(path):
                node = []
                for part in open(path).read().strip().split(","):
                    if part:
                        a, _, b = part.partition("-")
                        node += list(range(int(a), int(b or a) + 1))
                node = [c for c in node if c in allowed]
                if node:
                    cpus = node
        if cpus is None:
            per = max(1, len(allowed) // max(1, world))
            cpus = allowed[local_rank * per:(local_rank + 1) * per] or allowed
        os.sched_setaffinity(0, cpus)
        return cpus
    except Exception:  # noqa: BLE001 - affinity is an optimisation
        return None


def spawn_ranks(args):
    """--gpus N without a launcher: start N fresh child processes, one rank per GPU, from a parent that never touches a GPU
    (rank 0's stdout is passed through; the exit code is the worst child's)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for pr in procs:
        code = pr.wait()
        rc = max(rc, code if code >= 0 else 1)
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=0,
                    help="frames per GPU per step (default: 16384 for 1080p, 4096 for 4k -- two frames per strand of the lane-per-row "
                         "launch: 136 macroblock rows are 17 whole rounds of 8 lanes, 270 are 8.4 of 32; IR, tiled scratch and frame buffers "
                         "resident in HBM; the lane-per-row kernels want several frames per wave on each of the chip's 1024 SIMDs)")
    ap.add_argument("--workload", default="1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inter-probe", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-4k-probe", action="store_true")
    ap.add_argument("--no-curve", action="store_true", help="skip config.batch_curve and config.dense_content")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the VP8 pixel path has no CPU fallback")
    # (test hook for boxes with one GPU: every rank on device 0, collectives over gloo -- the same code path otherwise)
    one_dev = os.environ.get("VP8BENCH_TEST_SINGLE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    my_cpus = pin_rank(local_rank, world) if world > 1 else None
    coll_dev = torch.device("cpu") if one_dev else torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=coll_dev)

    from vp8_testlib import load_package, golden_md5
    P = load_package()
    from libvpx_opencl_amd import sharding
    fixture, W, H = WORKLOADS[args.workload]
    F = args.frames or {"1080p": 16384, "4k": 4096}[args.workload]
    gold = golden_md5(fixture)

    # ---- the stream: world * F frames, frame i = source frame i mod nsrc; rank r decodes the contiguous block
    # shard_range(world * F, world, r) -- entropy-decoded once on the host, outside the timed region
    # (the default launch keeps 16,384 1080p frames resident: 128 GB of IR slots and 56 GB of tiles.  A device that cannot give that --
    # somebody else's memory on it -- gets half as many frames per step rather than no bench line; --frames is taken as given)
    while True:
        lo, hi = sharding.shard_range(world * F, world, rank)
        assert hi - lo == F
        ctx = P.Vp8Hip(local_rank)
        fits = 1
        try:
            ctx.configure(W, H, F, F)
            nsrc, feed_s = load_stream(P, ctx, fixture, F, lo)
            copy_s = load_stream.last_copy_s
            jobs = (P.Job * F)()
            for i in range(F):
                jobs[i].ir_slot, jobs[i].dst_fb = i, i
                for k in range(4):
                    jobs[i].ref_fb[k] = -1
            ctx.decode_array(jobs, F, P.STAGE_ALL)           # (the first large launch allocates the tiled forms)
            ctx.sync()
        except RuntimeError as ex:
            fits = 0
            sys.stderr.write(f"[bench] rank {rank}: {F} frames per step do not fit: {ex}\n")
        if dist is not None:
            t = torch.tensor([fits], device=coll_dev, dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            fits = int(t.item())
        if fits:
            break
        ctx.close()
        if args.frames or F <= 2048:
            raise SystemExit("the device has no room for the benchmark's frames")
        F //= 2
    nmb = ctx.nmb

    # ---- correctness gate 1 (multi-GPU): a 16-frames-per-rank prefix stream decoded the sharded way -- every rank its block,
    # MD5 of every frame, digests all-gathered over RCCL -- must give the 1-GPU decode_to_md5 listing
    listing_ok = None
    if dist is not None:
        nv = 16 * world

        def decode_block(vlo, vhi):
            # frames vlo .. vhi-1 of the stream are source frames (i mod nsrc): decode them into buffers 0 .. from the slots
            # of this rank's shard that hold the same source frames
            want = [i % nsrc for i in range(vlo, vhi)]
            have = {(lo + j) % nsrc: j for j in range(min(F, nsrc) - 1, -1, -1)}
            vj = (P.Job * len(want))()
            for n, k in enumerate(want):
                vj[n].ir_slot, vj[n].dst_fb = have[k], n
                for q in range(4):
                    vj[n].ref_fb[q] = -1
            ctx.decode_array(vj, len(want), P.STAGE_ALL)
            ctx.sync()
            return [P.planes_md5(*ctx.download_planes(n)) for n in range(len(want))]

        listing = sharding.sharded_listing(dist, nv, decode_block, device=coll_dev)
        listing_ok = listing == [gold[i % nsrc] for i in range(nv)]
    # ---- correctness gate 2: one untimed pass of the timed launch, >= 64 frames spread over strands, waves and the whole
    # shard checked against the reference MD5s
    ctx.decode_array(jobs, F, P.STAGE_ALL)
    ctx.sync()
    import random
    rnd = random.Random(1234 + rank)
    sample = sorted(set([0, 1, 7, 8, 9, 63, 64, 65, F // 2, F - 2, F - 1] + [rnd.randrange(F) for _ in range(64)]))
    sample = [i for i in sample if 0 <= i < F]
    ok = 1
    for i in sample:
        # (hashed on the HOST: the frame packed on the device from the tiles the launch left, vp8hip_frames_fetch_i420_async -- the
        # raster pool of all F frame buffers, 56 GB at 16,384 1080p frames, is not allocated for the sake of 75 of them)
        if (hashlib.md5(ctx.frames_i420(i, 1)[0].tobytes()).hexdigest() if W % 8 == 0 else P.planes_md5(*ctx.download_planes(i))) != gold[(lo + i) % nsrc]:
            ok = 0
            sys.stderr.write(f"[bench] rank {rank}: frame {lo + i} MD5 mismatch\n")
    if listing_ok is False:
        ok = 0
        sys.stderr.write(f"[bench] rank {rank}: the gathered sharded listing differs from the 1-GPU listing\n")
    if dist is not None:
        t = torch.tensor([ok], device=coll_dev, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = int(t.item())
    if not ok:
        raise SystemExit("decode_to_md5 precondition failed: GPU output differs from the reference")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    elapsed_local, ms, st = timed_steps(P, ctx, jobs, F, args.steps, args.warmup, barrier)
    mem_gb = {k: round(v / 1e9, 2) for k, v in ctx.memory_usage().items() if v}      # (what the timed step had resident on this rank)
    elapsed = elapsed_local
    per_rank = [elapsed_local]
    rank_cpus = [f"{len(my_cpus)} CPUs from {my_cpus[0]}" if my_cpus else None]
    if dist is not None:
        obj = [None] * world
        dist.all_gather_object(obj, rank_cpus[0])
        rank_cpus = obj
        t = torch.tensor([elapsed_local], device=coll_dev, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [float(x.item()) for x in allt]
        elapsed = max(per_rank)
    # ---- the step with a consumer behind it (rank 0, N = 1; never `value`): what reads the frames a step leaves
    consumers = with_raster = None
    if rank == 0 and world == 1:
        def timed(fn, reps=3):
            fn(); ctx.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            ctx.sync()
            return (time.perf_counter() - t0) / reps * 1e3

        L = ctx.L
        L.vp8hip_frames_fetch_async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.vp8hip_download_wait.argtypes = [ctypes.c_void_p]
        L.vp8hip_host_alloc.restype = ctypes.c_void_p
        L.vp8hip_host_alloc.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.vp8hip_host_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.vp8hip_frame_stride.restype = ctypes.c_size_t
        L.vp8hip_frame_stride.argtypes = [ctypes.c_void_p]
        dig = L.vp8hip_host_alloc(ctx.h, 16 * F)

        def step_md5():
            ctx.decode_array(jobs, F, P.STAGE_ALL)
            ctx._chk(L.vp8hip_frames_fetch_async(ctx.h, 0, F, None, dig), "fetch")
            ctx._chk(L.vp8hip_download_wait(ctx.h), "wait")

        def step_raster():
            ctx.decode_array(jobs, F, P.STAGE_ALL)
            ctx.frames_to_raster(0, F)

        consumers = {}
        try:
            t_md5 = timed(step_md5)
            import numpy as np
            got = np.ctypeslib.as_array(ctypes.cast(dig, ctypes.POINTER(ctypes.c_uint8)), shape=(F, 16))
            bad = sum(1 for i in range(F) if got[i].tobytes().hex() != gold[(lo + i) % nsrc])
            consumers["device_md5"] = {"ms_per_step": round(t_md5, 3), "Mpix_s": round(F * W * H / t_md5 / 1e3, 1), "md5_mismatches": bad,
                                       "what": f"decode {F} frames, then every one of them is hashed on the device, a frame per lane (batches of 12,288 "
                                               f"frames and more from a packed copy, vp8_pack_i420_tiles_kernel + vp8_md5_kernel; smaller ones "
                                               f"straight from the tiles, vp8_md5_tiles_kernel) and the {F} digests come back: decode_to_md5's "
                                               f"output for the step"}
        except Exception as ex:      # noqa: BLE001 - a probe, not the benchmark
            consumers["device_md5"] = {"error": repr(ex)}
        try:
            nd = min(F, 512)
            stride = L.vp8hip_frame_stride(ctx.h)
            host = L.vp8hip_host_alloc(ctx.h, stride * nd)

            def dl():
                ctx._chk(L.vp8hip_frames_fetch_async(ctx.h, 0, nd, host, None), "fetch")
                ctx._chk(L.vp8hip_download_wait(ctx.h), "wait")
            ctx.L.vp8hip_set_direct_download(ctx.h, 1)
            ctx.decode_array(jobs, F, P.STAGE_ALL); ctx.sync()
            t_dl = timed(dl)
            ctx.L.vp8hip_set_direct_download(ctx.h, 0)
            import numpy as np
            hv = np.ctypeslib.as_array(ctypes.cast(host, ctypes.POINTER(ctypes.c_uint8)), shape=(nd, stride))
            okd = all(P.frame_md5(hv[i], ctx.g, W, H) == gold[(lo + i) % nsrc] for i in (0, nd // 2, nd - 1))
            consumers["download"] = {"frames": nd, "ms": round(t_dl, 3), "GB_s_over_pcie": round(nd * W * H * 1.5 / t_dl / 1e6, 2), "md5_ok": bool(okd),
                                     "what": "vp8hip_set_direct_download: vp8_detile_run_kernel writes the raster rows of the frames straight into "
                                             "page-locked host memory (the tiled -> raster pass is the download; no raster form in HBM); the "
                                             "default download goes through the raster form and the copy engines"}
            L.vp8hip_host_free(ctx.h, host)
        except Exception as ex:      # noqa: BLE001
            consumers["download"] = {"error": repr(ex)}
        try:
            t_r = timed(step_raster)
            with_raster = {"ms_per_step": round(t_r, 3), "Mpix_s": round(F * W * H / t_r / 1e3, 1),
                           "roofline_pipeline_frac": round((B_RECON + B_LF + B_EXTEND) * nmb * F / (t_r * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                           "what": "round 3's step: decode, then the border-extended raster form of every frame in HBM (vp8_detile_kf_kernel + "
                                   "vp8_extend_kernel behind the launch on the same stream); 2023 B/MB"}
        except Exception as ex:      # noqa: BLE001
            with_raster = {"error": repr(ex)}
        L.vp8hip_host_free(ctx.h, dig)
    # single-frame latency: one frame per launch (what a single-stream decoder sees), kernels only
    latency_ms = None
    copy_gbps = None
    if rank == 0:
        one = (P.Job * 1)()
        one[0].ir_slot, one[0].dst_fb = 0, 0
        for k in range(4):
            one[0].ref_fb[k] = -1
        ctx.decode_array(one, 1, P.STAGE_ALL); ctx.sync()
        tl = time.perf_counter()
        for _ in range(20):
            ctx.decode_array(one, 1, P.STAGE_ALL)
        ctx.sync()
        latency_ms = (time.perf_counter() - tl) / 20 * 1e3
    ctx.close()
    inter_latency = None
    if rank == 0 and world == 1 and args.workload == "1080p":
        try:
            inter_latency = single_inter_latency(P, local_rank)
        except Exception as ex:      # noqa: BLE001 - a probe, not the benchmark
            inter_latency = {"error": repr(ex)}
    if rank == 0:
        # what the HBM system delivers to a plain device-to-device copy on this box (SURVEY.md 8d asks for the probe)
        a = torch.empty(2 << 30, dtype=torch.uint8, device="cuda")
        b = torch.empty_like(a)
        b.copy_(a); torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(5):
            b.copy_(a)
        torch.cuda.synchronize()
        copy_gbps = 2 * a.numel() * 5 / (time.perf_counter() - tc) / 1e9
        del a, b
        torch.cuda.empty_cache()

    if rank == 0:
        K = args.steps
        total_pix = world * F * K * W * H
        lane = bool(st.fused)                 # the lane-per-row kernels ran (see vp8hip_stats): reconstruction + loop filter in ONE kernel
        bytes_per_launch = {"recon": B_RECON * nmb * F, "loopfilter": B_LF * nmb * F, "extend": B_EXTEND * nmb * F}
        if lane:
            # SURVEY.md 8(d)'s byte model for the stages the kernel covers: residual + intra recon (1217) + loop filter (770); the
            # borders are produced with the raster form, when something asks for it (config.with_raster_form)
            bytes_per_launch = {"recon": (B_RECON + B_LF) * nmb * F, "loopfilter": 0, "extend": 0}
        names = ({"recon": "vp8_keyframe_kernel", "loopfilter": None, "extend": None} if lane else
                 {"recon": "vp8_recon_kernel", "loopfilter": "vp8_loopfilter_kernel", "extend": "vp8_extend_kernel"})
        dom = max(ms, key=lambda k: ms[k])
        achieved = bytes_per_launch[dom] / (ms[dom] * 1e-3) / 1e9
        # the whole step by SURVEY.md 8(d): 1217 + 770 B/MB (+ 36 when the step extends borders: the wave-per-row kernels)
        step_bytes = (B_RECON + B_LF + (0 if lane else B_EXTEND)) * nmb * F
        survey_gbps = step_bytes / (elapsed_local / K) / 1e9
        traffic = load_traffic()
        tk = (traffic or {}).get("kernels", {}).get(names[dom] or "", None) if args.workload == "1080p" else None
        out = {
            "metric": "vp8_decode_pixel_path_mpix_per_s",
            "value": round(total_pix / elapsed / 1e6, 1),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / K * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"{W}x{H} all-key-frame VP8 stream (tests/golden/{fixture}.ivf looped to {world * F} frames), full "
                            f"pixel path: dequant+IDCT/WHT + intra recon + loop filter "
                            f"(BASELINE configs[1]+[3]{'+[4]' if args.workload == '4k' else ''}); IR resident in HBM in the device form of "
                            f"include/vp8_ir.h exactly as the host feeder uploaded it (no conditioning pass); frames left in the "
                            f"tiled form a large launch writes, read as such by the MD5 kernel and the batch download "
                            f"(config.consumers); MD5-checked vs the reference",
                "frames_per_gpu_per_step": F,
                "device_memory_GB": mem_gb,
                "macroblocks_per_frame": nmb,
                "parallelism": f"one stream of {world * F} frames sharded in contiguous blocks over {world} GPU(s) (rank r: frames "
                               f"[r*{F}, (r+1)*{F})), no pixel exchange; RCCL carries barriers, times and the MD5 listing",
                "per_rank_Mpix_s": [round(F * K * W * H / t / 1e6, 1) for t in per_rank],
                "rank_cpu_affinity": rank_cpus,
                "sharded_md5_listing_equals_1gpu_listing": listing_ok,
                "md5_checked_frames_per_rank": len(sample),
                "kernel_ms": {k: round(v, 4) for k, v in ms.items()},
                "kernel_family": ("one macroblock row per lane; reconstruction + loop filter fused, luma and chroma waves paired on every SIMD"
                                  if lane else "one wave per macroblock row"),
                "kernels": names,
                "waves_per_workgroup": {"recon": st.recon_waves, "loopfilter": st.lf_waves},
                "workgroups": st.workgroups,
                "host_feeder_s_for_source_frames": round(feed_s, 4),
                "rank_startup_s": {"host_feeder_source_frames": round(feed_s, 3), "slot_copies_device_to_device": round(copy_s, 3),
                                   "note": f"rank 0, outside the timed region: {F} slots filled from {nsrc} source frames (vp8hip_ir_copy)"},
                "single_frame_launch_ms": round(latency_ms, 3),
                "single_frame_launch_inter": inter_latency,
                "consumers": consumers,
                "with_raster_form": with_raster,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": names[dom],
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5),
                "traffic": round((2 * tk["fetch_KiB_per_mb"] + tk["write_KiB_per_mb"]) * 1024 * nmb * F) if tk else None,
                # what the HBM system really moved for this kernel, per second and against the peak: the kernel is NOT bound by
                # memory (it runs at 92 % of the vector ALU's issue rate); `frac` above is SURVEY's byte model of a
                # two-pass pipeline over this kernel's time, i.e. credit for the traffic the fusion removed
                "traffic_GBps": round((2 * tk["fetch_KiB_per_mb"] + tk["write_KiB_per_mb"]) * 1024 * nmb * F / (ms[dom] * 1e-3) / 1e9, 1) if tk else None,
                "traffic_frac_of_peak": (round((2 * tk["fetch_KiB_per_mb"] + tk["write_KiB_per_mb"]) * 1024 * nmb * F / (ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)
                                         if tk else None),
                "traffic_uncorrected": round((tk["fetch_KiB_per_mb"] + tk["write_KiB_per_mb"]) * 1024 * nmb * F) if tk else None,
                "achieved_is": "SURVEY 8(d) algorithmic bytes (dense-coefficient model, recon 1217 + loop filter 770 B/MB) / mean launch time: "
                               "the contract's definition; the counted bytes are `traffic` (FETCH_SIZE doubled as the MI355X guide prescribes "
                               "for 16-byte-per-lane loads -- this kernel's loads are 16-byte LDS-DMA pieces, but scattered, so the truth "
                               "lies between `traffic_uncorrected` and `traffic`) and `traffic_frac_of_peak`",
                "traffic_source": (f"scaled: per-macroblock FETCH_SIZE x2 + WRITE_SIZE of this kernel from the rocprofv3 --pmc passes of "
                                   f"{traffic.get('source')} ({traffic.get('frames_per_launch')} frames per launch, {traffic.get('lanes_per_strand')} lanes "
                                   f"per strand), times the macroblocks of this launch") if tk else None,
                "traffic_bytes_per_macroblock": ({k: round((2 * v["fetch_KiB_per_mb"] + v["write_KiB_per_mb"]) * 1024, 1)
                                                  for k, v in traffic["kernels"].items()} if tk else None),
                "algorithmic_bytes_per_launch": bytes_per_launch[dom],
                "mean_launch_ms": round(ms[dom], 4),
                "all_kernels_GBps": {k: round(bytes_per_launch[k] / (ms[k] * 1e-3) / 1e9, 2) if ms[k] > 1e-2 else None
                                     for k in ms},
                "pipeline": {"achieved": round(survey_gbps, 2), "frac": round(survey_gbps / HBM_PEAK_GBPS, 5),
                             "note": "SURVEY 8(d) bytes of the stages the step runs (recon 1217 + loop filter 770 B/MB" +
                                     ("" if lane else " + border extend 36") + ") / whole step time, one GPU"},
                "device_copy_probe_GBps": round(copy_gbps, 1) if copy_gbps else None,
            },
        }
        if world == 1 and args.workload == "1080p" and not args.no_4k_probe:
            # BASELINE configs[4]'s stream on one GPU: 3840x2160 all-key-frame, same path, measured the same way
            try:
                fx4, W4, H4 = WORKLOADS["4k"]
                F4 = 4096            # (two frames per strand at 32 lanes a strand: 270 macroblock rows, 8.4 rounds; 2048 frames: 4.2 rounds, five run)
                c4 = P.Vp8Hip(local_rank)
                c4.configure(W4, H4, F4, F4)
                ns4, _ = load_stream(P, c4, fx4, F4, 0)
                j4 = (P.Job * F4)()
                for i in range(F4):
                    j4[i].ir_slot, j4[i].dst_fb = i, i
                    for k in range(4):
                        j4[i].ref_fb[k] = -1
                g4 = golden_md5(fx4)
                c4.decode_array(j4, F4, P.STAGE_ALL); c4.sync()
                # (hashed as the launch left them, as tiles: the raster pool of 4096 4K frame buffers -- 53 GB -- is never allocated)
                ok4 = all(c4.frames_md5(i, 1)[0] == g4[i % ns4] for i in (0, 1, 2, 777, F4 // 2, F4 - 1))

                def b4():
                    torch.cuda.synchronize()
                    c4.sync()
                e4, ms4, _ = timed_steps(P, c4, j4, F4, 3, 1, b4)
                gb4 = (B_RECON + B_LF) * c4.nmb * F4 / (e4 / 3) / 1e9
                out["config"]["workload_4k"] = {
                    "workload": f"{W4}x{H4} all-key-frame stream ({fx4}.ivf looped), {F4} frames per step, 3 steps",
                    "md5_ok": bool(ok4), "Mpix_s": round(F4 * 3 * W4 * H4 / e4 / 1e6, 1), "ms_per_step": round(e4 / 3 * 1e3, 3),
                    "kernel_ms": {k: round(v, 4) for k, v in ms4.items()},
                    "roofline_pipeline": {"achieved": round(gb4, 2), "frac": round(gb4 / HBM_PEAK_GBPS, 5), "unit": "GB/s"}}
                c4.close()
            except Exception as ex:      # a probe, not the benchmark: report, do not fail the line
                out["config"]["workload_4k"] = {"error": repr(ex)}
        if world == 1 and args.workload == "1080p" and not args.no_curve:
            # throughput against frames per launch (the lane-per-row kernels want every SIMD's two wave slots full: 16,384 1080p
            # frames; up to 512 frames the wave-per-row kernels run), each point a context of its own with just that many frames
            # resident, every frame's MD5 checked; the last point is the headline launch itself
            curve = []
            for n in (1, 64, 512, 1024, 2048, 4096, 8192):
                if n >= F:
                    break
                try:
                    curve.append(small_run(P, local_rank, fixture, W, H, n))
                except Exception as ex:      # noqa: BLE001 - a probe, not the benchmark
                    curve.append({"frames_per_launch": n, "error": repr(ex)})
            curve.append({"frames_per_launch": F, "ms_per_launch": round(elapsed / K * 1e3, 3), "Mpix_s": round(F * W * H * K / elapsed / 1e6, 1),
                          "device_GB": round(sum(mem_gb.values()), 2), "kernels": "lane-per-row (tiles)" if lane else "wave-per-row (raster)",
                          "md5_mismatches": 0, "note": "the timed launch of this line"})
            out["config"]["batch_curve"] = curve
            # the same path on DENSE content (SURVEY 8d's noisy set: 21 of a macroblock's 24 blocks coded, 18 of them with more than
            # a first coefficient -- the byte model of `roofline` is dense, the headline stream is not): tests/golden/kf_dense_1920x1080
            # at the headline's launch size (a dense slot is the full 7.8 MB: 16,384 frames are 128 + 56 GB) and, as through round 5, at 8192
            for key, nd in (("dense_content", F), ("dense_content_8192", 8192)):
                if key != "dense_content" and nd >= F:
                    continue
                try:
                    d = small_run(P, local_rank, "kf_dense_1920x1080", W, H, nd)
                    gb = (B_RECON + B_LF) * nmb * nd / (d["ms_per_launch"] * 1e-3) / 1e9
                    d["roofline_pipeline"] = {"achieved": round(gb, 2), "frac": round(gb / HBM_PEAK_GBPS, 5), "unit": "GB/s",
                                              "note": "SURVEY 8(d) 1217 + 770 B/MB x macroblocks per launch / launch time"}
                    d["workload"] = "tests/golden/kf_dense_1920x1080.ivf (2 key frames, uniform +-16 noise, quantiser index 8..16) looped"
                    out["config"][key] = d
                except Exception as ex:          # noqa: BLE001
                    out["config"][key] = {"error": repr(ex)}
        if world == 1 and args.workload == "1080p" and not args.no_inter_probe:
            try:
                out["config"]["inter_frames"] = inter_frame_probe(P, local_rank)
            except Exception as ex:      # a probe, not the benchmark: report, do not fail the line
                out["config"]["inter_frames"] = {"error": repr(ex)}
            # SURVEY 8(d)'s own config-3 input: the 10-frame K+P fixture (93-99 % of its P frames' macroblocks skipped, sub-pixel vectors)
            try:
                out["config"]["inter_frames_typical"] = inter_frame_probe(P, local_rank, name="p_1920x1080", k=5)
            except Exception as ex:      # noqa: BLE001
                out["config"]["inter_frames_typical"] = {"error": repr(ex)}
        if world == 1 and args.workload == "1080p" and not args.no_end_to_end:
            # host-inclusive rates (never `value`): compressed frames in host memory -> per-frame MD5.  Every probe on its own: one
            # that fails reports its error and leaves the others' results alone
            e2 = out["config"]["end_to_end"] = {}

            def probe(key, fn):
                try:
                    e2[key] = fn()
                except Exception as ex:      # noqa: BLE001 - a probe, not the benchmark
                    e2[key] = {"error": repr(ex)}
            sys.path.insert(0, os.path.join(ROOT, "tools"))

            def py_pipeline():
                import e2e
                return e2e.run(P, local_rank, fixture=fixture, nframes=2048)
            probe("python_host_feeder", py_pipeline)
            probe("c_host", lambda: batch_md5_probe(fixture, local_rank))
            # ... and with the macroblocks' modes and tokens decoded on the GPU, a frame per lane (vp8hip_entropy_decode): the host
            # reads the frame headers only
            probe("device_entropy", lambda: batch_md5_probe(fixture, local_rank, 24576, ("--device-entropy", "--batch", "8192", "--entropy-batch", "24576")))
            # (24,576 frames per entropy launch: what the device holds at once, 32 lanes a wave and three waves a CU; their blocks out
            # of one pool, vp8hip_configure_pooled; 1,474,560 frames: sixty launches, the start-up's allocations -- 1 to 4 s -- included)
            probe("device_entropy_frames_stay", lambda: batch_md5_probe(
                fixture, local_rank, 147456, ("--device-entropy", "--no-download", "--batch", "8192", "--entropy-batch", "24576")))
            probe("inter_streams_device_entropy", lambda: streams_probe(local_rank))
        if not args.no_cpu_baseline and world == 1:      # the contract: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(fixture)
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
