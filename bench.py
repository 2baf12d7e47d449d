#!/usr/bin/env python3
"""bench.py -- VP8 decode pixel path on MI355X: Mpix/s on a 1080p all-key-frame stream.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--workload 1080p|4k]

One "step" = one pass of the whole pixel path (dequant+IDCT/WHT + intra reconstruction, in-loop
deblocking filter, border extension: the three kernels behind include/vp8hip.h) over a batch of F
independent key frames whose IR (modes, eobs, dense coefficients) is ALREADY RESIDENT IN HBM.  The
batch is the committed fixture tests/golden/kf_1920x1080.ivf (10 key frames, entropy-decoded once
on the host, outside the timed region) looped F/10 times -- legal because every key frame is
independently decodable (reference: vp8/decoder/decodframe.c:610-639).  Before timing, decoded
frames are checked bit-exactly against the reference decoder's per-frame MD5s
(tests/golden/*.md5).

N > 1: launched by torch.distributed.run, one rank per GPU, every rank decodes its own F frames
(frames shard one-per-GPU, no pixel exchange: "scaling": "weak").  RCCL (backend "nccl") carries
only the start/stop barriers, the max-over-ranks time and the per-rank verification flag.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field definitions):
  value      whole-job Mpix/s (display pixels) = N * F * K * w * h / seconds
  roofline   dominant kernel: algorithmic bytes per launch (SURVEY.md 8d byte model) / mean launch
             time measured with HIP events on the launch stream, vs 8 TB/s HBM peak
  cpu_baseline  the REAL reference decoder (oracle/_ref, generic-C build of /root/reference) timed on
             this host on the same stream, 1 core; falls back to the repo's C restatement ("port")
"""
import argparse
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
B_DETILE = 384 + 384 + 36  # lane-per-row pipeline only: tiled scratch read, raster frame + borders written
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# HBM traffic per macroblock of the lane-per-row kernels (1080p key frames, G = 8 as at the default launch size), from
# `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, KiB; FETCH_SIZE doubled: the gfx950 correction
# for 16-byte-per-lane loads, which the detile pass confirms -- it reads exactly 384 B/MB) over tools/pmc_one.py 7 1024
# with VP8HIP_SIMT_LGG=3: profiles/r01_*_pmc_*_1024frames_G8*.csv.  The counter passes crash or hang at 8192 frames per
# launch and under torch, so bench.py scales these per-macroblock figures instead of counting live.
PMC_TRAFFIC_B_PER_MB = {        # the loop filter writes the raster frame buffers, then vp8_extend_kernel (default)
    "recon": 2 * 447.7 + 384.0, "loopfilter": 2 * 242.2 + 634.4, "extend": 2 * 27.5 + 40.4}
PMC_TRAFFIC_B_PER_MB_DETILE = { # tiled -> raster pass (vp8_detile_kernel) after the loop filter
    "recon": 2 * 448.68 + 385.22, "loopfilter": 2 * 239.63 + 414.12, "extend": 2 * 192.08 + 466.87}

WORKLOADS = {
    "1080p": ("kf_1920x1080", 1920, 1080),
    "4k": ("kf_3840x2160", 3840, 2160),
}


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
            ncpu = os.cpu_count() or 1
            if ncpu > 1:
                r2 = max(1, reps // 6)
                procs = [subprocess.Popen([ref, "--time", str(r2), ivf], stdout=subprocess.PIPE, text=True)
                         for _ in range(ncpu)]
                agg = 0.0
                for pr in procs:
                    o, _ = pr.communicate(timeout=900)
                    _, px, sc = o.split()
                    agg += float(px) / float(sc) / 1e6
                res["all_cores"] = {"value": round(agg, 1), "unit": "Mpix/s", "cores": ncpu,
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



def inter_frame_probe(P, device, n=1024, name="p_1920x1080", k=5):
    """BASELINE configs[2] beside the headline: six-tap motion compensation + IDCT + loop filter on REAL inter
    frames.  The stream is decoded the normal way up to frame k-1, then n jobs decode frame k from the same
    references into n different frame buffers (n independent streams in lock step); whole-launch wall time."""
    from vp8_testlib import ivf_path
    w, h, frames = P.read_ivf(ivf_path(name))
    ctx = P.Vp8Hip(device)
    ctx.configure(w, h, n + 4, 2)
    parser = P.Parser()
    for data in frames[:k]:
        hdr = ctx.parse_into_slot(parser, data, 0)
        ctx.upload(0)
        r = parser.refs
        ctx.decode([(0, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx) if hdr.frame_type else None)], P.STAGE_ALL)
        ctx.sync()
        parser.swap(hdr)
    hdr = ctx.parse_into_slot(parser, frames[k], 1)
    ctx.upload(1)
    r = parser.refs
    jobs = (P.Job * n)()
    for i in range(n):
        jobs[i].ir_slot, jobs[i].dst_fb = 1, 4 + i
        jobs[i].ref_fb[1], jobs[i].ref_fb[2], jobs[i].ref_fb[3] = r.lst_idx, r.gld_idx, r.alt_idx
    ctx.decode_array(jobs, n, P.STAGE_ALL); ctx.sync()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.decode_array(jobs, n, P.STAGE_ALL)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    st = ctx.stats()
    parser.close()
    ctx.close()
    return {"workload": f"{name}.ivf frame {k} (inter, six-tap, normal loop filter) x {n} independent copies per launch",
            "Mpix_s": round(n * w * h / dt / 1e6, 1), "ms_per_launch": round(dt * 1e3, 3),
            "kernel_ms": {"recon": round(st.recon_ms, 3), "loopfilter": round(st.lf_ms, 3), "extend": round(st.extend_ms, 3)},
            "kernel_family": "one wave per macroblock row"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=0,
                    help="frames per GPU per step (default: 8192 for 1080p, 2048 for 4k -- about 165 GB of IR, "
                         "tiled scratch and frame buffers resident in HBM; the lane-per-row kernels want several "
                         "frames per wave on each of the chip's 1024 SIMDs)")
    ap.add_argument("--workload", default="1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inter-probe", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    dist = None
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the VP8 pixel path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from vp8_testlib import load_package, golden_md5, ivf_path
    P = load_package()
    fixture, W, H = WORKLOADS[args.workload]
    F = args.frames or {"1080p": 8192, "4k": 2048}[args.workload]

    # ---- host feeder once (outside the timed region): 10 key frames -> IR
    w, h, frames = P.read_ivf(ivf_path(fixture))
    assert (w, h) == (W, H)
    nsrc = len(frames)
    gold = golden_md5(fixture)
    ctx = P.Vp8Hip(local_rank)
    ctx.configure(W, H, F, F)
    parser = P.Parser()
    t_feed0 = time.time()
    for i, data in enumerate(frames):
        hdr = ctx.parse_into_slot(parser, data, i)
        assert hdr.frame_type == 0, "bench stream must be all key frames"
        parser.swap(hdr)
        ctx.upload(i)
    feed_s = time.time() - t_feed0
    for i in range(nsrc, F):
        ctx.ir_copy(i, i % nsrc)
    ctx.sync()
    parser.close()

    jobs = (P.Job * F)()
    for i in range(F):
        jobs[i].ir_slot, jobs[i].dst_fb = i, i
        for k in range(4):
            jobs[i].ref_fb[k] = -1
    nmb = ctx.nmb

    # ---- correctness gate: one untimed pass, check a spread of frames against the reference MD5s
    ctx.decode_array(jobs, F, P.STAGE_ALL)
    ctx.sync()
    ok = 1
    for i in sorted(set([0, 1, nsrc - 1, F // 2, F - 1])):
        if P.planes_md5(*ctx.download_planes(i)) != gold[i % nsrc]:
            ok = 0
            sys.stderr.write(f"[bench] rank {rank}: frame {i} MD5 mismatch\n")
    if dist is not None:
        t = torch.tensor([ok], device="cuda", dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = int(t.item())
    if not ok:
        raise SystemExit("decode_to_md5 precondition failed: GPU output differs from the reference")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    for _ in range(args.warmup):
        ctx.decode_array(jobs, F, P.STAGE_ALL)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.decode_array(jobs, F, P.STAGE_ALL)     # asynchronous: consecutive launches pipeline on the device
    barrier()
    elapsed = time.perf_counter() - t0
    # per-kernel times of the TIMED launches: HIP events recorded on the streams the kernels ran on, read back
    # now (reading a launch's events waits for it, which inside the loop would serialise the launches)
    KS = min(args.steps, 32)
    k_recon = k_lf = k_ext = 0.0
    for back in range(KS):
        st = ctx.stats(back)
        k_recon += st.recon_ms
        k_lf += st.lf_ms
        k_ext += st.extend_ms
    # the same kernels launched one at a time (the tiled -> raster pass of launch k otherwise overlaps the recon of
    # launch k+1 and both stretch): stand-alone durations, used to name the dominant kernel
    alone = [0.0, 0.0, 0.0]
    for _ in range(3):
        ctx.decode_array(jobs, F, P.STAGE_ALL)
        ctx.sync()
        st = ctx.stats()
        alone[0] += st.recon_ms / 3
        alone[1] += st.lf_ms / 3
        alone[2] += st.extend_ms / 3
    # single-frame latency: one frame per launch (what a single-stream decoder sees), kernels only
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
    # what the HBM system delivers to a plain device-to-device copy on this box (SURVEY.md 8d asks for the probe)
    copy_gbps = None
    if rank == 0:
        a = torch.empty(2 << 30, dtype=torch.uint8, device="cuda")
        b = torch.empty_like(a)
        b.copy_(a); torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(5):
            b.copy_(a)
        torch.cuda.synchronize()
        copy_gbps = 2 * a.numel() * 5 / (time.perf_counter() - tc) / 1e9
        del a, b
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        K = args.steps
        total_pix = world * F * K * W * H
        lane = st.recon_waves == 1            # the lane-per-row kernels ran (see vp8hip_stats)
        detile = bool(st.detile_pass)         # ... finished by the tiled -> raster pass instead of the loop filter's own raster output
        ms = {"recon": k_recon / KS, "loopfilter": k_lf / KS, "extend": k_ext / KS}
        bytes_per_launch = {"recon": B_RECON * nmb * F, "loopfilter": B_LF * nmb * F,
                            "extend": (B_DETILE if detile else B_EXTEND) * nmb * F}
        names = ({"recon": "vp8_recon_simt_kernel", "loopfilter": "vp8_loopfilter_simt_kernel",
                  "extend": "vp8_detile_kernel (tiled -> raster + border extension)" if detile else "vp8_extend_kernel"} if lane else
                 {"recon": "vp8_recon_kernel", "loopfilter": "vp8_loopfilter_kernel", "extend": "vp8_extend_kernel"})
        ms_alone = {"recon": alone[0], "loopfilter": alone[1], "extend": alone[2]}
        dom = max(ms_alone, key=lambda k: ms_alone[k])
        achieved = bytes_per_launch[dom] / (ms[dom] * 1e-3) / 1e9
        pipeline_gbps = sum(bytes_per_launch.values()) / (elapsed / K) / 1e9
        # SURVEY.md 8(d)'s own figure for the full key-frame path: 1217 + 770 + 36 = 2023 B/MB -- without the bytes of the
        # tiled -> raster pass, which is this implementation's extra pass, not part of the algorithm
        survey_gbps = (B_RECON + B_LF + B_EXTEND) * nmb * F / (elapsed / K) / 1e9
        pmc = PMC_TRAFFIC_B_PER_MB_DETILE if detile else PMC_TRAFFIC_B_PER_MB
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
                "workload": f"{W}x{H} all-key-frame VP8 stream (tests/golden/{fixture}.ivf looped), full pixel "
                            f"path: dequant+IDCT/WHT + intra recon + loop filter + border extend "
                            f"(BASELINE configs[1]+[3]); IR resident in HBM, MD5-checked vs the reference",
                "frames_per_gpu_per_step": F,
                "macroblocks_per_frame": nmb,
                "parallelism": f"frame-parallel, {world} GPU(s), no pixel exchange",
                "kernel_ms": {k: round(v, 4) for k, v in ms.items()},
                "kernel_ms_launched_alone": {k: round(v, 4) for k, v in ms_alone.items()},
                "kernel_family": "one macroblock row per lane, macroblock-tiled scratch frames" if lane
                                 else "one wave per macroblock row",
                "kernels": names,
                "waves_per_workgroup": {"recon": st.recon_waves, "loopfilter": st.lf_waves},
                "workgroups": st.workgroups,
                "host_feeder_s_for_source_frames": round(feed_s, 4),
                "single_frame_launch_ms": round(latency_ms, 3),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": names[dom],
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5),
                "traffic": round(pmc[dom] * nmb * F) if lane and args.workload == "1080p" else None,
                "traffic_note": "HBM bytes per launch = per-macroblock FETCH_SIZE x2 + WRITE_SIZE of the same kernel measured "
                                "with rocprofv3 --pmc at 1024 frames per launch (profiles/r01_g_pmc_*.csv) x macroblocks "
                                "per launch; null for configurations that were not counted",
                "traffic_bytes_per_macroblock": ({k: round(v, 1) for k, v in pmc.items()}
                                                 if lane and args.workload == "1080p" else None),
                "algorithmic_bytes_per_launch": bytes_per_launch[dom],
                "mean_launch_ms": round(ms[dom], 4),
                "all_kernels_GBps": {k: round(bytes_per_launch[k] / (ms[k] * 1e-3) / 1e9, 2) if ms[k] > 0 else None
                                     for k in ms},
                "pipeline": {"achieved": round(survey_gbps, 2), "frac": round(survey_gbps / HBM_PEAK_GBPS, 5),
                             "note": "SURVEY 8(d) bytes of the whole path (recon + loop filter + border extend = 2023 B/MB) / "
                                     "whole step time",
                             "achieved_counting_own_detile_pass": round(pipeline_gbps, 2),
                             "frac_counting_own_detile_pass": round(pipeline_gbps / HBM_PEAK_GBPS, 5)},
                "device_copy_probe_GBps": round(copy_gbps, 1) if copy_gbps else None,
            },
        }
        ctx.close()
        if world == 1 and args.workload == "1080p" and not args.no_inter_probe:
            try:
                out["config"]["inter_frames"] = inter_frame_probe(P, local_rank)
            except Exception as ex:      # a probe, not the benchmark: report, do not fail the line
                out["config"]["inter_frames"] = {"error": repr(ex)}
        if world == 1 and args.workload == "1080p" and not args.no_end_to_end:
            # host-inclusive rate (never `value`): compressed frames in host memory -> per-frame MD5, tools/e2e.py
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import e2e
                out["config"]["end_to_end"] = e2e.run(P, local_rank, fixture=fixture)
            except Exception as ex:      # a probe, not the benchmark: report, do not fail the line
                out["config"]["end_to_end"] = {"error": repr(ex)}
        if not args.no_cpu_baseline and world == 1:      # the contract: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(fixture)
        print(json.dumps(out))
    else:
        ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
