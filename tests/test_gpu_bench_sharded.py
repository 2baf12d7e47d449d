"""GPU (-m gpu): bench.py's N > 1 path on a one-GPU box -- two ranks spawned by bench.py itself, both on device 0
(VP8BENCH_TEST_SINGLE_DEVICE=1: collectives over gloo, everything else as on an 8-GPU node): the stream is sharded in
contiguous blocks, every rank MD5-checks its shard, and the MD5 listing of a sharded prefix stream, gathered over the
process group, must equal the 1-GPU decode_to_md5 listing."""
import json
import os
import subprocess
import sys

import pytest

from vp8_testlib import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("workload,frames", [("1080p", 64), ("4k", 32)])      # (4k: BASELINE configs[4]'s stream through the sharded path)
def test_bench_two_ranks_one_device(workload, frames):
    env = dict(os.environ, VP8BENCH_TEST_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload, "--frames", str(frames),
                        "--steps", "1", "--warmup", "1", "--no-inter-probe", "--no-4k-probe", "--no-end-to-end", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["sharded_md5_listing_equals_1gpu_listing"] is True
    assert d["config"]["md5_checked_frames_per_rank"] >= min(32, frames // 2)
    assert len(d["config"]["rank_cpu_affinity"]) == 2
    assert len(d["config"]["per_rank_Mpix_s"]) == 2 and d["value"] > 0


def test_rccl_smoke_when_two_devices_are_visible():
    """backend="nccl" (RCCL) itself: a barrier and an all_gather of 16 bytes between two ranks on two devices -- what bench.py's
    multi-GPU path uses it for.  Skips on a one-GPU box (the round's GPU box is one)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: RCCL needs two devices")
    code = (
        "import os, torch, torch.distributed as dist\n"
        "r = int(os.environ['RANK']); torch.cuda.set_device(r)\n"
        "dist.init_process_group(backend='nccl', device_id=torch.device('cuda', r))\n"
        "dist.barrier()\n"
        "t = torch.full((16,), r + 1, dtype=torch.uint8, device='cuda')\n"
        "out = [torch.zeros_like(t) for _ in range(2)]\n"
        "dist.all_gather(out, t)\n"
        "assert [int(o[0]) for o in out] == [1, 2]\n"
        "dist.barrier(); dist.destroy_process_group(); print('rccl ok', r)\n")
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for pr in procs:
        out, err = pr.communicate(timeout=300)
        assert pr.returncode == 0 and "rccl ok" in out, err[-2000:]
