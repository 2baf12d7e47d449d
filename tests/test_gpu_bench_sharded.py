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


def test_bench_two_ranks_one_device():
    env = dict(os.environ, VP8BENCH_TEST_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "64", "--steps", "1", "--warmup", "1",
                        "--no-inter-probe", "--no-4k-probe", "--no-end-to-end", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["sharded_md5_listing_equals_1gpu_listing"] is True
    assert d["config"]["md5_checked_frames_per_rank"] >= 32
    assert len(d["config"]["per_rank_Mpix_s"]) == 2 and d["value"] > 0
