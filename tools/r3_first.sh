#!/bin/bash
# round 3, first GPU run of the fused key-frame kernel: parity on its own test subsets, then A/B bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lane_shapes.py -k "fused" -x -q > $O/t_shapes.log 2>&1; echo "shapes rc=$?" | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -k "lane and not lane_2k and not lane_detile and not wave" -x -q > $O/t_parity.log 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt
B="--steps 5 --warmup 2 --no-inter-probe --no-4k-probe --no-end-to-end --no-cpu-baseline"
timeout 600 python bench.py $B > $O/bench_fused.json 2> $O/bench_fused.err; echo "bench fused rc=$?" | tee -a $O/summary.txt
VP8HIP_FUSED=0 timeout 600 python bench.py $B > $O/bench_2k.json 2> $O/bench_2k.err; echo "bench 2k rc=$?" | tee -a $O/summary.txt
tail -5 $O/t_shapes.log $O/t_parity.log
python - <<'PY'
import json
for n in ("fused", "2k"):
    try:
        d = json.loads(open(f"gpurun_out/r3a/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["config"]["kernel_ms"], d["roofline"]["pipeline"])
    except Exception as e:
        print(n, "ERR", e)
PY
