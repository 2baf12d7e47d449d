#!/bin/bash
# quick loop: fused-kernel parity subsets, A/B bench, stamps.  usage: tools/r3_quick.sh <tag> [notest] [no2k]
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r3q}; mkdir -p $O
if [ "$2" != "notest" ]; then
timeout 900 python -m pytest tests/test_gpu_lane_shapes.py -k "fused" -x -q > $O/t_shapes.log 2>&1; echo "shapes rc=$?" | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -k "lane and not lane_2k and not lane_detile and not wave" -x -q > $O/t_parity.log 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt
tail -n 3 $O/t_shapes.log; tail -n 3 $O/t_parity.log
fi
B="--steps 5 --warmup 2 --no-inter-probe --no-4k-probe --no-end-to-end --no-cpu-baseline"
timeout 600 python bench.py $B > $O/bench_fused.json 2> $O/bench_fused.err; echo "bench fused rc=$?" | tee -a $O/summary.txt
if [ "$3" != "no2k" ]; then VP8HIP_FUSED=0 timeout 600 python bench.py $B > $O/bench_2k.json 2> $O/bench_2k.err; echo "bench 2k rc=$?" | tee -a $O/summary.txt; fi
python - <<PY
import json
for n in ("fused", "2k"):
    try:
        d = json.loads(open("$O/bench_%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["config"]["kernel_ms"], d["roofline"]["pipeline"]["frac"])
    except Exception as e:
        print(n, "ERR", e)
PY
L=libvpx.opencl_amd/lib
if [ -f $L/var/libvp8hip_stamps.so ]; then
cp $L/libvp8hip.so /tmp/keep.so; cp $L/var/libvp8hip_stamps.so $L/libvp8hip.so
timeout 300 python3 tools/stamps_kf.py 8192 > $O/stamps_8192.txt 2>&1
cp /tmp/keep.so $L/libvp8hip.so
cat $O/stamps_8192.txt
fi
