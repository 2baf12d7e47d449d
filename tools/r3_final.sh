#!/bin/bash
# end-of-round set: GPU suite, the default bench line (unprofiled), the kernel trace of the bench command
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3final}; mkdir -p $O
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $O/suite.log
python bench.py > $O/bench_full.json 2> $O/bench_full.err
cd /tmp; export TMPDIR=/tmp
B="--steps 10 --warmup 2 --no-inter-probe --no-4k-probe --no-end-to-end --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py $B > $O/kt_bench.json 2> $O/kt.err
timeout 600 python3 $R/bench.py $B > $O/unprofiled_bench.json 2> $O/unprofiled.err
# the entropy decoder on the device: kernel trace of a batch_md5 run (8 batches of 4096 frames, frames downloaded)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ent -- $R/libvpx.opencl_amd/bin/batch_md5 --device-entropy --batch 4096 --loop 3277 $R/tests/golden/kf_1920x1080.ivf /tmp/ent.md5 > $O/kt_ent.log 2>&1
cd $R
cat $O/suite.log
tail -1 $O/kt_ent.log
find $O/kt_ent -name "*kernel_stats.csv" | head -1 | xargs cut -d, -f1-5 | head -8
python3 - <<PY
import json
for n in ("bench_full", "kt_bench", "unprofiled_bench"):
    try:
        d = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["config"]["kernel_ms"], d["roofline"]["pipeline"]["frac"], d["roofline"]["frac"])
        if "inter_frames" in d["config"]: print("   inter", {k: d["config"]["inter_frames"][k] for k in ("Mpix_s", "ms_per_launch", "chained", "kernel_ms")}, d["config"]["inter_frames"]["roofline"]["frac"])
        if "end_to_end" in d["config"]: print("   e2e", d["config"]["end_to_end"]["Mpix_s"], d["config"]["end_to_end"]["frames_per_s"], d["config"]["end_to_end"].get("md5_on"))
        if "end_to_end" in d["config"]:
            for k in ("c_host", "device_entropy", "device_entropy_frames_stay"): print("   e2e", k, d["config"]["end_to_end"].get(k))
        if "workload_4k" in d["config"]: print("   4k", d["config"]["workload_4k"]["Mpix_s"], d["config"]["workload_4k"]["ms_per_step"])
    except Exception as e:
        print(n, "ERR", e)
PY
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs cut -d, -f1-4 | head -6
