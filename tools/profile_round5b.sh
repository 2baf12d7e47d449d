#!/bin/bash
# Round 5, second part of the profile set: the kernel trace of the bench command itself (whose average kernel duration the bench line's
# HIP-event time must agree with), the same command unprofiled, and the inter launches' FETCH / WRITE / SQ passes with ONE repetition
# (tools/profile_round5.sh's ran into their time limit under the profiler).
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r05_b}; NJ=${2:-4096}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O/for_profiles
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
B="--steps 10 --warmup 2 --no-inter-probe --no-4k-probe --no-end-to-end --no-cpu-baseline --no-curve"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_bench -- python3 $R/bench.py $B > $O/kt_bench.json 2> $O/kt_bench.err; echo "kt_bench rc=$?" >> $O/summary.txt
timeout 600 python3 $R/bench.py $B > $O/unprofiled_bench.json 2> $O/unprofiled.err; echo "unprofiled rc=$?" >> $O/summary.txt
f=$(find $O/kt_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/for_profiles/${TAG}_kt_bench_kernel_stats.csv
cp $O/kt_bench.json $O/for_profiles/${TAG}_kt_bench.json; cp $O/unprofiled_bench.json $O/for_profiles/${TAG}_unprofiled_bench.json
pmc() {  # name macroblocks-per-dispatch command... -- counters...
    local name=$1 nmb=$2; shift; shift
    local cmd=(); while [ "$1" != "--" ]; do cmd+=("$1"); shift; done; shift
    timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python3 "${cmd[@]}" > $O/$name.log 2>&1
    echo "$name rc=$?" >> $O/summary.txt
    python3 $R/tools/pmc_summary.py $O/$name $nmb > $O/for_profiles/${TAG}_pmc_$name.summary.txt 2>&1
}
pmc inter_fetch_$NJ $((8160 * NJ)) $R/tools/inter_chain_time.py $NJ 1 -- FETCH_SIZE
pmc inter_write_$NJ $((8160 * NJ)) $R/tools/inter_chain_time.py $NJ 1 -- WRITE_SIZE
pmc inter_sq1_$NJ $((8160 * NJ)) $R/tools/inter_chain_time.py $NJ 1 -- SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
cd $R; cat $O/summary.txt
head -6 $O/for_profiles/${TAG}_kt_bench_kernel_stats.csv | cut -c1-150
python3 - <<PY
import json
for n in ("kt_bench", "unprofiled_bench"):
    try:
        d = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["config"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["mean_launch_ms"], d["roofline"].get("traffic_frac_of_peak"))
    except Exception as e:
        print(n, "ERR", e)
PY
for f in $O/for_profiles/*inter*.summary.txt; do echo "=== $f"; grep -A10 "vp8_inter_pred\|vp8_interframe\|vp8_detile_kf" $f | cut -c1-150 | head -60; done
