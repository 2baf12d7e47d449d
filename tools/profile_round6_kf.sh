#!/bin/bash
# Round 6: the key-frame kernel after the transform's v_mul_hi_i32 form: the kernel
# trace of the bench command (whose average kernel duration the bench line's HIP-event time must agree with), the same command
# unprofiled, and the key-frame kernel's counters at the benchmark's occupancy (8192 frames, a copy of the IR per slot; separate
# --pmc passes, --kernel-trace only beside them): FETCH_SIZE, WRITE_SIZE, two SQ sets, the instruction cache.
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r06_d}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O/for_profiles
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
    timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python3 "${cmd[@]}" > $O/$name.log 2>&1
    echo "$name rc=$?" >> $O/summary.txt
    python3 $R/tools/pmc_summary.py $O/$name $nmb > $O/for_profiles/${TAG}_pmc_$name.summary.txt 2>&1
}
N=8192
pmc kf_fetch_$N $((8160 * N)) $R/tools/pmc_one.py 7 $N kf_1920x1080 -- FETCH_SIZE
pmc kf_write_$N $((8160 * N)) $R/tools/pmc_one.py 7 $N kf_1920x1080 -- WRITE_SIZE
pmc kf_sq1_$N $((8160 * N)) $R/tools/pmc_one.py 7 $N kf_1920x1080 -- SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pmc kf_sq2_$N $((8160 * N)) $R/tools/pmc_one.py 7 $N kf_1920x1080 -- SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH
pmc kf_icache_$N $((8160 * N)) $R/tools/pmc_one.py 7 $N kf_1920x1080 shared -- SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
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
for f in $O/for_profiles/*kf_*.summary.txt; do echo "=== $f"; grep -A10 "vp8_keyframe_kernel" $f | cut -c1-150 | head -14; done
