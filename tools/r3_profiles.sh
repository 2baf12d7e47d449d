#!/bin/bash
# round-3 profile set: kernel-trace stats of the bench command (fused default and two-kernel pipeline), TCC traffic passes at
# full occupancy (1024 frames, 64 lanes per strand: 1024 luma + 1024 chroma waves), SQ / instruction-cache passes at 8192 frames
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3prof}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
B="--steps 10 --warmup 2 --no-inter-probe --no-4k-probe --no-end-to-end --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_fused -- python3 $R/bench.py $B > $O/kt_fused_bench.json 2> $O/kt_fused.err; echo "kt_fused rc=$?" >> $O/summary.txt
VP8HIP_FUSED=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_2k -- python3 $R/bench.py $B > $O/kt_2k_bench.json 2> $O/kt_2k.err; echo "kt_2k rc=$?" >> $O/summary.txt
timeout 600 python3 $R/bench.py $B > $O/unprofiled_fused_bench.json 2> $O/unprofiled_fused.err; echo "unprofiled rc=$?" >> $O/summary.txt
pmc() {  # name frames mbs lgg shared counters...
    local name=$1 nf=$2 lgg=$3 sh=$4; shift; shift; shift; shift
    VP8HIP_SIMT_LGG=$lgg timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python3 $R/tools/pmc_one.py 7 $nf kf_1920x1080 $sh > $O/$name.log 2>&1
    echo "$name rc=$?" >> $O/summary.txt
    python3 $R/tools/pmc_summary.py $O/$name $((8160 * nf)) > $O/$name.txt 2>&1
}
pmc fetch_1024_G64 1024 6 "" FETCH_SIZE
pmc write_1024_G64 1024 6 "" WRITE_SIZE
pmc fetch_1024_G8 1024 3 "" FETCH_SIZE
pmc write_1024_G8 1024 3 "" WRITE_SIZE
pmc sq1_8192_sharedIR 8192 0 shared SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pmc sq2_8192_sharedIR 8192 0 shared SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
pmc ic_8192_sharedIR 8192 0 shared SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
pmc write_8192_sharedIR 8192 0 shared WRITE_SIZE
cd $R; cat $O/summary.txt
for f in $O/*.txt; do echo "=== $f"; cat $f; done 2>/dev/null | grep -v "^$" | head -120
find $O -name "*kernel_stats.csv" | head; for f in $(find $O -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f; done
python3 - <<PY
import json
for n in ("kt_fused", "kt_2k", "unprofiled_fused"):
    try:
        d = json.loads(open("$O/%s_bench.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["config"]["kernel_ms"], d["roofline"]["pipeline"]["frac"], d["roofline"]["frac"])
    except Exception as e:
        print(n, "ERR", e)
PY
