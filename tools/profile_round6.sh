#!/bin/bash
# Round 6 profile passes, one kind per call (counter passes stay apart from each other and carry --kernel-trace only):
#   tools/profile_round6.sh TAG clock [jobs]        GRBM_GUI_ACTIVE of the launches with inter frames (the clock the chip holds: / 8 / wall)
#   tools/profile_round6.sh TAG sq [jobs]           SQ instruction / wait counters of the same launches
#   tools/profile_round6.sh TAG ta [jobs]           texture-addresser / vector-cache counters
#   tools/profile_round6.sh TAG fetch|write [jobs]  HBM traffic
#   tools/profile_round6.sh TAG whole_sq|whole_fetch [jobs]   the same counters on a stream of whole-pixel vectors (tools/whole_pixel_time.py)
#   tools/profile_round6.sh TAG kt_bench            kernel trace of the bench command + the same command unprofiled
#   tools/profile_round6.sh TAG kt_chain [jobs]     kernel trace of the chained launches
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r06_a}; WHAT=${2:-sq}; NJ=${3:-4096}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O/for_profiles
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
pmc() {  # name macroblocks-per-dispatch command... -- counters...
    local name=$1 nmb=$2; shift; shift
    local cmd=(); while [ "$1" != "--" ]; do cmd+=("$1"); shift; done; shift
    timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python3 "${cmd[@]}" > $O/$name.log 2>&1
    echo "$name rc=$?" >> $O/summary.txt
    python3 $R/tools/pmc_summary.py $O/$name $nmb > $O/for_profiles/${TAG}_pmc_$name.summary.txt 2>&1
    grep -A12 "vp8_inter_pred\|vp8_interframe\|vp8_keyframe" $O/for_profiles/${TAG}_pmc_$name.summary.txt | cut -c1-160
}
NMB=$((8160 * NJ))
case $WHAT in
clock) pmc inter_clock_$NJ $NMB $R/tools/inter_chain_time.py $NJ 2 -- GRBM_GUI_ACTIVE GRBM_COUNT ;;
sq)    pmc inter_sq1_$NJ $NMB $R/tools/inter_chain_time.py $NJ 1 -- SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY ;;
sq2)   pmc inter_sq2_$NJ $NMB $R/tools/inter_chain_time.py $NJ 1 -- SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM ;;
ta)    pmc inter_ta_$NJ $NMB $R/tools/inter_chain_time.py $NJ 1 -- TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum ;;
fetch) pmc inter_fetch_$NJ $NMB $R/tools/inter_chain_time.py $NJ 1 -- FETCH_SIZE ;;
write) pmc inter_write_$NJ $NMB $R/tools/inter_chain_time.py $NJ 1 -- WRITE_SIZE ;;
whole_sq)    pmc whole_sq1_$NJ $NMB $R/tools/whole_pixel_time.py $NJ 1 -- SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY ;;
whole_fetch) pmc whole_fetch_$NJ $NMB $R/tools/whole_pixel_time.py $NJ 1 -- FETCH_SIZE ;;
kt_chain)
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_chain -- python3 $R/tools/inter_chain_time.py $NJ 4 > $O/kt_chain.log 2>&1; echo "kt_chain rc=$?" >> $O/summary.txt
    f=$(find $O/kt_chain -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/for_profiles/${TAG}_kt_chain${NJ}_kernel_stats.csv
    cp $O/kt_chain.log $O/for_profiles/${TAG}_kt_chain${NJ}.log; cat $O/kt_chain.log; head -8 $f | cut -c1-160 ;;
kt_bench)
    B="--steps 10 --warmup 2 --no-inter-probe --no-4k-probe --no-end-to-end --no-cpu-baseline --no-curve"
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_bench -- python3 $R/bench.py $B > $O/kt_bench.json 2> $O/kt_bench.err; echo "kt_bench rc=$?" >> $O/summary.txt
    timeout 600 python3 $R/bench.py $B > $O/unprofiled_bench.json 2> $O/unprofiled.err; echo "unprofiled rc=$?" >> $O/summary.txt
    f=$(find $O/kt_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/for_profiles/${TAG}_kt_bench_kernel_stats.csv
    cp $O/kt_bench.json $O/for_profiles/${TAG}_kt_bench.json; cp $O/unprofiled_bench.json $O/for_profiles/${TAG}_unprofiled_bench.json
    head -6 $O/for_profiles/${TAG}_kt_bench_kernel_stats.csv | cut -c1-150 ;;
esac
cd $R; cat $O/summary.txt
