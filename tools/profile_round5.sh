#!/bin/bash
# Round 5's profile set (run on the GPU box: gpurun -- 'bash tools/profile_round5.sh r05_a'), on top of what tools/profile_round.sh
# collects for the key-frame kernel: the launches with inter frames -- references read as raster frames and as tiles, chained -- and
# the MD5 kernel, each with kernel-trace stats and TCC / SQ counter passes (separate runs, --kernel-trace only beside --pmc).
#   tools/inter_chain_time.py <jobs>: prediction from raster references, then chained launches with the references read as tiles
#   (vp8_inter_pred_tiles_kernel), through their raster form (vp8_detile_kf_kernel + vp8_extend_kernel + vp8_inter_pred_kernel), as tiles
#   tools/md5_time.py <frames ...>: vp8_md5_tiles_kernel over batches a large launch left
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r05_a}; NJ=${2:-4096}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O/for_profiles
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_chain -- python3 $R/tools/inter_chain_time.py $NJ 3 > $O/kt_chain.log 2>&1; echo "kt_chain rc=$?" >> $O/summary.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_chain8k -- python3 $R/tools/inter_chain_time.py 8192 3 > $O/kt_chain8k.log 2>&1; echo "kt_chain8k rc=$?" >> $O/summary.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_md5 -- python3 $R/tools/md5_time.py 16384 8192 4096 > $O/kt_md5.log 2>&1; echo "kt_md5 rc=$?" >> $O/summary.txt
for n in kt_chain kt_chain8k kt_md5; do f=$(find $O/$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/for_profiles/${TAG}_${n}_kernel_stats.csv; grep -v "rocprofv3\|^[EWI]2" $O/$n.log > $O/for_profiles/${TAG}_${n}.log; done
pmc() {  # name macroblocks-per-dispatch command... -- counters...
    local name=$1 nmb=$2; shift; shift
    local cmd=(); while [ "$1" != "--" ]; do cmd+=("$1"); shift; done; shift
    timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python3 "${cmd[@]}" > $O/$name.log 2>&1
    echo "$name rc=$?" >> $O/summary.txt
    python3 $R/tools/pmc_summary.py $O/$name $nmb > $O/for_profiles/${TAG}_pmc_$name.summary.txt 2>&1
}
pmc inter_fetch_$NJ $((8160 * NJ)) $R/tools/inter_chain_time.py $NJ 2 -- FETCH_SIZE
pmc inter_write_$NJ $((8160 * NJ)) $R/tools/inter_chain_time.py $NJ 2 -- WRITE_SIZE
pmc inter_sq1_$NJ $((8160 * NJ)) $R/tools/inter_chain_time.py $NJ 2 -- SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pmc inter_sq2_$NJ $((8160 * NJ)) $R/tools/inter_chain_time.py $NJ 2 -- SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
pmc inter_ta_$NJ $((8160 * NJ)) $R/tools/inter_chain_time.py $NJ 2 -- TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pmc md5_fetch_16384 $((8160 * 16384)) $R/tools/md5_time.py 16384 -- FETCH_SIZE
pmc md5_fetch_8192 $((8160 * 8192)) $R/tools/md5_time.py 8192 -- FETCH_SIZE
# the key-frame kernel at the benchmark's occupancy (two waves per SIMD: 8192 frames), one copy per slot in the setup
pmc kf_sq1_8192 $((8160 * 8192)) $R/tools/pmc_one.py 7 8192 kf_1920x1080 -- SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pmc kf_fetch_8192 $((8160 * 8192)) $R/tools/pmc_one.py 7 8192 kf_1920x1080 -- FETCH_SIZE
pmc kf_write_8192 $((8160 * 8192)) $R/tools/pmc_one.py 7 8192 kf_1920x1080 -- WRITE_SIZE
cd $R; cat $O/summary.txt
for f in $O/for_profiles/*.summary.txt; do echo "=== $f"; cat $f; done 2>/dev/null | grep -v "^$" | head -220
for f in $O/for_profiles/*kernel_stats.csv; do echo "== $f"; head -9 $f | cut -c1-150; done
for f in $O/for_profiles/*.log; do echo "== $f"; cat $f; done
