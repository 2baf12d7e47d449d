#!/bin/bash
# Dev aid (GPU box): instruction-cache counters of the key-frame kernel (gpurun -- 'bash tools/icache_probe.sh tag [libvariant]').
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-ic}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
R=$GRAFT_REPO_ROOT
[ -n "$2" ] && export VP8HIP_LIB=$2
cd /tmp; export TMPDIR=/tmp
for n in 8192 4096; do
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQ_IFETCH_LEVEL"; do
    name=ic_${n}_$(echo $set | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$name -- python3 $R/tools/pmc_one.py 7 $n kf_1920x1080 shared > $O/$name.log 2>&1
    echo "$name rc=$?"
    python3 $R/tools/pmc_summary.py $O/$name $((8160 * n)) 2>&1 | grep -A12 "vp8_keyframe_kernel"
  done
done
