#!/bin/bash
# SQ counter passes over the fused key-frame launch.  usage: tools/r3_pmc2.sh <tag> <frames> [LGG]
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3pmc}; O=gpurun_out/$TAG; mkdir -p $O
NF=${2:-8192}
FX=${4:-kf_1920x1080}
MBS=${5:-8160}
SH=${6:-}

cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {
    local name=$1; shift
    timeout 100 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$O/$name -- python3 $R/tools/pmc_one.py 7 $NF "$FX" $SH > $R/$O/$name.log 2>&1
    echo "$name rc=$?" >> $R/$O/summary.txt
    python3 $R/tools/pmc_summary.py $R/$O/$name $((MBS * NF)) > $R/$O/$name.txt 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
run ic SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R; cat $O/summary.txt; for f in sq1 sq2 ic fetch write; do cat $O/$f.txt | grep -A9 keyframe; done
