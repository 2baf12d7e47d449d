#!/bin/bash
# SQ / instruction-cache counter passes over one 1024-frame launch at full occupancy (G = 64: 1024 waves), fused kernel and
# two-kernel pipeline.  usage: tools/r3_pmc.sh <outdir-tag>
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3pmc}; O=gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export VP8HIP_SIMT_LGG=${LGG:-6}
run() {  # name, fused, counters...
    local name=$1 fused=$2; shift; shift
    VP8HIP_FUSED=$fused timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$O/$name -- python3 $R/tools/pmc_one.py 7 ${NF:-1024} > $R/$O/$name.log 2>&1
    echo "$name rc=$?" >> $R/$O/summary.txt
    python3 $R/tools/pmc_summary.py $R/$O/$name 8355840 > $R/$O/$name.txt 2>&1
}
for f in 1 0; do
  run sq1_f$f $f SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
  run sq2_f$f $f SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
  run ic_f$f $f SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
  run fetch_f$f $f FETCH_SIZE
  run write_f$f $f WRITE_SIZE
done
cd $R; cat $O/summary.txt; for f in $O/*.txt; do echo "=== $f"; cat $f; done 2>/dev/null | grep -v "^$" | head -150
