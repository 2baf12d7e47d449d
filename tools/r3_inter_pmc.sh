#!/bin/bash
# counter passes over the inter-frame probe (vp8_inter_pred_kernel + vp8_interframe_kernel).  usage: tools/r3_inter_pmc.sh <tag> [jobs]
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r3ipmc}; O=gpurun_out/$TAG; mkdir -p $O
N=${2:-1024}
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {
    local name=$1; shift
    timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/$O/$name -- python3 $R/tools/inter_probe.py $N > $R/$O/$name.log 2>&1
    echo "$name rc=$?" >> $R/$O/summary.txt
    python3 $R/tools/pmc_summary.py $R/$O/$name $((8160 * N)) > $R/$O/$name.txt 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr
cd $R; cat $O/summary.txt; for f in sq1 sq2 fetch write tcp; do grep -A12 "inter_pred\|interframe" $O/$f.txt; done
