#!/bin/bash
# TCC traffic of the key-frame launch (1024 frames, 8 lanes per strand as at the benchmark's launch size).  usage: tools/r3_pmc_kf.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3pmckf}; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
pmc() {
    local name=$1 nf=$2 lgg=$3; shift; shift; shift
    VP8HIP_SIMT_LGG=$lgg timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python3 $R/tools/pmc_one.py 7 $nf kf_1920x1080 > $O/$name.log 2>&1
    echo "$name rc=$?" >> $O/summary.txt
    python3 $R/tools/pmc_summary.py $O/$name $((8160 * nf)) > $O/$name.txt 2>&1
}
pmc fetch_1024_G8 1024 3 FETCH_SIZE
pmc write_1024_G8 1024 3 WRITE_SIZE
pmc req_1024_G8 1024 3 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
cd $R; cat $O/summary.txt; cat $O/fetch_1024_G8.txt $O/write_1024_G8.txt $O/req_1024_G8.txt | grep -A3 "keyframe\|detile_kf\|extend"
