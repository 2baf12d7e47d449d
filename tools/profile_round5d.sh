#!/bin/bash
# Round 5, fourth part of the profile set: the HBM traffic of the launches with inter frames (FETCH_SIZE / WRITE_SIZE, separate passes) at
# 1024 jobs a launch -- at 4096 the profiler's bookkeeping of the setup's frame copies ran the passes into their time limits twice
# (profile_round5.sh, profile_round5b.sh); bytes per macroblock do not depend on the launch size.
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r05_d}; NJ=${2:-1024}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O/for_profiles
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
pmc() {  # name macroblocks-per-dispatch command... -- counters...
    local name=$1 nmb=$2; shift; shift
    local cmd=(); while [ "$1" != "--" ]; do cmd+=("$1"); shift; done; shift
    timeout 500 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python3 "${cmd[@]}" > $O/$name.log 2>&1
    echo "$name rc=$?" >> $O/summary.txt
    python3 $R/tools/pmc_summary.py $O/$name $nmb > $O/for_profiles/${TAG}_pmc_$name.summary.txt 2>&1
}
pmc inter_fetch_$NJ $((8160 * NJ)) $R/tools/inter_chain_time.py $NJ 1 -- FETCH_SIZE
pmc inter_write_$NJ $((8160 * NJ)) $R/tools/inter_chain_time.py $NJ 1 -- WRITE_SIZE
cd $R; cat $O/summary.txt
for f in $O/for_profiles/*inter*.summary.txt; do echo "=== $f"; grep -A3 "vp8_inter_pred\|vp8_interframe\|vp8_detile_kf\|vp8_extend" $f | cut -c1-150 | head -40; done
