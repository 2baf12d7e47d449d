#!/bin/bash
# kernel trace of the inter-frame probe
cd /tmp && export TMPDIR=/tmp
N=${1:-4096}
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/inter_prof -o ip -- python3 $GRAFT_REPO_ROOT/tools/inter_probe.py $N > $GRAFT_REPO_ROOT/gpurun_out/inter_prof.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/inter_prof.log
find $GRAFT_REPO_ROOT/gpurun_out/inter_prof -name "*kernel_stats.csv" | head -1 | xargs cat | cut -d, -f1-8 | head -14
