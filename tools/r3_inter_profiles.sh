#!/bin/bash
# round-3 profile set for launches with inter frames: kernel-trace stats of the inter-frame probe (default: the tiled -> raster pass
# beside the next launch; and with the pass on the main stream: every kernel alone), counter passes at 1024 jobs
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3iprof}; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_default -o ip -- python3 $R/tools/inter_probe.py 4096 > $O/kt_default.json 2> $O/kt_default.err
VP8HIP_DETILE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial -o ip -- python3 $R/tools/inter_probe.py 4096 > $O/kt_serial.json 2> $O/kt_serial.err
timeout 300 python3 $R/tools/inter_probe.py 4096 > $O/unprofiled.json 2> $O/unprofiled.err
cd $R
bash tools/r3_inter_pmc.sh ${1:-r3iprof}/pmc 1024 > $O/pmc_summary.txt 2>&1
for f in kt_default kt_serial; do echo "== $f"; grep -o '"ms_per_launch": [0-9.]*' $O/$f.json | head -2; cut -d, -f1-4 $O/$f/ip_kernel_stats.csv | head -6; done
grep -o '"ms_per_launch": [0-9.]*' $O/unprofiled.json
cat $O/pmc_summary.txt | tail -60
