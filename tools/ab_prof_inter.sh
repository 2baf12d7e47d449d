#!/bin/bash
# Dev aid: kernel averages of the inter probe (every kernel alone) for prebuilt variants.   tools/ab_prof_inter.sh "name1 name2"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
L=$ROOT/libvpx.opencl_amd/lib
cp "$L/libvp8hip.so" "$L/var/.keep.so"
cd /tmp; export TMPDIR=/tmp
for v in $1; do
    cp "$L/var/libvp8hip_$v.so" "$L/libvp8hip.so"
    rm -rf /tmp/abp_$v
    VP8HIP_DETILE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp_$v -o ip -- python3 $ROOT/tools/inter_probe.py 4096 > /dev/null 2>&1
    echo "== $v"; find /tmp/abp_$v -name "*kernel_stats.csv" | head -1 | xargs cut -d, -f1-4 | grep -E "inter_pred|interframe|detile"
done
cp "$L/var/.keep.so" "$L/libvp8hip.so"
