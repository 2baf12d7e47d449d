#!/bin/bash
# Dev aid: like ab_variants.sh for the inter-frame probe.   tools/ab_inter.sh "name1 name2 ..." [rounds] [jobs]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
L=$ROOT/libvpx.opencl_amd/lib
NAMES=$1; ROUNDS=${2:-2}; N=${3:-4096}
cp "$L/libvp8hip.so" "$L/var/.keep.so"
for round in $(seq $ROUNDS); do
    for v in $NAMES; do
        cp "$L/var/libvp8hip_$v.so" "$L/libvp8hip.so"
        echo "== $v"
        python "$ROOT/tools/inter_probe.py" $N 2>&1 | grep -o '"ms_per_launch": [0-9.]*\|"kernel_ms": {[^}]*}\|"md5_ok": [a-z]*' | tr '\n' ' '; echo
    done
done
cp "$L/var/.keep.so" "$L/libvp8hip.so"
