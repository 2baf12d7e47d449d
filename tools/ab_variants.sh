#!/bin/bash
# Dev aid: time several prebuilt variants of libvp8hip.so (libvpx.opencl_amd/lib/var/libvp8hip_<name>.so) alternately on the
# same GPU box, so that box-to-box variance cancels.   tools/ab_variants.sh "name1 name2 ..." [rounds] [bench args]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
L=$ROOT/libvpx.opencl_amd/lib
NAMES=$1; ROUNDS=${2:-2}; shift; shift
cp "$L/libvp8hip.so" "$L/var/.keep.so"
for round in $(seq $ROUNDS); do
    for v in $NAMES; do
        cp "$L/var/libvp8hip_$v.so" "$L/libvp8hip.so"
        echo "== $v"
        python "$ROOT/bench.py" --no-cpu-baseline --no-inter-probe --no-end-to-end --no-4k-probe "$@" 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}'
    done
done
cp "$L/var/.keep.so" "$L/libvp8hip.so"
