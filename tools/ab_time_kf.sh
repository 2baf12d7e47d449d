#!/bin/bash
# Dev aid: tools/time_kf.py (no verification) over prebuilt variants, alternately.   tools/ab_time_kf.sh "name1 name2" [rounds]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
L=$ROOT/libvpx.opencl_amd/lib
NAMES=$1; ROUNDS=${2:-2}
cp "$L/libvp8hip.so" "$L/var/.keep.so"
for round in $(seq $ROUNDS); do
    for v in $NAMES; do
        cp "$L/var/libvp8hip_$v.so" "$L/libvp8hip.so"
        KF_NAME=$v python "$ROOT/tools/time_kf.py" 8192 5 2>&1 | tail -1
    done
done
cp "$L/var/.keep.so" "$L/libvp8hip.so"
