#!/bin/bash
# Dev aid: tools/entropy_probe.py (no verification) over prebuilt variants, alternately.   tools/ab_entropy.sh "name1 name2" [frames] [lanes]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
L=$ROOT/libvpx.opencl_amd/lib
NAMES=$1; N=${2:-640}; LANES=${3:-64}
cp "$L/libvp8hip.so" "$L/var/.keep.so"
for round in 1 2; do
    for v in $NAMES; do
        cp "$L/var/libvp8hip_$v.so" "$L/libvp8hip.so"
        echo -n "$v: "; python "$ROOT/tools/entropy_probe.py" $N kf_1920x1080 $LANES 2>&1 | tail -1
    done
done
cp "$L/var/.keep.so" "$L/libvp8hip.so"
