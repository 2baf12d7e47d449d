#!/bin/bash
# Dev aid: A/B the working-tree HIP library against the one built from git HEAD, on the same GPU box.
#   tools/ab_bench.sh build      (here, no GPU)   -> libvpx.opencl_amd/lib/var/libvp8hip_head.so
#   tools/ab_bench.sh run [bench args]   (on the GPU box, inside gpurun)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
L=$ROOT/libvpx.opencl_amd/lib
if [ "$1" = build ]; then
    T=$(mktemp -d /tmp/abhead.XXXX)
    git -C "$ROOT" archive HEAD libvpx.opencl_amd/csrc include | tar -x -C "$T"
    mkdir -p "$L/var"
    (cd "$T/libvpx.opencl_amd/csrc" && hipcc --offload-arch=gfx950 -O3 -fPIC -shared -fgpu-rdc -I../../include -Ihip \
        -o "$L/var/libvp8hip_head.so" hip/*.hip)
    rm -rf "$T"
    make -C "$ROOT/libvpx.opencl_amd/csrc" all > /dev/null
    cp "$L/libvp8hip.so" "$L/var/libvp8hip_work.so"
    exit 0
fi
shift || true
for round in 1 2; do
    for v in head work; do
        cp "$L/var/libvp8hip_$v.so" "$L/libvp8hip.so"
        echo "== $v"
        python "$ROOT/bench.py" --no-cpu-baseline --no-inter-probe --no-end-to-end "$@" 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}'
    done
done
