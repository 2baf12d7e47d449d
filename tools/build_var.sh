#!/bin/bash
# Dev aid: build the product library and a variant of it with extra compiler flags into lib/var/libvp8hip_<name>.so
#   tools/build_var.sh                       -> make all
#   tools/build_var.sh stamps -DVP8_STAMPS   -> make all + lib/var/libvp8hip_stamps.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/libvpx.opencl_amd/csrc"
make all 2>&1 | grep -E "error|Error" || true
if [ -n "$1" ]; then
    name=$1; shift
    mkdir -p ../lib/var
    SRC="hip/vp8hip.hip hip/vp8_recon.hip hip/vp8_recon_simt.hip hip/vp8_keyframe_simt.hip hip/vp8_inter_pred.hip hip/vp8_loopfilter.hip hip/vp8_loopfilter_simt.hip hip/vp8_detile.hip hip/vp8_rtcd_blocks.hip hip/vp8_lane_blocks.hip hip/vp8_postproc.hip hip/vp8_md5.hip hip/vp8_entropy.hip"
    hipcc --offload-arch=gfx950 -O3 -fPIC -shared -fgpu-rdc -I../../include -Ihip "$@" -o ../lib/var/libvp8hip_$name.so $SRC 2>&1 | grep -E "error" || true
    ls -la ../lib/var/libvp8hip_$name.so
fi
