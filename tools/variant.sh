#!/bin/bash
# Dev aid: build a VARIANT of the HIP library with extra compiler flags into lib/var/libvp8hip_<name>.so and run a command with it
# (VP8HIP_LIB, read by libvpx.opencl_amd/__init__.py: the product library lib/libvp8hip.so is never touched).
#   tools/variant.sh stamps -DVP8_STAMPS -- python3 tools/kf_diag.py waves key      build + run
#   tools/variant.sh stamps -DVP8_STAMPS                                            build only (e.g. before a gpurun)
#   VP8HIP_LIB=libvpx.opencl_amd/lib/var/libvp8hip_stamps.so python3 tools/...      run one that was built before
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
flags=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do flags+=("$1"); shift; done
[ "$1" == "--" ] && shift
make -s -C "$ROOT/libvpx.opencl_amd/csrc" var NAME="$name" HIPFLAGS="${flags[*]}"
if [ $# -gt 0 ]; then VP8HIP_LIB="$ROOT/libvpx.opencl_amd/lib/var/libvp8hip_$name.so" exec "$@"; fi
