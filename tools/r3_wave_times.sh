#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
L=libvpx.opencl_amd/lib
cp $L/libvp8hip.so /tmp/keep.so; cp $L/var/libvp8hip_stamps.so $L/libvp8hip.so
[ "$1" != "inter" ] && timeout 300 python3 tools/wave_times.py key 8192 2>&1 | tail -4
timeout 300 python3 tools/wave_times.py inter 4096 2>&1 | tail -4
timeout 300 python3 tools/stamps_inter.py 4096 2>&1 | tail -30
cp /tmp/keep.so $L/libvp8hip.so
