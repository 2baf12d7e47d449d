#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r3sti}; mkdir -p $O
L=libvpx.opencl_amd/lib
cp $L/libvp8hip.so /tmp/keep.so; cp $L/var/libvp8hip_stamps.so $L/libvp8hip.so
timeout 300 python3 tools/stamps_inter.py ${2:-4096} > $O/stamps_inter.txt 2>&1
cp /tmp/keep.so $L/libvp8hip.so
cat $O/stamps_inter.txt
