#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r3st}; mkdir -p $O
L=libvpx.opencl_amd/lib
cp $L/libvp8hip.so /tmp/keep.so; cp $L/var/libvp8hip_stamps.so $L/libvp8hip.so
VP8HIP_SIMT_LGG=3 timeout 300 python3 tools/stamps_kf.py 1024 > $O/stamps_g8.txt 2>&1
timeout 300 python3 tools/stamps_kf.py 8192 > $O/stamps_8192.txt 2>&1
cp /tmp/keep.so $L/libvp8hip.so
cat $O/stamps_g8.txt $O/stamps_8192.txt
