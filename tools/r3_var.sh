#!/bin/bash
# time variant builds (lib/var/libvp8hip_<name>.so) of the key-frame kernels alternately on one box; then stamps
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r3v}; mkdir -p $O
L=libvpx.opencl_amd/lib
cp $L/libvp8hip.so /tmp/keep.so
for round in 1 2; do
  for v in base $2; do
    if [ $v = base ]; then cp /tmp/keep.so $L/libvp8hip.so; else cp $L/var/libvp8hip_$v.so $L/libvp8hip.so; fi
    KF_NAME=$v timeout 300 python3 tools/time_kf.py 8192 4 2>&1 | tail -1 | tee -a $O/times.txt
  done
done
if [ -f $L/var/libvp8hip_stamps.so ]; then
cp $L/var/libvp8hip_stamps.so $L/libvp8hip.so
timeout 300 python3 tools/stamps_kf.py 8192 > $O/stamps_8192.txt 2>&1
cat $O/stamps_8192.txt
fi
cp /tmp/keep.so $L/libvp8hip.so
