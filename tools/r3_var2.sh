#!/bin/bash
# variants x (re)allocations: tools/realloc_kf.py per variant library, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r3v}; mkdir -p $O
L=libvpx.opencl_amd/lib
cp $L/libvp8hip.so /tmp/keep.so
for round in 1 2; do
  for v in base $2; do
    if [ $v = base ]; then cp /tmp/keep.so $L/libvp8hip.so; else cp $L/var/libvp8hip_$v.so $L/libvp8hip.so; fi
    echo "== $v" | tee -a $O/times.txt
    timeout 300 python3 tools/realloc_kf.py 8192 ${3:-5} 2>&1 | cut -d" " -f3- | tee -a $O/times.txt
  done
done
cp /tmp/keep.so $L/libvp8hip.so
