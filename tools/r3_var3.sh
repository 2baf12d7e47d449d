#!/bin/bash
# bench (kernel-only) of variant libraries, alternating: tools/r3_var3.sh <tag> "<names>" [rounds]
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r3v}; mkdir -p $O
L=libvpx.opencl_amd/lib
cp $L/libvp8hip.so /tmp/keep.so
B="--steps 8 --warmup 2 --no-inter-probe --no-4k-probe --no-end-to-end --no-cpu-baseline"
for round in $(seq ${3:-2}); do
  for v in base $2; do
    if [ $v = base ]; then cp /tmp/keep.so $L/libvp8hip.so; else cp $L/var/libvp8hip_$v.so $L/libvp8hip.so; fi
    python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['config']['kernel_ms'])" | tee -a $O/times.txt
  done
done
cp /tmp/keep.so $L/libvp8hip.so
