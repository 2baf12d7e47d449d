#!/bin/bash
# Dev aid: does it matter which NUMA node the page-locked download buffers are on?  (run on the GPU box)
R=$GRAFT_REPO_ROOT
lscpu | grep -i "numa\|^CPU(s)\|Model name" ; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
for d in /sys/class/drm/card*/device; do [ -f $d/numa_node ] && echo "$d numa_node=$(cat $d/numa_node) local_cpulist=$(cat $d/local_cpulist) $(cat $d/vendor)"; done
taskset -p $$
T="$R/libvpx.opencl_amd/bin/batch_md5 --device-entropy --batch 4096 --entropy-batch 24576 --loop 6144 $R/tests/golden/kf_1920x1080.ivf /tmp/o.md5"
echo "--- as started"; $T 2>&1 | tail -1
for d in /sys/class/drm/card*/device; do
  if [ -f $d/numa_node ] && [ "$(cat $d/vendor)" == "0x1002" ]; then
    L=$(cat $d/local_cpulist); echo "--- taskset -c $L"; taskset -c $L $T 2>&1 | tail -1; break
  fi
done
N=$(ls -d /sys/devices/system/node/node* | wc -l); echo "nodes: $N"
for n in /sys/devices/system/node/node*; do L=$(cat $n/cpulist); echo "--- $(basename $n): taskset -c $L"; taskset -c $L $T 2>&1 | tail -1; done
