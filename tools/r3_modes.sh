#!/bin/bash
# how often does a process land in the slow mode?  N runs of tools/time_kf.py per setting
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r3m}; mkdir -p $O
N=${2:-5}
run() { for i in $(seq $N); do env "$@" KF_NAME="$*" timeout 200 python3 tools/time_kf.py 8192 3 2>&1 | tail -1 | tee -a $O/modes.txt; done; }
run A=0
run VP8HIP_FB_PAD=4352 VP8HIP_SLOT_PAD=4352
run VP8HIP_FB_PAD=768 VP8HIP_SLOT_PAD=256
run GPU_MAX_HW_QUEUES=1
