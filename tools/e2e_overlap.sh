#!/bin/bash
# Dev aid (GPU): bin/batch_md5 --device-entropy --no-download as ONE pipeline of 24,576 frames in flight, and as 2 / 3 worker processes
# on the same device with a share of the frames in flight each (VP8BATCH_SINGLE_DEVICE=1: their entropy and pixel launches overlap)
cd "$GRAFT_REPO_ROOT" || exit 1
B=libvpx.opencl_amd/bin/batch_md5; F=tests/golden/kf_1920x1080.ivf; L=${1:-49152}
run() { echo "== $*"; "$@" 2>&1 | grep -E "frames in|Mpix|error|DIE|failed" | tail -3; }
run $B --device-entropy --no-download --batch 8192 --entropy-batch 24576 --loop $L $F /tmp/o1.md5
VP8BATCH_SINGLE_DEVICE=1 run $B --device-entropy --no-download --batch 4096 --entropy-batch 12288 --gpus 2 --loop $L $F /tmp/o2.md5
VP8BATCH_SINGLE_DEVICE=1 run $B --device-entropy --no-download --batch 4096 --entropy-batch 8192 --gpus 3 --loop $L $F /tmp/o3.md5
VP8BATCH_SINGLE_DEVICE=1 run $B --device-entropy --no-download --batch 8192 --entropy-batch 8192 --gpus 2 --loop $L $F /tmp/o4.md5
cmp /tmp/o1.md5 /tmp/o2.md5 && cmp /tmp/o1.md5 /tmp/o3.md5 && echo "listings equal"
