#!/bin/bash
# Dev aid: the pipeline with every frame downloaded -- copy streams, the direct download kernel, both at once (run on the GPU box)
R=$GRAFT_REPO_ROOT
run() { echo "$*"; env "$@" $R/libvpx.opencl_amd/bin/batch_md5 --device-entropy --batch ${B:-4096} --entropy-batch 24576 --loop ${LOOPS:-12288} $R/tests/golden/kf_1920x1080.ivf /tmp/o.md5 2>&1 | tail -1; }
export B=8192
run VP8HIP_D2H_STREAMS=2
run VP8HIP_D2H_STREAMS=1
run VP8HIP_D2H_STREAMS=2
run VP8HIP_D2H_STREAMS=3
run VP8HIP_D2H_STREAMS=4

LOOPS=24576 run VP8HIP_D2H_STREAMS=2
