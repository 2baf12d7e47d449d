#!/bin/bash
# Dev aid: the pipeline with every frame downloaded, by the number of copy streams (run on the GPU box)
R=$GRAFT_REPO_ROOT
for k in 1 2 3 4; do
  echo "VP8HIP_D2H_STREAMS=$k"
  VP8HIP_D2H_STREAMS=$k $R/libvpx.opencl_amd/bin/batch_md5 --device-entropy --batch 4096 --loop 4096 $R/tests/golden/kf_1920x1080.ivf /tmp/o.md5 2>&1 | tail -1
  VP8HIP_D2H_STREAMS=$k $R/libvpx.opencl_amd/bin/batch_md5 --device-entropy --batch 4096 --entropy-batch 24576 --loop 12288 $R/tests/golden/kf_1920x1080.ivf /tmp/o.md5 2>&1 | tail -1
done
