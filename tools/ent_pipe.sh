#!/bin/bash
# Dev aid: bin/batch_md5 --device-entropy --no-download at several frames per entropy launch (run on the GPU box)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ent_pipe; mkdir -p $O; : > $O/summary.txt
for e in ${1:-16384 24576}; do
  timeout 900 $R/libvpx.opencl_amd/bin/batch_md5 --device-entropy --no-download --batch 8192 --entropy-batch $e --loop ${2:-19661} $R/tests/golden/kf_1920x1080.ivf /tmp/o.md5 2>> $O/summary.txt
  echo "E=$e rc=$?" >> $O/summary.txt
done
cat $O/summary.txt
