#!/bin/bash
# Dev aid: kernels AND memory copies of bin/batch_md5 --device-entropy --no-download on one timeline (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ent_trace2; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/run -- $R/libvpx.opencl_amd/bin/batch_md5 --device-entropy --no-download --batch 8192 --entropy-batch ${1:-24576} --loop ${2:-12288} $R/tests/golden/kf_1920x1080.ivf /tmp/o.md5 > $O/log.txt 2>&1
python3 - $O/run > $O/timeline.txt <<'PY'
import csv, sys, glob
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:28], ""))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?"))))
ev.sort()
t0 = ev[0][0]
for s, e, n, x in ev:
    if (e - s) > 300000: print("%-34s %9.1f %9.1f %8.1f ms %s" % (n, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, x))
PY
rm -rf $O/run
grep "frames in" $O/log.txt; tail -60 $O/timeline.txt
