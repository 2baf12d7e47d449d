#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3pmcent1; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
VP8HIP_ENTROPY_LANES=1 timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/insts -- python3 $R/tools/entropy_probe.py 10 kf_1920x1080 1 > $O/insts.log 2>&1
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
tot = collections.defaultdict(float)
for f in glob.glob(f"{o}/insts/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "entropy" in r["Kernel_Name"]: tot[r["Counter_Name"]] += float(r["Counter_Value"])
print(dict(tot))
PY
