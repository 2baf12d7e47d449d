#!/bin/bash
# Instruction mix of vp8_entropy_kernel (640 1080p key frames, 10 waves).  usage: tools/r3_pmc_entropy.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3pmcent}; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
pmc() {
    local name=$1; shift
    timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python3 $R/tools/entropy_probe.py 640 kf_1920x1080 64 > $O/$name.log 2>&1
    echo "$name rc=$?" >> $O/summary.txt
}
pmc insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
pmc cycles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_BRANCH
pmc waits SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
cd $R; cat $O/summary.txt
python3 - $O <<'PY'
import csv, glob, sys, collections
o = sys.argv[1]
for name in ("insts", "cycles", "waits"):
    tot = collections.defaultdict(float); n = 0
    for f in glob.glob(f"{o}/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "entropy" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
    disp = len(set())
    print(name, {k: v for k, v in tot.items()})
PY
