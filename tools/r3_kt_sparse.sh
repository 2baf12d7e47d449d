#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/ktsparse; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $R/libvpx.opencl_amd/bin/batch_md5 --device-entropy --no-download --batch 8192 --entropy-batch 24576 --loop 13108 $R/tests/golden/kf_1920x1080.ivf /tmp/ent.md5 > $O/kt.log 2>&1
tail -1 $O/kt.log
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs cut -d, -f1-4 | head -8
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "entropy" in r["Kernel_Name"] or "expand" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows[:40]:
    print(r["Kernel_Name"][:28], "start %.3f s" % ((int(r["Start_Timestamp"]) - t0) / 1e9), "dur %.1f ms" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6), "grid", r.get("Grid_Size_X", r.get("Grid_Size")))
PY
