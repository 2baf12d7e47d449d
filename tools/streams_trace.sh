#!/bin/bash
# Dev aid: kernel timeline of bin/batch_md5 --streams S (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/streams_trace; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/run -- $R/libvpx.opencl_amd/bin/batch_md5 --streams ${1:-4096} $R/tests/golden/${2:-p_1920x1080}.ivf /tmp/o.md5 > $O/log.txt 2>&1
f=$(find $O/run -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s = (int(r["Start_Timestamp"]) - t0) / 1e6; e = (int(r["End_Timestamp"]) - t0) / 1e6
    if e - s > 0.5: print("%-28s %9.1f %9.1f %8.1f ms grid %s" % (r["Kernel_Name"][:28], s, e, e - s, r.get("Grid_Size_X", r.get("Grid_Size"))))
PY
rm -rf $O/run
grep "frames in" $O/log.txt; cat $O/timeline.txt
