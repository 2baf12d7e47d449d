#!/bin/bash
# Dev aid: vp8_entropy_kernel's duration against the frames per launch (kernel trace; run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ent_scale; mkdir -p $O
for n in ${2:-8192 16384 32768 65536}; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/n$n -- python3 $R/tools/entropy_probe.py $n ${1:-kf_640x360} 16 32 64 > $O/n$n.log 2>&1
  f=$(find $O/n$n -name "*kernel_trace.csv" | head -1)
  python3 - "$f" $n >> $O/summary.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("vp8_entropy")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(sys.argv[2], [(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 1)) for r in rows])
PY
  rm -rf $O/n$n
done
cat $O/summary.txt
