"""Dev aid: per-kernel sums of a rocprofv3 --pmc ... --kernel-trace --output-format csv run directory.
   python3 tools/pmc_summary.py <dir> [macroblocks for per-MB figures]"""
import csv, collections, glob, sys
d = sys.argv[1]; nmb = float(sys.argv[2]) if len(sys.argv) > 2 else 0
cc = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
kt = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); meta = {}
for f in cc:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "vp8" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        meta[k] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Grid_Size"], r["Workgroup_Size"])
dur = collections.defaultdict(list)
for f in kt:
    for r in csv.DictReader(open(f)):
        if "vp8" in r["Kernel_Name"]: dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k in agg:
    ds = dur.get(k, [0]); n = len(ds)
    print(f"{k}: {n} dispatch(es), mean {sum(ds)/n:.3f} ms; vgpr/sgpr/lds/grid/wg = {meta[k]}")
    for cn, v in sorted(agg[k].items()):
        extra = ""
        if nmb and cn in ("FETCH_SIZE", "WRITE_SIZE"):
            b = v * 1024 / n / nmb
            extra = f"   = {b:.1f} B/MB" + (f" (x2 = {2*b:.1f} for wide loads)" if cn == "FETCH_SIZE" else "")
        print(f"    {cn:24s} {v/n:16.0f} per dispatch{extra}")
