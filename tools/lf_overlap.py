"""Dev aid: from a rocprofv3 --kernel-trace CSV, start / end of the two loop-filter kernels of every launch relative to the
luma kernel's start (ms): do they overlap, and how long does each take?   python3 tools/lf_overlap.py <dir>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "loopfilter_simt" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "luma" if "luma" in r["Kernel_Name"] else "chroma"))
rows.sort()
luma = [r for r in rows if r[2] == "luma"]; chroma = [r for r in rows if r[2] == "chroma"]
for l, c in list(zip(luma, chroma))[-4:]:
    print(f"luma 0.00 .. {(l[1]-l[0])/1e6:6.2f}   chroma {(c[0]-l[0])/1e6:6.2f} .. {(c[1]-l[0])/1e6:6.2f}   pair {(max(l[1],c[1])-min(l[0],c[0]))/1e6:6.2f} ms")
