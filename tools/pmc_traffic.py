"""Dev aid: profiles/traffic_per_mb.json out of the summaries tools/profile_round.sh leaves (tools/pmc_summary.py's output of the
FETCH_SIZE and WRITE_SIZE passes): KiB per macroblock and kernel, as counted (FETCH_SIZE is doubled by the reader: the gfx950
correction for 16-byte-per-lane loads, MI355X guide).   python3 tools/pmc_traffic.py <dir> <tag> <frames> <lanes per strand>"""
import json, re, sys, os
d, tag, nf, g = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
kern = {}
def take(path, counter, key):
    name = None
    for ln in open(path):
        m = re.match(r"(vp8_\w+): (\d+) dispatch", ln)
        if m:
            name = m.group(1); continue
        m = re.match(r"\s+%s\s+([0-9.]+) per dispatch\s+=\s+([0-9.]+) B/MB" % counter, ln)
        if m and name:
            kern.setdefault(name, {})[key] = float(m.group(2)) / 1024.0
for variant in ("", "raster_"):
    f = os.path.join(d, f"{tag}_pmc_fetch_{variant}{nf}_G{g}.summary.txt"); w = os.path.join(d, f"{tag}_pmc_write_{variant}{nf}_G{g}.summary.txt")
    if os.path.exists(f): take(f, "FETCH_SIZE", "fetch_KiB_per_mb")
    if os.path.exists(w): take(w, "WRITE_SIZE", "write_KiB_per_mb")
kern = {k: v for k, v in kern.items() if "fetch_KiB_per_mb" in v and "write_KiB_per_mb" in v}
print(json.dumps({"source": f"profiles/{tag}_pmc_{{fetch,write}}[_raster]_{nf}_G{g}.summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                            f"over tools/pmc_one.py 7 {nf} kf_1920x1080)", "frames_per_launch": nf, "lanes_per_strand": g, "kernels": kern}, indent=1))
