#!/bin/bash
# The profile set of a round (run on the GPU box: gpurun -- 'bash tools/profile_round.sh r04_a'), written to gpurun_out/<tag>/ with
# the summaries that are to be judged copied to gpurun_out/<tag>/for_profiles/ (copy those into profiles/ and commit them):
#   1. kernel-trace stats of the bench command (short form) and of bin/batch_md5 --device-entropy --entropy-batch 24576 (and of
#      bin/batch_md5 --streams 4096)
#   2. TCC traffic passes (FETCH_SIZE, WRITE_SIZE: separate runs) of ONE launch at the benchmark's occupancy (tools/pmc_one.py),
#      of the launch followed by the raster form of its frames, and of the inter-frame launch; -> traffic_per_mb.json, which
#      bench.py reads (profiles/traffic_per_mb.json)
#   3. SQ passes (instructions per macroblock, wait shares)
# PMC passes are never combined with trace domains other than --kernel-trace.
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r04_a}; NF=${2:-4096}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O/for_profiles
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
B="--steps 10 --warmup 2 --no-inter-probe --no-4k-probe --no-end-to-end --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_bench -- python3 $R/bench.py $B > $O/kt_bench.json 2> $O/kt_bench.err; echo "kt_bench rc=$?" >> $O/summary.txt
timeout 600 python3 $R/bench.py $B > $O/unprofiled_bench.json 2> $O/unprofiled.err; echo "unprofiled rc=$?" >> $O/summary.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_batch_md5 -- $R/libvpx.opencl_amd/bin/batch_md5 --device-entropy --no-download --batch 8192 --entropy-batch 24576 --loop 12288 $R/tests/golden/kf_1920x1080.ivf /tmp/o.md5 > $O/kt_batch_md5.log 2>&1; echo "kt_batch_md5 rc=$?" >> $O/summary.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_inter -- python3 $R/bench.py --steps 2 --warmup 1 --no-4k-probe --no-end-to-end --no-cpu-baseline > $O/kt_inter.json 2> $O/kt_inter.err; echo "kt_inter rc=$?" >> $O/summary.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_streams -- $R/libvpx.opencl_amd/bin/batch_md5 --streams 4096 $R/tests/golden/p_1920x1080.ivf /tmp/o.md5 > $O/kt_streams.log 2>&1; echo "kt_streams rc=$?" >> $O/summary.txt
for n in kt_bench kt_batch_md5 kt_inter kt_streams; do f=$(find $O/$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/for_profiles/${TAG}_${n}_kernel_stats.csv; done
cp $O/kt_bench.json $O/for_profiles/${TAG}_kt_bench.json; cp $O/unprofiled_bench.json $O/for_profiles/${TAG}_unprofiled_bench.json
grep -v "rocprofv3\|^[EWI]2" $O/kt_batch_md5.log > $O/for_profiles/${TAG}_kt_batch_md5.log; grep -v "rocprofv3\|^[EWI]2" $O/kt_streams.log > $O/for_profiles/${TAG}_kt_streams.log; cp $O/kt_inter.json $O/for_profiles/${TAG}_kt_inter.json
pmc() {  # name frames lgg extra-args counters...
    local name=$1 nf=$2 lgg=$3 extra=$4; shift; shift; shift; shift
    VP8HIP_SIMT_LGG=$lgg timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python3 $R/tools/pmc_one.py 7 $nf kf_1920x1080 $extra > $O/$name.log 2>&1
    echo "$name rc=$?" >> $O/summary.txt
    python3 $R/tools/pmc_summary.py $O/$name $((8160 * nf)) > $O/for_profiles/${TAG}_pmc_$name.summary.txt 2>&1
}
# (lanes per strand as bench.py's launch of 8192 frames has them: 8; "" "raster": the launch, then the raster form of every frame)
pmc fetch_${NF}_G8 $NF 3 "" FETCH_SIZE
pmc write_${NF}_G8 $NF 3 "" WRITE_SIZE
pmc fetch_raster_${NF}_G8 $NF 3 "'' raster" FETCH_SIZE
pmc write_raster_${NF}_G8 $NF 3 "'' raster" WRITE_SIZE
pmc sq1_${NF}_G8 $NF 3 "" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pmc sq2_${NF}_G8 $NF 3 "" SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
python3 $R/tools/pmc_traffic.py $O/for_profiles $TAG $NF 8 > $O/for_profiles/traffic_per_mb.json 2> $O/pmc_traffic.err
cd $R; cat $O/summary.txt
for f in $O/for_profiles/*.summary.txt; do echo "=== $f"; cat $f; done 2>/dev/null | grep -v "^$" | head -150
for f in $O/for_profiles/*kernel_stats.csv; do echo "== $f"; head -12 $f; done
cat $O/for_profiles/traffic_per_mb.json
python3 - <<PY
import json
for n in ("kt_bench", "unprofiled_bench"):
    try:
        d = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["config"]["kernel_ms"], d["roofline"]["pipeline"]["frac"], d["roofline"]["frac"], d["config"].get("with_raster_form"), d["config"].get("consumers"))
    except Exception as e:
        print(n, "ERR", e)
PY
