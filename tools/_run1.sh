cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5e; mkdir -p $O
timeout 100 python3 tools/kf_time.py 16384 6 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH"; do
  name=sq_$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$name -- python3 $R/tools/pmc_one.py 7 8192 kf_1920x1080 shared > $O/$name.log 2>&1
  echo "$name rc=$?"
  python3 $R/tools/pmc_summary.py $O/$name $((8160 * 8192)) 2>&1 | grep -A12 "vp8_keyframe_kernel"
done
cd $R; export VP8HIP_LIB=$R/libvpx.opencl_amd/lib/var/libvp8hip_stamps.so; timeout 250 python3 tools/kf_diag.py stamps key 8192 2>&1 | tail -28; timeout 250 python3 tools/kf_diag.py waves key 8192 2>&1 | tail -2
