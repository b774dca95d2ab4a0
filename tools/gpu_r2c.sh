#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_tsdf_parity_gpu.py -m gpu -x -q 2>&1 | tail -15 > $O/r2c_pytest.log; cat $O/r2c_pytest.log
: > $O/r2c_kst.log; : > $O/r2c_tprobe.log
for e in "wg=512" "wg=768"; do
  EXTRA="$e" bash tools/gpu_kstats2.sh 2>&1 | grep -v amdgpu.ids | head -3 | tee -a $O/r2c_kst.log
done
for e in "wg=512" "wg=768"; do
  echo "== timing $e" >> $O/r2c_tprobe.log
  TSL_LIB=$GRAFT_REPO_ROOT/taichislam_amd/lib/libtaichislam_hip_timing.so timeout 120 python tools/timing_probe_int.py $e 2>&1 | grep -v amdgpu.ids >> $O/r2c_tprobe.log
done
cat $O/r2c_tprobe.log
