#!/bin/bash
# round-2 run B: A/B of workgroup sizes of the v2 brick kernel
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/r2b_pytest.log; cat $O/r2b_pytest.log
bash tools/gpu_ab.sh "--opt wg=256" "--opt wg=512" 2>&1 | tee $O/r2b_ab.log
: > $O/r2b_kst.log; : > $O/r2b_tprobe.log
for e in "wg=256" "wg=512"; do
  EXTRA="$e" bash tools/gpu_kstats2.sh 2>&1 | grep -v amdgpu.ids | head -4 | tee -a $O/r2b_kst.log
done
for e in "wg=256" "wg=512"; do
  echo "== timing $e" >> $O/r2b_tprobe.log
  TSL_LIB=$GRAFT_REPO_ROOT/taichislam_amd/lib/libtaichislam_hip_timing.so timeout 120 python tools/timing_probe_int.py $e 2>&1 | grep -v amdgpu.ids >> $O/r2b_tprobe.log
done
cat $O/r2b_tprobe.log
