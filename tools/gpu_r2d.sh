#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/r2d_pytest.log; cat $O/r2d_pytest.log
timeout 100 python tools/debug_temp.py 2>&1 | grep -v "amdgpu.ids\|Export\|initialized" | tee $O/r2d_debug.log
bash tools/gpu_ab.sh "--opt kern=1" "--opt kern=2" 2>&1 | tee $O/r2d_ab.log
: > $O/r2d_kst.log; : > $O/r2d_tprobe.log
for e in "kern=1" "kern=2"; do
  EXTRA="$e" bash tools/gpu_kstats2.sh 2>&1 | grep -v amdgpu.ids | head -3 | tee -a $O/r2d_kst.log
done
for e in "kern=2"; do
  echo "== timing $e" >> $O/r2d_tprobe.log
  TSL_LIB=$GRAFT_REPO_ROOT/taichislam_amd/lib/libtaichislam_hip_timing.so timeout 120 python tools/timing_probe_int.py $e 2>&1 | grep -v amdgpu.ids >> $O/r2d_tprobe.log
done
cat $O/r2d_tprobe.log
