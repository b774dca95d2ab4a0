#!/bin/bash
# round-2 run A: full GPU test suite, A/B of the integrate kernels, uncontended kernel times, per-part timeline
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/r2a_pytest.log; cat $O/r2a_pytest.log
bash tools/gpu_ab.sh "--opt kern=0" "--opt kern=1" "--opt kern=1 --opt psegs=1536" 2>&1 | tee $O/r2a_ab.log
for e in "kern=0" "kern=1" "kern=1 psegs=1536"; do
  EXTRA="$e" bash tools/gpu_kstats2.sh 2>&1 | grep -v amdgpu.ids | tee -a $O/r2a_kst.log
done
for e in "kern=1" "kern=1 psegs=1536"; do
  echo "== timing $e" >> $O/r2a_tprobe.log
  TSL_LIB=$GRAFT_REPO_ROOT/taichislam_amd/lib/libtaichislam_hip_timing.so timeout 120 python tools/timing_probe_int.py $e 2>&1 | grep -v amdgpu.ids >> $O/r2a_tprobe.log
done
cat $O/r2a_tprobe.log
