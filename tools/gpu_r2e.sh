#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/r2e_pytest.log; cat $O/r2e_pytest.log
bash tools/gpu_ab.sh "--opt chunks=1" "--opt chunks=2" "--opt chunks=4" "--opt wg=512 --opt chunks=1" "--opt wg=512 --opt chunks=2" 2>&1 | tee $O/r2e_ab.log
: > $O/r2e_kst.log; : > $O/r2e_tprobe.log
for e in "chunks=1" "chunks=2" "chunks=4" "wg=512 chunks=1" "wg=512 chunks=2"; do
  EXTRA="$e" bash tools/gpu_kstats2.sh 2>&1 | grep -v amdgpu.ids | head -3 | tee -a $O/r2e_kst.log
done
for e in "chunks=2" "wg=512 chunks=1"; do
  echo "== timing $e" >> $O/r2e_tprobe.log
  TSL_LIB=$GRAFT_REPO_ROOT/taichislam_amd/lib/libtaichislam_hip_timing.so timeout 120 python tools/timing_probe_int.py $e 2>&1 | grep -v amdgpu.ids >> $O/r2e_tprobe.log
done
cat $O/r2e_tprobe.log
