#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
: > $O/r2f.log
for lib in nowalk nosplit; do
  for e in "chunks=1" "chunks=2"; do
    echo "== lib $lib $e" >> $O/r2f.log
    TSL_LIB=$GRAFT_REPO_ROOT/taichislam_amd/lib/libtaichislam_hip_$lib.so EXTRA="$e" bash tools/gpu_kstats2.sh 2>&1 | grep -v amdgpu.ids | head -3 >> $O/r2f.log
  done
done
for e in "chunks=1" "chunks=2"; do
  echo "== wg timeline $e" >> $O/r2f.log
  TSL_LIB=$GRAFT_REPO_ROOT/taichislam_amd/lib/libtaichislam_hip_timing.so timeout 120 python tools/timing_probe_wg.py $e 2>&1 | grep -v "amdgpu.ids\|initialized" >> $O/r2f.log
done
cat $O/r2f.log
