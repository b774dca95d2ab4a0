#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3f; mkdir -p $O
timeout 600 python -m pytest tests/test_tsdf_parity_gpu.py tests/test_texture_gpu.py -x -q -m gpu --timeout=240 > $O/pytest.log 2>&1; rc=$?; echo "pytest rc $rc" >> $O/pytest.log; tail -3 $O/pytest.log
if [ $rc -ne 0 ]; then grep -v "^$" $O/pytest.log | grep -B2 -A25 "Error\|assert" | head -60; exit 1; fi
for o in "spt=2 unit=8192" "spt=2 unit=8192 chunks=4" "spt=2 unit=6144" "spt=2 unit=4096"; do bash tools/gpu_steady.sh "$o" 2>&1 | tail -10; done
