#!/bin/bash
# round 3, call E: sequential semantics parity + reference trace replay + steady-state timelines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3e; mkdir -p $O
timeout 600 python -m pytest tests/test_sequential_gpu.py tests/test_submap_trace.py tests/test_reference_callers.py -x -q -m gpu --timeout=240 > $O/pytest.log 2>&1; rc=$?; echo "pytest rc $rc" >> $O/pytest.log; tail -5 $O/pytest.log
if [ $rc -ne 0 ]; then grep -v "^$" $O/pytest.log | grep -B2 -A25 "Error\|assert" | head -80; fi
bash tools/gpu_steady.sh "spt=2 unit=8192" 2>&1 | tail -14
bash tools/gpu_steady.sh "" 2>&1 | tail -14
