#!/bin/bash
# round 3, call B: uncontended per-kernel times for the occupancy variants + steady-state trace of the default pipeline + merge tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3b; mkdir -p $O
for o in "" "spt=2" "wg=256" "spt=2 chunks=4" "split=4" "split=1"; do timeout 200 python tools/phase_probe.py $o 2>/dev/null | tail -1; done | tee $O/probe.log
bash tools/gpu_check2.sh 2>&1 | tee $O/steady.log
timeout 600 python -m pytest tests/test_merge_gpu.py tests/test_fusion_mesh_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.log
