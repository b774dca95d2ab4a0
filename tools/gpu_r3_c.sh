#!/bin/bash
# round 3, call C: private slab slots + k_apply_slab: parity, then the unit limit swept (uncontended per-kernel times and pipelined rates)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3c; mkdir -p $O
timeout 1200 python -m pytest tests/test_tsdf_parity_gpu.py tests/test_texture_gpu.py tests/test_esdf_gpu.py tests/test_merge_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
for o in "" "unit=16384" "unit=12288" "unit=8192" "unit=4096" "unit=8192 chunks=1" "spt=2" "spt=2 unit=8192" "spt=2 unit=4096"; do timeout 200 python tools/phase_probe.py $o 2>/dev/null | tail -1; done | tee $O/probe.log
bash tools/gpu_ab.sh "" "unit=12288" "unit=8192" "unit=4096" "spt=2 unit=8192" "spt=2 unit=4096" 2>&1 | tee $O/ab.log
