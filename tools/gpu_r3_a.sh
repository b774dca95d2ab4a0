#!/bin/bash
# round 3, call A: parity of the 16-waves-per-CU brick kernel (spt=2), then A/B of occupancy / split / grid options, then a steady-state trace
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests/test_tsdf_parity_gpu.py -x -q -m gpu -k "workgroup_and_part or units_and_parts or full_size_c2 or full_batches or kernel_variants or small_stream" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
bash tools/gpu_ab.sh "" "spt=2" "spt=2 chunks=4" "spt=2 bgrid=75" "spt=2 bgrid=50" "split=4" "spt=2 split=4" "spt=2 unit=49152" 2>&1 | tee $O/ab.log
