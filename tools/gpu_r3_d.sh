#!/bin/bash
# round 3, call D: inline pixel lists in the sensor-voxel table (no group_scan / group_fill) + the parallel k_apply_slab + boundary additions
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests/test_tsdf_parity_gpu.py tests/test_texture_gpu.py tests/test_merge_gpu.py tests/test_boundary_gpu.py tests/test_parity_vs_faithful_gpu.py tests/test_fusion_mesh_gpu.py tests/test_esdf_gpu.py tests/test_query_gpu.py tests/test_octomap_gpu.py -x -q -m gpu --timeout=240 > $O/pytest.log 2>&1; rc=$?; echo "pytest rc $rc" >> $O/pytest.log; tail -5 $O/pytest.log
if [ $rc -ne 0 ]; then grep -v "^$" $O/pytest.log | grep -B2 -A25 "Error\|assert" | head -80; exit 1; fi
for o in "" "unit=12288" "unit=8192" "spt=2 unit=8192" "spt=2 unit=12288" "spt=2 unit=6144"; do timeout 120 python tools/phase_probe.py $o 2>/dev/null | tail -1; done | tee $O/probe.log
bash tools/gpu_ab.sh "" "unit=12288" "unit=8192" "spt=2 unit=8192" "spt=2 unit=12288" 2>&1 | tee $O/ab.log
