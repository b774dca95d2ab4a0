#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_tsdf_parity_gpu.py tests/test_texture_gpu.py -x -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -4
bash tools/gpu_ab.sh "" "bgrid=60" "bgrid=75"
