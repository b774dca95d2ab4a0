cd $GRAFT_REPO_ROOT; timeout 600 python -m pytest tests/test_esdf_gpu.py -x -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -40
