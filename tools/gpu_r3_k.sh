#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_tsdf_parity_gpu.py tests/test_texture_gpu.py tests/test_sequential_gpu.py -x -q -m gpu --timeout=240 2>&1 | tail -3
timeout 120 python tools/phase_probe.py 2>/dev/null | tail -2
bash tools/gpu_steady.sh "" 2>&1 | tail -10
