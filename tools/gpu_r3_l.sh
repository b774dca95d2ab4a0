#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_tsdf_parity_gpu.py tests/test_texture_gpu.py -x -q -m gpu --timeout=240 2>&1 | tail -3
for o in "" "unit_half=8192" "unit_half=3072 unit=8192" "unit_half=4096 unit=12288" "unit_half=3072 unit=6144"; do timeout 120 python tools/phase_probe.py $o 2>/dev/null | tail -2; done
bash tools/gpu_ab.sh "" "unit_half=8192" "unit_half=3072" "unit_half=4096 unit=12288" "unit_half=3072 unit=6144" 2>&1
