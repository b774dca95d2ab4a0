#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_tsdf_parity_gpu.py -x -q -m gpu --timeout=240 -k "units_and_parts or full_batches or workgroup" 2>&1 | tail -2
for o in "spt=2 unit=8192" "spt=2 unit=6144" "spt=2 unit=4096"; do timeout 120 python tools/phase_probe.py $o 2>/dev/null | tail -2; done
bash tools/gpu_steady.sh "spt=2 unit=8192" 2>&1 | tail -10
bash tools/gpu_steady.sh "spt=2 unit=6144" 2>&1 | tail -10
