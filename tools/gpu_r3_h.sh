#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_tsdf_parity_gpu.py -x -q -m gpu --timeout=240 -k "units_and_parts or full_batches or workgroup or small_stream" 2>&1 | tail -2
TSL_LIB=$GRAFT_REPO_ROOT/taichislam_amd/lib/libtaichislam_hip_timing.so timeout 200 python tools/timing_probe.py spt=2 unit=8192 2>&1 | tail -22
bash tools/gpu_steady.sh "spt=2 unit=8192" 2>&1 | tail -10
