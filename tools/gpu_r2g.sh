#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/r2g_pytest.log; cat $O/r2g_pytest.log
timeout 300 python tools/parity_report.py 12 $O/parity_vs_faithful.json 2>&1 | grep -v "amdgpu.ids\|initialized\|Export" | tee $O/r2g_parity.log
bash tools/gpu_ab.sh "" "--opt chunks=2" 2>&1 | tee $O/r2g_ab.log
