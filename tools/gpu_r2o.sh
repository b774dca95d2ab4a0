#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $O/r2o_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/gpu_profiles_r02.sh 2>&1 | tail -40
