#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_esdf_gpu.py -m gpu -x -q 2>&1 | tail -8 | tee $O/r2n_esdf.log
timeout 300 python bench.py --config 4 --steps 100 --warmup 20 2>&1 | grep -v "amdgpu.ids\|initialized\|Total tri" | tail -1 > $O/r2n_c4.json
python - << 'PY'
import json
d=json.loads(open("gpurun_out/r2n_c4.json").read().strip().splitlines()[-1])
print("C4 fps", d["value"], {k:v for k,v in d["config"].items() if k!="workload"})
PY
