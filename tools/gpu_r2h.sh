#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/r2h_pytest.log; cat $O/r2h_pytest.log
timeout 600 python bench.py --steps 100 --warmup 20 --merge > $O/r2h_bench.json 2> $O/r2h_bench.err; echo "bench rc=$?"; tail -3 $O/r2h_bench.err
python - << 'PY'
import json
d=json.loads(open("gpurun_out/r2h_bench.json").read().strip().splitlines()[-1])
print("fps", d["value"], "host", d["value_host_input"], "merge", d["config"]["merge"])
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline_allcore",{}).get("value"), d.get("cpu_baseline_allcore",{}).get("cores"))
print("parity", d.get("parity_vs_faithful"))
print("roof", d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
PY
