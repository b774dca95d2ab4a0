#!/bin/bash
# first GPU session: smoke, parity tests, variant sweep, rocprof summary
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out
{ rocm-smi --showproductname | head -8; nproc; lscpu | grep "Model name"; } > $O/box.txt 2>&1
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
for vs in "1 4" "0 4" "1 1" "1 8" "1 16" "0 1"; do
  set -- $vs
  timeout 300 python bench.py --steps 100 --warmup 10 --variant $1 --split $2 --no-cpu-baseline > $O/bench_v$1_s$2.json 2> $O/bench_v$1_s$2.err
done
timeout 600 python bench.py --steps 200 --warmup 20 > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof1 -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof1.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R $O | head -50
tail -5 $O/smoke.log; tail -15 $O/pytest_gpu.log; cat $O/bench_*.json
