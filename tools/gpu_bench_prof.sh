#!/bin/bash
# bench + rocprofv3 kernel stats; args: tag [bench args...]
TAG=${1:-run}; shift
mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp
{ rocm-smi --showproductname 2>/dev/null | grep -E "Card Series|GFX" | head -2; echo "nproc $(nproc)"; lscpu | grep "Model name"; } > $O/box.txt 2>&1
timeout 600 python bench.py "$@" > $O/bench_$TAG.json 2> $O/bench_$TAG.err; echo "bench rc=$?"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > $O/prof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT; find $O/prof_$TAG -name "*kernel_stats*.csv" | head -3
f=$(find $O/prof_$TAG -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && head -20 "$f" | cut -c1-200
cat $O/bench_$TAG.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['kernels_us'], d.get('cpu_baseline',{}).get('value'))"
cat $O/box.txt
