#!/bin/bash
# state check of HEAD: GPU tests, default bench, kernel stats of the bench command
O=$GRAFT_REPO_ROOT/gpurun_out/r2s; mkdir -p $O; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20.json 2>> $O/bench_default.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o p -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/ks.log 2>&1
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv; rm -rf $O/ks
head -20 $O/bench_kernel_stats.csv
