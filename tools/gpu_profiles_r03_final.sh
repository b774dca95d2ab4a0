#!/bin/bash
# the default-pipeline lines of the round-3 profile set once more after the work list changed order (units before parts):
# the driver's command and the default run, and the kernel stats of the two commands the roofline durations are checked against
O=$GRAFT_REPO_ROOT/gpurun_out/r03prof; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
stats() {  # tag, command...
  tag=$1; shift
  cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$tag -o p -- "$@" > $O/ks_$tag.log 2>&1
  f=$(find $O/ks_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${tag}_kernel_stats.csv
  rm -rf $O/ks_$tag; cd $R
}
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
stats bench_driver_cmd python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady 0
stats bench python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --steady 0
timeout 240 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for f in $O/bench_default.json $O/bench_driver_cmd.json; do tail -1 $f | python tools/bench_brief.py $(basename $f); done
grep -h k_integrate_batch $O/bench_kernel_stats.csv $O/bench_driver_cmd_kernel_stats.csv | cut -c1-60,200-
