#!/bin/bash
# the bench lines of the round-3 profile set again (final code): default run, the driver's command, configs 1 / 3 / 4, the dry run of rank 3 of 8
O=$GRAFT_REPO_ROOT/gpurun_out/r03prof; mkdir -p $O; O4=$GRAFT_REPO_ROOT/gpurun_out/r03prof_c4; mkdir -p $O4; cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
timeout 300 python bench.py --as-rank 3 --of 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_dry_rank3of8.json 2> /dev/null
timeout 300 python bench.py --config 3 --steps 100 --warmup 10 2>/dev/null | tail -1 > $O/bench_c3.json
timeout 300 python bench.py --config 1 --steps 100 --warmup 10 2>/dev/null | tail -1 > $O4/bench_c1.json
timeout 300 python bench.py --config 4 --steps 100 --warmup 10 2>/dev/null | tail -1 > $O4/bench_c4.json
for f in $O/bench_default.json $O/bench_driver_cmd.json $O/bench_c3.json $O4/bench_c1.json $O4/bench_c4.json; do tail -1 $f | python tools/bench_brief.py $(basename $f); done
