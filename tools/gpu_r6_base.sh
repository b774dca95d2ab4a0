#!/bin/bash
# round 6 baseline: ESDF iteration (tests, rounds trace, timing build, config 4) + kernel trace of config 4 with the memsets named
O=$GRAFT_REPO_ROOT/gpurun_out/r6base; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/gpu_esdf_iter.sh 2>&1 | tee $O/esdf_iter.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/c4 -o c4 -- python $GRAFT_REPO_ROOT/bench.py --config 4 --steps 40 --warmup 10 --no-cpu-baseline > $O/c4.log 2>&1
find $O/c4 -name "*stats*" | head; f=$(find $O/c4 -name "*kernel_stats.csv" | head -1); head -25 $f
