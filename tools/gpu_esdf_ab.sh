#!/bin/bash
# ESDF A/B: for every variant library lib/libtaichislam_hip_<tag>.so given as argument: ESDF tests, per-round trace of one update, config 4
O=$GRAFT_REPO_ROOT/gpurun_out/esdf; mkdir -p $O; cd $GRAFT_REPO_ROOT
for tag in "$@"; do
  export TSL_LIB=$GRAFT_REPO_ROOT/taichislam_amd/lib/libtaichislam_hip_$tag.so
  echo "== $tag"
  timeout 300 python -m pytest tests/test_esdf_gpu.py -x -q 2>&1 | grep "passed\|failed" | tail -1
  bash tools/gpu_esdf_probe.sh 2>&1 | grep "k_esdf\|^23 " | awk '{printf "%s ", $0; if ($0 ~ /^23/) print ""; else if ($1 ~ /k_esdf_round/) printf "| "; } END {print ""}' | sed 's/tsl::k_esdf_//g; s/start *[0-9.]* us  dur//g'
  cd $GRAFT_REPO_ROOT
  timeout 300 python bench.py --config 4 --steps 100 --warmup 10 2>/dev/null | tail -1 > $O/bench_c4_$tag.json
  python -c "
import json; j=json.load(open('$O/bench_c4_$tag.json')); print('config4', round(j['value'],1), 'fps', 'esdf ms', round(j['config']['esdf_ms_per_update'],3))"
done
