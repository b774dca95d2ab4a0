#!/bin/bash
# config 4 with backend option sets (each argument one set, e.g. "chunks=2 unit_floor=2048")
O=$GRAFT_REPO_ROOT/gpurun_out/c4; mkdir -p $O; cd $GRAFT_REPO_ROOT
for o in "$@"; do
  TSL_C4_OPTS="$o" timeout 300 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/b.json
  python -c "
import json; j=json.load(open('$O/b.json')); print('config4 [$o]', round(j['value'],1), 'fps', 'esdf ms', round(j['config']['esdf_ms_per_update'],3))"
done
