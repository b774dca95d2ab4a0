#!/bin/bash
# config 4 under backend option sets: bash tools/gpu_c4_opts2.sh "esdf_mode=1 esdf_grid=256" "esdf_mode=0" ...
cd $GRAFT_REPO_ROOT
for o in "$@"; do
TSL_C4_OPTS="$o" timeout 300 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('config4 [$o]', round(j['value'],1), 'fps', {k:(round(v,3) if isinstance(v,float) else v) for k,v in j['config'].items() if k.startswith('esdf_ms') or k.startswith('esdf_brick')})"
done
