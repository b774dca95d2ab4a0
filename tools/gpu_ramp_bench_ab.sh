#!/bin/bash
# developer A/B: the driver's bench command under different ramp-up settings (value, frames per launch, roofline fraction), three runs each, interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for o in "ramp=2" "ramp=1" "ramp=0"; do
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --steady 0 --opt $o 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$o', round(j['value']), 'frames/s  launch', round(r['avg_launch_us'],1), 'us x', r.get('frames_per_launch'), ' frac', round(r['frac'],4))"
done; done
