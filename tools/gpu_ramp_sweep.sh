#!/bin/bash
# the driver's 20-frame burst under other ramp settings (half batches behind a dry pipeline): value / steady per setting, three runs each
for opts in "ramp=2 ramp_size=4" "ramp=0 ramp_size=4" "ramp=1 ramp_size=4" "ramp=3 ramp_size=4" "ramp=2 ramp_size=2" "ramp=4 ramp_size=2" "ramp=2 ramp_size=6"; do
  o=""; for kv in $opts; do o="$o --opt $kv"; done
  for i in 1 2 3; do
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --steady 0 $o 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$opts burst %.0f' % j['value'])"
  done
done
