#!/bin/bash
# A/B of the pipeline events' fence scope: bursts of 20 frames and a steady stream, default path
for i in 1 2 3; do
  for v in sys dev; do
    if [ $v = sys ]; then export TSL_EV_SYS=1 TSL_LIB=$GRAFT_REPO_ROOT/taichislam_amd/lib/libtaichislam_hip_testhooks.so; else unset TSL_EV_SYS TSL_LIB; fi
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --steady 300 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v burst %.0f steady %.0f host %s launch_us %.1f' % (j['value'], j['value_steady']['value'], {k: round(v) for k, v in j['value_host_input'].items() if k != 'note'}, j['roofline']['avg_launch_us']))"
  done
done
