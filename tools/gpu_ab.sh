#!/bin/bash
# A/B of bench.py option strings: gpu_ab.sh "opt1=v opt2=v" "..."   ("" = defaults); two runs each
cd $GRAFT_REPO_ROOT
for o in "$@"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); k=j['config']['kernels_us']; print('[$o]', round(j['value'],1), 'fps | integrate us', round(j['roofline']['avg_launch_us'],1), '| host input', {a:round(b) for a,b in j['value_host_input'].items() if a!='note'})"; done
done
