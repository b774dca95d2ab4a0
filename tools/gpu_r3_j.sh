#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3j; mkdir -p $O
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "driver cmd rc $?"; python -c "
import json; j=json.load(open('$O/bench_driver_cmd.json')); print('value', round(j['value']), 'steady', j['value_steady'], 'seq', j.get('value_sequential'), 'cpu', j['cpu_baseline']['value'], 'roof', {k: j['roofline'][k] for k in ('frac','avg_launch_us','frames_per_launch','valu_frac')})"
timeout 300 python bench.py --as-rank 3 --of 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_dry.json 2> $O/bench_dry.err; echo "dry rc $?"; python -c "
import json; j=json.load(open('$O/bench_dry.json')); print('dry value', round(j['value']), j['n_gpus'], j['dry_run'], j['config']['merge'])"
for c in 1 3 4; do timeout 300 python bench.py --config $c --steps 100 --warmup 10 > $O/bench_c$c.json 2> $O/bench_c$c.err; echo "c$c rc $?"; python -c "
import json; j=json.load(open('$O/bench_c$c.json')); print('c$c', round(j['value'],1), j['unit'], 'cpu', j.get('cpu_baseline'), 'roof', j['roofline']['avg_launch_us'], j['roofline']['frac'])"; done
timeout 600 python tools/parity_report.py 12 $O/parity_vs_faithful.json 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(json.dumps(j['fusion'], indent=0)[:1800])"
