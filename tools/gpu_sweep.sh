#!/bin/bash
# tests + variant sweep; usage: gpu_sweep.sh "v s" "v s" ...
mkdir -p gpurun_out; O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -12 $O/pytest_gpu.log
for vs in "$@"; do
  set -- $vs
  timeout 300 python bench.py --steps 200 --warmup 20 --variant $1 --split $2 --no-cpu-baseline > $O/bench_v$1_s$2.json 2> $O/bench_v$1_s$2.err
  python - << PY
import json
try:
    d=json.loads(open("$O/bench_v$1_s$2.json").read().strip().splitlines()[-1])
    k=d['config']['kernels_us']
    print("variant $1 split $2: %.0f fps  "%d['value'] + "  ".join(f"{n}={v['avg_us']:.1f}" for n,v in k.items()))
except Exception as e:
    print("variant $1 split $2 FAILED", e); print(open("$O/bench_v$1_s$2.err").read()[-1500:])
PY
done
