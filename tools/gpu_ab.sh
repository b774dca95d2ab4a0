#!/bin/bash
# A/B bench lines; usage: gpu_ab.sh "bench args" "bench args" ...
mkdir -p gpurun_out; O=gpurun_out
i=0
for a in "$@"; do
  i=$((i+1))
  timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline $a > $O/ab_$i.json 2> $O/ab_$i.err
  python - << PY
import json
try:
    d=json.loads(open("$O/ab_$i.json").read().strip().splitlines()[-1])
    k=d['config']['kernels_us']
    print("[$a] %.0f fps  "%d['value'] + "  ".join(f"{n}={v['avg_us']:.1f}" for n,v in k.items()))
except Exception as e:
    print("[$a] FAILED", e); print(open("$O/ab_$i.err").read()[-1500:])
PY
done
