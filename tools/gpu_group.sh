#!/bin/bash
# tests + A/B of the pixel grouping path (hash table vs rocPRIM sort)
mkdir -p gpurun_out; O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -12 $O/pytest_gpu.log
for g in hash sort; do
  if [ $g = sort ]; then export TSL_GROUP_SORT=1; else unset TSL_GROUP_SORT; fi
  timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_group_$g.json 2> $O/bench_group_$g.err
  python - << PY
import json
try:
    d=json.loads(open("$O/bench_group_$g.json").read().strip().splitlines()[-1])
    k=d['config']['kernels_us']
    print("group $g: %.0f fps  "%d['value'] + "  ".join(f"{n}={v['avg_us']:.1f}" for n,v in k.items()))
except Exception as e:
    print("group $g FAILED", e); print(open("$O/bench_group_$g.err").read()[-1500:])
PY
done
