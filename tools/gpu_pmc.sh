#!/bin/bash
# rocprofv3 PMC passes over a short bench run.  usage: gpu_pmc.sh tag "CTR1 CTR2 ..." ["CTR..." ...]  (one pass per quoted group)
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
n=0
for grp in "$@"; do
  n=$((n+1))
  cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_${TAG}_$n -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline $BENCH_ARGS > $O/pmc_${TAG}_$n.log 2>&1
  f=$(find $O/pmc_${TAG}_$n -name "*counter_collection.csv" | head -1)
  echo "== pass $n: $grp -> $f"
  python - "$f" << 'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    k = k.split("(")[0].replace("void ", "")
    if "rocprim" in k: k = "rocprim::" + k.split("::")[-1][:40]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(f"{k[:60]:60s} " + "  ".join(f"{c}={sum(v)/len(v):.4g}(n={len(v)})" for c, v in sorted(d.items())))
PY
done
