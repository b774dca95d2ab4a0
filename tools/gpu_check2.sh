#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/ks; mkdir -p $O
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/ks.log 2>&1
f=$(find $O/t -name "*kernel_stats.csv" | head -1); python - "$(find $O/t -name '*kernel_trace.csv' | head -1)" << 'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ib = [r for r in rows if "k_integrate_batch" in r["Kernel_Name"]]
lo, hi = 12, 48                                   # batches inside the timed region (5 warm-up batches + 50 timed)
t0, t1 = int(ib[lo]["Start_Timestamp"]), int(ib[hi]["Start_Timestamp"])
n = hi - lo
print(f"steady state: {n} batches in {(t1-t0)/1e3:.0f} us = {(t1-t0)/1e3/n:.1f} us per batch")
import collections
dur = collections.defaultdict(list)
for r in rows:
    s0, s1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 <= s0 < t1: dur[r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]].append((s1 - s0) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])): print(f"  {k:40s} n {len(v):3d}  mean {sum(v)/len(v):7.1f} us  per batch {sum(v)/n:7.1f}")
gaps = [int(ib[i+1]["Start_Timestamp"]) - int(ib[i]["End_Timestamp"]) for i in range(lo, hi)]
print("  gap between consecutive B launches: mean %.1f us  min %.1f  max %.1f" % (sum(gaps)/len(gaps)/1e3, min(gaps)/1e3, max(gaps)/1e3))
PY
rm -rf $O/t
