#!/bin/bash
# steady-state kernel timeline of the pipelined bench: gpu_steady.sh "opt=v opt=v" [tag]   (rocprofv3 --kernel-trace, middle of the timed region)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/steady; mkdir -p $O
args=""; for kv in $1; do args="$args --opt $kv"; done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 40 --no-cpu-baseline $args > $O/run.log 2>&1
python - "$(find $O/t -name '*kernel_trace.csv' | head -1)" "$1" << 'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ib = [r for r in rows if "k_integrate_batch" in r["Kernel_Name"]]
# the timed region = the first 55 batches (5 warm-up + 50 timed of eight); take its middle
lo, hi = 15, 45
t0, t1 = int(ib[lo]["Start_Timestamp"]), int(ib[hi]["Start_Timestamp"])
n = hi - lo
print(f"[{sys.argv[2]}] steady state: {n} batches in {(t1-t0)/1e3:.0f} us = {(t1-t0)/1e3/n:.1f} us per batch = {8e6*n/((t1-t0)/1e3):.0f} frames/s")
dur = collections.defaultdict(list)
for r in rows:
    s0, s1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 <= s0 < t1: dur[r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]].append((s1 - s0) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])): print(f"  {k:44s} n {len(v):3d}  mean {sum(v)/len(v):7.1f} us  per batch {sum(v)/n:7.1f}")
gaps = [int(ib[i+1]["Start_Timestamp"]) - int(ib[i]["End_Timestamp"]) for i in range(lo, hi)]
print("  end of a B launch -> start of the next: mean %.1f us  min %.1f  max %.1f" % (sum(gaps)/len(gaps)/1e3, min(gaps)/1e3, max(gaps)/1e3))
PY
rm -rf $O/t
