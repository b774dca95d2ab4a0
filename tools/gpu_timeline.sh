#!/bin/bash
# kernel timeline of a short bench run (the driver's command line): what the 20 timed frames actually wait for
O=$GRAFT_REPO_ROOT/gpurun_out/tl; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/run.log 2>&1
python - "$(find $O/t -name '*kernel_trace.csv' | head -1)" << 'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "tsl::" in r["Kernel_Name"] and "verify_div" not in r["Kernel_Name"]]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
for r in rows[:90]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tsl::", "")[:30]
    s0, s1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s0 - prev_end) / 1e3
    print(f"{n:30s} start {(s0-t0)/1e3:9.1f}  end {(s1-t0)/1e3:9.1f}  frames {r.get('Grid_Size_Y','')}" + (f"   <-- idle {gap:.0f} us" if gap > 150 else ""))
    prev_end = max(prev_end, s1)
PY
grep -o '"value": [0-9.]*' $O/run.log | head -1
rm -rf $O/t
