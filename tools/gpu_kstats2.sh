#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kst_b -o p -- python $GRAFT_REPO_ROOT/tools/batch_serial_probe.py $EXTRA > $O/kst_b.log 2>&1
f=$(find $O/kst_b -name "*kernel_stats.csv" | head -1)
echo "== $(tail -1 $O/kst_b.log)"
python - "$f" << 'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n = r["Name"].split("(")[0].replace("void ", "")[:60]
    print(f"{n:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
