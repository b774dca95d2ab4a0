#!/bin/bash
# per-kernel durations (rocprofv3 kernel trace) of the frame pipeline, serial (overlap=0) and pipelined (overlap=4)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for ov in 0 4; do
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kst_ov$ov -o p -- python $GRAFT_REPO_ROOT/tools/host_overhead_probe.py overlap=$ov $EXTRA > $O/kst_ov$ov.log 2>&1
  f=$(find $O/kst_ov$ov -name "*kernel_stats.csv" | head -1)
  echo "== overlap=$ov  $(tail -1 $O/kst_ov$ov.log)"
  python - "$f" << 'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    n = r["Name"].split("(")[0].replace("void ", "")[:70]
    print(f"{n:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  total% {r['Percentage']}")
PY
done
