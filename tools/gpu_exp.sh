#!/bin/bash
# timing-only experiments: libs built with -DTSL_EXP=n are in taichislam_amd/lib/exp_n.so
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for n in "$@"; do
  cp taichislam_amd/lib/exp_$n.so taichislam_amd/lib/libtaichislam_hip.so
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/exp_$n -o p -- python $GRAFT_REPO_ROOT/tools/host_overhead_probe.py overlap=0 > $O/exp_$n.log 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find $O/exp_$n -name "*kernel_stats.csv" | head -1)
  echo "== TSL_EXP=$n  $(tail -1 $O/exp_$n.log)"
  python - "$f" << 'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:3]:
    n = r["Name"].split("(")[0].replace("void ", "")[:70]
    print(f"{n:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
done
