#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/esdf; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o p -- python $R/tools/esdf_rounds_probe.py "$@" > $O/probe.log 2>&1
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python - "$f" << 'PY' > $O/rounds.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last update: from the last k_esdf_collect on
idx = max(i for i, r in enumerate(rows) if "k_esdf_collect" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    n = r["Kernel_Name"].split("(")[0]
    print(f"{n[:40]:40s} start {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us")
PY
rm -rf $O/kt
grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" $O/probe.log | tail -6; cat $O/rounds.txt
