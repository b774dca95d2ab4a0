#!/bin/bash
# config 4 A/B of backend options (TSL_C4_OPTS), plus a kernel trace of the host probe for the first option set
O=$GRAFT_REPO_ROOT/gpurun_out/c4; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for o in; do
  TSL_C4_OPTS="$o" timeout 300 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/b.json
  python -c "
import json; j=json.load(open('$O/b.json')); print('bench config4 [$o]', round(j['value'],1), 'fps', 'esdf ms', round(j['config']['esdf_ms_per_update'],3))"
  timeout 200 python tools/c4_host_probe.py $o 2>&1 | tail -1
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o p -- python $GRAFT_REPO_ROOT/tools/c4_host_probe.py $1 > $O/probe.log 2>&1
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python - "$f" << 'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last three ESDF updates: from the third-last k_esdf_collect on
idx = [i for i, r in enumerate(rows) if "k_esdf_collect" in r["Kernel_Name"]]
i0 = idx[-6]; t0 = int(rows[i0]["Start_Timestamp"])
first = None
for r in rows[i0:idx[-3]]:
    n = r["Kernel_Name"].split("(")[0].replace("tsl::", "").replace("void ", "")
    st, en = (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3
    if "k_esdf_round" in n:
        if first is None: first = st
        last = en; continue
    if first is not None: print(f"{'  k_esdf_round x N':44s}      start {first:9.1f} us  end {last:9.1f} us"); first = None
    print(f"{n[:44]:44s} q{r.get('Queue_Id','?'):3s} start {st:9.1f} us  end {en:9.1f} us")
PY
rm -rf $O/kt
