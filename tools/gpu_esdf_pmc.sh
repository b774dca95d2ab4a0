#!/bin/bash
# PMC passes over the ESDF probe (k_esdf_round): instruction mix, LDS activity, waits
O=$GRAFT_REPO_ROOT/gpurun_out/esdf; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
n=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  n=$((n+1))
  cd /tmp && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc$n -o p -- python $R/tools/esdf_rounds_probe.py 1.0 24 > $O/pmc$n.log 2>&1
  f=$(find $O/pmc$n -name "*counter_collection.csv" | head -1)
  python - "$f" << 'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
rows = list(csv.DictReader(open(sys.argv[1])))
# the last update only: dispatches after the last k_esdf_collect
ids = [int(r["Dispatch_Id"]) for r in rows if "k_esdf_collect" in r["Kernel_Name"]]
last = max(ids)
for r in rows:
    if int(r["Dispatch_Id"]) < last or "k_esdf_round" not in r["Kernel_Name"]: continue
    agg["round"][r["Counter_Name"]] += float(r["Counter_Value"])
print("  ".join(f"{c}={v:.4g}" for c, v in sorted(agg["round"].items())))
PY
  rm -rf $O/pmc$n
done
