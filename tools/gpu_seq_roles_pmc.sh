#!/bin/bash
# VALU wave-instructions of the two roles of k_seq_replay launched apart (TSL_SEQ_SPLIT_ROLES=1): short runs first, long runs second
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/roles; mkdir -p $O
cd /tmp && TSL_SEQ_SPLIT_ROLES=1 TSL_LIB=$GRAFT_REPO_ROOT/taichislam_amd/lib/libtaichislam_hip_testhooks.so timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/t -o p -- python $GRAFT_REPO_ROOT/tools/seq_probe.py --frames 72 > $O/pmc.log 2>&1
python - "$(find $O/t -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_seq_replay" in r["Kernel_Name"]]
d = collections.OrderedDict()
for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
    d.setdefault(r["Dispatch_Id"], {"grid": r["Grid_Size"]})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in d.items():
    print(k, "short" if int(v["grid"]) > 500000 else "long ", {a: (f"{b:.3g}" if isinstance(b, float) else b) for a, b in v.items()})
PY
rm -rf $O/t
