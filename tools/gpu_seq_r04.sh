#!/bin/bash
# round 4: the sequential semantics on the brick pipeline -- tests, throughput, kernel table
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/seq; mkdir -p $O
timeout 600 python -m pytest "tests/test_tsdf_parity_gpu.py::test_arithmetic_shortcuts_hold_for_every_float" tests/test_sequential_gpu.py tests/test_ref_golden.py -x -q -m gpu 2>&1 | tail -15 | tee $O/tests.log
timeout 200 python tools/seq_probe.py --frames 72 --check 12 2>&1 | grep -v "^TSDF\|^Export" | tee $O/probe1.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o p -- python $GRAFT_REPO_ROOT/tools/seq_probe.py --frames 72 > $O/prof.log 2>&1
cp $(find $O/t -name '*kernel_stats.csv' | head -1) $O/seq_kernel_stats.csv 2>/dev/null
python - $O/seq_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void ", "")
    if "rocprim" in n: n = "rocprim::" + n.split("detail::")[-1][:50]
    print(f"{n[:70]:70s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms  {r['Percentage']}%")
PY
rm -rf $O/t
