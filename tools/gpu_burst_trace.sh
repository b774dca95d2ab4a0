#!/bin/bash
# kernel timeline of the driver's 20-frame burst (bench.py --gpus 1 --steps 20 --warmup 5): the launches between the warm-up's last apply and the end
O=$GRAFT_REPO_ROOT/gpurun_out/burst; mkdir -p $O; export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --steady 0 "$@" > $O/bench.json 2>/dev/null
tail -1 $O/bench.json | python $GRAFT_REPO_ROOT/tools/bench_brief.py burst
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python - "$f" << 'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("tsl::", "").replace("void ", "") for r in rows]
# the timed region of the contract: the third k_integrate_batch-bearing group from the start is warm-up ... simply print everything from the first k_set_params after the 2nd k_apply_slab
ap = [i for i, n in enumerate(names) if n.startswith("k_apply_slab")]
i0 = ap[1] + 1 if len(ap) > 1 else 0
# stop before the breakdown pass: the timed region has at most 4 integrate launches
ib = [i for i, n in enumerate(names) if n.startswith("k_integrate_batch") and i >= i0][:4]
i1 = [i for i in ap if i > ib[-1]][0] + 1
t0 = int(rows[i0]["Start_Timestamp"])
for r, n in zip(rows[i0:i1], names[i0:i1]):
    print(f"{n[:40]:40s} q{r.get('Queue_Id','?'):3s} {(int(r['Start_Timestamp'])-t0)/1e3:8.1f} .. {(int(r['End_Timestamp'])-t0)/1e3:8.1f} us  ({(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:6.1f})")
PY
rm -rf $O/kt
