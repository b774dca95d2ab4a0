#!/bin/bash
# variants of k_integrate_batch's LDS accumulation (lib/libtaichislam_hip_<tag>.so, built with -DTSL_EXP_*), each: bit-exactness tests,
# the driver's bench command twice, the steady leg, and the SQ / LDS counters of the brick kernel -> gpurun_out/brickexp/summary.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/brickexp; mkdir -p $O; : > $O/summary.txt
B20="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady 0"
for tag in ${TAGS:-base oddskip slot16 both}; do
  if [ "$tag" = base ]; then unset TSL_LIB; else export TSL_LIB=$GRAFT_REPO_ROOT/taichislam_amd/lib/libtaichislam_hip_$tag.so; fi
  echo "== $tag" >> $O/summary.txt
  timeout 300 python -m pytest tests/test_tsdf_parity_gpu.py -x -q -m gpu -k "stream or batch or bit_exact or full" 2>&1 | tail -1 >> $O/summary.txt
  for i in 1 2; do timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --steady 300 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('  burst', round(j['value']), 'frames/s, steady', round((j.get('value_steady') or {}).get('value',0)), ', launch', round(r['avg_launch_us'],1), 'us x', r.get('frames_per_launch'), ', frac', round(r['frac'],4))" >> $O/summary.txt; done
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
    cd /tmp && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p -o p -- $B20 > $O/pmc.log 2>&1
    python - "$(find $O/p -name '*counter_collection.csv' | head -1)" >> $O/summary.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_integrate_batch" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("  " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(agg.items())) + f"  (n={len(next(iter(agg.values())))})")
PY
    rm -rf $O/p; cd $GRAFT_REPO_ROOT
  done
done
cat $O/summary.txt
