#!/bin/bash
# the config-4 and config-1 parts of the round-3 profile set again (after the ESDF and marching-cubes rewrites): bench line, kernel stats, FETCH / WRITE / SQ / LDS counter passes.
# Output in gpurun_out/r03prof_c4; tools/make_r03_profiles.py merges the sections into the set of tools/gpu_profiles_r03.sh.
O=$GRAFT_REPO_ROOT/gpurun_out/r03prof_c4; mkdir -p $O; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
stats() {  # tag, command...
  tag=$1; shift
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$tag -o p -- "$@" > $O/ks_$tag.log 2>&1
  f=$(find $O/ks_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${tag}_kernel_stats.csv
  rm -rf $O/ks_$tag
}
pmc() {  # tag, counters, command...
  tag=$1; ctr=$2; shift; shift
  cd /tmp && timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- "$@" > $O/pmc_$tag.log 2>&1
  f=$(find $O/pmc_$tag -name "*counter_collection.csv" | head -1)
  echo "== $tag: rocprofv3 --pmc $ctr --kernel-trace -- $*" | sed "s|$R/||g" >> $O/pmc_summary.txt
  python - "$f" >> $O/pmc_summary.txt << 'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    if k.startswith("tsl::") or k.startswith("k_octo"): print(f"{k[:58]:58s} " + "  ".join(f"{c}={sum(v)/len(v):.6g}(n={len(v)})" for c, v in sorted(d.items())))
PY
  rm -rf $O/pmc_$tag
}
: > $O/pmc_summary.txt
C4="python $R/bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline"
stats c4 python $R/bench.py --config 4 --steps 60 --warmup 10 --no-cpu-baseline
pmc fetch_c4 "FETCH_SIZE" $C4
pmc write_c4 "WRITE_SIZE" $C4
pmc sq_c4 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" $C4
pmc lds_c4 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" $C4
stats c1 python $R/bench.py --config 1 --steps 60 --warmup 10 --no-cpu-baseline
pmc fetch_c1 "FETCH_SIZE" python $R/bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline
pmc write_c1 "WRITE_SIZE" python $R/bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline
cd $R
timeout 300 python bench.py --config 1 --steps 100 --warmup 10 2>/dev/null | tail -1 > $O/bench_c1.json
timeout 300 python bench.py --config 4 --steps 100 --warmup 10 2>/dev/null | tail -1 > $O/bench_c4.json
TSL_C4_OPTS="esdf_overlap=0" timeout 300 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4_no_overlap.json
ls $O; grep "k_esdf_round" $O/pmc_summary.txt | cut -c1-300
