#!/bin/bash
# round-2 profile set: kernel stats of the bench command, PMC passes (HBM traffic, SQ, LDS) for the brick kernel, kernel stats of the other configs
O=$GRAFT_REPO_ROOT/gpurun_out/r02prof; mkdir -p $O; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{ rocm-smi --showproductname 2>/dev/null | grep -E "Card Series|GFX" | head -2; echo "nproc $(nproc)"; lscpu | grep "Model name"; } > $O/box.txt 2>&1
stats() {  # tag, command...
  tag=$1; shift
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$tag -o p -- "$@" > $O/ks_$tag.log 2>&1
  f=$(find $O/ks_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${tag}_kernel_stats.csv
  rm -rf $O/ks_$tag
}
pmc() {  # tag, counters, bench args...
  tag=$1; ctr=$2; shift; shift
  cd /tmp && timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $O/pmc_$tag.log 2>&1
  f=$(find $O/pmc_$tag -name "*counter_collection.csv" | head -1)
  echo "== $tag: rocprofv3 --pmc $ctr --kernel-trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline $*" >> $O/pmc_summary.txt
  python - "$f" >> $O/pmc_summary.txt << 'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    if k.startswith("tsl::"): print(f"{k[:58]:58s} " + "  ".join(f"{c}={sum(v)/len(v):.5g}(n={len(v)})" for c, v in sorted(d.items())))
PY
  rm -rf $O/pmc_$tag
}
: > $O/pmc_summary.txt
cd $R && timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
stats bench python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline
for cfg in ""; do
  t=$(echo "c$cfg" | tr -d ' =-')
  pmc fetch_$t "FETCH_SIZE" $cfg
  pmc write_$t "WRITE_SIZE" $cfg
done
pmc sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
pmc lds "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
pmc tcc "TCC_HIT_sum TCC_MISS_sum"
stats c1 python $R/bench.py --config 1 --steps 50 --warmup 5
stats c3 python $R/bench.py --config 3 --steps 100 --warmup 10
stats c4 python $R/bench.py --config 4 --steps 60 --warmup 10
stats merge python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --merge
cd $R
for c in 1 3 4; do timeout 300 python bench.py --config $c --steps 100 --warmup 10 2>/dev/null | tail -1 > $O/bench_c$c.json; done
ls $O; cat $O/pmc_summary.txt | grep -i "integrate_bricks\|=="
