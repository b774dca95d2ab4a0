#!/bin/bash
# round-6 profile set (everything lands in gpurun_out/r06prof, the summaries are then copied to profiles/r06_*):
#   bench lines (default, the driver's command, configs 1 / 3 / 4, the dry run of rank 3 of 8), rocprofv3 kernel stats of each,
#   PMC passes (separate runs, --kernel-trace only): HBM traffic of every kernel of the default pipeline and of the other configs' kernels,
#   SQ / LDS counters of the brick kernel.
O=$GRAFT_REPO_ROOT/gpurun_out/r06prof; mkdir -p $O; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{ rocm-smi --showproductname 2>/dev/null | grep -E "Card Series|GFX" | head -2; echo "nproc $(nproc)"; lscpu | grep "Model name"; } > $O/box.txt 2>&1
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
# (first pass without the CPU legs: make_r06_profiles.py only needs the frame statistics of the driver's command; the full lines are printed at the end,
#  once the counters of THESE sources exist)
cd $R && timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
cd $R && timeout 300 python bench.py --as-rank 3 --of 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_dry_rank3of8.json 2> /dev/null
B20="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady 0"
stats bench python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --steady 0
stats bench_driver_cmd $B20
pmc fetch "FETCH_SIZE" $B20
pmc write "WRITE_SIZE" $B20
pmc sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" $B20
pmc lds "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" $B20
pmc tcc "TCC_HIT_sum TCC_MISS_sum" $B20
for c in 1 3 4; do
  stats c$c python $R/bench.py --config $c --steps 60 --warmup 10 --no-cpu-baseline
  pmc fetch_c$c "FETCH_SIZE" python $R/bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline
  pmc write_c$c "WRITE_SIZE" python $R/bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline
done
# configs[3] with the ESDF update as the raise / lower wavefront (esdf_mode 1)
TSL_C4_OPTS="esdf_mode=1" stats c4w python $R/bench.py --config 4 --steps 60 --warmup 10 --no-cpu-baseline
TSL_C4_OPTS="esdf_mode=1" pmc fetch_c4w "FETCH_SIZE" python $R/bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline
TSL_C4_OPTS="esdf_mode=1" pmc write_c4w "WRITE_SIZE" python $R/bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline
stats merge python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady 0 --merge
pmc fetch_merge "FETCH_SIZE" python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady 0 --merge
pmc write_merge "WRITE_SIZE" python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady 0 --merge
# the sequential semantics (rounds 4-5): kernel table of a 400-frame stream, SQ counters of the replay and the grouping kernels
SEQ="python $R/tools/seq_probe.py --frames 400 --warmup 40"
stats seq $SEQ
pmc sq_seq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" python $R/tools/seq_probe.py --frames 72 --warmup 8
pmc fetch_seq "FETCH_SIZE" python $R/tools/seq_probe.py --frames 72 --warmup 8
pmc write_seq "WRITE_SIZE" python $R/tools/seq_probe.py --frames 72 --warmup 8
cd $R && timeout 120 python tools/seq_probe.py --frames 400 --warmup 40 2>&1 | grep seq_impl > $O/seq_probe.txt
cd $R && timeout 120 python tools/host_input_probe.py 400 2>&1 | tail -1 > $O/host_input_probe.txt
cd $R
timeout 600 python tools/parity_report.py 12 $O/parity_vs_faithful.json > $O/parity_summary.json 2>/dev/null
python - > $O/lib_source_hash.txt << 'PY'
import sys; sys.path.insert(0, ".")
from taichislam_amd import build
print(build.source_hash())
PY
# the counters are in: reduce them here too (profiles/r06_traffic.json of THIS copy, stamped with these sources' hash) and print the bench lines
# again, now with `roofline.traffic` from counters collected on the same sources a minute ago
cd $R && python tools/make_r06_profiles.py > /dev/null 2>&1
cd $R && timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd $R && timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
for c in 1 3 4; do timeout 300 python bench.py --config $c --steps 100 --warmup 10 2>/dev/null | tail -1 > $O/bench_c$c.json; done
TSL_C4_OPTS="esdf_mode=1" timeout 300 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4_wavefront.json
timeout 300 python bench.py --config 3 --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c3_1000.json
ls $O; grep -c . $O/pmc_summary.txt
