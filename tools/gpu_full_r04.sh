cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/full; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $O/tests.log
timeout 200 python tools/seq_probe.py --frames 400 --warmup 40 2>&1 | grep -v "^TSDF\|^Export" | tee $O/probe_steady.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 600 $O/bench_driver_cmd.err
python - $O/bench_driver_cmd.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: j.get(k) for k in ("value", "ms_per_step")}, "steady", j.get("value_steady"), "host", j.get("value_host_input"))
print("roofline", {k: j["roofline"].get(k) for k in ("frac", "avg_launch_us", "frames_per_launch", "valu_frac")})
print("seq", j.get("value_sequential"))
print("parity", {k: j["parity_vs_faithful"].get(k) for k in ("tsdf_rel_frac_le_1e-4", "frames")})
PY
timeout 100 python tools/host_input_probe.py 400 2>&1 | grep -v "^TSDF\|^Export" | tail -2
