#!/bin/bash
# round 6 ESDF iteration: exactness probe, ESDF tests, per-round trace, timing build, config 4 in both modes
O=$GRAFT_REPO_ROOT/gpurun_out/esdf; mkdir -p $O; cd $GRAFT_REPO_ROOT
F='^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL\|initialized'
[ -z "$SKIP_CHECK" ] && python tools/esdf_wave_check.py ${CHECK:-c2} 2>&1 | grep -v "$F" | tail -${CHECK_TAIL:-4}
[ -z "$SKIP_TESTS" ] && timeout 600 python -m pytest tests/test_esdf_gpu.py tests/test_boundary_gpu.py -x -q 2>&1 | grep -v "$F" | tail -4
bash tools/gpu_esdf_probe.sh 2>&1 | grep -v "rocprofv3\|Opened result" | tail -22
cd $GRAFT_REPO_ROOT
[ -f taichislam_amd/lib/libtaichislam_hip_timing.so ] && TSL_LIB=$GRAFT_REPO_ROOT/taichislam_amd/lib/libtaichislam_hip_timing.so python tools/esdf_rounds_probe.py 1.0 24 2>&1 | grep -A20 "esdf timing" | tail -12
for mode in ${MODES:-1 0}; do
TSL_C4_OPTS="esdf_mode=$mode" timeout 300 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4_$mode.json
python -c "
import json; j=json.load(open('$O/bench_c4_$mode.json')); print('config4 esdf_mode=$mode', round(j['value'],1), 'fps', {k:(round(v,3) if isinstance(v,float) else v) for k,v in j['config'].items() if k.startswith('esdf') or k.startswith('mesh_ms')})"
done
