#!/bin/bash
# ESDF iteration: tests, per-round trace, phase timing (timing build), config 4 with the update on its own stream and on the handle's
O=$GRAFT_REPO_ROOT/gpurun_out/esdf; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_esdf_gpu.py tests/test_boundary_gpu.py -x -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -4
bash tools/gpu_esdf_probe.sh 2>&1 | grep -v "rocprofv3\|Opened result"
cd $GRAFT_REPO_ROOT
[ -f taichislam_amd/lib/libtaichislam_hip_timing.so ] && TSL_LIB=$GRAFT_REPO_ROOT/taichislam_amd/lib/libtaichislam_hip_timing.so python tools/esdf_rounds_probe.py 1.0 24 2>&1 | grep -A20 "esdf timing" | tail -7
for ov in 1 0; do
TSL_C4_OPTS="esdf_overlap=$ov" timeout 300 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4_$ov.json
python -c "
import json; j=json.load(open('$O/bench_c4_$ov.json')); print('config4 overlap=$ov', round(j['value'],1), 'fps', {k:(round(v,3) if isinstance(v,float) else v) for k,v in j['config'].items() if k.startswith('esdf') or k.startswith('mesh_ms')})"
done
