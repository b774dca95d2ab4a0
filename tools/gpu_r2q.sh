#!/bin/bash
# fused batch kernel: parity first (under timeout), then timing
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_tsdf_parity_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee $O/r2q_pytest.log
for o in "chunks=2 unit=4096" "chunks=2 unit=3072" "chunks=2 unit=2048" "chunks=1 unit=3072" "chunks=4 unit=3072"; do
  echo "$o: $(python tools/host_overhead_probe.py $o 2>&1 | tail -1)"
done
python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps',d['steps'],'fps',round(d['value']),'host',d['value_host_input'], d['roofline']['avg_launch_us'])"
for o in "chunks=2 unit=4096" "chunks=2 unit=3072"; do
  echo "== wg timeline $o"
  TSL_LIB=$GRAFT_REPO_ROOT/taichislam_amd/lib/libtaichislam_hip_timing.so timeout 120 python tools/timing_probe_wg.py $o 2>&1 | grep -v "amdgpu.ids\|initialized"
done
