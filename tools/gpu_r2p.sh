#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/host_overhead_probe.py 2>&1 | tail -1
python tools/host_overhead_probe.py chunks=2 2>&1 | tail -1
for s in "300 30" "100 20" "20 5"; do set -- $s; python bench.py --steps $1 --warmup $2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps',d['steps'],'fps',round(d['value']),'host',d['value_host_input'])"; done
python - << 'PY'
import sys, time, numpy as np, torch, cProfile, pstats
sys.path.insert(0, ".")
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH)
N = 330
frames = list(syn.sphere_room_stream(N))
dev = torch.from_numpy(np.stack([d for _, _, d in frames]).view(np.int16)).cuda()
for i in range(30): m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
m.sync()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for i in range(30, N): m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
t1 = time.perf_counter(); m.sync(); t2 = time.perf_counter()
pr.disable()
print(f"enqueue {1e6*(t1-t0)/300:.1f} us/frame total {1e6*(t2-t0)/300:.1f} us/frame")
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
PY
