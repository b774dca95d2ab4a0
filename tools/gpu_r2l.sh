#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_esdf_gpu.py -m gpu -x -q 2>&1 | tail -8 | tee $O/r2l_esdf.log
timeout 200 python - << 'PY' 2>&1 | grep -v "amdgpu.ids\|initialized" | tee $O/r2l_time.log
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
from taichislam_amd import _lib
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
frames = list(syn.sphere_room_stream(60))
dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
for md, full in ((1.0, 0), (1.0, 1), (0.4, 0), (5.0, 1)):
    m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH); m.set_option("esdf_full", full)
    for i in range(30):
        m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None); m.update_esdf(max_dist=md)
    m.sync(); m.enable_profiling(True, only=[_lib.K_ESDF]); t0 = time.perf_counter()
    for i in range(30, 60):
        m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None); m.update_esdf(max_dist=md)
    m.sync(); dt = time.perf_counter() - t0
    ems, en = m.kernel_time(_lib.K_ESDF)
    print(f"max_dist {md} full {full}: {30/dt:.0f} frames/s with an ESDF update per frame, ESDF kernels {1e3*ems/en:.0f} us per update, stats {m.esdf_stats()}")
    del m
PY

