#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_esdf_gpu.py -m gpu -x -q 2>&1 | tail -30 > $O/r2i_esdf.log; cat $O/r2i_esdf.log
timeout 900 python -m pytest tests/test_fusion_mesh_gpu.py tests/test_reference_callers.py -m gpu -q 2>&1 | tail -8 > $O/r2i_pytest.log; cat $O/r2i_pytest.log
timeout 120 python - << 'PY' 2>&1 | grep -v "amdgpu.ids\|initialized" | tee $O/r2i_esdf_time.log
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from taichislam_amd.mapping import DenseTSDF
from taichislam_amd.utils import synthetic as syn
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
frames = list(syn.sphere_room_stream(60))
dev = [torch.from_numpy(d.view(np.int16)).cuda() for _, _, d in frames]
for md, full in ((1.0, 0), (1.0, 1), (0.4, 0), (0.4, 1), (5.0, 1)):
    m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH); m.set_option("esdf_full", full)
    for i in range(30):
        m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None); m.update_esdf(max_dist=md)
    m.sync(); t0 = time.perf_counter(); te = 0.0
    for i in range(30, 60):
        m.recast_depth_to_map(frames[i][0], frames[i][1], dev[i], None)
        m.sync(); a = time.perf_counter(); m.update_esdf(max_dist=md); te += time.perf_counter() - a
    dt = time.perf_counter() - t0
    print(f"max_dist {md} full {full}: {30/dt:.0f} frames/s with an ESDF update per frame, {1e3*te/30:.3f} ms per update, stats {m.esdf_stats()}")
PY
