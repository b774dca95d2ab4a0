#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TSL_LIB=$GRAFT_REPO_ROOT/taichislam_amd/lib/libtaichislam_hip_timing.so timeout 120 python tools/esdf_probe.py 1.0 2>&1 | grep -v "amdgpu.ids\|initialized" | tee $O/r2k_esdf.log
timeout 600 python -m pytest tests/test_fusion_mesh_gpu.py tests/test_golden.py tests/test_texture_gpu.py tests/test_esdf_gpu.py -m gpu -q 2>&1 | tail -8 | tee $O/r2k_pytest.log
timeout 200 python - << 'PY' 2>&1 | grep -v "amdgpu.ids\|initialized\|Total tri" | tee $O/r2k_mesh.log
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from taichislam_amd.mapping import DenseTSDF, MarchingCubeMesher
from taichislam_amd.utils import synthetic as syn
from taichislam_amd import _lib
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
m = DenseTSDF(**C2); m.set_dep_camera_intrinsic(syn.K_DEPTH)
for R, T, d in syn.sphere_room_stream(30): m.recast_depth_to_map(R, T, d, None)
g = DenseTSDF(map_scale=[6.4, 6.4], voxel_scale=0.05, num_voxel_per_blk_axis=16); g.init_sphere(voxels=100, radius=1.5)
for name, mp, thres in (("C2 30 frames", m, 0.1), ("C1 sphere 128^3", g, 0.1)):
    for gather in (1, 0):
        mp.set_option("mesh_gather", gather)
        ms = MarchingCubeMesher(mp, 4000000, tsdf_surface_thres=thres)
        ms.generate_mesh(1); mp.sync()
        mp.enable_profiling(True, only=[_lib.K_MESH])
        for _ in range(10): ms.generate_mesh(1)
        t, n = mp.kernel_time(_lib.K_MESH); mp.enable_profiling(False)
        print(f"{name}: gather={gather} k_marching_cubes {1000*t/n:.1f} us per launch, {ms.num_facelets[None]} triangles, {mp.count_active()} voxels")
PY
