"""bench.py --config 1 / 3 / 4: the other single-GPU configurations of BASELINE.json, each as one JSON line in bench.py's format.

  1  tests/marching_cube_test.py geometry: analytic sphere TSDF in a 128^3 map, generate_mesh(1)           metric: meshes/s
  3  the 640x480 stream into Octomap 1024^3 / 5 cm (K = 2)                                                metric: depth-frames/s
  4  the stream into DenseTSDF 512^3 / 2 cm + incremental ESDF after EVERY frame + generate_mesh(1) with tsdf_surface_thres = 5 voxels
     every 10th frame (scripts/taichislam_node.py:209, :338)                                              metric: depth-frames/s

Byte models are SURVEY.md section 8d's: marching cubes 3*A + 72*Tri, Octomap 2*P_used + 8*P_valid, ESDF >= 8 bytes per cell relaxed.
Kernel times come from HIP events on the handle's stream inside the run where the handle has them (marching cubes, ESDF); the Octomap
kernel is timed by rocprofv3 (profiles/r02_octomap_kernel_stats.csv) and through the wall clock here."""
import gc
import os
import time

import numpy as np

HBM_PEAK_GBS = 8000.0
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0,
          min_ray_length=0.3, internal_voxels=10, recast_step=2, texture_enabled=False)


def _line(metric, value, unit, steps, warmup, ms_per_step, workload, extra, roofline):
    return {"metric": metric, "value": value, "unit": unit, "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16 storage / f32 arithmetic", "data": "synthetic",
            "config": dict({"workload": workload}, **extra), "roofline": roofline}


def _roof(kernel, alg_bytes, us, note):
    ach = alg_bytes / (us * 1e-6) / 1e9 if us else None
    return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS if ach else None,
            "traffic": None, "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": us, "note": note}


def run(config, steps, warmup, dev, host=None):
    """`host`: frames of the stream made by the caller (bench.py's default line runs the configurations as compact sub-legs over its own frames)"""
    import torch
    from taichislam_amd import _lib
    from taichislam_amd.mapping import DenseTSDF, MarchingCubeMesher, Octomap
    from taichislam_amd.utils import synthetic as syn
    if config == 1:
        g = DenseTSDF(map_scale=[6.4, 6.4], voxel_scale=0.05, num_voxel_per_blk_axis=16, device=dev)
        g.init_sphere(voxels=100, radius=1.5)
        ms = MarchingCubeMesher(g, 1000000)
        for _ in range(max(1, warmup)):
            ms.generate_mesh(1)
        g.sync()
        g.enable_profiling(True, only=[_lib.K_MESH])
        gc.collect(); gc.disable()      # as in bench.py: no ~35 ms full collection inside the timed region
        t0 = time.perf_counter()
        for _ in range(steps):
            ms.generate_mesh(1)
        g.sync()
        dt = time.perf_counter() - t0
        gc.enable()
        kms, kn = g.kernel_time(_lib.K_MESH)
        a, tri = g.count_active(), ms.num_facelets[None]
        us = 1000.0 * kms / kn
        return _line("marching-cubes meshes/s (128^3 sphere TSDF)", steps / dt, "meshes/s", steps, warmup, 1000.0 * dt / steps,
                     "BASELINE configs[0]: analytic sphere SDF r = 1.5 m in a 128^3 / 5 cm map (100^3 observed voxels), generate_mesh(1), count read back per call",
                     {"observed_voxels": a, "triangles": tri}, _roof("tsl::k_marching_cubes_lds", 3 * a + 72 * tri, us, "3 B per scanned voxel + 72 B per triangle"))
    nframes = warmup + steps
    host = list(host[:nframes]) if host is not None and len(host) >= nframes else list(syn.sphere_room_stream(nframes))
    depth_dev = torch.from_numpy(np.stack([d for _, _, d in host]).view(np.int16)).cuda(dev)
    if config == 3:
        oc = Octomap(map_scale=[51.2, 51.2], voxel_scale=0.05, K=2, max_ray_length=5.0, max_submap_num=4, device=dev)
        oc.set_dep_camera_intrinsic(syn.K_DEPTH)
        for f in range(warmup):
            oc.recast_depth_to_map(host[f][0], host[f][1], depth_dev[f], None)
        oc.sync()
        gc.collect(); gc.disable()      # as in bench.py: no ~35 ms full collection inside the timed region
        t0 = time.perf_counter()
        for f in range(warmup, nframes):
            oc.recast_depth_to_map(host[f][0], host[f][1], depth_dev[f], None)
        oc.sync()
        dt = time.perf_counter() - t0
        gc.enable()
        st = oc.last_frame_stats()
        # the same frames as host numpy images (the reference API's calling form): visited pixels copied into pinned, device-mapped slots, no copy call per frame
        nh = min(steps, 200)
        gc.collect(); gc.disable()
        th = time.perf_counter()
        for f in range(warmup, warmup + nh):
            oc.recast_depth_to_map(host[f][0], host[f][1], host[f][2], None)
        oc.sync()
        host_fps = nh / (time.perf_counter() - th)
        gc.enable()
        alg = 2 * st["p_used"] + 8 * st["p_valid"]
        return _line("depth-frames/s inserted (640x480 -> Octomap 1024^3 / 5 cm)", steps / dt, "frames/s", steps, warmup, 1000.0 * dt / steps,
                     "BASELINE configs[2]: the 640x480 sphere-room stream into Octomap(map_scale=[51.2, 51.2], voxel_scale=0.05, K=2, max_ray 5 m), recast_step 2",
                     {"frame_stats": st, "leaves": int(oc.export_leaves()[0].shape[0]), "host_input_frames_per_s": host_fps},
                     _roof("tsl::k_octo_depth", alg, 1e6 * dt / steps, "2 B per visited pixel + 8 B per valid point; the time is the whole per-frame wall time (eight queued frames per "
                           "launch): kernel-only duration in profiles/r06_c3_kernel_stats.csv"))
    if config == 4:
        m = DenseTSDF(**C2, device=dev)
        m.set_dep_camera_intrinsic(syn.K_DEPTH)
        mesher = MarchingCubeMesher(m, 4000000, tsdf_surface_thres=5 * C2["voxel_scale"])
        for kv in os.environ.get("TSL_C4_OPTS", "").split():          # developer A/B of backend options
            k, v = kv.split("="); m.set_option(k, int(v))
        md = float(os.environ.get("TSL_ESDF_MAX_DIST", "1.0"))      # developer sweep; the reported configuration is 1 m

        def step(f):
            m.recast_depth_to_map(host[f][0], host[f][1], depth_dev[f], None)
            m.update_esdf(max_dist=md, wait=False)       # the per-frame hook of dense_esdf.py:400-402: enqueued behind the frame
            if f % 10 == 9:
                mesher.generate_mesh(1)

        for f in range(warmup):
            step(f)
        m.sync()
        t_before = m.esdf_totals()
        m.enable_profiling(True, only=[_lib.K_ESDF, _lib.K_MESH])
        gc.collect(); gc.disable()      # as in bench.py: no ~35 ms full collection inside the timed region
        t0 = time.perf_counter()
        for f in range(warmup, nframes):
            step(f)
        tot = m.esdf_totals()           # waits for the ESDF updates still in flight
        m.sync()
        dt = time.perf_counter() - t0
        gc.enable()
        relax, pushes, region = (tot[k] - t_before[k] for k in ("brick_relaxations", "voxel_pushes", "region_bricks"))
        assert tot["updates"] - t_before["updates"] == steps
        ems, en = m.kernel_time(_lib.K_ESDF)
        mms, mn = m.kernel_time(_lib.K_MESH)
        a, tri = m.count_active(), mesher.num_facelets[None]
        cells = region / max(1, steps) * 4096
        return _line("depth-frames/s (TSDF 512^3 + incremental ESDF every frame + marching cubes every 10th)", steps / dt, "frames/s", steps, warmup,
                     1000.0 * dt / steps,
                     f"BASELINE configs[3]: the configs[1] stream; after every frame tsl_esdf_update(gamma = voxel, max_dist = {md:g} m, asynchronous), every 10th frame "
                     "generate_mesh(1) with tsdf_surface_thres = 5 voxels",
                     {"esdf_ms_per_update": ems / max(1, en), "esdf_brick_relaxations_per_update": relax / max(1, steps),
                      "esdf_voxel_pushes_per_update": pushes / max(1, steps), "esdf_region_bricks_per_update": region / max(1, steps),
                      "mesh_ms_per_call": mms / max(1, mn), "mesh_triangles": tri, "observed_voxels": a, "bricks": m.bricks_in_use(), "esdf_max_dist_m": md},
                     _roof(("tsl::k_esdf_wave" if m.get_option("esdf_mode") == 1 else "tsl::k_esdf_round") + " (all launches of one update)", 8 * cells, 1000.0 * ems / max(1, en),
                           "(4 + 4) B per cell of the bricks the update reaches; the update is a chain of dependent rounds (latency-bound), not a stream"))
    raise ValueError(config)
