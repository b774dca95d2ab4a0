#!/usr/bin/env python3
"""bench.py -- depth-frames/s integrated (640x480 -> 512^3 TSDF @ 2 cm) on MI355X, with the HBM roofline fraction of the dominant
kernel, the CPU restatement timed beside it and the measured deviation from the reference-literal semantics.

    python bench.py --gpus N --steps K --warmup W [--config 2|1|3|4]

A "step" is one pass of the hot path over one synthetic depth frame (BASELINE.json configs[1], the default --config 2): frames and
poses are generated up front and are resident in HBM before the timed region starts.  --gpus N > 1 launches one process per GPU (this
script re-executes itself under torch.distributed.run when it is not already running under it): each rank integrates its own stream
into its own submap -- the path shards by submap with no data-path collective, so scaling is weak -- and after
the timed region the ranks merge their submaps into one global map with ONE exchange over RCCL (configs[4]; reported as `merge`).
Rank 0 prints exactly one JSON line.  The other BASELINE configs (1: marching cubes on a 128^3 sphere, 3: Octomap 1024^3 / 5 cm,
4: TSDF + incremental ESDF per frame + mesh every 10th frame) print their own line with --config; they are single-GPU."""
import argparse
import gc
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0,
          min_ray_length=0.3, internal_voxels=10, recast_step=2, texture_enabled=False)
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable copy
WORKLOAD = ("BASELINE configs[1]: single 640x480 u16 depth stream (sphere room r=3 m, 1 deg/frame) -> DenseTSDF 512^3 / 2 cm, "
            "recast_step 2, max_ray 5 m; one stream+submap per GPU")


# ---- CPU legs (the oracle is only the measured baseline / the checker here, never the product path) --------------------------------
def cpu_baselines(frames, budget_s=10.0):
    """Time the CPU restatement of the reference on this box's host cores for a bounded sample of the same stream: FAITHFUL on one
    thread (sequential f16 replay = the restatement proper) and the brick-binned OpenMP port of the same sums on every core.
    Returns (one-thread dict, all-core dict, FAITHFUL oracle handle, frames it integrated)."""
    from oracle import FAITHFUL, OracleTSDF
    from taichislam_amd.utils import synthetic as syn
    o = OracleTSDF(**C2)
    o.set_intrinsics(syn.K_DEPTH)
    n, t0 = 0, time.perf_counter()
    for R, T, d in frames:
        o.integrate_depth(R, T, d, mode=FAITHFUL)
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    one = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"first {n} frames of the same 640x480 stream into the same 512^3/2cm map, {dt:.1f} s, 1 thread, oracle FAITHFUL mode "
                     f"(sequential f16 replay of dense_tsdf.py:188-270; Taichi itself unavailable)"}
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)      # (bench.py binds itself to one NUMA node)
    p = OracleTSDF(**C2)
    p.set_intrinsics(syn.K_DEPTH)
    p.integrate_depth_mt(*frames[0], ncpu)                 # first touch of the thread pool and the bricks
    # the port has serial stretches (pixel accumulate, counting sort, brick allocation): more threads than it can feed only add
    # fork/join cost, so the thread count is swept on three frames each and the best one is used for the sample
    trial, f0 = {}, 1
    for nt in sorted({min(ncpu, c) for c in (8, 32, 128, ncpu)}):
        t0 = time.perf_counter()
        for R, T, d in frames[f0:f0 + 3]:
            p.integrate_depth_mt(R, T, d, nt)
        trial[nt] = 3.0 / (time.perf_counter() - t0)
        f0 += 3
    cores = max(trial, key=trial.get)
    m, t0 = 0, time.perf_counter()
    for R, T, d in frames[f0:]:
        p.integrate_depth_mt(R, T, d, cores)
        m += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt2 = time.perf_counter() - t0
    allc = {"value": m / max(dt2, 1e-9), "unit": "frames/s", "cores": cores, "kind": "port", "host_threads": ncpu,
            "frames_per_s_by_threads": {str(k): v for k, v in trial.items()},
            "sample": f"{m} frames of the same stream, {dt2:.1f} s, {cores} of {ncpu} OpenMP threads (best of the sweep): brick-binned BATCHED sums, bit-identical to the one-thread BATCHED map"}
    return one, allc, o, n


def parity_vs_faithful(dev, frames, oracle_map):
    """HIP map of the frames the FAITHFUL baseline just integrated vs that map: the measured deviation (taichislam_amd.utils.parity)."""
    from taichislam_amd.mapping import DenseTSDF
    from taichislam_amd.utils import parity
    from taichislam_amd.utils import synthetic as syn
    g = DenseTSDF(**C2, device=dev)
    g.set_dep_camera_intrinsic(syn.K_DEPTH)
    for R, T, d in frames:
        g.recast_depth_to_map(R, T, d, None)
    rep = parity.deviation_report(g.export_submap(), oracle_map.export_sparse(), C2["voxel_scale"], sensor_xyz=frames[-1][1])
    out = parity.short_summary(rep)
    out["frames"] = len(frames)
    out["note"] = ("HIP == oracle BATCHED bit for bit (tests); this is HIP vs the reference-literal sequential f16 replay (oracle FAITHFUL). "
                   "Growth with the stream length, fusion and mesh deviation: profiles/r05_parity_vs_faithful.json; the reference's own schedule-to-schedule spread: profiles/r04_parity_envelope.json")
    return out


def cpu_baseline_config(config, budget_s=10.0):
    """The CPU restatement timed on one host core for a bounded sample of configs 1 / 3 / 4 (the oracle is the measured baseline here, nothing else)."""
    from oracle import FAITHFUL, OracleOctomap, OracleTSDF
    from taichislam_amd.utils import synthetic as syn
    if config == 1:
        o = OracleTSDF(map_scale=[6.4, 6.4], voxel_scale=0.05, num_voxel_per_blk_axis=16)
        r = np.arange(-50, 50, dtype=np.int16)
        ii, jj, kk = np.meshgrid(r, r, r, indexing="ij")
        idx = np.stack([ii, jj, kk], -1).reshape(-1, 3)
        p = idx.astype(np.float32) * np.float32(0.05)
        t = (np.sqrt((p * p).sum(1)) - np.float32(1.5)).astype(np.float16)
        o.import_sparse(0, idx, t, np.ones(len(t), np.float16), np.zeros(len(t), np.int8))
        n, t0 = 0, time.perf_counter()
        while True:
            tri = o.generate_mesh(1, 0.1, 1000000)[3]
            n += 1
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
        return {"value": n / dt, "unit": "meshes/s", "cores": 1, "kind": "port", "sample": f"{n} meshes of the same 128^3 sphere map ({tri} triangles), {dt:.1f} s, 1 thread, "
                "CPU restatement of marching_cube_mesher.py:127-187"}
    frames = list(syn.sphere_room_stream(40))
    if config == 3:
        o = OracleOctomap(map_scale=[51.2, 51.2], voxel_scale=0.05, K=2, max_ray_length=5.0, max_submap_num=4)
        o.set_intrinsics(syn.K_DEPTH)
        n, t0 = 0, time.perf_counter()
        for R, T, d in frames:
            o.integrate_depth(R, T, d)
            n += 1
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
        return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port", "sample": f"first {n} frames of the same stream into the same Octomap, {dt:.1f} s, 1 thread, "
                "CPU restatement of taichi_octomap.py:116-169"}
    o = OracleTSDF(**C2)
    o.set_intrinsics(syn.K_DEPTH)
    n, t0 = 0, time.perf_counter()
    for f, (R, T, d) in enumerate(frames):
        o.integrate_depth(R, T, d, mode=FAITHFUL)
        o.esdf(max_dist=1.0)                                   # the restatement has no incremental form: a full Dijkstra per frame
        if f % 10 == 9:
            o.generate_mesh(1, 5 * C2["voxel_scale"], 4000000)
        n += 1
        if time.perf_counter() - t0 > 2 * budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port", "sample": f"first {n} frames of the same stream: FAITHFUL integration + full ESDF recompute (Dijkstra over the "
            f"observed voxels; the CPU restatement has no incremental update) per frame + mesh every 10th, {dt:.1f} s, 1 thread"}


def sequential_leg(dev, frames, oracle_map, more=()):
    """Option semantics = 1 (csrc/tsl_sequential.hip): the reference-literal sequential replay on the GPU, for the frames the FAITHFUL
    baseline integrated -- its rate, and whether the map equals the FAITHFUL map bit for bit."""
    import torch
    from taichislam_amd import _lib
    from taichislam_amd.mapping import DenseTSDF
    from taichislam_amd.utils import synthetic as syn
    g = DenseTSDF(**C2, device=dev)
    g.set_dep_camera_intrinsic(syn.K_DEPTH)
    g.set_option("semantics", 1)
    dd = [torch.from_numpy(d.view(np.int16)).cuda(dev) for _, _, d in frames]
    g.recast_depth_to_map(frames[0][0], frames[0][1], dd[0], None)      # first touch: scratch allocation
    g.sync()
    t0 = time.perf_counter()
    for (R, T, _), d in zip(frames[1:], dd[1:]):
        g.recast_depth_to_map(R, T, d, None)
    g.sync()
    dt = time.perf_counter() - t0
    a, b = g.export_submap(), oracle_map.export_sparse()
    # the steady rate: the stream goes on (`more`: the frames behind the ones the CPU leg integrated) on the same map, behind a full pipeline
    more = list(more)
    dm = [torch.from_numpy(d.view(np.int16)).cuda(dev) for _, _, d in more]
    g.sync()
    g.enable_profiling(True, only=[_lib.K_INTEGRATE, _lib.K_RAYS, _lib.K_SORT])
    t1 = time.perf_counter()
    for (R, T, _), d in zip(more, dm):
        g.recast_depth_to_map(R, T, d, None)
    g.sync()
    dts = time.perf_counter() - t1
    chain = None
    try:      # what bounds the mode: the replay of a batch is as long as the chain of its most-visited voxel (the one next to the sensor)
        kt = {k: g.kernel_time(i) for k, i in (("replay", _lib.K_INTEGRATE), ("group", _lib.K_RAYS), ("rank_sort", _lib.K_SORT))}
        run = g.get_option("seq_longest_run")
        rep_us = 1000.0 * kt["replay"][0] / max(1, kt["replay"][1])
        chain = {"replay_us_per_batch": rep_us, "replay_launches": kt["replay"][1], "group_us_per_batch": 1000.0 * kt["group"][0] / max(1, kt["group"][1]),
                 "rank_sort_us_per_batch": 1000.0 * kt["rank_sort"][0] / max(1, kt["rank_sort"][1]),
                 "longest_voxel_run_updates_last_batch": run, "ns_per_update_if_the_chain_is_the_launch": 1000.0 * rep_us / max(1, run),
                 "note": "replay / grouping / rank-sort time per batch of eight frames (HIP events), DESIGN.md section 4"}
    except Exception as e:
        chain = {"error": repr(e)[:200]}
    g.enable_profiling(False)

    def srt(e):
        i = e["indices"].astype(np.int64)
        o = np.argsort(((i[:, 0] + 32768) << 32) | ((i[:, 1] + 32768) << 16) | (i[:, 2] + 32768))
        return e["indices"][o], np.asarray(e["TSDF"])[o].view(np.uint16), np.asarray(e["W_TSDF"])[o].view(np.uint16), e["occupy"][o]
    x, y = srt(a), srt(b)
    exact = all(u.shape == v.shape and np.array_equal(u, v) for u, v in zip(x, y))
    names = ("indices", "TSDF", "W_TSDF", "occupy")
    where = None if exact else first_difference(dict(zip(names, x)), dict(zip(names, y)), names[1:], sensor_xyz=frames[-1][1], voxel_scale=C2["voxel_scale"])
    rate, steady = (len(frames) - 1) / max(dt, 1e-9), (len(more) / max(dts, 1e-9) if len(more) >= 64 else None)
    return {"value": rate, "unit": "frames/s", "frames": len(frames), "voxels": int(x[0].shape[0]),
            "bit_exact_with_oracle_FAITHFUL": bool(exact), "differences": where,
            "value_steady": steady, "steady_frames": len(more), "chain_floor": chain,
            "north_star": {"rate_target_frames_per_s": 2000, "rate_met": bool(min(rate, steady if steady else rate) >= 2000.0), "tsdf_tolerance": "1e-4 relative",
                           "tolerance_met": bool(exact), "how": "every TSDF / W bit equals the reference's struct-for serialisation (oracle FAITHFUL, pinned to the "
                           "reference's own source by tests/golden/ref_*.npz)"},
            "note": "tsl_tsdf_set_option(semantics, 1): rays in struct-for order, every ray step applied on its own in f16 with the W clamp (dense_tsdf.py:264-267); "
                    "`value` = these frames behind an empty pipeline, `value_steady` = the frames that follow them, same map"}


def default_exact_leg(dev, frames):
    """The benchmarked default path against the oracle's BATCHED mode (its definition), bit for bit, on the first frames of the stream -- queued back to
    back as host images and as device tensors, one synchronisation at the end, so phase A of a batch really runs beside phase B of the batch before."""
    import torch
    from oracle import BATCHED, OracleTSDF
    from taichislam_amd.mapping import DenseTSDF
    from taichislam_amd.utils import synthetic as syn
    o = OracleTSDF(**C2)
    o.set_intrinsics(syn.K_DEPTH)
    for R, T, d in frames:
        o.integrate_depth(R, T, d, mode=BATCHED)

    def srt(e):
        i = e["indices"].astype(np.int64)
        k = np.argsort(((i[:, 0] + 32768) << 32) | ((i[:, 1] + 32768) << 16) | (i[:, 2] + 32768))
        return {"indices": e["indices"][k], "TSDF": np.asarray(e["TSDF"])[k].view(np.uint16), "W_TSDF": np.asarray(e["W_TSDF"])[k].view(np.uint16), "occupy": e["occupy"][k]}
    want, res, where = srt(o.export_sparse()), {}, {}
    dd = [torch.from_numpy(d.view(np.int16)).cuda(dev) for _, _, d in frames]
    for kind in ("device", "host"):
        g = DenseTSDF(**C2, device=dev)
        g.set_dep_camera_intrinsic(syn.K_DEPTH)
        for (R, T, d), t in zip(frames, dd):
            g.recast_depth_to_map(R, T, t if kind == "device" else d, None)
        got = srt(g.export_submap())
        res[kind] = all(got[k].shape == want[k].shape and np.array_equal(got[k], want[k]) for k in want)
        if not res[kind]:
            where[kind] = first_difference(got, want, ("TSDF", "W_TSDF", "occupy"), sensor_xyz=frames[-1][1], voxel_scale=C2["voxel_scale"])
    return {"frames": len(frames), "voxels": int(want["indices"].shape[0]), "bit_exact_with_oracle_BATCHED": all(res.values()), "per_input": res, "differences": where or None,
            "note": "semantics = 0 (the path `value` measures) == oracle BATCHED on every index, TSDF / W bit and occupancy byte; frames queued back to back"}


def configs_leg(dev, host, cpu=True):
    """BASELINE.json's other single-GPU configurations as compact sub-legs of the default line, so that the driver's one command puts all of `configs` under
    its clock: [0] marching cubes on the 128^3 sphere, [2] the stream into the Octomap 1024^3 / 5 cm, [3] TSDF + incremental ESDF every
    frame + a mesh every 10th.  The full-length forms with their own lines: --config 1 / 3 / 4."""
    from taichislam_amd.utils import bench_configs
    out = {}
    for key, config, steps, warmup in (("c1_marching_cubes_128", 1, 200, 20), ("c3_octomap_1024", 3, 100, 10), ("c4_tsdf_esdf_mesh", 4, 60, 10)):
        try:
            t0 = time.perf_counter()
            line = bench_configs.run(config, steps, warmup, dev, host=host)
            r = line.get("roofline") or {}
            out[key] = {"metric": line["metric"], "value": line["value"], "unit": line["unit"], "steps": steps, "warmup": warmup, "ms_per_step": line["ms_per_step"],
                        "roofline": {"kernel": r.get("kernel"), "bound": "hbm", "achieved": r.get("achieved"), "peak": r.get("peak"), "unit": "GB/s", "frac": r.get("frac"),
                                     "algorithmic_bytes_per_launch": r.get("algorithmic_bytes_per_launch"), "avg_launch_us": r.get("avg_launch_us")},
                        "detail": {k: v for k, v in line["config"].items() if k not in ("workload", "frame_stats")}, "leg_wall_s": time.perf_counter() - t0}
            if cpu:
                out[key]["cpu_baseline"] = cpu_baseline_config(config, budget_s=1.5)
        except Exception as e:
            out[key] = {"error": repr(e)[:200]}
    # configs[3] once more with the ESDF update as the raise / lower wavefront with parent directions (option esdf_mode = 1; exact as well: tests/test_esdf_gpu.py)
    try:
        os.environ["TSL_C4_OPTS"] = (os.environ.get("TSL_C4_OPTS", "") + " esdf_mode=1").strip()
        line = bench_configs.run(4, 40, 10, dev, host=host)
        out["c4_esdf_wavefront"] = {"value": line["value"], "unit": line["unit"], "steps": 40, "esdf_ms_per_update": line["config"].get("esdf_ms_per_update"),
                                    "esdf_voxel_pushes_per_update": line["config"].get("esdf_voxel_pushes_per_update"),
                                    "esdf_region_bricks_per_update": line["config"].get("esdf_region_bricks_per_update")}
    except Exception as e:
        out["c4_esdf_wavefront"] = {"error": repr(e)[:200]}
    finally:
        os.environ["TSL_C4_OPTS"] = os.environ.get("TSL_C4_OPTS", "").replace("esdf_mode=1", "").strip()
    return out


def reference_source_leg(dev):
    """tests/golden/ref_*.npz -- what the reference's OWN source produces when run on the sequential Taichi stand-in of tools/ti_seq (generated on the dev
    box by tools/gen_ref_golden.py; the reference tree is not needed here).  Checked live: the oracle's FAITHFUL mode and the HIP path with semantics = 1
    against the six integration vectors (depth stream, point clouds, weight clamp, depth + colour, coloured point clouds, two submaps made of 10^3 blocks) and
    the two fused global maps (tilted and axis-aligned base poses), bit for bit.  (The whole set, fusion / Octomap / mesh / exports / session included: tests/test_ref_golden.py.)"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
    import test_ref_golden as tr
    from oracle import FAITHFUL
    names, ora_ok, hip_ok, voxels, detail = ["depth_stream", "point_clouds", "weight_clamp", "textured", "textured_points", "blk10_two_submaps", "two_submaps_fused", "aligned_submap_fused"], {}, {}, 0, {}
    for name in names:
        cfg, K, Kc, steps, want = tr.load(name)
        keys = ("indices", "TSDF", "W_TSDF", "occupy") + (("color",) if "color" in want else ())
        same = lambda got: all(got[k].shape == want[k].shape and np.array_equal(got[k], want[k]) for k in keys)
        ora_ok[name] = bool(same(tr.replay(lambda over: tr._Ora({**cfg, **over}, K, Kc), steps, K, Kc, {"mode": FAITHFUL}, lambda g, m: g.o.fuse_submaps(m.o, mode=FAITHFUL))))
        got = tr.replay(lambda over: tr._Hip({**cfg, **over, "device": dev}, K, Kc, 1), steps, K, Kc, {}, lambda g, m: g.m.fuse_submaps(m.m))
        hip_ok[name] = bool(same(got))
        if not hip_ok[name]:
            detail[name] = first_difference(got, want, keys[1:])
        voxels += int(want["indices"].shape[0])
    return {"vectors": names, "voxels": voxels, "oracle_FAITHFUL_bit_exact": all(ora_ok.values()), "hip_semantics_1_bit_exact": all(hip_ok.values()),
            "per_vector": {n: {"oracle_FAITHFUL": ora_ok[n], "hip_semantics_1": hip_ok[n]} for n in names}, "differences": detail or None,
            "note": "golden maps made by the reference's own source on tools/ti_seq (not by Taichi itself)"}


def first_difference(got, want, keys, sensor_xyz=None, voxel_scale=None):
    """Where two sorted sparse exports differ: voxel counts, or per field the number of differing voxels, the first one and -- with a sensor position --
    how far from the sensor the differing voxels lie (a false exactness flag has to say where)."""
    if got["indices"].shape != want["indices"].shape or not np.array_equal(got["indices"], want["indices"]):
        return {"voxel_sets_differ": True, "voxels_got": int(got["indices"].shape[0]), "voxels_want": int(want["indices"].shape[0])}
    out = {}
    for k in keys:
        bad = np.nonzero(np.atleast_1d((got[k] != want[k]).reshape(got[k].shape[0], -1).any(axis=1)))[0]
        if bad.size:
            idx = want["indices"][bad].astype(np.int64)
            out[k] = {"differing_voxels": int(bad.size), "first_index": idx[0].tolist(), "got_bits": np.atleast_1d(got[k][bad[0]]).tolist(), "want_bits": np.atleast_1d(want[k][bad[0]]).tolist()}
            if sensor_xyz is not None and voxel_scale:
                d = np.linalg.norm(idx * float(voxel_scale) - np.asarray(sensor_xyz, dtype=np.float64)[None, :], axis=1)
                out[k]["distance_from_sensor_m_min_median_max"] = [float(d.min()), float(np.median(d)), float(d.max())]
    return out


def stored_counters():
    """profiles/r06_traffic.json -- PMC counters of separate rocprofv3 passes over the driver's command (tools/gpu_profiles_r05.sh) -- or None when
    the file was collected on OTHER kernels: it carries the hash of the kernel sources it was measured on (taichislam_amd.build.source_hash), and
    a line never replays counters of kernels that have changed since."""
    path = os.path.join(ROOT, "profiles", "r06_traffic.json")
    if not os.path.exists(path):
        return None, "profiles/r06_traffic.json is missing"
    try:
        tj = json.load(open(path))
        from taichislam_amd import build
        have = build.source_hash()
        if tj.get("lib_source_hash") != have:
            return None, f"profiles/r06_traffic.json was collected on kernel sources {tj.get('lib_source_hash')}, these are {have}: counters not replayed"
        return tj, None
    except Exception as e:
        return None, repr(e)[:200]


def bind_to_gpu_numa_node():
    """Bind this process to the CPUs of the NUMA node its pinned host allocations land on (the driver prefers one node per GPU): the host side of the
    host-input legs -- a strided copy out of the caller's image -- runs at local- or remote-memory speed depending on where the scheduler happens to have put
    the thread, 1.7x apart on the two-socket box (profiles/r06_pinned_probe.txt).  Returns what was done, for the line."""
    try:
        import re
        import torch
        t = torch.empty(1 << 20, dtype=torch.uint8).pin_memory()
        t.fill_(1)
        addr, best = t.data_ptr(), None
        for ln in open("/proc/self/numa_maps"):
            a = int(ln.split()[0], 16)
            if a <= addr and (best is None or a > best[0]):
                best = (a, ln)
        nodes = re.findall(r"N(\d+)=(\d+)", best[1]) if best else []
        if not nodes:
            return {"bound": False, "why": "no node in numa_maps"}
        node = max(nodes, key=lambda x: int(x[1]))[0]
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return {"bound": False, "why": "no allowed cpu on that node"}
        os.sched_setaffinity(0, cpus)
        return {"bound": True, "node": int(node), "cpus": len(cpus)}
    except Exception as e:
        return {"bound": False, "why": repr(e)[:120]}


def relaunch(args):
    """--gpus N > 1 from a bare shell: one rank per GPU under torch.distributed.run on this node."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    # the contract is ONE JSON line on stdout: everything the shims print on the way (the reference's own progress prints) goes to stderr
    real_stdout, sys.stdout = sys.stdout, sys.stderr

    def emit(obj):
        real_stdout.write(json.dumps(obj) + "\n"); real_stdout.flush()

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 3, 4], help="BASELINE.json configs index + 1 (2 = the headline metric)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="backend option for A/B runs (tsl_tsdf_set_option)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--merge", action="store_true", help="also run the configs[4] global-map merge on one GPU (always on when --gpus > 1)")
    ap.add_argument("--as-rank", type=int, default=None, metavar="R", help="dry run of the multi-rank branch on ONE GPU: behave as rank R of --of N (stream offset, "
                    "submap id, pose table of N submaps, merge leg) without any collective; the line is marked dry_run and its value is this rank's alone")
    ap.add_argument("--of", type=int, default=8, metavar="N", help="world size of the --as-rank dry run")
    ap.add_argument("--bursts", type=int, default=7, help="timed regions of K steps; `value` is their median (1 = the single region of the contract)")
    ap.add_argument("--steady", type=int, default=300, help="frames of the steady-state leg behind the contract region (0 = off)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch(args))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    dry = args.as_rank is not None
    if dry:
        if distributed or args.gpus != 1 or not (0 <= args.as_rank < args.of):
            raise SystemExit("--as-rank R --of N is a single-process dry run: --gpus 1, 0 <= R < N")
        rank, world = args.as_rank, args.of             # every rank-dependent line below runs as that rank; `distributed` stays False
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if distributed and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = local_rank if distributed else 0
    torch.cuda.set_device(dev)
    host_numa = bind_to_gpu_numa_node()            # before any data is made: images, poses and the library's staging buffers are first touched from that node
    if args.config != 2:
        if distributed:
            raise SystemExit("--config 1/3/4 are single-GPU measurements")
        from taichislam_amd.utils import bench_configs
        line = bench_configs.run(args.config, args.steps, args.warmup, dev)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_config(args.config)
        tall, why = stored_counters()
        if tall is None and line.get("roofline"):
            line["roofline"]["traffic_source"] = why
        if tall is not None and line.get("roofline"):
            try:
                wave = args.config == 4 and "esdf_mode=1" in os.environ.get("TSL_C4_OPTS", "")
                tj = tall.get(f"config{args.config}" + ("_wavefront" if wave else ""))
                if tj:
                    line["roofline"]["traffic"] = tj["hbm_bytes_per_launch"] / tj.get("frames_per_launch", 1.0)      # (config 3: eight queued frames per launch, the line's unit is a frame)
                    line["roofline"]["traffic_source"] = tj.get("command")
            except Exception:
                pass
        emit(line)
        return

    from taichislam_amd import _lib
    from taichislam_amd.mapping import DenseTSDF
    from taichislam_amd.utils import synthetic as syn

    # ---- synthetic stream: each rank starts 45 degrees further round the room ----
    nframes = args.warmup + args.steps
    from taichislam_amd import distributed as D
    nsteady = args.steady if (args.steady > 0 and (rank == 0 or dry)) else 0
    ngen = nframes + max(0, nsteady - args.steps)            # the steady-state leg goes over the timed frames again and on along the stream
    host = list(syn.sphere_room_stream(ngen, start_deg=D.stream_start_deg(rank)))
    depth_dev = torch.from_numpy(np.stack([d for _, _, d in host]).view(np.int16)).cuda(dev)   # resident in HBM
    poses = [(np.ascontiguousarray(R), np.ascontiguousarray(T)) for R, T, _ in host]

    nsub = max(8, world) + 1
    m = DenseTSDF(**C2, device=dev, max_submap_num=nsub, max_bricks=32768)
    m.set_dep_camera_intrinsic(syn.K_DEPTH)
    # config 5: rank r integrates into submap id r whose base pose is its first camera pose
    m.active_submap_id[None] = rank
    m.set_base_pose_submap(rank, poses[0][0], poses[0][1])
    for kv in args.opt:
        k, v = kv.split("=")
        if not k.startswith("fuse_"):
            m.set_option(k, int(v))

    frames_dev = [depth_dev[f] for f in range(ngen)]        # one [480, 640] view per frame, made before the timed region

    def step(f):
        R, T = poses[f]
        m.recast_depth_to_map(R, T, frames_dev[f], None)

    for f in range(args.warmup):
        step(f)
    m.sync()
    # timed region: only the dominant kernel is bracketed by HIP events (one pair per batch of its launches)
    if not os.environ.get("TSL_BENCH_NOPROF"):      # developer A/B: what the event pair around every phase-B launch costs
        m.enable_profiling(True, only=[_lib.K_INTEGRATE])

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        m.sync()

    # Python's cyclic collector is paused for the timed regions (as timeit does): a full collection of this process (torch, numpy, the
    # frame lists) takes ~35 ms, two thirds of a 300-frame run, and would land in them at random
    def timed_burst():
        """the contract's timed region: EXACTLY K steps behind W warm-up steps, barrier + synchronise on both sides, MAX over ranks"""
        barrier()
        gc.collect(); gc.disable()
        t0 = time.perf_counter()
        for f in range(args.warmup, nframes):
            step(f)
        m.sync()
        barrier()
        dt = time.perf_counter() - t0
        gc.enable()
        dt_own = dt
        if distributed:
            t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{dev}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, dt_own

    # `value` is the MEDIAN of `--bursts` such regions (one 1 ms burst of 20 frames has +-8 % box noise, more than a kernel change moves it): the
    # first is the contract's region as it always was, the others repeat it on the same map over the same frames (the same rays, the same work), each one
    # behind a drained pipeline like the first.  Every one is listed in `burst_runs`; `ms_per_step` x `steps` is the median burst.
    bursts = [timed_burst() for _ in range(max(1, args.bursts))]
    order = sorted(range(len(bursts)), key=lambda i: bursts[i][0])
    dt, dt_rank = bursts[order[(len(bursts) - 1) // 2]]
    dom_ms, dom_n = m.kernel_time(_lib.K_INTEGRATE)
    stats = m.last_frame_stats()
    # second, untimed pass over some of the same frames with every kernel bracketed: the per-kernel breakdown
    m.enable_profiling(True)
    for f in range(args.warmup, min(nframes, args.warmup + 60)):
        step(f)
    m.sync()
    kern = {}
    for kid, name in _lib.KERNEL_NAMES.items():
        ms, n = m.kernel_time(kid)
        if n:
            kern[name] = {"avg_us": 1000.0 * ms / n, "launches": n}
    m.enable_profiling(False)
    if dom_n:
        kern["integrate"] = {"avg_us": 1000.0 * dom_ms / dom_n, "launches": dom_n, "measured_in": "timed region"}

    # ---- the same frames handed over as HOST buffers, as the reference API does (H2D copy inside the timed loop; never `value`) ----
    host_rates = None
    if rank == 0:
        nsrc = min(args.steps, 200)                        # distinct host images
        nh = max(nsrc, 300)                                # calls: a stream, like value_steady, not a burst (the images are handed over again and again)
        pinned = torch.from_numpy(np.stack([host[args.warmup + i][2] for i in range(nsrc)]).view(np.int16)).pin_memory().numpy().view(np.uint16)
        host_rates = {}
        for label, src in (("pageable", [host[args.warmup + i][2] for i in range(nsrc)]), ("pinned", [pinned[i] for i in range(nsrc)])):
            m.sync()
            gc.collect(); gc.disable()
            th = time.perf_counter()
            for i in range(nh):
                R, T = poses[args.warmup + i % nsrc]
                m.recast_depth_to_map(R, T, src[i % nsrc], None)
            m.sync()
            host_rates[label] = nh / (time.perf_counter() - th)
            gc.enable()
        host_rates["note"] = f"{nh} calls over {nsrc} images: a 614 kB uint16 numpy image per call through tsl_tsdf_integrate_depth (taichislam_node.py:381-382), rank 0 only"

    # ---- steady state, driver-visible: >= 300 frames behind the contract region, same map, same clock (the contract's K may be a 20-frame
    #      burst, which is dominated by filling and draining the batch pipeline) ----
    steady = None
    if nsteady > 0:
        m.sync()
        gc.collect(); gc.disable()
        ts = time.perf_counter()
        for i in range(nsteady):
            step(args.warmup + i)
        m.sync(); torch.cuda.synchronize()
        steady_dt = time.perf_counter() - ts
        gc.enable()
        steady = {"value": nsteady / steady_dt, "unit": "frames/s", "frames": nsteady,
                  "note": "same handle, same map, frames resident in HBM, timed like `value` (sync on both sides); rank 0 only"}

    per_rank = None
    if distributed:
        tr = torch.zeros(world, dtype=torch.float64, device=f"cuda:{dev}")
        tr[rank] = args.steps / dt_rank
        dist.all_reduce(tr, op=dist.ReduceOp.SUM)
        per_rank = [float(x) for x in tr.tolist()]

    # ---- configs[4]: ONE exchange at merge time, outside the timed region.  The multi-rank form of this leg cannot be exercised
    #      on the one-GPU boxes this was developed on, so it runs under a watchdog: if it does not finish, the line is still printed
    #      (merge = {"error": "timeout"}) and every rank leaves without further collectives ----
    merge_box = {"merge": None, "done": False}

    def merge_leg():
        torch.cuda.set_device(dev)                                   # the current device is per thread
        merge = None
        if True:
            ok = 1
            g = comm = None
            try:
                g = DenseTSDF(**C2, device=dev, is_global_map=True, max_submap_num=nsub, max_bricks=65536)
                for kv in args.opt:                                      # (A/B options of the global map's side: --opt fuse_direct=1)
                    if kv.startswith("fuse_"):
                        g.set_option(kv.split("=")[0], int(kv.split("=")[1]))
                for r in range(world):
                    Rb, Tb = syn.camera_pose(0, start_deg=D.stream_start_deg(r))
                    g.set_base_pose_submap(r, Rb, Tb)
                m.active_submap_id[None] = rank + 1                      # the rank's submap is closed (create_new_submap)
                if distributed:
                    uid = [D.Communicator.unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(uid, src=0)
                    comm = D.Communicator(uid[0], world, rank, device=dev)
            except Exception as e:                                       # every rank must agree before entering the collective
                ok, merge = 0, {"error": repr(e)[:300]}
            if distributed:
                flag = torch.tensor([ok], dtype=torch.int32, device=f"cuda:{dev}")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            if ok:
                try:
                    g.allreduce_merge(m, comm)                           # first call: scratch allocation, RCCL channel set-up
                    barrier()
                    tm = time.perf_counter()
                    nbytes = g.allreduce_merge(m, comm)
                    torch.cuda.synchronize()
                    merge = {"ms": 1000.0 * (time.perf_counter() - tm), "allreduce_bytes_per_rank": nbytes, "global_voxels": g.count_active(),
                             "global_bricks": g.bricks_in_use(), "ranks": world,
                             "note": "splat of the rank's submap + all-reduce(MAX) of the brick mask + all-reduce(SUM) of the packed union bricks + finalise, second call"}
                except Exception as e:
                    merge = {"error": repr(e)[:300]}
            elif merge is None:
                merge = {"error": "another rank failed to set the merge up"}
            if comm is not None:
                comm.close()

        merge_box["merge"] = merge
        merge_box["done"] = True

    merge_timed_out = False
    if distributed or args.merge or dry or not args.no_cpu_baseline:          # (the default N = 1 line carries configs[4]'s one-GPU form as well)
        import threading
        th = threading.Thread(target=merge_leg, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("TSL_BENCH_MERGE_TIMEOUT", "120")))
        merge_timed_out = not merge_box["done"]
    merge = {"error": "timeout: the merge leg did not finish (first multi-rank run on hardware?)"} if merge_timed_out else merge_box["merge"]

    parity_failed = False
    if rank == 0 or dry:
        fps = (1 if dry else world) * args.steps / dt
        # algorithmic bytes (SURVEY.md section 8d / DESIGN.md): phase A = 2*P_used + 24*P_valid, phase B = 9*U + V_pcl
        bytes_a = 2 * stats["p_used"] + 24 * stats["p_valid"]
        bytes_b = 9 * stats["unique"] + stats["v_pcl"]
        roof = None
        if "integrate" in kern:
            # one launch of the brick kernel integrates a whole batch of queued frames (up to 8): algorithmic bytes per launch =
            # 9 B per distinct voxel a frame updates (4 B read + 4 B + 1 B written) x the frames the launch covers
            fpl = args.steps * len(bursts) / max(1, kern["integrate"]["launches"])      # (the events were on in every burst)
            alg = 9 * stats["unique"] * fpl
            us = kern["integrate"]["avg_us"]
            ach = alg / (us * 1e-6) / 1e9
            traffic, traffic_src, valu = None, None, None
            tj, why = stored_counters()
            if tj is None:
                traffic_src = why
            else:
                try:
                    # the counters are averages per launch of the PMC command (its launches cover other frame counts than this run's): per frame, x this run's frames per launch
                    traffic = tj["integrate"]["hbm_bytes_per_launch"] / tj["integrate"]["frames_per_launch"] * fpl
                    traffic_src = tj.get("command") + f" (per frame there x {fpl:g} frames per launch here; kernel sources {tj.get('lib_source_hash')})"
                    # what actually bounds the kernel: its VALU issue rate.  SQ_INSTS_VALU (wave instructions per launch, PMC pass of the same
                    # command) x 4 cycles per wave64 instruction / (SIMDs x cycles of THIS run's average launch)
                    vi, fpl_p = tj["integrate"].get("valu_wave_insts_per_launch"), tj["integrate"].get("frames_per_launch")
                    if vi and fpl_p:
                        simds, mhz = 4 * 256, 2400.0
                        valu = {"valu_wave_insts_per_frame": vi / fpl_p, "cycles_per_inst": 4, "simds": simds, "clock_mhz": mhz,
                                "frac": (vi / fpl_p * fpl) * 4.0 / (simds * us * mhz)}
                except Exception:
                    traffic = None
            roof = {"bound": "hbm", "kernel": "tsl::k_integrate_batch", "frames_per_launch": fpl, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                    "valu_frac": valu["frac"] if valu else None, "valu": valu,
                    "algorithmic_bytes_per_launch": alg, "avg_launch_us": us,
                    "frame_bytes": bytes_a + bytes_b, "frame_gbs": (bytes_a + bytes_b) * fps / world / 1e9,
                    "frame_frac": (bytes_a + bytes_b) * fps / world / 1e9 / HBM_PEAK_GBS,
                    "note": "one 640x480 frame touches ~12 MB: the launch is bound by the exact per-step arithmetic and load imbalance, not by bytes (DESIGN.md section 4)"}
        out = {
            "metric": "depth-frames/s integrated (640x480->512^3 TSDF)", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16 storage / f32 + int64 fixed-point arithmetic",
            "data": "synthetic",
            "config": {"workload": WORKLOAD, "frame_stats": stats, "kernels_us": kern,
                       "updates_per_s": stats["steps"] * fps, "per_rank_frames_per_s": per_rank, "merge": merge},
            "roofline": roof,
            "burst_runs": {"frames_per_s": [(1 if dry else world) * args.steps / b[0] for b in bursts], "value_is": "median",
                           "note": "each: W warm-up steps done once, then EXACTLY K steps, barrier + sync on both sides; [0] is the first region behind the warm-up"},
            "value_steady": steady,
            "value_host_input": host_rates,
            "host_numa": host_numa,
        }
        if dry:
            out["dry_run"] = {"as_rank": rank, "of": world, "note": "single-process dry run of the multi-rank branch: this rank's stream offset, submap id and pose table, "
                              "merge leg without a communicator; `value` is this rank's rate alone"}
        if not args.no_cpu_baseline and world == 1:              # the CPU legs: rank 0 at N = 1 only (the other ranks of a multi-GPU run would wait at the barrier)
            sample = [(R, T, d) for R, T, d in host[: max(8, min(len(host), 200))]]
            one, allc, omap, n_done = cpu_baselines(sample)
            out["cpu_baseline"] = one
            out["cpu_baseline_allcore"] = allc
            try:
                out["parity_vs_faithful"] = parity_vs_faithful(dev, sample[:n_done], omap)
            except Exception as e:
                out["parity_vs_faithful"] = {"error": repr(e)[:200]}
            try:
                out["value_sequential"] = sequential_leg(dev, sample[:n_done], omap, more=host[n_done:n_done + 320])
            except Exception as e:
                out["value_sequential"] = {"error": repr(e)[:200]}
            try:
                out["reference_source_vectors"] = reference_source_leg(dev)
            except Exception as e:
                out["reference_source_vectors"] = {"error": repr(e)[:200]}
            try:
                out["configs"] = configs_leg(dev, host)
                out["configs"]["c5_merge_one_gpu"] = merge
            except Exception as e:
                out["configs"] = {"error": repr(e)[:200]}
            try:
                out["default_path_exact"] = default_exact_leg(dev, sample[:24])
            except Exception as e:
                out["default_path_exact"] = {"error": repr(e)[:200]}
            # ONE word for every live exactness check of this run, at the top level of the line, and a non-zero exit code behind the line when it is
            # false (a false flag three levels down stopped nobody)
            checks = {"value_sequential.bit_exact_with_oracle_FAITHFUL": out["value_sequential"].get("bit_exact_with_oracle_FAITHFUL"),
                      "reference_source_vectors.hip_semantics_1_bit_exact": out["reference_source_vectors"].get("hip_semantics_1_bit_exact"),
                      "reference_source_vectors.oracle_FAITHFUL_bit_exact": out["reference_source_vectors"].get("oracle_FAITHFUL_bit_exact"),
                      "default_path_exact.bit_exact_with_oracle_BATCHED": out["default_path_exact"].get("bit_exact_with_oracle_BATCHED")}
            out["parity_checks"] = checks
            out["parity_ok"] = all(v is True for v in checks.values())
            parity_failed = not out["parity_ok"]
            # the parity-qualified rate beside `value`: semantics = 1 is the mode that is bit-exact with the reference's struct-for schedule
            vs = out["value_sequential"]
            out["value_conforming"] = {"value": vs.get("value_steady") or vs.get("value"), "unit": "frames/s", "mode": "semantics = 1 (reference-literal), steady stream",
                                       "bit_exact_with_oracle_FAITHFUL": vs.get("bit_exact_with_oracle_FAITHFUL")}
        # LAST in the line, so that a reader who only keeps the line's tail still has every rate
        def _v(d, *ks):
            for k in ks:
                d = d.get(k) if isinstance(d, dict) else None
            return d
        cf = out.get("configs") or {}
        out["summary"] = {"value_burst_median": fps, "burst_runs": out["burst_runs"]["frames_per_s"], "value_steady": _v(steady, "value"),
                          "value_sequential": {"value": _v(out, "value_sequential", "value"), "value_steady": _v(out, "value_sequential", "value_steady"),
                                               "bit_exact": _v(out, "value_sequential", "bit_exact_with_oracle_FAITHFUL")},
                          "value_host_input": {"pageable": _v(host_rates, "pageable"), "pinned": _v(host_rates, "pinned")},
                          "c1_meshes_per_s": _v(cf, "c1_marching_cubes_128", "value"), "c3_octomap_frames_per_s": _v(cf, "c3_octomap_1024", "value"), "c3_host_images_frames_per_s": _v(cf, "c3_octomap_1024", "detail", "host_input_frames_per_s"),
                          "c4_tsdf_esdf_mesh_frames_per_s": _v(cf, "c4_tsdf_esdf_mesh", "value"), "c4_esdf_ms_per_update": _v(cf, "c4_tsdf_esdf_mesh", "detail", "esdf_ms_per_update"),
                          "c4_wavefront": {"frames_per_s": _v(cf, "c4_esdf_wavefront", "value"), "esdf_ms": _v(cf, "c4_esdf_wavefront", "esdf_ms_per_update"), "voxel_writes": _v(cf, "c4_esdf_wavefront", "esdf_voxel_pushes_per_update")},
                          "c5_merge_one_gpu_ms": _v(merge, "ms"), "roofline_frac": _v(roof, "frac"), "cpu_baseline_frames_per_s": _v(out, "cpu_baseline", "value"),
                          "parity_ok": out.get("parity_ok")}
        emit(out)
    if merge_timed_out:
        sys.stderr.write("bench.py: merge leg timed out; leaving without further collectives\n"); real_stdout.flush()
        os._exit(0)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:
        sys.stderr.write("bench.py: a live exactness check FAILED (parity_ok false, see parity_checks in the line): exit code 3\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
