"""The reference's own nondeterminism envelope -- CPU only, no GPU needed.

dense_tsdf.py:239 is a PARALLEL struct-for whose body updates TSDF / W with an unsynchronised f16 read-modify-write per ray step (:264-267):
every interleaving of the rays' step sequences is a legal outcome.  The oracle's FAITHFUL mode replays ONE of them (struct-for order, the
schedule tools/ti_seq executes and the GPU's literal mode reproduces bit for bit).  This tool replays the same stream under several legal
schedules (oracle/tsl_oracle.c, ora_tsdf_set_schedule): random ray orders, and P threads over contiguous shares of the struct-for order
advancing one step per turn -- and measures
  * how far two legal schedules are from each other (same statistics as the HIP-vs-FAITHFUL report, taichislam_amd/utils/parity.py),
  * the per-voxel envelope [min, max] over the schedules, and whether the ORDER-FREE map (oracle BATCHED == the default HIP path, bit for
    bit: tests/test_tsdf_parity_gpu.py) lies inside it, per distance from the sensor -- with the same question asked of every schedule
    against the envelope of the OTHER schedules as the calibration of what "inside" means.

python tools/parity_envelope.py [frames] [out.json]       (BASELINE configs[1] stream; ~1 minute on 8 cores for 77 frames)"""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
C2 = dict(map_scale=[10.24, 10.24], voxel_scale=0.02, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3, internal_voxels=10, recast_step=2)
SCHEDULES = [("struct_for", 0, 0, 0), ("random_rays_1", 1, 0, 1), ("random_rays_2", 1, 0, 2), ("random_rays_3", 1, 0, 3),
             ("threads_8", 2, 8, 0), ("threads_64", 2, 64, 0), ("threads_512", 2, 512, 0), ("threads_4096", 2, 4096, 0)]
DIST_BINS = (0.0, 0.1, 0.3, 1.0, 2.0, 3.0, 1e9)


def run(job):
    name, mode_name, kind, param, seed, n, checkpoints, cfg, small = job
    from oracle import BATCHED, FAITHFUL, IDEAL, OracleTSDF
    from taichislam_amd.utils import synthetic as syn
    mode = {"faithful": FAITHFUL, "batched": BATCHED, "ideal": IDEAL}[mode_name]
    o = OracleTSDF(**cfg)
    if small:
        K = syn.scaled_intrinsics(*small)
        frames = [(R, T, syn.sphere_room_depth(R, T, small[0], small[1], K=K)) for R, T in (syn.camera_pose(f) for f in range(n))]
    else:
        K = syn.K_DEPTH
        frames = list(syn.sphere_room_stream(n))
    o.set_intrinsics(K)
    o.set_schedule(kind, param, seed)
    out = {}
    for f, (R, T, d) in enumerate(frames):
        o.integrate_depth(R, T, d, mode=mode)
        if f + 1 in checkpoints:
            out[f + 1] = sorted_export(o.export_sparse())
    return name, out


def sorted_export(e):
    i = e["indices"].astype(np.int64)
    order = np.argsort(((i[:, 0] + 32768) << 32) | ((i[:, 1] + 32768) << 16) | (i[:, 2] + 32768), kind="stable")
    return {"indices": e["indices"][order], "TSDF": np.asarray(e["TSDF"])[order].view(np.uint16), "W_TSDF": np.asarray(e["W_TSDF"])[order].view(np.uint16),
            "occupy": e["occupy"][order]}


def envelope_stats(maps, batched, ideal, cams, vs):
    """maps: {schedule: export}; returns the JSON-able report of one checkpoint."""
    from taichislam_amd.utils import parity
    names = list(maps)
    ref = maps["struct_for"]
    for k in names:
        assert np.array_equal(maps[k]["indices"], ref["indices"]), f"voxel index set of schedule {k} differs"
    same_b = np.array_equal(batched["indices"], ref["indices"])
    T = {k: maps[k]["TSDF"].view(np.float16).astype(np.float64) for k in names}
    ulp = parity.f16_ulp(T["struct_for"])
    stack = np.stack([T[k] for k in names])
    lo, hi = stack.min(0), stack.max(0)
    p = ref["indices"].astype(np.float64) * vs
    dist = np.min(np.linalg.norm(p[:, None, :] - cams[None, :, :], axis=2), axis=1) if len(cams) <= 128 else None
    rep = {"voxels": int(lo.size), "schedules": names, "index_sets_identical_across_schedules": True, "batched_index_set_identical": bool(same_b),
           "occupancy_identical_across_schedules": bool(all(np.array_equal(maps[k]["occupy"], ref["occupy"]) for k in names))}
    # 1. schedule against schedule (each against the struct-for serialisation), and the order-free map against it: the same yardstick
    rep["vs_struct_for"] = {k: parity.short_summary(parity.deviation_report(maps[k], ref, vs)) for k in names if k != "struct_for"}
    rep["vs_struct_for"]["batched (= default HIP path)"] = parity.short_summary(parity.deviation_report(batched, ref, vs))
    # 2. the envelope
    spread = (hi - lo) / ulp
    rep["envelope_width_f16_ulps"] = parity._pct(spread)
    rep["envelope_width_m"] = parity._pct(hi - lo)
    b = batched["TSDF"].view(np.float16).astype(np.float64)

    def inside(x, l, h, u, tol):
        return (x >= l - tol * u) & (x <= h + tol * u)

    def outside_by(x, l, h):
        return np.maximum(np.maximum(l - x, x - h), 0.0)

    rows = []
    for a, z in zip(DIST_BINS[:-1], DIST_BINS[1:]):
        s = (dist >= a) & (dist < z)
        if not s.any():
            continue
        row = {"from_m": a, "to_m": min(z, 1e9), "n": int(s.sum()),
               "envelope_width_ulps": parity._pct(spread[s]),
               "batched_inside": float(np.mean(inside(b[s], lo[s], hi[s], ulp[s], 0))), "batched_inside_or_1ulp": float(np.mean(inside(b[s], lo[s], hi[s], ulp[s], 1))),
               "batched_outside_by_m": parity._pct(outside_by(b[s], lo[s], hi[s]))}
        loo, loo1, lood = [], [], []
        for k in names:      # calibration: a legal schedule against the envelope of the OTHER schedules
            others = np.stack([T[j][s] for j in names if j != k])
            l2, h2 = others.min(0), others.max(0)
            loo.append(float(np.mean(inside(T[k][s], l2, h2, ulp[s], 0)))); loo1.append(float(np.mean(inside(T[k][s], l2, h2, ulp[s], 1))))
            lood.append(parity._pct(outside_by(T[k][s], l2, h2))["p99"])
        row["schedule_inside_others_min_mean_max"] = [min(loo), float(np.mean(loo)), max(loo)]
        row["schedule_inside_others_or_1ulp_min_mean_max"] = [min(loo1), float(np.mean(loo1)), max(loo1)]
        row["schedule_outside_others_by_m_p99_max_over_schedules"] = max(lood)
        if ideal is not None and np.array_equal(ideal["indices"], ref["indices"]):
            iv = ideal["TSDF"].view(np.float16).astype(np.float64)[s]
            row["mean_abs_m_vs_float64_sequence"] = {"batched": float(np.mean(np.abs(b[s] - iv))), "struct_for": float(np.mean(np.abs(T["struct_for"][s] - iv))),
                                                     "schedules_mean": float(np.mean([np.mean(np.abs(T[k][s] - iv)) for k in names]))}
        rows.append(row)
    rep["by_distance_from_the_sensor_path"] = rows
    rep["batched_inside_envelope"] = float(np.mean(inside(b, lo, hi, ulp, 0)))
    rep["batched_inside_envelope_or_1ulp"] = float(np.mean(inside(b, lo, hi, ulp, 1)))
    rep["w_identical_across_schedules"] = bool(all(np.array_equal(maps[k]["W_TSDF"], ref["W_TSDF"]) for k in names))
    rep["w_batched_vs_struct_for_within_1ulp"] = parity.field_deviation(batched["W_TSDF"].view(np.float16), ref["W_TSDF"].view(np.float16))["within_1ulp"]
    return rep


def study(n, checkpoints, cfg=C2, small=None, procs=None, keep_maps=False):
    from taichislam_amd.utils import synthetic as syn
    jobs = [(name, "faithful", kind, param, seed, n, checkpoints, cfg, small) for name, kind, param, seed in SCHEDULES]
    jobs += [("batched", "batched", 0, 0, 0, n, checkpoints, cfg, small), ("ideal", "ideal", 0, 0, 0, n, checkpoints, cfg, small)]
    with mp.get_context("spawn").Pool(procs or min(len(jobs), os.cpu_count() or 1)) as pool:
        res = dict(pool.map(run, jobs))
    out = {}
    for cp in checkpoints:
        cams = np.stack([syn.camera_pose(f)[1] for f in range(cp)])
        out[f"after_{cp}_frames"] = envelope_stats({k: res[k][cp] for k, *_ in SCHEDULES}, res["batched"][cp], res["ideal"][cp], cams, cfg["voxel_scale"])
    return (out, res) if keep_maps else out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 77
    path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r04_parity_envelope.json")
    cps = sorted({c for c in (12, 25, 50, n) if c <= n})
    rep = {"config": f"BASELINE configs[1], first {n} frames of the synthetic sphere-room stream; oracle FAITHFUL under {len(SCHEDULES)} legal schedules of "
                     "dense_tsdf.py:239's parallel ray loop; batched = oracle BATCHED = the default HIP path bit for bit; ideal = the struct-for sequence in float64",
           "schedules": {name: {"kind": ["struct-for order", "random ray order", "threads over contiguous shares, one step per turn"][kind], "param": param, "seed": seed}
                         for name, kind, param, seed in SCHEDULES}}
    rep.update(study(n, cps))
    with open(path, "w") as f:
        json.dump(rep, f, indent=1)
    last = rep[f"after_{cps[-1]}_frames"]
    print(json.dumps({"frames": cps[-1], "voxels": last["voxels"], "envelope_width_ulps": last["envelope_width_f16_ulps"], "batched_inside": last["batched_inside_envelope"],
                      "batched_inside_or_1ulp": last["batched_inside_envelope_or_1ulp"],
                      "by_distance": [{k: r[k] for k in ("from_m", "to_m", "n", "batched_inside", "batched_inside_or_1ulp", "schedule_inside_others_min_mean_max")} for r in last["by_distance_from_the_sensor_path"]]}, indent=1))
