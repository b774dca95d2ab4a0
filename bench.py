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
    ncpu = os.cpu_count() or 1
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
            "sample": f"{m} frames of the same stream, {dt2:.1f} s, {cores} of {ncpu} OpenMP threads (best of the sweep): BATCHED sums, rays cut into "
                      f"per-brick segments and bricks integrated in parallel like the GPU path (ray-parallel atomics scale negatively); "
                      f"bit-identical to the one-thread BATCHED map"}
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
                   "Parity unpinned by the reference (no golden vectors, Taichi not installable). Histogram: profiles/r02_parity_vs_faithful.json")
    return out


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
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch(args))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
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
    if args.config != 2:
        if distributed:
            raise SystemExit("--config 1/3/4 are single-GPU measurements")
        from taichislam_amd.utils import bench_configs
        emit(bench_configs.run(args.config, args.steps, args.warmup, dev))
        return

    from taichislam_amd import _lib
    from taichislam_amd.mapping import DenseTSDF
    from taichislam_amd.utils import synthetic as syn

    # ---- synthetic stream: each rank starts 45 degrees further round the room ----
    nframes = args.warmup + args.steps
    from taichislam_amd import distributed as D
    host = list(syn.sphere_room_stream(nframes, start_deg=D.stream_start_deg(rank)))
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
        m.set_option(k, int(v))

    frames_dev = [depth_dev[f] for f in range(nframes)]     # one [480, 640] view per frame, made before the timed region

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

    barrier()
    # Python's cyclic collector is paused for the timed region (as timeit does): a full collection of this process (torch, numpy, the
    # frame lists) takes ~35 ms, two thirds of a 300-frame run, and would land in it at random
    gc.collect(); gc.disable()
    t0 = time.perf_counter()
    for f in range(args.warmup, nframes):
        step(f)
    m.sync()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    dt_rank = dt
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{dev}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
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
        nh = min(args.steps, 200)
        pinned = torch.from_numpy(np.stack([host[args.warmup + i][2] for i in range(nh)]).view(np.int16)).pin_memory().numpy().view(np.uint16)
        host_rates = {}
        for label, src in (("pageable", [host[args.warmup + i][2] for i in range(nh)]), ("pinned", [pinned[i] for i in range(nh)])):
            m.sync()
            th = time.perf_counter()
            for i in range(nh):
                R, T = poses[args.warmup + i]
                m.recast_depth_to_map(R, T, src[i], None)
            m.sync()
            host_rates[label] = nh / (time.perf_counter() - th)
        host_rates["note"] = f"{nh} frames, 614 kB uint16 numpy image per call through tsl_tsdf_integrate_depth (copy + stream sync per frame), rank 0 only"

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
    if distributed or args.merge:
        import threading
        th = threading.Thread(target=merge_leg, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("TSL_BENCH_MERGE_TIMEOUT", "120")))
        merge_timed_out = not merge_box["done"]
    merge = {"error": "timeout: the merge leg did not finish (first multi-rank run on hardware?)"} if merge_timed_out else merge_box["merge"]

    if rank == 0:
        fps = world * args.steps / dt
        # algorithmic bytes (SURVEY.md section 8d / DESIGN.md): phase A = 2*P_used + 24*P_valid, phase B = 9*U + V_pcl
        bytes_a = 2 * stats["p_used"] + 24 * stats["p_valid"]
        bytes_b = 9 * stats["unique"] + stats["v_pcl"]
        roof = None
        if "integrate" in kern:
            # one launch of the brick kernel integrates a whole batch of queued frames (up to 8): algorithmic bytes per launch =
            # 9 B per distinct voxel a frame updates (4 B read + 4 B + 1 B written) x the frames the launch covers
            fpl = args.steps / max(1, kern["integrate"]["launches"])
            alg = 9 * stats["unique"] * fpl
            us = kern["integrate"]["avg_us"]
            ach = alg / (us * 1e-6) / 1e9
            traffic, traffic_src = None, None
            tpath = os.path.join(ROOT, "profiles", "r02_traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    traffic, traffic_src = tj["integrate"]["hbm_bytes_per_launch"], tj.get("command")
                except Exception:
                    traffic = None
            roof = {"bound": "hbm", "kernel": "tsl::k_integrate_batch", "frames_per_launch": fpl, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": alg, "avg_launch_us": us,
                    "frame_bytes": bytes_a + bytes_b, "frame_gbs": (bytes_a + bytes_b) * fps / world / 1e9,
                    "frame_frac": (bytes_a + bytes_b) * fps / world / 1e9 / HBM_PEAK_GBS,
                    "note": "one 640x480 frame touches ~12 MB: the launch is bounded by the exact per-step arithmetic (walk VALU-bound), load "
                            "imbalance of the bricks next to the sensor and the merge of split bricks, not by bytes (DESIGN.md section 4)"}
        out = {
            "metric": "depth-frames/s integrated (640x480->512^3 TSDF)", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16 storage / f32 + int64 fixed-point arithmetic",
            "data": "synthetic",
            "config": {"workload": WORKLOAD, "frame_stats": stats, "kernels_us": kern,
                       "kernels_us_note": "every kernel is launched once per batch of up to 8 queued frames",
                       "updates_per_s": stats["steps"] * fps, "per_rank_frames_per_s": per_rank, "merge": merge},
            "roofline": roof,
            "value_host_input": host_rates,
        }
        if not args.no_cpu_baseline:
            sample = [(R, T, d) for R, T, d in host[: max(8, min(len(host), 200))]]
            one, allc, omap, n_done = cpu_baselines(sample)
            out["cpu_baseline"] = one
            out["cpu_baseline_allcore"] = allc
            try:
                out["parity_vs_faithful"] = parity_vs_faithful(dev, sample[:n_done], omap)
            except Exception as e:
                out["parity_vs_faithful"] = {"error": repr(e)[:200]}
        emit(out)
    if merge_timed_out:
        sys.stderr.write("bench.py: merge leg timed out; leaving without further collectives\n"); real_stdout.flush()
        os._exit(0)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
