"""Round-5 reproduction of VERDICT r4's leading finding: semantics = 1 reported NOT bit-exact by bench.py (the 75-frame C2 stream on the driver's box,
the reference-source vectors in every round-4 bench run) while the same calls are green in pytest.  Does what the bench process does, stage by stage,
and reports every comparison on its own (GPU box only; the oracle is the checker).

  python tools/repro_r5.py [stage ...]      stages: vec0 stream vec1 bisect default   (default: all)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

VECTORS = ["depth_stream", "point_clouds", "weight_clamp", "textured", "textured_points", "blk10_two_submaps", "two_submaps_fused", "aligned_submap_fused"]


def say(**kw):
    print(json.dumps(kw), flush=True)


def diff(got, want, keys=("TSDF", "W_TSDF", "occupy")):
    """got / want: sorted exports (tests/util.sort_export or test_ref_golden.sorted_bits)"""
    if got["indices"].shape != want["indices"].shape or not np.array_equal(got["indices"], want["indices"]):
        return {"ok": False, "voxels": [int(got["indices"].shape[0]), int(want["indices"].shape[0])], "what": "voxel sets differ"}
    out = {"ok": True, "voxels": int(want["indices"].shape[0])}
    for k in keys:
        if k not in want:
            continue
        bad = np.nonzero(np.atleast_1d((got[k] != want[k]).reshape(got[k].shape[0], -1).any(axis=1)))[0]
        if bad.size:
            out["ok"] = False
            idx = want["indices"][bad].astype(np.int64)
            out[k] = {"n": int(bad.size), "first_index": idx[0].tolist(), "got": np.atleast_1d(got[k][bad[0]]).tolist(), "want": np.atleast_1d(want[k][bad[0]]).tolist(),
                      "index_min": idx.min(axis=0).tolist(), "index_max": idx.max(axis=0).tolist(),
                      "bricks": int(np.unique((idx + 4096) >> 4, axis=0).shape[0])}
    return out


def run_vectors(tag, names=VECTORS, opts=()):
    import test_ref_golden as tr
    for name in names:
        cfg, K, Kc, steps, want = tr.load(name)

        def make(over):
            h = tr._Hip({**cfg, **over, "device": 0}, K, Kc, 1)
            for k, v in opts:
                h.m.set_option(k, v)
            return h
        try:
            got = tr.replay(make, steps, K, Kc, {}, lambda g, m: g.m.fuse_submaps(m.m))
            say(stage=tag, vector=name, opts=list(opts), **diff(got, want, ("TSDF", "W_TSDF", "occupy", "color")))
        except Exception as e:
            say(stage=tag, vector=name, opts=list(opts), error=repr(e)[:300])


def c2_stream(n):
    from taichislam_amd.utils import synthetic as syn
    return list(syn.sphere_room_stream(n))


def hip_stream(frames, dev_tensors, opts=(), semantics=1, host=False):
    """queue every frame back to back (the pipeline really overlaps), one sync at the end"""
    import torch
    from taichislam_amd.mapping import DenseTSDF
    from taichislam_amd.utils import synthetic as syn
    from util import C2, sort_export
    g = DenseTSDF(**C2, device=0)
    g.set_dep_camera_intrinsic(syn.K_DEPTH)
    if semantics:
        g.set_option("semantics", 1)
    for k, v in opts:
        g.set_option(k, v)
    t0 = time.perf_counter()
    for (R, T, d), dd in zip(frames, dev_tensors):
        g.recast_depth_to_map(R, T, d if host else dd, None)
    g.sync()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    e = sort_export(g.export_submap())
    st = g.last_frame_stats()
    del g
    return e, dt, st


def main():
    stages = sys.argv[1:] or ["vec0", "stream", "vec1", "default"]
    import torch
    from oracle import BATCHED, FAITHFUL, OracleTSDF
    from taichislam_amd.utils import synthetic as syn
    from util import C2, sort_export
    say(stage="start", device=torch.cuda.get_device_name(0), stages=stages)
    if "vec0" in stages:
        run_vectors("vec0")
    if "stream" in stages or "default" in stages:
        N, NCMP = 305, 75
        frames = c2_stream(N)
        dev = [torch.from_numpy(d.view(np.int16)).cuda(0) for _, _, d in frames]
    if "stream" in stages:
        t0 = time.perf_counter()
        o = OracleTSDF(**C2); o.set_intrinsics(syn.K_DEPTH)
        for R, T, d in frames[:NCMP]:
            o.integrate_depth(R, T, d, mode=FAITHFUL)
        want = sort_export(o.export_sparse())
        say(stage="oracle", frames=NCMP, mode="FAITHFUL", s=time.perf_counter() - t0)
        variants = [(), (("ramp", 0),), (("ramp", 16),), (("overlap", 3),), (("overlap", 1),), (("overlap", 0),), (("seq_impl", 0),)]
        for rep in range(3):
            for opts in (variants if rep == 0 else variants[:2]):
                try:
                    e, dt, st = hip_stream(frames[:NCMP], dev[:NCMP], opts)
                    say(stage="stream75", rep=rep, opts=list(opts), fps=NCMP / dt, **diff(e, want))
                except Exception as ex:
                    say(stage="stream75", rep=rep, opts=list(opts), error=repr(ex)[:300])
        # host images through the mapped pinned buffers, back to back
        for rep in range(2):
            e, dt, st = hip_stream(frames[:NCMP], dev[:NCMP], (), host=True)
            say(stage="stream75_host", rep=rep, fps=NCMP / dt, **diff(e, want))
        # the long stream: GPU against GPU (no oracle at this length): repeatability of the pipelined replay, and against the serial form
        ref, dt, _ = hip_stream(frames, dev, (("overlap", 0),))
        say(stage="stream305_serial", fps=N / dt, voxels=int(ref["indices"].shape[0]))
        for rep in range(3):
            e, dt, _ = hip_stream(frames, dev, ())
            say(stage="stream305_vs_serial", rep=rep, fps=N / dt, **diff(e, ref))
    if "stress" in stages:
        # the bench's sequential leg under timing jitter: first frame alone + sync, then the rest back to back with random pauses, device-wide
        # synchronisations, garbage handles destroyed in between (what Python's collector does to bench.py at random)
        import gc
        import random
        from taichislam_amd.mapping import DenseTSDF
        NS = 76
        frames = c2_stream(NS)
        dev = [torch.from_numpy(d.view(np.int16)).cuda(0) for _, _, d in frames]
        o = OracleTSDF(**C2); o.set_intrinsics(syn.K_DEPTH)
        for R, T, d in frames:
            o.integrate_depth(R, T, d, mode=FAITHFUL)
        want = sort_export(o.export_sparse())
        rng = random.Random(5)
        nrep = int(os.environ.get("STRESS_REPS", "30"))
        for rep in range(nrep):
            mode = rep % 6
            g = DenseTSDF(**C2, device=0)
            g.set_dep_camera_intrinsic(syn.K_DEPTH)
            g.set_option("semantics", 1)
            g.recast_depth_to_map(frames[0][0], frames[0][1], dev[0], None)
            g.sync()
            junk = None
            for f in range(1, NS):
                R, T, _ = frames[f]
                g.recast_depth_to_map(R, T, dev[f], None)
                if mode == 1 and rng.random() < 0.3:
                    time.sleep(rng.random() * 0.002)
                if mode == 2 and rng.random() < 0.1:
                    torch.cuda.synchronize()
                if mode == 3 and f in (9, 30, 55):
                    junk = DenseTSDF(**C2, device=0); junk.set_dep_camera_intrinsic(syn.K_DEPTH)
                    junk.recast_depth_to_map(R, T, frames[f][2], None); junk.sync(); junk = None; gc.collect()
                if mode == 4 and rng.random() < 0.2:
                    time.sleep(rng.random() * 0.0003)
                if mode == 5 and f % 11 == 0:
                    g.last_frame_stats()
            g.sync()
            e = sort_export(g.export_submap())
            say(stage="stress", rep=rep, mode=mode, **diff(e, want))
            del g
            gc.collect()
    if "benchlike" in stages:
        # what bench.py does around its sequential leg, many times over in one process: a default-semantics handle fed host images and left to the cyclic
        # collector, a burst of the all-core OpenMP port (its worker threads keep spinning for a while), then the literal stream compared with FAITHFUL
        import gc
        from taichislam_amd.mapping import DenseTSDF
        NS = int(os.environ.get("BL_FRAMES", "76"))
        frames = c2_stream(NS)
        o = OracleTSDF(**C2); o.set_intrinsics(syn.K_DEPTH)
        for R, T, d in frames:
            o.integrate_depth(R, T, d, mode=FAITHFUL)
        want = sort_export(o.export_sparse())
        ncpu = os.cpu_count() or 1
        p = OracleTSDF(**C2); p.set_intrinsics(syn.K_DEPTH)
        nrep = int(os.environ.get("BL_REPS", "30"))
        outdir = os.path.join(ROOT, "gpurun_out", os.environ.get("BL_OUT", "r5e"))
        os.makedirs(outdir, exist_ok=True)
        for rep in range(nrep):
            def parity_like():
                g = DenseTSDF(**C2, device=0)
                g.set_dep_camera_intrinsic(syn.K_DEPTH)
                for R, T, d in frames[:40]:
                    g.recast_depth_to_map(R, T, d, None)
                return g.export_submap()["TSDF"].shape[0]
            if not os.environ.get("BL_NOJUNK"):
                nv = parity_like()                               # (the handle is garbage now: a reference cycle, freed by the collector whenever it runs)
            if rep % 2 == 0 and not os.environ.get("BL_NOOMP"):
                for R, T, d in frames[1:4]:
                    p.integrate_depth_mt(R, T, d, min(ncpu, 32))
            g = DenseTSDF(**C2, device=0)
            g.set_dep_camera_intrinsic(syn.K_DEPTH)
            g.set_option("semantics", 1)
            for kv in os.environ.get("BL_OPTS", "").split():
                g.set_option(kv.split("=")[0], int(kv.split("=")[1]))
            dd = [torch.from_numpy(d.view(np.int16)).cuda(0) for _, _, d in frames]
            g.recast_depth_to_map(frames[0][0], frames[0][1], dd[0], None)
            g.sync()
            t0 = time.perf_counter()
            for (R, T, _), d in zip(frames[1:], dd[1:]):
                g.recast_depth_to_map(R, T, d, None)
            g.sync()
            dt = time.perf_counter() - t0
            e = sort_export(g.export_submap())
            r = diff(e, want)
            shape = None
            try:
                shape = [g.get_option("batch_shape_hash"), g.get_option("dry_launches"), g.get_option("seq_verify_mismatches")]
            except Exception:
                pass
            say(stage="benchlike", rep=rep, fps=(NS - 1) / dt, shape=shape, **r)
            if not r["ok"] and "TSDF" in r or "W_TSDF" in r:
                bad = np.nonzero((e["TSDF"] != want["TSDF"]) | (e["W_TSDF"] != want["W_TSDF"]))[0]
                np.savez(os.path.join(outdir, f"diff_rep{rep}.npz"), indices=want["indices"][bad], got_T=e["TSDF"][bad], got_W=e["W_TSDF"][bad], want_T=want["TSDF"][bad], want_W=want["W_TSDF"][bad])
            del g, dd
            if rep % 10 == 9:
                gc.collect()
    if "vec1" in stages:
        run_vectors("vec1")
    if "bisect" in stages:
        for opts in ((("seq_impl", 0),), (("overlap", 0),)):
            run_vectors("bisect", opts=opts)
    if "default" in stages:
        # the default path behind a really overlapped pipeline, against oracle BATCHED (missing 2)
        NB = 100
        t0 = time.perf_counter()
        o = OracleTSDF(**C2); o.set_intrinsics(syn.K_DEPTH)
        for R, T, d in frames[:NB]:
            o.integrate_depth(R, T, d, mode=BATCHED)
        want = sort_export(o.export_sparse())
        say(stage="oracle", frames=NB, mode="BATCHED", s=time.perf_counter() - t0)
        for host in (False, True):
            for rep in range(2):
                e, dt, st = hip_stream(frames[:NB], dev[:NB], (), semantics=0, host=host)
                say(stage="default100", host=host, rep=rep, fps=NB / dt, **diff(e, want))
    say(stage="done")


if __name__ == "__main__":
    main()
