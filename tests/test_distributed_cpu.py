"""world_size-2 gloo test of the multi-GPU merge path (taichislam_amd.distributed.allreduce_merge) on CPU.

The HIP kernels cannot run here, so each rank's submap and the accumulate/finalize steps are provided by the CPU oracle
(same method names as DenseTSDF); what is under test is the sharding, the all-reduce plumbing and the claim that the
integer merge is bit-identical to a single process fusing every submap."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(map_scale=[5.12, 5.12], voxel_scale=0.04, num_voxel_per_blk_axis=16, max_ray_length=5.0, min_ray_length=0.3,
           internal_voxels=10, recast_step=2, max_submap_num=4)


def _build_submap(rank):
    from oracle import BATCHED, OracleTSDF
    from taichislam_amd import distributed as D
    from taichislam_amd.utils import synthetic as syn
    h, w = 60, 80
    K = syn.scaled_intrinsics(h, w)
    sub = OracleTSDF(**CFG)
    sub.set_intrinsics(K)
    sub.set_active_submap(rank)                       # rank r owns submap id r
    poses = []
    for f in range(2):
        R, T = syn.camera_pose(f, start_deg=D.stream_start_deg(rank))
        if f == 0:
            sub.set_base_pose_submap(rank, R, T)
            base = (R, T)
        sub.integrate_depth(R, T, syn.sphere_room_depth(R, T, h, w, K=K), mode=BATCHED)
    sub.set_active_submap(rank + 1)                   # closed, as create_new_submap would
    return sub, base


def _global(bases):
    from oracle import OracleTSDF
    g = OracleTSDF(**dict(CFG, is_global_map=True))
    for sid, (R, T) in enumerate(bases):
        g.set_base_pose_submap(sid, R, T)
    return g


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from taichislam_amd import distributed as D
    from taichislam_amd.utils import synthetic as syn
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    sub, _ = _build_submap(rank)
    bases = [syn.camera_pose(0, start_deg=D.stream_start_deg(r)) for r in range(world)]
    g = _global(bases)
    nbytes = D.allreduce_merge(g, sub)                       # brick-sparse exchange (default)
    e = g.export_sparse()
    g2 = _global(bases)
    nbytes_dense = D.allreduce_merge(g2, sub, sparse=False)   # dense exchange: same map
    e2 = g2.export_sparse()
    for k in ("indices", "TSDF", "W_TSDF", "occupy"):
        a, b = np.asarray(e[k]), np.asarray(e2[k])
        assert np.array_equal(a.view(np.uint16) if a.dtype == np.float16 else a, b.view(np.uint16) if b.dtype == np.float16 else b), k
    np.savez(os.path.join(out, f"rank{rank}.npz"), idx=e["indices"], t=e["TSDF"].view(np.uint16), w=e["W_TSDF"].view(np.uint16), occ=e["occupy"],
             nbytes=nbytes, nbytes_dense=nbytes_dense)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_merge_equals_single_process(tmp_path):
    from oracle import BATCHED, OracleTSDF
    from taichislam_amd import distributed as D
    from taichislam_amd.utils import synthetic as syn
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k in ("idx", "t", "w", "occ"):
        assert np.array_equal(r0[k], r1[k]), f"ranks disagree on {k}"
    assert r0["idx"].shape[0] > 5000
    # single process: one collection holding both submaps, fused directly
    h, w = 60, 80
    K = syn.scaled_intrinsics(h, w)
    both = OracleTSDF(**CFG); both.set_intrinsics(K)
    bases = []
    for rank in range(2):
        both.set_active_submap(rank)
        for f in range(2):
            R, T = syn.camera_pose(f, start_deg=D.stream_start_deg(rank))
            if f == 0:
                both.set_base_pose_submap(rank, R, T); bases.append((R, T))
            both.integrate_depth(R, T, syn.sphere_room_depth(R, T, h, w, K=K), mode=BATCHED)
    both.set_active_submap(2)
    g = _global(bases)
    g.fuse_submaps(both, mode=BATCHED)
    e = g.export_sparse()
    assert np.array_equal(e["indices"], r0["idx"])
    t = e["TSDF"].view(np.uint16)
    ok = ~np.isnan(e["TSDF"].astype(np.float32))
    assert np.array_equal(t[ok], r0["t"][ok]) and np.array_equal(e["W_TSDF"].view(np.uint16), r0["w"]) and np.array_equal(e["occupy"], r0["occ"])
    nvox = g.N * g.N * g.Nz
    assert int(r0["nbytes_dense"]) == nvox * 20
    assert 0 < int(r0["nbytes"]) < int(r0["nbytes_dense"]) // 2 and int(r0["nbytes"]) == int(r1["nbytes"])


def test_stream_sharding_is_disjoint():
    from taichislam_amd import distributed as D
    assert [D.stream_start_deg(r) for r in range(8)] == [0.0, 45.0, 90.0, 135.0, 180.0, 225.0, 270.0, 315.0]
