"""world_size-2 gloo test of the multi-GPU merge path (taichislam_amd.distributed.allreduce_merge) on CPU.

The HIP kernels cannot run here, so each rank's submap is held by the CPU oracle and the step protocol (merge_begin / merge_pack /
merge_finish, the methods DenseTSDF implements on device buffers) is provided by a small adapter over the oracle's dense fusion sums.
What is under test is the sharding, the brick-sparse exchange and the two all-reduces, and the claim that the integer merge is
bit-identical to a single process fusing every submap.  tests/test_merge_gpu.py runs the same protocol on real DenseTSDF handles."""
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


class _OracleGlobal:
    """Step protocol of DenseTSDF's merge on top of the oracle's dense fusion sums (numpy)."""

    def __init__(self, bases):
        from oracle import OracleTSDF
        self.o = OracleTSDF(**dict(CFG, is_global_map=True))
        for sid, (R, T) in enumerate(bases):
            self.o.set_base_pose_submap(sid, R, T)
        self.N, self.Nz = self.o.N, self.o.Nz

    def _bricks(self, a, tail=()):
        nb, nbz = self.N // 16, self.Nz // 16
        return a.reshape((nb, 16, nb, 16, nbz, 16) + tail)

    def merge_begin(self, sub):
        nvox = self.N * self.N * self.Nz
        self.acc, self.cnt = np.zeros((nvox, 2), np.int64), np.zeros(nvox, np.int32)
        self.o.fuse_accumulate(sub, self.acc, self.cnt)
        return np.ascontiguousarray(self._bricks(self.cnt).astype(bool).any(axis=(1, 3, 5)).reshape(-1).astype(np.uint8))

    def merge_pack(self, mask):
        nb, nbz = self.N // 16, self.Nz // 16
        self.union = np.nonzero(mask.reshape(nb, nb, nbz))                     # ascending brick order
        bi, bj, bk = self.union
        pa = np.ascontiguousarray(self._bricks(self.acc, (2,))[bi, :, bj, :, bk].reshape(len(bi), 4096, 2))
        pc = np.ascontiguousarray(self._bricks(self.cnt)[bi, :, bj, :, bk].reshape(len(bi), 4096))
        return pa, pc

    def merge_finish(self, acc, cnt):
        bi, bj, bk = self.union
        A, Cn = np.zeros_like(self.acc), np.zeros_like(self.cnt)
        self._bricks(A, (2,))[bi, :, bj, :, bk] = acc.reshape(len(bi), 16, 16, 16, 2)
        self._bricks(Cn)[bi, :, bj, :, bk] = cnt.reshape(len(bi), 16, 16, 16)
        self.o.fuse_finalize(A, Cn)

    # the reduce-scatter + all-gather form (tsl_tsdf_merge_finalize_slice / _finish_records): a brick's record is 4096 u32 {f16 TSDF | f16 W << 16},
    # 4096 occupancy bytes, 4096 "written" bits
    def merge_finalize_slice(self, acc, cnt):
        k = len(acc)
        rec = np.zeros((k, 4096 * 4 + 4096 + 512), np.uint8)
        with np.errstate(all="ignore"):
            num = (acc[..., 0].astype(np.float64) * 2.0 ** -24).astype(np.float32)
            den = (acc[..., 1].astype(np.float64) * 2.0 ** -24).astype(np.float32)
            t = (num / den).astype(np.float16).view(np.uint16).astype(np.uint32)
            w = den.astype(np.float16).view(np.uint16).astype(np.uint32)
        on = cnt != 0
        tw = np.where(on, t | (w << 16), 0).astype(np.uint32)
        occ = np.where(on, (cnt & 0xffff).astype(np.uint16).view(np.int16).astype(np.int8), 0).astype(np.int8)
        rec[:, :4096 * 4] = tw.view(np.uint8).reshape(k, -1)
        rec[:, 4096 * 4:4096 * 5] = occ.view(np.uint8)
        rec[:, 4096 * 5:] = np.packbits(on, axis=1, bitorder="little")
        return rec

    def merge_finish_records(self, rec):
        bi, bj, bk = self.union
        n = len(bi)
        rec = np.asarray(rec)[:n]
        tw = np.ascontiguousarray(rec[:, :4096 * 4]).view(np.uint32).reshape(n, 4096)
        occ = np.ascontiguousarray(rec[:, 4096 * 4:4096 * 5]).view(np.int8).reshape(n, 4096)
        on = np.unpackbits(np.ascontiguousarray(rec[:, 4096 * 5:]), axis=1, bitorder="little").astype(bool)
        u, l = np.nonzero(on)
        idx = np.stack([bi[u] * 16 + (l >> 8) - self.N // 2, bj[u] * 16 + ((l >> 4) & 15) - self.N // 2, bk[u] * 16 + (l & 15) - self.Nz // 2], 1).astype(np.int16)
        v = tw[u, l]
        self.o.import_sparse(0, idx, (v & 0xffff).astype(np.uint16).view(np.float16), (v >> 16).astype(np.uint16).view(np.float16), occ[u, l])

    def export_sparse(self):
        return self.o.export_sparse()


def _global(bases):
    return _OracleGlobal(bases)


class _FailingGlobal(_OracleGlobal):
    """Rank-local failure injected into the step protocol."""

    def __init__(self, bases, fail_in):
        super().__init__(bases)
        self.fail_in = fail_in

    def empty_merge_mask(self):
        return np.zeros((self.N // 16) ** 2 * (self.Nz // 16), np.uint8)

    def merge_begin(self, sub):
        if self.fail_in == "begin":
            raise MemoryError("injected: brick pool exhausted")
        return super().merge_begin(sub)

    def merge_pack(self, mask):
        if self.fail_in == "pack":
            raise MemoryError("injected: no memory for the packed planes")
        return super().merge_pack(mask)


def _worker_failing(rank, world, port, out, fail_in):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from taichislam_amd import distributed as D
    from taichislam_amd.utils import synthetic as syn
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    sub, _ = _build_submap(rank)
    bases = [syn.camera_pose(0, start_deg=D.stream_start_deg(r)) for r in range(world)]
    g = _FailingGlobal(bases, fail_in if rank == 1 else None)      # only rank 1 fails
    try:
        D.allreduce_merge(g, sub)
        msg = "no error"
    except RuntimeError as e:
        msg = str(e)
    open(os.path.join(out, f"fail_{fail_in}_rank{rank}.txt"), "w").write(msg)
    dist.barrier()                                                # both ranks get here: nobody is stuck in a collective
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_in", ["begin", "pack"])
def test_a_failing_rank_does_not_hang_the_others(tmp_path, fail_in):
    """One rank raises inside merge_begin / merge_pack: every rank still runs every exchange and every rank raises."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker_failing, args=(2, port, str(tmp_path), fail_in), nprocs=2, join=True)
    m0 = open(tmp_path / f"fail_{fail_in}_rank0.txt").read(); m1 = open(tmp_path / f"fail_{fail_in}_rank1.txt").read()
    assert "another rank failed" in m0 and "this rank failed" in m1, (m0, m1)


def _worker(rank, world, port, out, exchange="allreduce"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from taichislam_amd import distributed as D
    from taichislam_amd.utils import synthetic as syn
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    sub, _ = _build_submap(rank)
    bases = [syn.camera_pose(0, start_deg=D.stream_start_deg(r)) for r in range(world)]
    g = _global(bases)
    nbytes = D.allreduce_merge(g, sub, exchange=exchange)     # brick-sparse exchange over gloo
    e = g.export_sparse()
    np.savez(os.path.join(out, f"rank{rank}.npz"), idx=e["indices"], t=e["TSDF"].view(np.uint16), w=e["W_TSDF"].view(np.uint16), occ=e["occupy"],
             nbytes=nbytes)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["allreduce", "scatter_gather"])
def test_two_rank_merge_equals_single_process(tmp_path, exchange):
    """Both forms of the exchange (all-reduce of the packed sums; reduce-scatter, finalise the slice, all-gather of the finalised voxels) over a 2-rank gloo
    group: both ranks end with the map ONE process gets from fusing both submaps."""
    from oracle import BATCHED, OracleTSDF
    from taichislam_amd import distributed as D
    from taichislam_amd.utils import synthetic as syn
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path), exchange), nprocs=2, join=True)
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
    from oracle import OracleTSDF as _O
    g = _O(**dict(CFG, is_global_map=True))
    for sid, (R, T) in enumerate(bases):
        g.set_base_pose_submap(sid, R, T)
    g.fuse_submaps(both, mode=BATCHED)
    e = g.export_sparse()
    assert np.array_equal(e["indices"], r0["idx"])
    t = e["TSDF"].view(np.uint16)
    ok = ~np.isnan(e["TSDF"].astype(np.float32))
    assert np.array_equal(t[ok], r0["t"][ok]) and np.array_equal(e["W_TSDF"].view(np.uint16), r0["w"]) and np.array_equal(e["occupy"], r0["occ"])
    nvox = g.N * g.N * g.Nz
    assert 0 < int(r0["nbytes"]) < nvox * 20 // 2 and int(r0["nbytes"]) == int(r1["nbytes"])


def test_stream_sharding_is_disjoint():
    from taichislam_amd import distributed as D
    assert [D.stream_start_deg(r) for r in range(8)] == [0.0, 45.0, 90.0, 135.0, 180.0, 225.0, 270.0, 315.0]
