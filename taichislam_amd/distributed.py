"""Multi-GPU operation: one submap collection (one depth stream) per GPU, ONE exchange at merge time.

Integration shards by submap with no data-path collective.  The only exchange is the global-map merge
(include/taichislam_hip.h "multi-GPU global-map merge", csrc/tsl_merge.hip): every rank splats its own submaps into exact 2^-24
fixed-point sums per touched 16^3 brick, a byte mask of touched bricks is all-reduced (MAX), the union bricks are packed in
ascending brick order and all-reduced (SUM), and every rank writes the same global TSDF.  Integer sums make the result independent
of the number of ranks and of the reduction order, i.e. bit-identical to a single GPU fusing all submaps (reference:
taichi_slam/mapping/dense_tsdf.py:272-318 run once over every agent's submaps; the reference ships submaps between agents as
zlib-compressed numpy dicts over LCM, submap_mapping.py:226-253, utils/communication.py:9-43).

Two ways to run the two reductions:
  * `Communicator` -- the library's own RCCL communicator (ncclCommInitRank from a 128-byte id distributed by the caller); the whole
    merge is then one C call, `tsl_tsdf_allreduce_merge` (what bench.py --gpus N uses);
  * any `torch.distributed` process group (nccl = RCCL over xGMI, or gloo) through the step protocol
    `merge_begin -> all_reduce(MAX) -> merge_union -> merge_pack -> all_reduce(SUM) x2 -> merge_finish`, which DenseTSDF implements
    on device buffers and the CPU oracle adapter of tests/test_distributed_cpu.py implements on numpy arrays."""
import ctypes as C


def stream_start_deg(rank):
    """Start angle of rank `rank`'s synthetic stream (45 degrees apart)."""
    return 45.0 * rank


class Communicator:
    """RCCL communicator owned by libtaichislam_hip (bound at run time).  `unique_id()` on one rank, ship the 128 bytes to the
    others by any means (torch.distributed.broadcast_object_list, MPI, a file), then `Communicator(id, nranks, rank, device)` on all."""

    def __init__(self, uid, nranks, rank, device=0):
        from . import _lib
        self._lib = _lib
        self.L = _lib.lib()
        self.h = C.c_void_p()
        self.nranks, self.rank, self.device = int(nranks), int(rank), int(device)
        buf = C.create_string_buffer(bytes(uid), 128)
        _lib.check(self.L.tsl_comm_create(buf, self.nranks, self.rank, self.device, C.byref(self.h)))

    @staticmethod
    def unique_id():
        from . import _lib
        buf = C.create_string_buffer(128)
        _lib.check(_lib.lib().tsl_comm_unique_id(buf))
        return buf.raw

    @property
    def handle(self):
        return self.L.tsl_comm_handle(self.h)

    def close(self):
        if self.h:
            self.L.tsl_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def allreduce_merge(global_map, submaps, group=None, comm=None, exchange="allreduce"):
    """Merge every rank's `submaps` into every rank's `global_map`.  Returns the bytes all-reduced per rank.

    exchange: "allreduce" -- the packed sums of the union bricks are all-reduced (20 B per voxel out and back) -- or "scatter_gather" -- they are
    reduce-scattered, every rank finalises its slice and the finalised voxels (5.1 B each) are all-gathered; the maps are identical.

    comm: a `Communicator` -> one native call (RCCL).  Otherwise the step protocol over `torch.distributed` (`group`, default
    group when None); with no initialised process group (or world size 1) the same steps run without the reductions.
    The global map's pose table must hold the base pose of every submap id used by any rank (set_base_pose_submap)."""
    if exchange not in ("allreduce", "scatter_gather"):
        raise ValueError(exchange)
    if comm is not None:
        global_map.set_option("merge_exchange", 1 if exchange == "scatter_gather" else 0)
        return global_map.allreduce_merge(submaps, comm)
    import torch
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1

    def agree(err, like):
        """A rank that failed locally must not leave the others inside a collective: every rank runs every exchange, a failed one with
        empty contributions, and a status word (MAX) travels next to them; the error is raised on every rank afterwards."""
        if not multi:
            return bool(err)
        flag = torch.tensor([1 if err else 0], dtype=torch.int32, device=like.device if isinstance(like, torch.Tensor) else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        return bool(int(flag.item()))

    err = None
    try:
        mask = global_map.merge_begin(submaps)             # uint8 [bricks]: torch CUDA tensor (DenseTSDF) or numpy array (oracle adapter)
    except Exception as e:                                 # contribute an empty mask of the right shape
        err = e
        mask = global_map.empty_merge_mask() if hasattr(global_map, "empty_merge_mask") else None
        if mask is None:
            raise
    as_t = (lambda a: a) if isinstance(mask, torch.Tensor) else torch.from_numpy
    nbytes = 0
    if multi:
        dist.all_reduce(as_t(mask), op=dist.ReduceOp.MAX, group=group)
        nbytes += mask.size if not isinstance(mask, torch.Tensor) else mask.numel()
    if agree(err, mask):
        raise RuntimeError(f"allreduce_merge: {'this rank' if err else 'another rank'} failed before the exchange; nothing was merged") from err
    if isinstance(mask, torch.Tensor) and mask.is_cuda:
        torch.cuda.current_stream(mask.device).synchronize()      # the library reads the reduced mask on its own stream
    try:
        acc, cnt = global_map.merge_pack(mask)              # int64 [n,4096,2], int32 [n,4096] of the union bricks
    except Exception as e:
        err, acc, cnt = e, None, None
    if agree(err, mask):
        raise RuntimeError(f"allreduce_merge: {'this rank' if err else 'another rank'} failed while packing; nothing was merged") from err
    if exchange == "scatter_gather" and hasattr(global_map, "merge_finalize_slice"):
        world = dist.get_world_size(group) if multi else 1
        me = dist.get_rank(group) if multi else 0
        n = len(acc)
        nper = (n + world - 1) // world
        if n:
            ta, tc = as_t(acc), as_t(cnt)
            if nper * world > n:                                   # whole bricks per rank: the last slice is padded with zero bricks
                ta = torch.cat([ta, torch.zeros((nper * world - n,) + tuple(ta.shape[1:]), dtype=ta.dtype, device=ta.device)])
                tc = torch.cat([tc, torch.zeros((nper * world - n,) + tuple(tc.shape[1:]), dtype=tc.dtype, device=tc.device)])
            native = multi and dist.get_backend(group) == "nccl"
            if native:
                sa, sc = torch.empty_like(ta[:nper]), torch.empty_like(tc[:nper])
                dist.reduce_scatter_tensor(sa, ta, op=dist.ReduceOp.SUM, group=group)
                dist.reduce_scatter_tensor(sc, tc, op=dist.ReduceOp.SUM, group=group)
            else:                                                   # gloo has no reduce-scatter: an all-reduce and this rank's slice of it is the same thing
                if multi:
                    dist.all_reduce(ta, op=dist.ReduceOp.SUM, group=group); dist.all_reduce(tc, op=dist.ReduceOp.SUM, group=group)
                sa, sc = ta[me * nper:(me + 1) * nper].contiguous(), tc[me * nper:(me + 1) * nper].contiguous()
            if sa.is_cuda:
                torch.cuda.current_stream(sa.device).synchronize()
            rec = global_map.merge_finalize_slice(sa if isinstance(acc, torch.Tensor) else sa.numpy(), sc if isinstance(cnt, torch.Tensor) else sc.numpy())
            tr = as_t(rec)
            allrec = torch.zeros((nper * world,) + tuple(tr.shape[1:]), dtype=tr.dtype, device=tr.device)
            if native:
                dist.all_gather_into_tensor(allrec, tr, group=group)
            else:                                                   # disjoint slices: their sum is their concatenation
                allrec[me * nper:(me + 1) * nper] = tr
                if multi:
                    dist.all_reduce(allrec, op=dist.ReduceOp.SUM, group=group)
            if multi:
                nbytes += int((world - 1) / world * (nper * world * 4096 * 20 + allrec.numel()))      # what a rank sends round a ring
            if allrec.is_cuda:
                torch.cuda.current_stream(allrec.device).synchronize()
            global_map.merge_finish_records(allrec if isinstance(rec, torch.Tensor) else allrec.numpy())
        else:
            global_map.merge_finish(acc, cnt)
        return int(nbytes)
    if multi and len(acc):
        dist.all_reduce(as_t(acc), op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(as_t(cnt), op=dist.ReduceOp.SUM, group=group)
        nbytes += len(acc) * 4096 * 20
    if isinstance(mask, torch.Tensor) and mask.is_cuda:
        torch.cuda.current_stream(mask.device).synchronize()
    global_map.merge_finish(acc, cnt)
    return int(nbytes)
