"""Multi-GPU operation: one submap (one depth stream) per GPU, one exchange at merge time.

Integration shards by submap with no data-path collective.  The only exchange is the global-map
merge: every rank splats its own submaps into exact int64 fixed-point accumulators over the dense global grid
(`fuse_accumulate`), the accumulators are all-reduced (sum) -- RCCL over xGMI when the process group is "nccl" -- and
every rank finalises the same global TSDF (`fuse_finalize`).  Integer sums make the result independent of the number
of ranks and of the reduction order, i.e. bit-identical to a single GPU fusing all submaps
(reference: taichi_slam/mapping/dense_tsdf.py:272-318 run once over every agent's submaps; the reference itself ships
submaps between agents as zlib-compressed numpy dicts over LCM, submap_mapping.py:226-253)."""
import numpy as np


def stream_start_deg(rank):
    """Start angle of rank `rank`'s synthetic stream (45 degrees apart)."""
    return 45.0 * rank


def merge_buffers(global_map, device=None):
    """Zero-initialised accumulators for `global_map`: torch tensors on `device` (CUDA) or numpy arrays (device=None)."""
    nvox = int(global_map.N) * int(global_map.N) * int(global_map.Nz)
    if device is None:
        return np.zeros((nvox, 2), np.int64), np.zeros(nvox, np.int32)
    import torch
    return (torch.zeros((nvox, 2), dtype=torch.int64, device=device), torch.zeros(nvox, dtype=torch.int32, device=device))


def _brick_exchange(acc, cnt, N, Nz, group):
    """All-reduce only the 16^3 bricks some rank wrote to: a byte mask of the bricks is reduced first (MAX), the union of
    touched bricks is packed, summed over the ranks and written back.  Returns the bytes all-reduced."""
    import torch
    import torch.distributed as dist
    nbx, nbz = N // 16, Nz // 16
    c6 = cnt.view(nbx, 16, nbx, 16, nbz, 16)
    mask = c6.ne(0).any(dim=5).any(dim=3).any(dim=1).to(torch.uint8).contiguous()          # [nbx, nbx, nbz]
    dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=group)
    bi, bj, bk = torch.nonzero(mask, as_tuple=True)
    nbytes = mask.numel()
    if bi.numel() == 0:
        return nbytes
    a7 = acc.view(nbx, 16, nbx, 16, nbz, 16, 2)
    pa = a7[bi, :, bj, :, bk].contiguous()                                                   # [n, 16, 16, 16, 2]
    pc = c6[bi, :, bj, :, bk].contiguous()                                                   # [n, 16, 16, 16]
    dist.all_reduce(pa, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(pc, op=dist.ReduceOp.SUM, group=group)
    a7[bi, :, bj, :, bk] = pa
    c6[bi, :, bj, :, bk] = pc
    return nbytes + pa.numel() * pa.element_size() + pc.numel() * pc.element_size()


def allreduce_merge(global_map, submaps, group=None, device=None, sparse=True):
    """Merge every rank's `submaps` into every rank's `global_map`.  Returns the bytes all-reduced per rank.

    `global_map` / `submaps` need `fuse_accumulate(submaps, acc, cnt)` and `fuse_finalize(acc, cnt)` (DenseTSDF on
    the GPU; the CPU oracle in the gloo tests).  The global map's pose table must hold the base pose of every submap
    id used by any rank (set_base_pose_submap), exactly as for a single-process fuse_submaps.  With `sparse` (default)
    only the bricks touched by some rank travel (a 512^3 map: 2.7 GB dense, typically a few hundred MB sparse); the
    merged map is the same either way."""
    import torch
    import torch.distributed as dist
    acc, cnt = merge_buffers(global_map, device)
    on_gpu = isinstance(acc, torch.Tensor) and acc.is_cuda
    if on_gpu:
        torch.cuda.synchronize(acc.device)          # the accumulators were cleared on torch's stream, the splat runs on the map's
    global_map.fuse_accumulate(submaps, acc, cnt)     # returns after its stream has drained
    nbytes = 0
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        tacc = acc if isinstance(acc, torch.Tensor) else torch.from_numpy(acc)
        tcnt = cnt if isinstance(cnt, torch.Tensor) else torch.from_numpy(cnt)
        N, Nz = int(global_map.N), int(global_map.Nz)
        if sparse and N % 16 == 0 and Nz % 16 == 0:
            nbytes = _brick_exchange(tacc, tcnt, N, Nz, group)
        else:
            dist.all_reduce(tacc, op=dist.ReduceOp.SUM, group=group)
            dist.all_reduce(tcnt, op=dist.ReduceOp.SUM, group=group)
            nbytes = tacc.numel() * tacc.element_size() + tcnt.numel() * tcnt.element_size()
    if on_gpu:
        torch.cuda.synchronize(acc.device)          # the collectives ran on torch's stream
    global_map.fuse_finalize(acc, cnt)
    return int(nbytes)
