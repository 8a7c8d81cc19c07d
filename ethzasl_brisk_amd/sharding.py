"""Multi-GPU plumbing of the batch path (one process per GPU, torch.distributed; backend nccl = RCCL on ROCm).

The detect+describe path has no exchange step: frames are independent units, every rank runs the same kernels on
its own shard.  The only collective is the result gather of BASELINE config 3: per-frame counts (all_gather), then
the packed (keypoint, descriptor) rows to rank 0 (gather of slabs padded to the largest rank's payload).
Works on CPU tensors with gloo (tests) and on device tensors with nccl (bench.py).
"""
import torch
import torch.distributed as dist


def shard_frames(n_frames, rank, world, mode="block"):
    """Frame indices owned by `rank`.  block: contiguous blocks (rank r gets [r*n/w, (r+1)*n/w)); cyclic: r, r+w, ..."""
    if mode == "cyclic":
        return list(range(rank, n_frames, world))
    lo = (n_frames * rank) // world
    hi = (n_frames * (rank + 1)) // world
    return list(range(lo, hi))


def pack_results(counts, kps, desc, strings):
    """counts int32[B]; kps float32[B, cap, 7]; desc uint8[B, cap, pitch] -> packed rows in frame order."""
    cap = kps.shape[1]
    mask = torch.arange(cap, device=counts.device)[None, :] < counts[:, None].to(torch.int64)
    return kps[mask], desc[mask][:, :strings]


def gather_results(counts, kps, desc, strings, dst=0, group=None):
    """Gathers every rank's packed results on `dst`.

    Returns on dst a list (one entry per rank) of (counts[B] int32, keypoints[n,7] float32, descriptors[n,strings] u8);
    None on the other ranks.  Two small all_gathers (counts, totals) + two gathers of equally sized slabs.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = counts.contiguous()
    pk, pd = pack_results(counts, kps, desc, strings)
    n = torch.tensor([pk.shape[0]], device=counts.device, dtype=torch.int64)
    all_counts = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    all_n = [torch.empty_like(n) for _ in range(world)]
    dist.all_gather(all_n, n, group=group)
    nmax = max(int(x.item()) for x in all_n)
    slab_k = torch.zeros((nmax, 7), device=counts.device, dtype=torch.float32)
    slab_d = torch.zeros((nmax, strings), device=counts.device, dtype=torch.uint8)
    slab_k[:pk.shape[0]] = pk
    slab_d[:pd.shape[0]] = pd
    gk = [torch.empty_like(slab_k) for _ in range(world)] if rank == dst else None
    gd = [torch.empty_like(slab_d) for _ in range(world)] if rank == dst else None
    dist.gather(slab_k, gk, dst=dst, group=group)
    dist.gather(slab_d, gd, dst=dst, group=group)
    if rank != dst:
        return None
    out = []
    for r in range(world):
        m = int(all_n[r].item())
        out.append((all_counts[r], gk[r][:m], gd[r][:m]))
    return out


def gather_results_padded(counts, kps, desc, strings, kpad, dst=0, group=None):
    """Same exchange without any host synchronisation (what bench.py times): every rank contributes fixed-size
    slabs [B, kpad] cut from its result buffers plus the per-frame counts; rank `dst` slices them with the counts
    later.  The caller guarantees counts.max() <= kpad (checked outside the timed region); use gather_results()
    otherwise.  Returns on dst (counts [world, B], keypoints [world, B, kpad, 7], descriptors [world, B, kpad, strings])."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = counts.contiguous()
    slab_k = kps[:, :kpad, :].contiguous()
    slab_d = desc[:, :kpad, :strings].contiguous()
    flat = torch.empty((world * counts.numel(),), device=counts.device, dtype=counts.dtype)
    dist.all_gather_into_tensor(flat, counts.reshape(-1), group=group)
    all_counts = flat.view((world,) + tuple(counts.shape))
    gk = [torch.empty_like(slab_k) for _ in range(world)] if rank == dst else None
    gd = [torch.empty_like(slab_d) for _ in range(world)] if rank == dst else None
    dist.gather(slab_k, gk, dst=dst, group=group)
    dist.gather(slab_d, gd, dst=dst, group=group)
    if rank != dst:
        return None
    return all_counts, torch.stack(gk), torch.stack(gd)
