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


class PaddedGather:
    """The fixed-slab exchange of gather_results_padded as an asynchronous, double-buffered pipeline: start() copies
    the rank's slabs out of the engine's result buffers (which the next batch overwrites) and launches the collectives
    with async_op=True, so the transfer of batch i runs on the communicator's stream while batch i+1 is computed;
    a slot is reused only after its previous transfer has been waited for.  finish() waits for everything in flight
    and returns the last batch's (counts [world, B], keypoints [world][B, kpad, 7], descriptors [world][B, kpad, strings])
    on dst, None elsewhere (B = frames_max, the largest shard; a rank's surplus rows have count 0)."""

    def __init__(self, counts, kps, desc, strings, kpad, dst=0, group=None, slots=2, frames_max=None):
        self.counts, self.kps, self.desc = counts, kps, desc
        self.strings, self.kpad, self.dst, self.group = strings, kpad, dst, group
        # ranks may own shards of different sizes (512 frames over 3 ranks): every rank cuts slabs for `frames_max`
        # frames (the largest shard), the frames it does not own stay at count 0
        self.bmax = int(frames_max) if frames_max else int(counts.shape[0])
        assert self.bmax >= counts.shape[0]
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.slots = [None] * slots
        self.i = 0
        self.last = None

    def _alloc(self):
        b = self.bmax
        dev = self.counts.device
        sl = {"c": torch.zeros((b,), device=dev, dtype=self.counts.dtype),
              "k": torch.zeros((b, self.kpad, self.kps.shape[2]), device=dev, dtype=self.kps.dtype),
              "d": torch.zeros((b, self.kpad, self.strings), device=dev, dtype=self.desc.dtype),
              "ac": torch.empty((self.world * b,), device=dev, dtype=self.counts.dtype), "works": [], "kpad": self.kpad}
        if self.rank == self.dst:
            sl["gk"] = [torch.empty_like(sl["k"]) for _ in range(self.world)]
            sl["gd"] = [torch.empty_like(sl["d"]) for _ in range(self.world)]
        else:
            sl["gk"] = sl["gd"] = None
        return sl

    def _wait(self, sl):
        for w in sl["works"]:
            w.wait()
        sl["works"] = []

    def start(self):
        j = self.i % len(self.slots)
        sl = self.slots[j]
        if sl is not None:
            self._wait(sl)
        if sl is None or sl["kpad"] != self.kpad:
            sl = self.slots[j] = self._alloc()
        b = self.counts.shape[0]
        sl["c"][:b].copy_(self.counts)
        sl["k"][:b].copy_(self.kps[:, :self.kpad, :])
        sl["d"][:b].copy_(self.desc[:, :self.kpad, :self.strings])
        sl["works"] = [dist.all_gather_into_tensor(sl["ac"], sl["c"], group=self.group, async_op=True),
                       dist.gather(sl["k"], sl["gk"], dst=self.dst, group=self.group, async_op=True),
                       dist.gather(sl["d"], sl["gd"], dst=self.dst, group=self.group, async_op=True)]
        self.last = sl
        self.i += 1

    def finish(self):
        for sl in self.slots:
            if sl is not None:
                self._wait(sl)
        sl = self.last
        if sl is None or self.rank != self.dst:
            return None
        return sl["ac"].view(self.world, -1), sl["gk"], sl["gd"]
