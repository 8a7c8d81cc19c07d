#!/usr/bin/env python3
"""bench.py - frames/sec detect+describe @1080p on N MI355X GPUs (BASELINE.json metric).

A "step" is one pass of the hot path (pyramid -> AGAST detect -> NMS/refine -> integral -> describe) over one
batch of synthetic 1080p frames that already sit in HBM (BASELINE config 2: Appendix-C recipe, 4 octaves,
threshold 80, ~1k keypoints/frame).  Frames shard over ranks (one process per GPU, no data-path collective in the
detect/describe path itself); with N > 1 every step ends with the RCCL gather of the packed keypoints+descriptors
to rank 0 that BASELINE config 3 describes.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

W, H, OCTAVES, THRESHOLD = 1920, 1080, 4, 80
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable


def algorithmic_bytes(w, h, octaves, kp):
    """SURVEY §8(d) stage model.  Returns (total per frame, detect-kernel bytes per frame)."""
    sizes = [(w, h)]
    if octaves:
        sizes.append((2 * (w // 3), 2 * (h // 3)))
        for i in range(2, 2 * octaves):
            sizes.append((sizes[i - 2][0] // 2, sizes[i - 2][1] // 2))
    px = [a * b for a, b in sizes]
    P = sum(px)
    parents = sum(px[0 if i == 1 else i - 2] for i in range(1, len(px)))
    s1 = parents + (P - px[0])
    s2 = 2 * P
    s3 = P + 28 * kp
    s4 = px[0] + 4 * (w + 1) * (h + 1)
    s5 = kp * 8524
    return s1 + s2 + s3 + s4 + s5, s2


def cpu_baseline(frames, seconds_budget=20.0):
    """Oracle (CPU port of the reference path) on a bounded sample of the same workload, one process per core."""
    import multiprocessing as mp
    cores = max(1, min(len(os.sched_getaffinity(0)), 64))
    sample = frames[:max(cores, 4)]
    t0 = time.time()
    _cpu_one(sample[0])                      # single-thread time for one frame
    t1 = time.time() - t0
    reps = max(1, int(seconds_budget / max(t1, 1e-3) / 2))
    work = [sample[i % len(sample)] for i in range(min(cores * reps, cores * 8))]
    ctx = mp.get_context("fork")
    t0 = time.time()
    with ctx.Pool(cores) as pool:
        counts = pool.map(_cpu_one, work, chunksize=1)
    dt = time.time() - t0
    return {"value": round(len(work) / dt, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "single_thread_fps": round(1.0 / t1, 3),
            "sample": "%d synthetic 1080p frames (same recipe/params), oracle detect+describe, %d processes; "
                      "mean %d keypoints/frame" % (len(work), cores, int(np.mean(counts)))}


_EXT = None


def _cpu_one(img):
    global _EXT
    import oracle_lib as O
    if _EXT is None:
        _EXT = O.Extractor()
    k = O.detect(img, THRESHOLD, OCTAVES)
    k2, _ = _EXT.compute(img, k)
    return len(k2)


def main():
    # Exactly ONE line may reach stdout (the JSON).  Libraries (RCCL prints a version banner at exit) write to the
    # C-level stdout, so fd 1 is pointed at stderr for the whole run and the JSON is written to the saved fd.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per step per GPU")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic frames per GPU (ring)")
    ap.add_argument("--streams", type=int, default=1, help="internal stream slices per batch (1..8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-gather", action="store_true", help="run the RCCL result gather even with one rank (self-test)")
    ap.add_argument("--debug-flags", type=int, default=0, help="timing experiments only (results become wrong)")
    ap.add_argument("--pattern-version", type=int, default=2, help="2 = default 66-point pattern (the metric's workload), 1 = legacy 60-point pattern (timing experiments)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist
    import synth
    import ethzasl_brisk_amd as B

    if world > 1 or args.force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    # ---- synthetic frame ring, resident in HBM before the timed region
    nd = max(1, min(args.distinct, args.batch))
    host = np.stack([synth.frame_1080p(rank * 100000 + i) for i in range(nd)])
    ring = torch.from_numpy(host).to(dev)
    idx = torch.arange(args.batch, device=dev) % nd
    frames = ring[idx].contiguous()          # [batch, H, W] u8 in HBM
    del ring

    ctx = B.Context(local_rank)
    ctx.set_streams(args.streams)
    ctx.debug_set_flags(args.debug_flags)
    ext = B.BriskDescriptorExtractor(version=args.pattern_version, context=ctx)
    # everything below runs on ONE explicit torch stream: the engine's launches, the slab copies of the gather and
    # (through torch.distributed's stream synchronisation) the RCCL transfers are ordered on it.  (The legacy NULL
    # stream would make the engine fall back to its own non-blocking stream, which torch's work is not ordered with.)
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    stream = work_stream.cuda_stream
    strings = ext.descriptorSize()

    # result buffers as torch views (for the multi-GPU gather)
    def step():
        ctx.detect_describe_batch(ext, frames.data_ptr(), args.batch, W, H, W * H, W, THRESHOLD, OCTAVES, stream)

    gather = None
    gather_note = ""
    if world > 1 or args.force_gather:
        step()                               # allocates the engine's result buffers
        torch.cuda.synchronize()
        gather = ResultGather(ctx, args.batch, strings, dev, rank, world)

    try:
        for _ in range(args.warmup):
            step()
            if gather:
                gather.run()
        torch.cuda.synchronize()
    except Exception as e:                   # a failing collective must not take the throughput measurement with it
        if gather is None or args.force_gather:
            raise
        print("result gather failed in warm-up (%r): measuring without it" % (e,), file=sys.stderr)
        gather = None
        gather_note = " [RCCL gather failed in warm-up and was left out: %s]" % type(e).__name__
        torch.cuda.synchronize()
    assert ctx.batch_status(args.batch) == 0 or args.debug_flags
    if gather:
        gather.check_kpad()
    kps0, _ = ctx.batch_download(0, True, strings)
    mean_kp = float(np.mean([len(ctx.batch_download(f, True, strings)[0]) for f in range(min(args.batch, 8))]))

    ctx.profile_enable(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if gather:
            gather.run()
    if gather:
        gather.finish()                      # the last transfers are part of the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    stage_ms, ncalls = ctx.profile_read()
    fpl = ctx.profile_frames_per_launch() or args.batch   # frames per timed kernel launch (one stream slice)
    ctx.profile_enable(False)

    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    total_frames = args.batch * args.steps * world
    fps = total_frames / dt
    per_frame_bytes, detect_bytes = algorithmic_bytes(W, H, OCTAVES, int(round(mean_kp)))
    det_ms = stage_ms.get("k_detect", 0.0)
    achieved = (detect_bytes * fpl) / (det_ms * 1e-3) / 1e9 if det_ms > 0 else 0.0
    out = {
        "metric": "frames/sec detect+describe @1080p (1/2/4/8 GPU); % HBM roofline",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "1080p synthetic textured stream (SURVEY App. C recipe, 300 rects), 4 octaves, "
                               "AGAST threshold 80, default 66-point pattern (48-byte descriptors)",
                   "frames_per_step_per_gpu": args.batch, "stream_slices": args.streams, "frames_per_kernel_launch": fpl, "distinct_frames_per_gpu": nd,
                   "mean_keypoints_per_frame": round(mean_kp, 1),
                   "parallelism": "frames sharded over %d rank(s)%s" % (world, (", asynchronous RCCL gather of keypoints+descriptors to rank 0 each step (overlaps the next batch)" if (world > 1 and gather) else "") + gather_note),
                   "algorithmic_MB_per_frame": round(per_frame_bytes / 1e6, 3),
                   "pipeline_achieved_GBps": round(per_frame_bytes * fps / 1e9, 2),
                   "pipeline_frac_of_hbm_peak": round(per_frame_bytes * fps / 1e9 / (HBM_PEAK_GBS * world), 5),
                   "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
                   "stage_note": "HIP-event intervals on the launch stream; k_integral_final runs on a second stream beside "
                                 "k_tie_resolve / k_finalize / k_desc_prepare, so its own entry is only the join and those "
                                 "three entries include the sharing (kernel durations: profiles/*kernel_stats*)"},
        "roofline": {"bound": "hbm", "kernel": "k_detect", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": load_traffic(fpl),
                     "algorithmic_bytes_per_launch": detect_bytes * fpl, "avg_launch_ms": round(det_ms, 4),
                     "launches_timed": ncalls},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(list(host[:16]))
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if gather is not None and args.force_gather:
        # self-test on content the previous steps did not produce (a stale or half-written slab would show)
        frames2 = frames.flip(0).contiguous()
        ctx.detect_describe_batch(ext, frames2.data_ptr(), args.batch, W, H, W * H, W, THRESHOLD, OCTAVES, stream)
        gather.run()
        gather.finish()
        torch.cuda.synchronize()
    if gather is not None and rank == 0 and args.force_gather:
        ac, gk, gd = gather.last
        k0, d0 = ctx.batch_download(0, True, strings)
        n0 = int(ac[0, 0].item())
        assert n0 == len(k0) and np.array_equal(gk[0][0, :n0].cpu().numpy().view(np.uint32), np.stack([k0[f].view(np.uint32) for f in k0.dtype.names], 1))
        assert np.array_equal(gd[0][0, :n0].cpu().numpy(), d0)
        fl = args.batch - 1                      # a frame whose content differs from the timed steps' frame at that slot
        kl, dl = ctx.batch_download(fl, True, strings)
        nl = int(ac[0, fl].item())
        assert nl == len(kl) and np.array_equal(gd[0][fl, :nl].cpu().numpy(), dl)
        print('gather self-test ok', file=sys.stderr)
    if world > 1 or args.force_gather:
        dist.destroy_process_group()


def load_traffic(frames_per_launch):
    """HBM bytes per k_detect launch from the committed rocprofv3 PMC passes (profiles/traffic_k_detect.json, made by
    tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE runs), scaled to this run's frames per launch; or None."""
    p = os.path.join(ROOT, "profiles", "traffic_k_detect.json")
    try:
        d = json.load(open(p))
        return round(d["hbm_bytes_per_launch"] * frames_per_launch / d["frames_per_launch"])
    except Exception:
        return None


class ResultGather:
    """RCCL gather of the packed per-frame results to rank 0 (BASELINE config 3).  Counts first (all_gather),
    then one gather of fixed-size padded slabs sized to the largest rank's payload."""

    def __init__(self, ctx, batch, strings, dev, rank, world):
        import ctypes as C
        import torch
        self.torch = torch
        self.dev, self.rank, self.world, self.batch, self.strings = dev, rank, world, batch, strings
        self.kpad = 1536
        self.pg = None
        self.last = None
        L = ctx._L
        vp = C.c_void_p
        d_det, d_desc_n, d_kd, d_kp, d_desc = vp(), vp(), vp(), vp(), vp()
        stride, cap, pitch = C.c_int(), C.c_int(), C.c_int()
        ctx.check(L.brisk_hip_batch_results(ctx._h, C.byref(d_det), C.byref(d_desc_n), C.byref(stride), C.byref(d_kd),
                                            C.byref(d_kp), C.byref(d_desc), C.byref(cap), C.byref(pitch)))
        self.cap, self.pitch, self.cstride = cap.value, pitch.value, stride.value
        self.counts = self._wrap(d_desc_n.value, (batch, self.cstride // 4), torch.int32, (self.cstride // 4, 1))[:, 0]
        self.kps = self._wrap(d_kp.value, (batch, self.cap, 7), torch.float32, (self.cap * 7, 7, 1))
        self.desc = self._wrap(d_desc.value, (batch, self.cap, self.pitch), torch.uint8, (self.cap * self.pitch, self.pitch, 1))

    def _wrap(self, ptr, shape, dtype, strides):
        torch = self.torch
        itemsize = torch.tensor([], dtype=dtype).element_size()
        typestr = {torch.int32: "<i4", torch.float32: "<f4", torch.uint8: "|u1"}[dtype]

        class _A:
            pass
        a = _A()
        a.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 3,
                                      "strides": tuple(s * itemsize for s in strides)}
        return torch.as_tensor(a, device=self.dev)

    def run(self):
        """fixed-size slabs, no host synchronisation inside the timed region; asynchronous: the transfer of this batch
        overlaps the next batch's kernels (sharding.PaddedGather), finish() waits for what is still in flight"""
        from ethzasl_brisk_amd import sharding
        if self.pg is None or self.pg.kpad != self.kpad:
            if self.pg is not None:
                self.pg.finish()
            self.pg = sharding.PaddedGather(self.counts, self.kps, self.desc, self.strings, self.kpad, dst=0)
        self.pg.start()

    def finish(self):
        if self.pg is not None:
            self.last = self.pg.finish()

    def check_kpad(self):
        """outside the timed region: the slab size must cover every frame of the batch (rounded up to 128)"""
        self.finish()
        m = self.counts.max().to(self.torch.int64).reshape(1)
        if self.world > 1:  # every rank must cut slabs of the same shape
            import torch.distributed as dist
            dist.all_reduce(m, op=dist.ReduceOp.MAX)
        self.kpad = min(self.cap, (int(m.item()) + 127) // 128 * 128)


if __name__ == "__main__":
    main()
