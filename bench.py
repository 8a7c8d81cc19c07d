#!/usr/bin/env python3
"""bench.py - frames/sec detect+describe @1080p on N MI355X GPUs (BASELINE.json metric).

A "step" is one pass of the hot path (pyramid -> AGAST detect -> NMS/refine -> describe) over one batch of synthetic
1080p frames that already sit in HBM (BASELINE config 2: Appendix-C recipe, 4 octaves, threshold 80, ~1k keypoints per
frame).  The batch of a step is `--batch x --inner` frames (default 512 x 16 = 8192 per GPU; 256-frame chunks until the
end of round 4: 2 % slower, `tools/sweep_batch.py`): the engine takes it in
chunks of `--batch` frames through one workspace, so that a step is ~0.1-0.2 s of GPU work and the timed region
lasts seconds (clocks settled), while the per-chunk time stays comparable between rounds (`config.ms_per_chunk`).

Launch: `python bench.py --gpus N ...` starts N ranks itself (fresh child processes, one per GPU, before anything has
touched the GPU) unless it already runs under a launcher (RANK set, e.g. `python -m torch.distributed.run`).  Frames
shard over ranks with no data-path collective; with N > 1 every chunk ends with the asynchronous RCCL gather of the
keypoints + descriptors to rank 0 that BASELINE config 3 describes.  Default is weak scaling (fixed frames per GPU);
`--frames 512` is config 3 literally: 512 frames split by sharding.shard_frames (strong scaling).
`--dry --backend gloo` runs the same launcher / rendezvous / sharding / gather / timing plumbing on CPU tensors without
the engine (CI without a GPU: tests/test_bench_launcher.py).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

W, H, OCTAVES, THRESHOLD = 1920, 1080, 4, 80
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable with a float4 copy
METRIC = "frames/sec detect+describe @1080p (1/2/4/8 GPU); % HBM roofline"


def layer_sizes(w, h, octaves):
    sizes = [(w, h)]
    if octaves:
        sizes.append((2 * (w // 3), 2 * (h // 3)))
        for i in range(2, 2 * octaves):
            sizes.append((sizes[i - 2][0] // 2, sizes[i - 2][1] // 2))
    return sizes


def algorithmic_bytes(w, h, octaves, kp):
    """SURVEY §8(d) stage model (every stage reads its inputs once and writes its outputs once), bytes per frame.
    Returned per kernel group of the engine: the groups partition the model's S1..S5 (sum == the survey's total)."""
    px = [a * b for a, b in layer_sizes(w, h, octaves)]
    P = sum(px)
    parents = sum(px[0 if i == 1 else i - 2] for i in range(1, len(px)))
    groups = {
        "pyramid": parents + (P - px[0]),          # S1
        "detect": 2 * P,                           # S2: threshold map + detection + score
        "nms": P + 28 * kp,                        # S3: NMS / refinement
        "integral": px[0] + 4 * (w + 1) * (h + 1),  # S4
        "describe": kp * 8524,                     # S5: K (2*66*(4*1+12*4) + 2*792 + 48 + 28)
    }
    return groups


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (CPU port of the reference path) on the GPU box's host cores.  The worker processes are
# forked BEFORE this process loads torch / HIP, build their own extractor and frames, and then wait; the measurement
# itself is a steady-state window (every worker loops over its frames until the deadline), so process start-up, table
# construction and frame generation are outside the timed interval.
# ------------------------------------------------------------------------------------------------------------------
def _cpu_worker(idx, conn, nframes):
    import numpy as np  # noqa: F401
    import oracle_lib as O
    import synth
    ext = O.Extractor()
    frames = [synth.frame_1080p(500000 + idx * 16 + i) for i in range(nframes)]

    def one(img):
        k = O.detect(img, THRESHOLD, OCTAVES)
        k2, _ = ext.compute(img, k)
        return len(k2)

    one(frames[0])  # warm-up (page faults, lazy binding)
    conn.send(("ready", idx))
    while True:
        msg = conn.recv()
        if msg[0] == "quit":
            return
        seconds = msg[1]
        n = 0
        kp = 0
        t0 = time.perf_counter()
        while True:
            kp += one(frames[n % len(frames)])
            n += 1
            dt = time.perf_counter() - t0
            if dt >= seconds:
                break
        conn.send(("done", n, dt, kp))


def gen_frames(fn, seeds, sharers=1):
    """synthetic frames on a thread pool (NumPy releases the GIL in the array passes: 0.4 s per 1080p frame, 1.4 s per 4K
    frame on one thread); `sharers` ranks of one node split the CPU quota"""
    from concurrent.futures import ThreadPoolExecutor
    nthreads = max(1, min(len(seeds), usable_cores() // max(sharers, 1)))
    if nthreads == 1:
        return [fn(s) for s in seeds]
    with ThreadPoolExecutor(nthreads) as ex:
        return list(ex.map(fn, seeds))


def usable_cores():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (cpu.max) - on the GPU
    boxes 256 hardware threads are visible but the container's quota is 16 CPUs."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


class CpuBaseline:
    def __init__(self, max_workers=128, frames_per_worker=3):
        import multiprocessing as mp
        self.visible = len(os.sched_getaffinity(0))
        self.cores = max(1, min(usable_cores(), max_workers))
        ctx = mp.get_context("fork")
        self.procs, self.conns = [], []
        for i in range(self.cores):
            a, b = ctx.Pipe()
            p = ctx.Process(target=_cpu_worker, args=(i, b, frames_per_worker), daemon=True)
            p.start()
            b.close()
            self.procs.append(p)
            self.conns.append(a)

    def _window(self, conns, seconds):
        for c in conns:
            c.send(("go", seconds))
        res = [c.recv() for c in conns]
        fps = [r[1] / r[2] for r in res]
        frames = sum(r[1] for r in res)
        kps = sum(r[3] for r in res)
        return fps, frames, kps

    def measure(self, single_seconds=3.0, all_seconds=8.0):
        for c in self.conns:  # every worker has built its tables and frames
            assert c.recv()[0] == "ready"
        fps1, n1, _ = self._window(self.conns[:1], single_seconds)
        fps, n, kps = self._window(self.conns, all_seconds)
        for c in self.conns:
            c.send(("quit",))
        for p in self.procs:
            p.join(timeout=10)
        model = "unknown"
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        agg = sum(fps)
        return {"value": round(agg, 2), "unit": "frames/s", "cores": self.cores, "kind": "port",
                "single_thread_fps": round(fps1[0], 3), "per_core_fps_all_busy": round(agg / self.cores, 3),
                "cpu_model": model, "hardware_threads_visible": self.visible,
                "sample": "oracle detect+describe on synthetic 1080p frames (same recipe/params), one process per core "
                          "(%d), steady-state window of %.0f s after start-up (tables, frames, one warm-up frame outside "
                          "the window): %d frames, mean %d keypoints/frame; single-thread window %.0f s alone: %d frames"
                          % (self.cores, all_seconds, n, kps // max(n, 1), single_seconds, n1)}


# ------------------------------------------------------------------------------------------------------------------
# launcher: N fresh child ranks, started before this process touches torch / HIP
# ------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, deadline_s=None):
    """N fresh children, one per GPU.  All of them are watched: the first rank that exits non-zero (bad GPU, import
    error, RCCL initialisation ...) ends the run - the others would otherwise sit in a rendezvous or a collective until
    some timeout - and an overall deadline bounds the whole launch.  Only child processes are ever killed."""
    import threading
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    deadline_s = deadline_s or float(os.environ.get("BRISK_BENCH_DEADLINE_S", "1800"))
    children = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "BRISK_BENCH_CHILD": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                         stdout=subprocess.PIPE if r == 0 else sys.stderr))
    chunks = []
    drain = threading.Thread(target=lambda: chunks.append(children[0].stdout.read()), daemon=True)
    drain.start()
    t_end = time.monotonic() + deadline_s
    failed = None
    while True:
        rcs = [c.poll() for c in children]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = "rank %d exited with code %d" % bad[0]
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() > t_end:
            failed = "deadline of %.0f s exceeded" % deadline_s
            break
        time.sleep(0.05)
    if failed:
        for c in children:
            if c.poll() is None:
                c.terminate()
        t_kill = time.monotonic() + 5.0
        for c in children:
            try:
                c.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                c.kill()
                c.wait()
        sys.stderr.write("bench.py: %s; rank exit codes %r\n" % (failed, [c.returncode for c in children]))
        sys.exit(1)
    drain.join(timeout=10)
    out0 = b"".join(chunks)
    lines = [ln for ln in out0.decode().splitlines() if ln.strip().startswith("{")]
    if len(lines) != 1:
        sys.stderr.write("bench.py: expected one JSON line from rank 0, got %d\n" % len(lines))
        sys.exit(1)
    sys.stdout.write(lines[0] + "\n")
    sys.stdout.flush()


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512, help="frames per engine call (chunk) per GPU")
    ap.add_argument("--inner", type=int, default=16, help="chunks per step: a step is batch x inner frames per GPU")
    ap.add_argument("--frames", type=int, default=0,
                    help="BASELINE config 3: this many frames in total per step, split over the ranks by "
                         "sharding.shard_frames (strong scaling); 0 = weak scaling with --batch x --inner per GPU")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic frames per GPU (ring)")
    ap.add_argument("--streams", type=int, default=1, help="internal stream slices per chunk (1..8)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo with --dry)")
    ap.add_argument("--dry", action="store_true", help="no engine, no GPU: launcher / sharding / gather / timing plumbing only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the PCIe-fed (host frames) measurement")
    ap.add_argument("--force-gather", action="store_true", help="run the RCCL result gather even with one rank (self-test)")
    ap.add_argument("--gather", default="torch", choices=["torch", "capi"],
                    help="result gather of the N-GPU batch path: torch.distributed (RCCL through torch) or the engine's own "
                         "C-ABI communicator (brisk_hip_comm_*: RCCL directly, what a C++ host uses)")
    ap.add_argument("--debug-flags", type=lambda v: int(v, 0), default=0, help="timing experiments only (results become wrong)")
    ap.add_argument("--pattern-version", type=int, default=2)
    ap.add_argument("--integral-format", default="auto", choices=["auto", "24", "32"],
                    help="brisk_hip_set_integral_format: element size of the integral image (auto = from the previous batch's candidate "
                         "density: 24 bits on this sparse stream); reported in config.integral_format_bits")
    ap.add_argument("--min-region-s", type=float, default=2.0,
                    help="--frames mode: the step is repeated until the timed region is at least this long")
    ap.add_argument("--no-other-configs", action="store_true", help="skip BASELINE configs 1 / 4 / 5 and the dense regimes (config.other_configs)")
    ap.add_argument("--config", default="", help="profiling: run ONLY this entry of config.other_configs (1, 4, 4_uniform, 5, dense30, dense50, threads, multi_image) "
                                                 "for --config-seconds and print its object")
    ap.add_argument("--config-seconds", type=float, default=0.5)
    ap.add_argument("--fail-rank", type=int, default=-1, help="launcher self-test: this rank exits with code 7 at start-up")
    return ap.parse_args(argv)


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args.gpus, argv)
    # Exactly ONE line may reach stdout (the JSON).  Libraries (RCCL prints a version banner at exit) write to the
    # C-level stdout, so fd 1 is pointed at stderr for the whole run and the JSON is written to the saved fd.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        sys.exit(2)
    if args.fail_rank == rank:
        sys.stderr.write("bench.py: rank %d fails on request (--fail-rank)\n" % rank)
        sys.exit(7)

    # the CPU baseline workers are forked before torch / HIP exist in this process (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.dry and not args.config:
        cpu = CpuBaseline()

    import numpy as np
    import torch
    import torch.distributed as dist
    from ethzasl_brisk_amd import sharding

    use_dist = world > 1 or args.force_gather
    ctl = None  # gloo control group: barriers, timing reduction and the collective "keep the gather?" decision
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.dry:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(args.backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        ctl = dist.new_group(backend="gloo")
        assert dist.get_world_size() == args.gpus

    # ---- work split
    if args.frames:
        mine = sharding.shard_frames(args.frames, rank, world)
        chunk = len(mine)
        chunk_max = max(len(sharding.shard_frames(args.frames, r, world)) for r in range(world))
        inner = 1
        scaling = "strong"
        frames_per_step_total = args.frames
    else:
        mine = list(range(args.batch))
        chunk = chunk_max = args.batch
        inner = args.inner
        scaling = "weak"
        frames_per_step_total = args.batch * args.inner * world
    assert chunk > 0, "rank without frames"

    if args.dry:
        return dry_run(args, rank, world, chunk, chunk_max, inner, scaling, frames_per_step_total, ctl, real_stdout)

    import synth
    import ethzasl_brisk_amd as B
    if torch.cuda.device_count() <= local_rank:
        sys.stderr.write("bench.py: rank %d wants cuda:%d but only %d device(s) are visible\n" % (rank, local_rank, torch.cuda.device_count()))
        sys.exit(4)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    if args.config:
        o = other_configs(local_rank, only=args.config, seconds=args.config_seconds)
        os.write(real_stdout, (json.dumps(o) + "\n").encode())
        return
    # ---- synthetic frame ring, resident in HBM before the timed region (frame seed = global frame index)
    nd = max(1, min(args.distinct, chunk))
    seeds = [(mine[i] if args.frames else rank * 100000 + i) for i in range(nd)]
    host = np.stack(gen_frames(synth.frame_1080p, seeds, world))
    ring = torch.from_numpy(host).to(dev)
    idx = torch.arange(chunk, device=dev) % nd
    frames = ring[idx].contiguous()          # [chunk, H, W] u8 in HBM
    del ring

    ctx = B.Context(local_rank)
    ctx.set_integral_format({"auto": 0, "24": 24, "32": 32}[args.integral_format])
    ctx.set_streams(args.streams)
    ctx.debug_set_flags(args.debug_flags)
    ext = B.BriskDescriptorExtractor(version=args.pattern_version, context=ctx)
    # everything below runs on ONE explicit torch stream: the engine's launches, the slab copies of the gather and
    # (through torch.distributed's stream synchronisation) the RCCL transfers are ordered on it.
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    stream = work_stream.cuda_stream
    strings = ext.descriptorSize()

    def run_chunk():
        ctx.detect_describe_batch(ext, frames.data_ptr(), chunk, W, H, W * H, W, THRESHOLD, OCTAVES, stream)

    gather = None
    gather_note = ""
    if use_dist:
        run_chunk()                          # allocates the engine's result buffers
        torch.cuda.synchronize()
        if args.gather == "capi":
            gather = CapiGather(ctx, chunk, strings, dev, rank, world, chunk_max, ctl, stream)
        else:
            gather = ResultGather(ctx, chunk, strings, dev, rank, world, chunk_max)

    def step():
        for _ in range(inner):
            run_chunk()
            if gather:
                gather.run()

    ok = 1
    try:
        for _ in range(max(args.warmup, 1) if gather else args.warmup):
            step()
        torch.cuda.synchronize()
    except Exception as e:
        if gather is None or args.force_gather:
            raise
        print("rank %d: result gather failed in warm-up (%r)" % (rank, e), file=sys.stderr)
        ok = 0
    if ctl is not None:
        # the decision is collective and taken on the gloo control group: either every rank keeps the gather or none
        # does.  A broken RCCL communicator is never reused (barriers and the timing reduction run on gloo).
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=ctl)
        if int(flag.item()) == 0 and gather is not None:
            gather = None
            gather_note = " [RCCL gather failed in warm-up on some rank and was left out on all ranks]"
            try:
                torch.cuda.synchronize()
            except Exception:
                sys.exit(3)
    assert ctx.batch_status(chunk) == 0 or args.debug_flags
    if gather:
        gather.check_kpad(ctl)
        run_chunk()
        gather.run()                         # slabs of the final size exist before the timed region
        gather.finish()
        torch.cuda.synchronize()
    mean_kp = float(np.mean([len(ctx.batch_download(f, True, strings)[0]) for f in range(min(chunk, 8))]))

    # --frames (config 3, strong scaling): one step is a few milliseconds per rank; it is repeated until the timed region
    # is at least --min-region-s long (every rank uses the same count: the slowest rank's probe decides)
    reps = 1
    if args.frames:
        if ctl is not None:
            dist.barrier(group=ctl)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        step()
        if gather:
            gather.finish()
        torch.cuda.synchronize()
        tp = torch.tensor([time.perf_counter() - tp], dtype=torch.float64)
        if ctl is not None:
            dist.all_reduce(tp, op=dist.ReduceOp.MAX, group=ctl)
        reps = max(1, int(np.ceil(1.1 * args.min_region_s / max(float(tp.item()) * args.steps, 1e-9))))
    for attempt in range(4):
        ctx.profile_enable(True)             # (resets the accumulated stage times)
        if ctl is not None:
            dist.barrier(group=ctl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps * reps):
            step()
        if gather:
            gather.finish()                      # the last transfers are part of the timed region
        torch.cuda.synchronize()
        dt_rank = time.perf_counter() - t0       # this rank's own time (before the barrier)
        if ctl is not None:
            dist.barrier(group=ctl)
        dt = time.perf_counter() - t0
        if ctl is not None:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=ctl)
            dt = float(t.item())
        if not args.frames or dt >= args.min_region_s or attempt == 3:
            break
        reps = int(np.ceil(reps * args.min_region_s / dt * 1.15))  # the probe was not representative: a longer region
    stage_ms, ncalls = ctx.profile_read()
    fpl = ctx.profile_frames_per_launch() or chunk   # frames per timed kernel launch (one stream slice)
    ctx.profile_enable(False)

    # per-rank throughput (own frames / own time) and what the result gather costs when nothing hides it
    nsteps = args.steps * reps
    rank_fps = torch.zeros(world, dtype=torch.float64)
    rank_fps[rank] = chunk * inner * nsteps / dt_rank
    if ctl is not None:
        dist.all_reduce(rank_fps, op=dist.ReduceOp.SUM, group=ctl)
    gather_ms = None
    if gather:
        torch.cuda.synchronize()
        if ctl is not None:
            dist.barrier(group=ctl)
        tg = time.perf_counter()
        for _ in range(4):
            gather.run()
        gather.finish()
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) / 4 * 1e3 * inner
        gather.verify_counts()   # outside every timed region: no frame had more rows than the slabs carry

    total_frames = frames_per_step_total * nsteps
    fps = total_frames / dt
    integral_bits = ctx.debug_integral_bits(0)   # what the timed region's batches used (every batch alike: same stream)
    groups = algorithmic_bytes(W, H, OCTAVES, int(round(mean_kp)))
    per_frame_bytes = sum(groups.values())
    traffic = load_traffic(ctx)
    kgroups = kernel_groups(stage_ms, groups, fpl, traffic)
    # the dominant KERNEL (largest HIP-event interval of a single stage) carries the roofline object; its algorithmic
    # bytes are those of its SURVEY 8(d) stage
    group_of_stage = {"k_pyramid": "pyramid", "k_detect": "detect", "k_classify_refine": "nms", "k_tie_resolve": "nms",
                      "k_finalize": "nms", "k_postfilter": "nms", "k_integral_final": "integral", "k_desc_prepare": "describe", "k_describe": "describe"}
    dom_stage = max(stage_ms, key=lambda k: stage_ms[k])
    dom = group_of_stage[dom_stage]
    dom_ms = stage_ms[dom_stage]
    dom_alg = groups[dom] * fpl
    dom_gb = dom_alg / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    dom_traffic = None
    if traffic and dom_stage in traffic.get("kernels", {}):
        dom_traffic = round(traffic["kernels"][dom_stage]["hbm_bytes_per_launch"] * fpl / traffic["frames_per_launch"])
    out = {
        "metric": METRIC,
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / nsteps * 1e3, 4), "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "1080p synthetic textured stream (SURVEY App. C recipe, 300 rects), 4 octaves, "
                               "AGAST threshold 80, default 66-point pattern (48-byte descriptors)",
                   "frames_per_step_per_gpu": chunk * inner, "frames_per_step_total": frames_per_step_total,
                   "chunk_frames": chunk, "chunks_per_step": inner,
                   "ms_per_chunk": round(dt / nsteps / inner * 1e3, 4),
                   "timed_region_s": round(dt, 3), "steps_effective": nsteps,
                   "per_rank_frames_per_s": [round(float(v), 1) for v in rank_fps],
                   "gather_ms_per_step": None if gather_ms is None else round(gather_ms, 4),
                   "gather_note": None if gather_ms is None else "result gather of one step run alone after the timed region (inside it the transfers overlap the next chunk's kernels)",
                   "stream_slices": args.streams, "frames_per_kernel_launch": fpl, "distinct_frames_per_gpu": nd,
                   "mean_keypoints_per_frame": round(mean_kp, 1),
                   "integral_format_bits": integral_bits, "integral_format_setting": args.integral_format,
                   "parallelism": "frames sharded over %d rank(s)%s" % (world, ((", asynchronous RCCL gather of keypoints+descriptors to rank 0 after every chunk (overlaps the next chunk; %s)" % ("C-ABI communicator brisk_hip_comm_*" if args.gather == "capi" else "torch.distributed")) if (world > 1 and gather) else "") + gather_note),
                   "algorithmic_MB_per_frame": round(per_frame_bytes / 1e6, 3),
                   "pipeline_achieved_GBps": round(per_frame_bytes * fps / 1e9, 2),
                   "pipeline_frac_of_hbm_peak": round(per_frame_bytes * fps / 1e9 / (HBM_PEAK_GBS * world), 5),
                   "stage_ms_per_chunk": {k: round(v, 4) for k, v in stage_ms.items()},
                   "kernel_groups": kgroups,
                   "stage_note": "HIP-event intervals on the launch stream of one chunk (average over the timed "
                                 "region); kernel_groups: algorithmic bytes of SURVEY 8(d) per group x frames per launch / "
                                 "interval; hbm_bytes from the committed rocprofv3 PMC passes when they belong to this "
                                 "kernel revision, else null"},
        # "bound": the roofline the fraction is PRICED against (the contract knows "hbm" and "mfma"; there is no MFMA work on
        # this path); "bound_by": what the kernel's time actually follows
        "roofline": {"bound": "hbm", "bound_by": "l1_gather" if dom_stage == "k_describe" else ("valu_issue" if dom_stage in ("k_detect", "k_classify_refine") else "hbm"),
                     "kernel": dom_stage, "achieved": round(dom_gb, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(dom_gb / HBM_PEAK_GBS, 5), "traffic": dom_traffic,
                     # the same fraction on the COUNTER-measured HBM bytes of that kernel (committed PMC passes of this kernel
                     # revision; null when they belong to another revision)
                     "hbm_frac": None if not dom_traffic or dom_ms <= 0 else round(dom_traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                     "algorithmic_bytes_per_launch": dom_alg, "avg_launch_ms": round(dom_ms, 4), "launches_timed": ncalls,
                     "note": ("k_describe is bound by the CU's texture path - the address unit is busy 86 %, the data-return unit 93 %, the L1 96 % of the "
                              "kernel's clocks (round 6 counters: profiles/r06_describe_overlap.txt) - i.e. by the NUMBER of gather look-ups (ten "
                              "gathers per sample, 88 bytes used; one look-up per lane whose line no neighbouring lane shares, which is why the two "
                              "lanes of a lane pair read the same row since round 5), not by HBM bandwidth, occupancy, instruction issue or exposed "
                              "latency (a two-deep gather pipeline gains 1.7 %): DESIGN.md 5, profiles/r05_microbench_il2.txt; "
                              if dom_stage == "k_describe" else "")
                             + "every kernel group: config.kernel_groups"},
    }
    if rank == 0:
        try:
            cp, rd = ctx.stream_ceiling(1 << 30)
            out["config"]["box_streaming_ceiling_GBps"] = {"float4_copy_read_plus_write": round(cp, 1),
                                                           "float4_read_only": round(rd, 1)}
        except Exception as e:  # reported, never fatal for the measurement
            out["config"]["box_streaming_ceiling_GBps"] = "failed: %r" % (e,)
        if world == 1 and not args.no_host_fed and hasattr(ctx, "detect_describe_batch_host"):
            out["config"]["pcie_fed"] = host_fed(ctx, ext, host, chunk, strings)
        if world == 1 and not args.no_other_configs:
            try:
                out["config"]["other_configs"] = other_configs(local_rank)
            except Exception as e:  # reported, never fatal for the bench line
                out["config"]["other_configs"] = "failed: %r" % (e,)
        if cpu is not None:
            out["cpu_baseline"] = cpu.measure()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if gather is not None and args.force_gather:
        gather_selftest(ctx, ext, frames, chunk, strings, stream, gather, rank)
    if gather is not None:
        gather.close()   # (the C-ABI communicator: brisk_hip_comm_destroy)
    if use_dist:
        dist.destroy_process_group()


def _timed_calls(fn, sync, seconds, min_reps=3):
    """fn() repeated for about `seconds` (after one untimed call); returns seconds per call"""
    fn()
    sync()
    t0 = time.perf_counter()
    fn()
    sync()
    one = max(time.perf_counter() - t0, 1e-6)
    reps = max(min_reps, int(seconds / one))
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps, reps


def config5_keypoints(n=100000, seed=7):
    """x, y uniform, size log-uniform in [8.64, 200] (scale indices 5..63), angle -1"""
    import numpy as np
    import ethzasl_brisk_amd as B
    rng = np.random.default_rng(seed)
    kp = np.zeros(n, B.KEYPOINT)
    kp["size"] = np.exp(rng.uniform(np.log(8.64), np.log(200.0), n)).astype(np.float32)
    kp["x"] = rng.uniform(0, W, n).astype(np.float32)
    kp["y"] = rng.uniform(0, H, n).astype(np.float32)
    kp["angle"] = -1
    kp["class_id"] = -1
    return kp


def other_configs(device, only="", seconds=0.4):
    try:
        return _other_configs(device, only, seconds)
    finally:
        import torch
        torch.cuda.empty_cache()


def _other_configs(device, only="", seconds=0.4):
    """The BASELINE configurations that are not the bench line, and the dense regimes of config 2 - measured after the
    timed region, never part of `value`.  Every entry: workload, rate, algorithmic MB (SURVEY 8(d) model), fraction of
    the 8 TB/s peak that rate x bytes is, and the engine's HIP-event stage intervals."""
    import numpy as np
    import torch
    import synth
    import ethzasl_brisk_amd as B
    dev = torch.device("cuda", device)
    sync = torch.cuda.synchronize
    stream = torch.cuda.current_stream().cuda_stream
    res = {}

    def want(name):
        return not only or only == name

    def entry(workload, per_s, unit, alg_bytes, units_per_call, stage_ms, extra=None):
        # alg_bytes: SURVEY 8(d) bytes per unit (frame or descriptor); algorithmic_MB: per call
        gb = alg_bytes * per_s / 1e9
        e = {"workload": workload, "value": round(per_s, 1), "unit": unit, "algorithmic_MB": round(alg_bytes * units_per_call / 1e6, 3),
             "algorithmic_MB_per": "call of %d %s" % (units_per_call, unit.split("/")[0]),
             "achieved_GBps": round(gb, 1), "frac": round(gb / HBM_PEAK_GBS, 5),
             "stage_ms": {k: round(v, 4) for k, v in stage_ms.items() if v > 0}, "units_per_call": units_per_call}
        if extra:
            e.update(extra)
        return e

    def guarded(fn, *a):
        try:
            return fn(*a)
        except Exception as e:  # one configuration failing must not hide the others
            return {"failed": repr(e)}

    def batch_config(name, frames_np, w, h, octaves, thr, radius, nbatch, workload):
        ctx = B.Context(device, max_candidates=262144 if name.startswith("dense") else 65536,
                        max_keypoints=65536 if name.startswith("dense") else 16384)
        ext = B.BriskDescriptorExtractor(context=ctx)
        d = torch.from_numpy(frames_np).to(dev)
        batch = d[torch.arange(nbatch, device=dev) % len(frames_np)].contiguous()
        del d
        ctx.set_uniformity(radius)
        call = lambda: ctx.detect_describe_batch(ext, batch.data_ptr(), nbatch, w, h, w * h, w, thr, octaves, stream)
        call()
        sync()
        assert ctx.batch_status(nbatch) == 0
        kp = float(np.mean([ctx.debug_counters(f)["described"] for f in range(min(nbatch, 16))]))
        det = float(np.mean([ctx.debug_counters(f)["keypoints"] for f in range(min(nbatch, 16))]))
        cand = float(np.mean([ctx.debug_counters(f)["candidates"] for f in range(min(nbatch, 16))]))
        ctx.profile_enable(True)
        dt, reps = _timed_calls(call, sync, seconds)
        stage_ms, _ = ctx.profile_read()
        ctx.profile_enable(False)
        alg = sum(algorithmic_bytes(w, h, octaves, int(round(kp))).values())
        e = entry(workload, nbatch / dt, "frames/s", alg, nbatch, stage_ms,
                  {"ms_per_frame": round(dt / nbatch * 1e3, 4), "distinct_frames": len(frames_np), "calls_timed": reps,
                   "mean_candidates": round(cand, 1), "mean_keypoints_detected": round(det, 1), "mean_keypoints_described": round(kp, 1)})
        ctx.close()
        del batch
        torch.cuda.empty_cache()
        return e

    if want("1"):
        # config 1: one 640 x 480 frame through the host-buffer calls (what the drop-in classes do)
        ctx = B.Context(device)
        ext = B.BriskDescriptorExtractor(context=ctx)
        det = B.BriskFeatureDetector(70, 4, context=ctx)
        img = synth.frame_vga(1)
        k = det.detect(img)
        k2, _ = ext.compute(img, k)
        ctx.profile_enable(True)
        dt, reps = _timed_calls(lambda: ext.compute(img, det.detect(img)), sync, seconds)
        stage_ms, _ = ctx.profile_read()   # (average over both kinds of call: a stage only counts in the call that runs it)
        ctx.profile_enable(False)
        alg = sum(algorithmic_bytes(640, 480, 4, len(k2)).values())
        res["1"] = entry("BASELINE config 1: one 640x480 frame, threshold 70, 4 octaves, host-buffer detect() + compute() "
                         "(upload, kernels, download; one frame per call: latency-bound)", 1.0 / dt, "frames/s", alg, 1, stage_ms,
                         {"ms_per_frame": round(dt * 1e3, 4), "keypoints_described": int(len(k2)), "calls_timed": reps})
        ctx.close()
    if want("4") or want("4_uniform") or want("4_uniform_single"):
        frames4 = np.stack(gen_frames(synth.frame_4k, [2 + i for i in range(16)]))
        for name, radius in (("4", 0.0), ("4_uniform", 8.0)):
            if want(name):
                res[name] = guarded(batch_config, name, frames4, 3840, 2160, 6, 80, radius, 64,
                                         "BASELINE config 4: 3840x2160, 6 octaves (12 layers), threshold 80, 64 resident frames per call "
                                         "(16 distinct)" + (", uniformity enforcement radius 8 px" if radius else ", no post-filter"))
        if want("4_uniform_single"):  # latency of ONE 4K frame per call (the post-filter is one workgroup per frame)
            res["4_uniform_single"] = guarded(batch_config, "4_uniform_single", frames4[:1], 3840, 2160, 6, 80, 8.0, 1,
                                                   "BASELINE config 4, ONE resident 3840x2160 frame per call, uniformity enforcement radius 8 px "
                                                   "(latency: stage_ms.k_postfilter is the filter alone)")
        del frames4
    if want("dense30") or want("dense50"):
        framesd = np.stack(gen_frames(synth.frame_1080p, [700000 + i for i in range(16)]))
        for name, thr in (("dense50", 50), ("dense30", 30)):
            if want(name):
                res[name] = guarded(batch_config, name, framesd, W, H, OCTAVES, thr, 0.0, 64,
                                         "dense regime: config 2's frames at AGAST threshold %d, 64 resident frames per call (16 distinct)" % thr)
        del framesd
    if want("5"):
        # config 5: descriptor only, 100 000 provided keypoints on one 1080p frame, orientation estimated
        ctx = B.Context(device, max_candidates=65536, max_keypoints=131072)
        ext = B.BriskDescriptorExtractor(context=ctx)
        img = synth.frame_1080p(0)
        kp = config5_keypoints()
        k2, dd = ext.compute(img, kp)
        nd = len(k2)
        dt_py, _ = _timed_calls(lambda: ext.compute(img, kp), sync, seconds / 2)
        # the C ABI itself (include/brisk_hip.h: brisk_hip_describe) on buffers the caller reuses, as a C++ caller's
        # vectors would be: keypoints in / out in one array, packed 48-byte descriptor rows
        import ctypes as C
        kbuf = kp.copy()
        dbuf = np.empty((len(kp), ext.descriptorSize()), np.uint8)
        nio = C.c_int()

        def raw():
            kbuf.view(np.uint8)[:] = kp.view(np.uint8)   # (byte views: NumPy copies structured arrays field by field, 25 x slower)
            nio.value = len(kp)
            ctx.check(ctx._L.brisk_hip_describe(ctx._h, ext._h, img.ctypes.data_as(C.c_void_p), W, H, W, kbuf.ctypes.data_as(C.c_void_p),
                                                C.byref(nio), dbuf.ctypes.data_as(C.c_void_p), dbuf.shape[1], 1, 1))
        raw()
        assert nio.value == nd and np.array_equal(dbuf[:nd], dd)
        ctx.profile_enable(True)
        dt, reps = _timed_calls(raw, sync, seconds)
        stage_ms, _ = ctx.profile_read()
        ctx.profile_enable(False)
        dev_ms = sum(stage_ms.values())
        # integral build 10.38 + one read of image and integral 10.38 + kp / descriptor I/O (the reference's 51.9 MB
        # LUT is 1.2 MB of factorised tables here and not counted)
        alg = (W * H + 4 * (W + 1) * (H + 1)) * 2 + len(kp) * 28 + nd * (28 + 48)
        res["5"] = entry("BASELINE config 5: descriptor only, 100000 provided keypoints (size log-uniform 8.64..200, angle -1) on one "
                         "1080p frame, brisk_hip_describe on host buffers incl. transfers", nd / dt, "descriptors/s", alg / max(nd, 1), nd, stage_ms,
                         {"ms_per_call": round(dt * 1e3, 4), "described": nd, "calls_timed": reps,
                          "device_ms_per_call": round(dev_ms, 4),
                          "device_descriptors_per_s": round(nd / (dev_ms * 1e-3), 1) if dev_ms > 0 else None,
                          "device_frac": round(alg / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if dev_ms > 0 else None,
                          "python_wrapper_ms_per_call": round(dt_py * 1e3, 4),
                          "note": "value / frac: the whole C-ABI call on pageable host buffers (refilling the 2.8 MB keypoint array, H2D of "
                                  "the 2 MB image + keypoints, kernels, D2H of keypoints + descriptors); device_*: the HIP-event stage "
                                  "intervals only; python_wrapper: the same through ethzasl_brisk_amd.BriskDescriptorExtractor.compute "
                                  "(fresh numpy arrays per call)"})
        ctx.close()
    if want("threads"):
        res["threads"] = guarded(threads_table, seconds)
    if want("multi_image"):
        res["multi_image"] = guarded(multi_image_table, seconds)
    return res


def multi_image_table(seconds):
    """The classes' multi-image overloads (cv::FeatureDetector::detect(vector<Mat>), cv::DescriptorExtractor::compute(vector<Mat>) of the
    reference's OpenCV bases) on N separate pageable 1080p buffers per call: tests/cpp/test_multi_image --time, a child process per row"""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_cpp_classes as tc
    exe = tc.build_binary("test_multi_image")
    rows = []
    for n, flags in ((1, []), (8, []), (64, []), (256, []), (64, ["--same-image"]), (256, ["--same-image"])):
        r = subprocess.run([exe, "--time", str(n), str(max(1.0, 3 * seconds))] + flags, capture_output=True, text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        rows.append(json.loads(line[-1]) if (r.returncode == 0 and line) else {"images_per_call": n, "failed": (r.stdout + r.stderr)[-300:]})
    return {"workload": "brisk::BriskFeatureDetector::detect(vector<Mat>) + BriskDescriptorExtractor::compute(vector<Mat>, ...) on N separate "
                        "pageable 1080p buffers per call, threshold 80, 4 octaves, one host thread; two uploads per frame (detect and compute "
                        "are separate calls in the reference's API) unless same_image = 1 (compute() under ScopedSameImage: the caller's word "
                        "that it gets detect()'s unchanged buffers)", "unit": "frames/s", "rows": rows}


def threads_table(seconds):
    """The drop-in C++ classes (include/brisk/*.h) under 1 ... 32 host threads: tests/cpp/test_threads --time, a child process per
    row; distinct 1080p frames per thread, detect() + compute() per frame, every result compared with the serial run inside the
    child.  Rows: the classes' default policy (a context per thread; the device's shared call-combining pool from more concurrent
    callers than CPUs on), then the same thread counts with the policy forced one way or the other and with ScopedSameImage (the
    caller's word that compute() gets detect()'s unchanged buffer)."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_cpp_classes as tc
    exe = tc.build_binary("test_threads")
    secs = str(max(1.0, 3 * seconds))

    def run(n, *flags):
        r = subprocess.run([exe, "--time", str(n), secs] + list(flags), capture_output=True, text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"threads": n, "flags": list(flags), "failed": (r.stdout + r.stderr)[-300:]}
        d = json.loads(line[-1])
        d["flags"] = list(flags)
        return d
    rows = [run(n) for n in (1, 2, 4, 8, 16, 32)]
    extra = [run(16, "--pool-threshold", "1"), run(32, "--pool-threshold", "0"), run(16, "--same-image"), run(32, "--same-image"),
             run(32, "--same-image", "--pool-threshold", "1")]
    ok = [r for r in rows if "frames_per_s" in r]
    base = ok[0]["frames_per_s"] if ok and ok[0]["threads"] == 1 else None
    return {"workload": "drop-in classes brisk::BriskFeatureDetector::detect + BriskDescriptorExtractor::compute, one 1080p frame per "
                        "call pair, threshold 80, 4 octaves, N host threads, pageable cv::Mat-like buffers; aggregate over threads. "
                        "rows: default policy (own context per thread; shared pool from more callers than CPUs on); "
                        "forced: --pool-threshold 1 = every call through the pool, 0 = never; --same-image = ScopedSameImage",
            "unit": "frames/s", "rows": rows, "forced": extra,
            "speedup_vs_1_thread": None if not base else {str(r["threads"]): round(r["frames_per_s"] / base, 2) for r in ok},
            "limiter": "the HIP runtime serialises the API calls of a process (hipLaunchKernel 4.6 -> 26 us from 1 to 8 threads): "
                       "profiles/r06_threads_limiter.txt",
            "host_cpus": len(os.sched_getaffinity(0))}


def kernel_groups(stage_ms, groups, fpl, traffic):
    """{group: {alg_bytes, ms, GBps, frac, hbm_bytes}} per launch (fpl frames), from the engine's per-stage HIP-event
    intervals.  Stage names -> groups of SURVEY 8(d)."""
    stage_of = {"pyramid": ["k_pyramid"], "detect": ["k_detect"],
                "nms": ["k_classify_refine", "k_tie_resolve", "k_finalize", "k_postfilter"],
                "integral": ["k_integral_final"], "describe": ["k_desc_prepare", "k_describe"]}
    out = {}
    for g, stages in stage_of.items():
        ms = sum(stage_ms.get(s, 0.0) for s in stages)
        alg = groups[g] * fpl
        gb = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        hb = None
        if traffic and g in traffic.get("groups", {}):
            hb = round(traffic["groups"][g] * fpl / traffic["frames_per_launch"])
        # frac: algorithmic bytes / interval / peak; hbm_frac: the counter-measured HBM bytes instead (what the memory
        # system really moved: below frac where the model charges a re-read the kernel does not do, above it where
        # gathers fetch whole lines)
        out[g] = {"alg_bytes": alg, "ms": round(ms, 4), "GBps": round(gb, 2), "frac": round(gb / HBM_PEAK_GBS, 5),
                  "hbm_bytes": hb,
                  "hbm_frac": None if (hb is None or ms <= 0) else round(hb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
    return out


def load_traffic(ctx):
    """HBM bytes per kernel group from the committed rocprofv3 PMC passes (profiles/traffic.json, made by
    tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE runs); only used when it was measured on this kernel
    revision (brisk_hip_kernel_revision), otherwise None."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        d = json.load(open(p))
        rev = ctx.kernel_revision() if hasattr(ctx, "kernel_revision") else None
        if d.get("kernel_revision") != rev:
            return None
        return d
    except Exception:
        return None


def host_fed(ctx, ext, host, chunk, strings, seconds=1.5):
    """PCIe-fed rates: frames start in pinned HOST memory, the engine's host-batch entry moves them in
    slices over a copy stream while the previous slice computes.  `fps`: results left in HBM (the H2D-only rate);
    `host_to_host_fps`: keypoints + descriptors of every frame back in pinned host memory as well
    (brisk_hip_detect_describe_batch_host_results: exact prefix-summed rows, written by the device on the egress stream
    beside the next batch), two destination sets alternating, every batch waited for before its set is reused.  Never
    `value`: reported next to it."""
    import numpy as np
    import torch
    import ethzasl_brisk_amd as B
    n = min(chunk, 256)
    src = torch.from_numpy(np.ascontiguousarray(host[np.arange(n) % len(host)])).pin_memory()
    ctx.detect_describe_batch_host(ext, src.data_ptr(), n, W, H, W * H, W, THRESHOLD, OCTAVES)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < seconds:
        ctx.detect_describe_batch_host(ext, src.data_ptr(), n, W, H, W * H, W, THRESHOLD, OCTAVES)
        reps += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"fps": round(reps * n / dt, 1), "frames_per_call": n,
           "note": "fps: frames in pinned host memory, H2D on a copy stream overlapped with compute, results left in HBM; "
                   "host_to_host_fps: the same with every frame's keypoints + descriptors back in pinned host memory "
                   "(brisk_hip_detect_describe_batch_host_results, transfer of batch n beside batch n + 1)"}
    if not hasattr(ctx, "detect_describe_batch_host_results"):
        return out
    # rows of one batch (results are still in HBM from the loop above), with room to spare
    rows = int(1.25 * sum(len(ctx.batch_download(f, True, strings)[0]) for f in range(0, n, max(1, n // 8))) * max(1, n // 8)) + 4096
    dsts = [B.HostResults(n, rows, strings, pinned=True) for _ in range(2)]
    tk = [0, 0]

    def issue(i):
        if tk[i & 1]:
            assert ctx.batch_download_wait(tk[i & 1]) == 0
        tk[i & 1] = ctx.detect_describe_batch_host_results(ext, src.data_ptr(), n, W, H, W * H, W, THRESHOLD, OCTAVES, dsts[i & 1])
    issue(0)
    issue(1)
    issue(2)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < seconds:
        issue(reps + 3)
        reps += 1
    for i in (0, 1):
        if tk[i]:
            assert ctx.batch_download_wait(tk[i]) == 0
    dt = time.perf_counter() - t0
    out["host_to_host_fps"] = round(reps * n / dt, 1)
    out["host_to_host_vs_h2d_only"] = round(out["host_to_host_fps"] / out["fps"], 4)
    # outside the timed region: the rows the last transfer delivered against the per-frame download of the same batch
    last = dsts[(reps + 2) & 1]
    checked = []
    for f in (0, n // 2, n - 1):
        k0, d0 = ctx.batch_download(f, True, strings)
        k1, d1 = last.frame(f, strings)
        assert len(k0) == len(k1) and np.array_equal(k0.view(np.uint8), np.ascontiguousarray(k1).view(np.uint8)) and np.array_equal(d0, d1), f
        checked.append(f)
    out["host_to_host_checked_frames"] = checked
    out["host_to_host_MB_per_batch"] = round(int(last.offsets[n]) * (28 + strings) / 1e6, 2)
    return out


def gather_selftest(ctx, ext, frames, chunk, strings, stream, gather, rank):
    import numpy as np
    import torch
    # content the previous steps did not produce (a stale or half-written slab would show)
    frames2 = frames.flip(0).contiguous()
    ctx.detect_describe_batch(ext, frames2.data_ptr(), chunk, W, H, W * H, W, THRESHOLD, OCTAVES, stream)
    gather.run()
    gather.finish()
    torch.cuda.synchronize()
    if rank == 0:
        ac, gk, gd = gather.last
        k0, d0 = ctx.batch_download(0, True, strings)
        n0 = int(ac[0, 0].item())
        assert n0 == len(k0) and np.array_equal(gk[0][0, :n0].cpu().numpy().view(np.uint32), np.stack([k0[f].view(np.uint32) for f in k0.dtype.names], 1))
        assert np.array_equal(gd[0][0, :n0].cpu().numpy(), d0)
        fl = chunk - 1                      # a frame whose content differs from the timed steps' frame at that slot
        kl, dl = ctx.batch_download(fl, True, strings)
        nl = int(ac[0, fl].item())
        assert nl == len(kl) and np.array_equal(gd[0][fl, :nl].cpu().numpy(), dl)
        print('gather self-test ok', file=sys.stderr)


def dry_run(args, rank, world, chunk, chunk_max, inner, scaling, frames_per_step_total, ctl, real_stdout):
    """The N-rank plumbing without the engine: every rank fabricates result buffers on the CPU, the steps run the same
    PaddedGather pipeline, barriers and max-over-ranks timing; value is 0 (nothing was measured)."""
    import torch
    import torch.distributed as dist
    from ethzasl_brisk_amd import sharding
    cap, strings, pitch = 32, 48, 64
    g = torch.Generator().manual_seed(rank)
    counts = torch.randint(0, cap + 1, (chunk,), generator=g, dtype=torch.int32)
    kps = torch.randn((chunk, cap, 7), generator=g)
    desc = torch.randint(0, 256, (chunk, cap, pitch), generator=g, dtype=torch.uint8)
    pg = sharding.PaddedGather(counts, kps, desc, strings, cap, dst=0, frames_max=chunk_max) if world > 1 else None
    def step():
        for _ in range(inner):
            if pg:
                pg.start()
            else:
                time.sleep(0.001)

    reps = 1
    if args.frames:  # same rule as the real run: repeat the step until the timed region is long enough
        if ctl is not None:
            dist.barrier(group=ctl)
        tp = time.perf_counter()
        step()
        if pg:
            pg.finish()
        tp = torch.tensor([time.perf_counter() - tp], dtype=torch.float64)
        if ctl is not None:
            dist.all_reduce(tp, op=dist.ReduceOp.MAX, group=ctl)
        reps = max(1, int(-(-args.min_region_s // max(float(tp.item()) * args.steps, 1e-9))))
    for attempt in range(4):
        if ctl is not None:
            dist.barrier(group=ctl)
        t0 = time.perf_counter()
        for _ in range(args.steps * reps):
            step()
        last = pg.finish() if pg else None
        if ctl is not None:
            dist.barrier(group=ctl)
        dt = time.perf_counter() - t0
        if ctl is not None:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=ctl)
            dt = float(t.item())
        if not args.frames or dt >= args.min_region_s or attempt == 3:
            break
        reps = int(-(-(reps * args.min_region_s * 1.15) // dt))
    if rank == 0:
        if last is not None:
            ac, gk, gd = last
            assert ac.shape == (world, chunk_max) and len(gk) == world and torch.equal(ac[0][:chunk], counts)
        out = {"metric": METRIC, "value": 0.0, "unit": "frames/s", "n_gpus": dist.get_world_size() if world > 1 else 1,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / max(args.steps * reps, 1) * 1e3, 4),
               "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "dry": True,
               "config": {"workload": "DRY RUN: launcher / sharding / gather plumbing on CPU tensors, no engine",
                          "frames_per_step_total": frames_per_step_total, "chunk_frames": chunk, "chunks_per_step": inner,
                          "timed_region_s": round(dt, 3), "steps_effective": args.steps * reps,
                          "shard_sizes": [len(sharding.shard_frames(args.frames, r, world)) for r in range(world)] if args.frames else None,
                          "backend": args.backend}}
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


class ResultGather:
    """RCCL gather of the packed per-frame results to rank 0 (BASELINE config 3).  Counts first (all_gather),
    then one gather of fixed-size padded slabs sized to the largest rank's payload."""

    def __init__(self, ctx, batch, strings, dev, rank, world, frames_max=None):
        import ctypes as C
        import torch
        self.torch = torch
        self.dev, self.rank, self.world, self.batch, self.strings = dev, rank, world, batch, strings
        self.kpad = 1536
        self.frames_max = frames_max or batch
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
        """fixed-size slabs, no host synchronisation inside the timed region; asynchronous: the transfer of this chunk
        overlaps the next chunk's kernels (sharding.PaddedGather), finish() waits for what is still in flight"""
        from ethzasl_brisk_amd import sharding
        if self.pg is None or self.pg.kpad != self.kpad:
            if self.pg is not None:
                self.pg.finish()
            self.pg = sharding.PaddedGather(self.counts, self.kps, self.desc, self.strings, self.kpad, dst=0, frames_max=self.frames_max)
        self.pg.start()

    def finish(self):
        if self.pg is not None:
            self.last = self.pg.finish()

    def verify_counts(self):
        """outside the timed regions, rank 0: the gathered counts are the frames' FULL counts - one above the slab size would
        mean rows that were not sent (the slab size is fixed before the timed region from the warm-up batches)"""
        if self.rank == 0 and self.last is not None:
            m = int(self.last[0].max().item())
            if m > self.kpad:
                raise RuntimeError("result gather: a frame has %d described keypoints, the slabs carry %d rows" % (m, self.kpad))

    def close(self):
        pass

    def check_kpad(self, ctl):
        """outside the timed region: the slab size must cover every frame of the batch (rounded up to 128)"""
        self.finish()
        m = self.counts.max().to(self.torch.int64).reshape(1).cpu()
        if self.world > 1:  # every rank must cut slabs of the same shape
            import torch.distributed as dist
            dist.all_reduce(m, op=dist.ReduceOp.MAX, group=ctl)
        self.kpad = min(self.cap, (int(m.item()) + 127) // 128 * 128)


class CapiGather(ResultGather):
    """The same exchange through the engine's C ABI (include/brisk_hip.h: brisk_hip_comm_*): RCCL directly, no
    torch.distributed on the data path - torch only provides the destination buffers on rank 0 and, through the gloo
    control group, carries the 128-byte unique id to the other ranks."""

    def __init__(self, ctx, batch, strings, dev, rank, world, frames_max, ctl, stream):
        import ctypes as C
        super().__init__(ctx, batch, strings, dev, rank, world, frames_max)
        torch = self.torch
        self.ctx, self.stream, self.C = ctx, stream, C
        L = ctx._L
        uid = (C.c_uint8 * 128)()
        if rank == 0:
            ctx.check(L.brisk_hip_comm_unique_id(uid))
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor(list(uid), dtype=torch.uint8)
            dist.broadcast(t, 0, group=ctl)
            uid = (C.c_uint8 * 128)(*[int(v) for v in t])
        self.comm = C.c_void_p()
        ctx.check(L.brisk_hip_comm_create(ctx._h, rank, world, uid, C.byref(self.comm)))
        self.slots = [None, None]
        self.i = 0

    def _dst(self):
        torch = self.torch
        j = self.i & 1
        sl = self.slots[j]
        if self.rank != 0:
            return None
        if sl is None or sl["kpad"] != self.kpad:
            b = self.frames_max
            sl = self.slots[j] = {"kpad": self.kpad,
                                  "ac": torch.zeros((self.world, b), device=self.dev, dtype=torch.int32),
                                  "gk": torch.zeros((self.world, b, self.kpad, 7), device=self.dev, dtype=torch.float32),
                                  "gd": torch.zeros((self.world, b, self.kpad, self.strings), device=self.dev, dtype=torch.uint8)}
        return sl

    def run(self):
        C = self.C
        sl = self._dst()
        p = (lambda t: C.c_void_p(t.data_ptr())) if sl else (lambda t: None)
        self.ctx.check(self.ctx._L.brisk_hip_comm_gather_results(
            self.ctx._h, self.comm, 0, self.frames_max, self.kpad, self.strings,
            p(sl["ac"]) if sl else None, p(sl["gk"]) if sl else None, p(sl["gd"]) if sl else None, C.c_void_p(self.stream)))
        self.cur = sl
        self.i += 1

    def finish(self):
        if self.i:
            self.ctx.check(self.ctx._L.brisk_hip_comm_wait(self.comm, None))
            sl = getattr(self, "cur", None)
            self.last = (sl["ac"], sl["gk"], sl["gd"]) if sl else None

    def close(self):
        if getattr(self, "comm", None) is not None and self.comm:
            self.ctx._L.brisk_hip_comm_destroy(self.comm)
            self.comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 (interpreter shutdown)
            pass


if __name__ == "__main__":
    main()
