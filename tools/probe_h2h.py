#!/usr/bin/env python3
"""Host-to-host batch rate (brisk_hip_detect_describe_batch_host_results) against the H2D-only rate, for several frames per
call; run under rocprofv3 --kernel-trace --memory-copy-trace --stats for the durations of k_export_* and of the copies.
usage: probe_h2h.py [frames_per_call ...]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ethzasl_brisk_amd as B, synth
W, H = 1920, 1080
ns = [int(a) for a in sys.argv[1:]] or [256]
host = np.stack([synth.frame_1080p(100 + i) for i in range(8)])
ctx = B.Context(0)
ext = B.BriskDescriptorExtractor(context=ctx)
for n in ns:
    src = torch.from_numpy(np.ascontiguousarray(host[np.arange(n) % len(host)])).pin_memory()
    def h2d():
        ctx.detect_describe_batch_host(ext, src.data_ptr(), n, W, H, W * H, W, 80, 4)
    h2d(); torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < 1.5:
        h2d(); reps += 1
    torch.cuda.synchronize()
    fps0 = reps * n / (time.perf_counter() - t0)
    rows = n * 1400
    dsts = [B.HostResults(n, rows, 48, pinned=True) for _ in range(2)]
    tk = [0, 0]
    def issue(i):
        if tk[i & 1]:
            ctx.batch_download_wait(tk[i & 1])
        tk[i & 1] = ctx.detect_describe_batch_host_results(ext, src.data_ptr(), n, W, H, W * H, W, 80, 4, dsts[i & 1])
    for i in range(3):
        issue(i)
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < 1.5:
        issue(reps + 3); reps += 1
    for i in (0, 1):
        ctx.batch_download_wait(tk[i])
    fps1 = reps * n / (time.perf_counter() - t0)
    # the transfer alone: results of a resident batch, egress of one batch at a time
    ctx.detect_describe_batch_host(ext, src.data_ptr(), n, W, H, W * H, W, 80, 4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.batch_download_wait(ctx.batch_download_all(dsts[0]))
    dl = (time.perf_counter() - t0) / 10
    mb = int(dsts[0].offsets[n]) * 76 / 1e6
    print(json.dumps({"frames_per_call": n, "h2d_only_fps": round(fps0, 1), "host_to_host_fps": round(fps1, 1), "ratio": round(fps1 / fps0, 4),
                      "download_all_alone_ms": round(dl * 1e3, 3), "MB": round(mb, 2), "alone_GBps": round(mb / 1e3 / dl, 2)}), flush=True)
