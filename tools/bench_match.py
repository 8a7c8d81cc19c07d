#!/usr/bin/env python3
"""Times the device-resident Hamming k-NN matcher (brisk_hip_match_knn_device) on synthetic 48-byte descriptors.
Usage: python tools/bench_match.py [nq nt k]...   (default: a frame pair, a 20k x 20k set and a 100k x 100k set)"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ethzasl_brisk_amd as B


def run(nq, nt, k, reps=5):
    ctx = B.default_context(0)
    g = torch.Generator(device="cuda").manual_seed(1)
    q = torch.randint(0, 256, (nq, 48), dtype=torch.uint8, device="cuda", generator=g)
    t = torch.randint(0, 256, (nt, 48), dtype=torch.uint8, device="cuda", generator=g)
    out = torch.zeros((nq, k, 4), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(nq, dtype=torch.int32, device="cuda")
    stream = torch.cuda.Stream()      # (a NULL stream argument would select the context's own stream)
    torch.cuda.synchronize()
    torch.cuda.set_stream(stream)
    st = stream.cuda_stream
    call = lambda: ctx.check(ctx._L.brisk_hip_match_knn_device(ctx._h, q.data_ptr(), nq, 48, t.data_ptr(), nt, 48, 48, k,
                                                               out.data_ptr(), cnt.data_ptr(), st))
    call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    pairs = nq * nt
    print("knn  nq=%d nt=%d k=%d: %.3f ms  (%.1f G pairs/s, %.1f GB/s of descriptor reads if every pair re-read both rows)"
          % (nq, nt, k, ms, pairs / ms / 1e6, pairs * 96 / ms / 1e6))


if __name__ == "__main__":
    args = [int(a) for a in sys.argv[1:]]
    cases = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(1000, 1000, 2), (20000, 20000, 2), (100000, 100000, 2)]
    for c in cases:
        run(*c)
