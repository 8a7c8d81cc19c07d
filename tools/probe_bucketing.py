#!/usr/bin/env python3
"""KeyPointBucketing as a post-filter of a 4K batch (64 frames, 6 octaves): time of the filter kernels"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ethzasl_brisk_amd as B
import synth
stream = torch.cuda.current_stream().cuda_stream
base = np.stack([synth.frame_4k(2 + i) for i in range(4)])
d = torch.from_numpy(base).cuda()
for nb in (64, 1):
    batch = d[torch.arange(nb, device="cuda") % 4].contiguous()
    ctx = B.Context(0)
    ctx.set_bucketing(8, 6, 3000)
    for rep in range(2):
        ctx.detect_batch(batch.data_ptr(), nb, 3840, 2160, 3840 * 2160, 3840, 80, 6, stream)
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    for rep in range(5):
        ctx.detect_batch(batch.data_ptr(), nb, 3840, 2160, 3840 * 2160, 3840, 80, 6, stream)
    torch.cuda.synchronize()
    ms, _ = ctx.profile_read()
    print("frames", nb, "kept", ctx.debug_counters(0)["keypoints"], "k_postfilter %.3f ms" % ms["k_postfilter"])
    ctx.close()
