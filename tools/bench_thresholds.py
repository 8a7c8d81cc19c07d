#!/usr/bin/env python3
"""Stage times of a 64-frame 1080p batch at several AGAST thresholds (dense-keypoint regimes: capacity and scaling
behaviour of the sparse kernels)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import ethzasl_brisk_amd as B
import synth

ctx = B.Context(0, max_candidates=262144, max_keypoints=65536)
ctx.debug_set_flags(int(os.environ.get("BRISK_DEBUG_FLAGS", "0"), 0))   # (k_describe launch knobs: tools/sweep_describe.py)
ext = B.BriskDescriptorExtractor(context=ctx)
frames = np.stack([synth.frame_1080p(i) for i in range(4)])
d = torch.from_numpy(frames).cuda()
batch = d[torch.arange(64, device="cuda") % 4].contiguous()
n, h, w = batch.shape
st = torch.cuda.current_stream().cuda_stream
for thr in [int(a) for a in sys.argv[1:]] or (80, 50, 30):
    for _ in range(2):
        ctx.detect_describe_batch(ext, batch.data_ptr(), n, w, h, w * h, w, thr, 4, st)
    torch.cuda.synchronize()
    flags = ctx.batch_status(n)
    ctx.profile_enable(True)
    for _ in range(3):
        ctx.detect_describe_batch(ext, batch.data_ptr(), n, w, h, w * h, w, thr, 4, st)
    torch.cuda.synchronize()
    ms, _ = ctx.profile_read()
    nd, nk = len(ctx.batch_download(0, False)[0]), len(ctx.batch_download(0, True)[0])
    print("thr %d" % thr, ctx.debug_counters(0))
    print("thr %d flags %d detected %d described %d total %.2f ms / 64 frames %s"
          % (thr, flags, nd, nk, sum(ms.values()), {k: round(v, 2) for k, v in ms.items()}))
