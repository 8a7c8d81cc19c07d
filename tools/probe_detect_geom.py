#!/usr/bin/env python3
"""k_detect / k_integral_final / k_tie_resolve per megapixel for different frame geometries with the same total pixel count
(why does a 4K batch run 1.5 x slower per pixel than a 1080p batch?).  Stage intervals of the engine's profiler."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ethzasl_brisk_amd as B
import synth
stream = torch.cuda.current_stream().cuda_stream
base = np.stack([synth.frame_4k(2 + i) for i in range(4)])
out = {}
for (w, h, nb, octaves) in [(1920, 1080, 256, 4), (3840, 2160, 64, 4), (3840, 2160, 64, 6), (3840, 1080, 128, 4), (1920, 2160, 128, 4), (2048, 1080, 240, 4), (3584, 2160, 68, 4)]:
    for thr in (80, 250):
        ctx = B.Context(0)
        ext = B.BriskDescriptorExtractor(context=ctx)
        crop = np.ascontiguousarray(base[:, :h, :w])
        d = torch.from_numpy(crop).cuda()
        batch = d[torch.arange(nb, device="cuda") % 4].contiguous()
        for rep in range(2):
            ctx.detect_describe_batch(ext, batch.data_ptr(), nb, w, h, w * h, w, thr, octaves, stream)
        torch.cuda.synchronize()
        ctx.profile_enable(True)
        for rep in range(5):
            ctx.detect_describe_batch(ext, batch.data_ptr(), nb, w, h, w * h, w, thr, octaves, stream)
        torch.cuda.synchronize()
        ms, _ = ctx.profile_read()
        cand = ctx.debug_counters(0)["candidates"]
        out["%dx%d x%d oct%d thr%d" % (w, h, nb, octaves, thr)] = {"cand0": cand, **{k: round(v, 3) for k, v in ms.items() if v > 0.005}}
        ctx.close(); del batch, d
        torch.cuda.empty_cache()
for k, v in out.items():
    print(k, json.dumps(v))
