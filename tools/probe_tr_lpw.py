#!/usr/bin/env python3
"""config 4 (64 x 4K, 6 octaves): stage intervals for the BRISK_TR_LPW knob given in the environment + tie counts per layer"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ethzasl_brisk_amd as B
import synth
stream = torch.cuda.current_stream().cuda_stream
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
base = np.stack([synth.frame_4k(2 + i) for i in range(4)])
ctx = B.Context(0)
ext = B.BriskDescriptorExtractor(context=ctx)
d = torch.from_numpy(base).cuda()
batch = d[torch.arange(nb, device="cuda") % 4].contiguous()
w, h = 3840, 2160
for rep in range(2):
    ctx.detect_describe_batch(ext, batch.data_ptr(), nb, w, h, w * h, w, 80, 6, stream)
torch.cuda.synchronize()
import time
ctx.profile_enable(True)
t0 = time.perf_counter()
for rep in range(10):
    ctx.detect_describe_batch(ext, batch.data_ptr(), nb, w, h, w * h, w, 80, 6, stream)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
ms, _ = ctx.profile_read()
print("LPW", os.environ.get("BRISK_TR_LPW", "auto"), "nb", nb, "ms/call %.3f" % (dt * 1e3), {k: round(v, 3) for k, v in ms.items() if v > 0.005})
print("ties per layer, frame 0:", ctx.debug_counters(0)["ties"], "cands", ctx.debug_counters(0)["candidates"])
