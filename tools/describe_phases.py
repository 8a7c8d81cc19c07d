#!/usr/bin/env python3
"""Per-phase wave time of k_describe's run loop (needs the DS_TIMING build variant:
python -c "from ethzasl_brisk_amd import build; build.build_variant('libbrisk_dstiming', ['DS_TIMING'])";
BRISK_HIP_LIB=ethzasl_brisk_amd/libbrisk_dstiming.so python3 tools/describe_phases.py [frames] [threshold])."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import ethzasl_brisk_amd as B
import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 80
ctx = B.Context(0)
ext = B.BriskDescriptorExtractor(context=ctx)
nd = 16
frames = np.stack([synth.frame_1080p(i) for i in range(nd)])
d = torch.from_numpy(frames).cuda()
batch = d[torch.arange(n, device="cuda") % nd].contiguous()
_, h, w = batch.shape
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    ctx.detect_describe_batch(ext, batch.data_ptr(), n, w, h, w * h, w, thr, 4, st)
torch.cuda.synchronize()
ctx.batch_status(n)
base = np.array([int(v) for v in ctx.debug_counters_raw(0)[48:56]], np.int64)
ctx.profile_enable(True)
reps = 3
for _ in range(reps):
    ctx.detect_describe_batch(ext, batch.data_ptr(), n, w, h, w * h, w, thr, 4, st)
torch.cuda.synchronize()
ms, _ = ctx.profile_read()
e = np.array([int(v) for v in ctx.debug_counters_raw(0)[48:56]], np.int64)
if np.all(e >= base):
    e = e - base if e[5] > base[5] else e
names = ["top of the run (frame, records)", "orientation pass (3 rounds of gathers)", "long pairs + atan2", "rotated pass",
         "ticket decode, bit tests, stores, next first round"]
runs = max(int(e[5]), 1)
kd = ms.get("k_describe", 0)
print("%d frames, threshold %d: k_describe %.3f ms per launch; %d runs summed" % (n, thr, kd, runs))
tot = 0.0
for i, nm in enumerate(names):
    t = e[i] * 16 * 0.01 / runs
    tot += t
    print("  %-52s %7.2f us per run (s_memtime at 100 MHz)" % (nm, t))
print("  total %.2f us per run" % tot)
