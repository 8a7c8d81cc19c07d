#!/usr/bin/env python3
"""k_detect on 64 4K frames with one of its three global side effects removed (BRISK_HIP_LIB = a -DDT_EXP_* build)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ethzasl_brisk_amd as B
import synth
stream = torch.cuda.current_stream().cuda_stream
base = np.stack([synth.frame_4k(2 + i) for i in range(4)])
for (w, h, nb) in [(3840, 2160, 64), (3584, 2160, 68), (1920, 1080, 256)]:
    ctx = B.Context(0)
    crop = np.ascontiguousarray(base[:, :h, :w])
    d = torch.from_numpy(crop).cuda()
    batch = d[torch.arange(nb, device="cuda") % 4].contiguous()
    for rep in range(2):
        ctx.detect_batch(batch.data_ptr(), nb, w, h, w * h, w, 80, 4, stream)
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    for rep in range(5):
        ctx.detect_batch(batch.data_ptr(), nb, w, h, w * h, w, 80, 4, stream)
    torch.cuda.synchronize()
    ms, _ = ctx.profile_read()
    print(os.environ.get("BRISK_HIP_LIB", "default").split("/")[-1], w, h, nb, "k_detect %.3f" % ms["k_detect"])
    ctx.close(); del batch, d
