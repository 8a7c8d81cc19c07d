#!/usr/bin/env python3
"""Per-phase wave time of k_score_blocks (needs the SB_TIMING build variant:
python -c "from ethzasl_brisk_amd import build; build.build_variant('libbrisk_sbtiming', ['SB_TIMING'])";
BRISK_HIP_LIB=ethzasl_brisk_amd/libbrisk_sbtiming.so python3 tools/score_block_phases.py [threshold] [frames])."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import ethzasl_brisk_amd as B
import synth

thr = int(sys.argv[1]) if len(sys.argv) > 1 else 80
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ctx = B.Context(0, max_candidates=262144, max_keypoints=65536)
frames = np.stack([synth.frame_1080p(i) for i in range(4)])
d = torch.from_numpy(frames).cuda()
batch = d[torch.arange(n, device="cuda") % 4].contiguous()
_, h, w = batch.shape
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    ctx.detect_batch(batch.data_ptr(), n, w, h, w * h, w, thr, 4, st)
torch.cuda.synchronize()
ctx.batch_status(n)
e = [int(v) for v in ctx.debug_counters_raw(0)[48:56]]   # BriskFrameCounters::sphase (SB_TIMING build)
names = ["headers, addresses, patch + map loads (2 round trips)", "patches -> LDS", "evaluation of the round's 3 candidates + stores"]
print("thr %d, %d frames: frame 0, %d wave rounds of 3 candidates" % (thr, n, e[3]))
for i, nm in enumerate(names):
    print("  %-58s %7.2f us per round" % (nm, e[i] * 0.01 / max(e[3], 1)))
