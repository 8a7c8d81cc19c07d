#!/usr/bin/env python3
"""Per-phase wave time of k_detect (needs the DT_TIMING build variant:
python -c "from ethzasl_brisk_amd import build; build.build_variant('libbrisk_dttiming', ['DT_TIMING'])";
BRISK_HIP_LIB=ethzasl_brisk_amd/libbrisk_dttiming.so python3 tools/detect_phases.py [frames] [threshold]).
ISA instruction counts of the same phases: tools/detect_phase_isa.sh."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import ethzasl_brisk_amd as B
import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 80
ctx = B.Context(0)
nd = 16
frames = np.stack([synth.frame_1080p(i) for i in range(nd)])
d = torch.from_numpy(frames).cuda()
batch = d[torch.arange(n, device="cuda") % nd].contiguous()
_, h, w = batch.shape
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    ctx.detect_batch(batch.data_ptr(), n, w, h, w * h, w, thr, 4, st)
torch.cuda.synchronize()
ctx.batch_status(n)
ctx.profile_enable(True)
ctx.detect_batch(batch.data_ptr(), n, w, h, w * h, w, thr, 4, st)
torch.cuda.synchronize()
ms, _ = ctx.profile_read()
e = np.zeros(8, np.int64)
for f in range(n):
    e += np.array([int(v) for v in ctx.debug_counters_raw(f)[48:56]], np.int64)
names = ["tile decode + staging (global loads -> LDS)", "barrier (staging of the other waves)", "window: LDS dword reads + centre byte pairs",
         "W / E byte pairs + packed pre-gate, 16 pixels", "compaction of the survivors", "barrier in front of phase B",
         "phase B (exact contrast + segment test)"]
waves = max(int(e[7]), 1)
kd = ms.get("k_detect", 0)
tot = float(sum(e[:7]))
print("%d frames, threshold %d: k_detect %.3f ms per launch; %d waves sampled (one 64 x 64 tile in 16, 4 waves each)" % (n, thr, kd, waves))
for i, nm in enumerate(names):
    print("  %-48s %8.0f ticks per wave  %5.1f %%" % (nm, e[i] / waves, 100.0 * e[i] / tot))
print("  total %.0f s_memtime ticks per wave (launch %.3f ms)" % (tot / waves, kd))
