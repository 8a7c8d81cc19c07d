#!/usr/bin/env python3
"""Per-phase time of k_tie_resolve's decision loop (needs the TR_TIMING build variant:
python -c "from ethzasl_brisk_amd import build; build.build_variant('libbrisk_trtiming', ['TR_TIMING'])";
BRISK_HIP_LIB=ethzasl_brisk_amd/libbrisk_trtiming.so python3 tools/tie_phases.py [threshold] [frames])."""
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
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
which = sys.argv[3] if len(sys.argv) > 3 else "1080p"   # 1080p | 4k (6 octaves) | vga
octaves = 6 if which == "4k" else 4
gen = {"1080p": synth.frame_1080p, "4k": lambda i: synth.frame_4k(2 + i), "vga": lambda i: synth.frame_vga(1 + i)}[which]
ctx = B.Context(0, max_candidates=262144, max_keypoints=65536)
frames = np.stack([gen(i) for i in range(min(n, 4))])
d = torch.from_numpy(frames).cuda()
batch = d[torch.arange(n, device="cuda") % len(frames)].contiguous()
_, h, w = batch.shape
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    ctx.detect_batch(batch.data_ptr(), n, w, h, w * h, w, thr, octaves, st)
torch.cuda.synchronize()
ctx.batch_status(n)
ctx.profile_enable(True)
for _ in range(3):
    ctx.detect_batch(batch.data_ptr(), n, w, h, w * h, w, thr, octaves, st)
torch.cuda.synchronize()
ms, _ = ctx.profile_read()
c = ctx.debug_counters(0)
e = [int(v) for v in ctx.debug_counters_raw(0)[48:56]]   # BriskFrameCounters::tphase (TR_TIMING build)
names = ["resolve -> decision", "layer-below need + poll", "next prefetch issue", "window -> LDS + search",
         "spin on earlier ties", "static replay", None, "wait for the layer below + window read"]
pair = n < 32 and os.environ.get("BRISK_TR_PAIR", "1") != "0"
if pair:  # k_tie_resolve_pair: an iteration decides two ties; [1] counts passes through the resolve step, [4] is the polling loop around them
    names = ["resolve -> decision (both halves' passes)", None, "next prefetch issue (incl. layer-below poll)", "windows -> LDS + searches",
             "polling for earlier ties", "static replay", None, "wait for the layer below + window read"]
print("thr %d, %d %s frames%s: k_tie_resolve %.3f ms; frame 0 ties %s" % (thr, n, which, " (two ties per wave)" if pair else "", ms.get("k_tie_resolve", 0), c["ties"]))
if e[6]:
    for i, nm in enumerate(names):
        if nm:
            print("  %-52s %7.2f us per tie iteration" % (nm, e[i] * 0.01 / e[6]))
    print("  total %.2f us per iteration, %d iterations" % ((sum(e) - e[6] - (e[1] if pair else 0)) * 0.01 / e[6], e[6]))
    if pair:
        print("  passes through the resolve step per iteration: %.2f" % (e[1] / e[6]))
