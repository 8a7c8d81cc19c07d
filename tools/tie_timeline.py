#!/usr/bin/env python3
"""Per-layer wall-clock stamps of k_tie_resolve for ONE frame (needs the TR_TIMELINE build variant:
python -c "from ethzasl_brisk_amd import build; build.build_variant('libbrisk_trtl', ['TR_TIMELINE'])";
BRISK_HIP_LIB=ethzasl_brisk_amd/libbrisk_trtl.so python3 tools/tie_timeline.py [vga|1080p|4k])."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ethzasl_brisk_amd as B
import synth

which = sys.argv[1] if len(sys.argv) > 1 else "vga"
img = {"vga": lambda: synth.frame_vga(1), "1080p": lambda: synth.frame_1080p(0), "4k": lambda: synth.frame_4k(2)}[which]()
thr = 70 if which == "vga" else 80
det = B.BriskFeatureDetector(thr, 6 if which == "4k" else 4)
for _ in range(5):
    k = det.detect(img)
ctx = det._ctx
ctx.profile_enable(True)
for _ in range(3):
    k = det.detect(img)
ms, _ = ctx.profile_read()
raw = ctx.debug_counters_raw(0)
c = ctx.debug_counters(0)
tl = raw[48:48 + 16 * 8].reshape(16, 8)
nl = len(c["ties"])
t0 = min(int(tl[l, 0]) for l in range(nl))
print("%s: %d keypoints, k_tie_resolve %.1f us, ties %s" % (which, len(k), ms.get("k_tie_resolve", 0) * 1e3, c["ties"]))
print("layer   start  setup-done  first-decision  wave0-done  writer-done  barrier   (us after the first layer's start)")
for l in range(nl):
    print("%5d " % l + " ".join("%10.2f" % (((int(tl[l, i]) - t0) & 0xFFFFFFFF) * 0.01) if tl[l, i] else "         -" for i in range(6)))
