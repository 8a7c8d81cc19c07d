#!/usr/bin/env python3
"""thresholds below 20: which frames stay on the fast path (no detection stored a score <= 2) and what a call costs"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ethzasl_brisk_amd as B
import oracle_lib as O
import synth
from test_gpu_parity import same_kps
ctx = B.Context(0, max_candidates=1 << 21, max_keypoints=1 << 19)
for name, img in (("1080p", synth.frame_1080p(0)), ("vga", synth.frame_vga(2))):
    for thr in (19, 15, 10, 8):
        det = B.BriskFeatureDetector(thr, 4, context=ctx)
        k = det.detect(img, capacity=1 << 19)
        t0 = time.perf_counter()
        k = det.detect(img, capacity=1 << 19)
        dt = time.perf_counter() - t0
        c = ctx.debug_counters(0)
        ko = O.detect(img, thr, 4)
        print("%s threshold %d: %d candidates, %d keypoints, low-score flag %d, %.1f ms per call, equal to the oracle: %s"
              % (name, thr, c["candidates"], len(k), c["experiment"][3], dt * 1e3, same_kps(k, ko)), flush=True)
