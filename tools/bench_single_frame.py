#!/usr/bin/env python3
"""Latency of the host-buffer calls the drop-in classes use (one frame per call: H2D copy, kernels, D2H copy)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import ethzasl_brisk_amd as B
import synth

for name, img, octaves in (("1080p", synth.frame_1080p(0), 4), ("vga", synth.frame_vga(1), 4)):
    det, ext = B.BriskFeatureDetector(80, octaves), B.BriskDescriptorExtractor()
    for _ in range(5):
        k = det.detect(img)
        k2, d = ext.compute(img, k)
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        k = det.detect(img)
    t1 = time.perf_counter()
    for _ in range(n):
        k2, d = ext.compute(img, k)
    t2 = time.perf_counter()
    for _ in range(n):   # the caller states that compute() sees the buffer detect() just saw (no second upload)
        k = det.detect(img)
        k2, d = ext.compute(img, k, same_image=True)
    t3 = time.perf_counter()
    print("%s: detect %.3f ms  describe %.3f ms  (%d / %d keypoints)  -> %.0f frames/s one frame per call; detect + compute(same_image) %.3f ms -> %.0f frames/s"
          % (name, (t1 - t0) / n * 1e3, (t2 - t1) / n * 1e3, len(k), len(k2), n / (t2 - t0), (t3 - t2) / n * 1e3, n / (t3 - t2)))
