#!/usr/bin/env python3
"""where one host-buffer compute() call of a 1080p frame spends its time: with / without the second upload"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ethzasl_brisk_amd as B
import synth
img = synth.frame_1080p(0)
det, ext = B.BriskFeatureDetector(80, 4), B.BriskDescriptorExtractor()
k = det.detect(img)
for same in (False, True, False):
    for _ in range(5):
        ext.compute(img, k, same_image=same)
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        ext.compute(img, k, same_image=same)
    print("compute(same_image=%s): %.3f ms" % (same, (time.perf_counter() - t0) / n * 1e3))
