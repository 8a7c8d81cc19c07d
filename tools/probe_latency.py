#!/usr/bin/env python3
"""One frame per call through the host-buffer entry points (BASELINE config 1): wall time per call, for a trace."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ethzasl_brisk_amd as B
import synth

which = sys.argv[1] if len(sys.argv) > 1 else "vga"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
img = synth.frame_vga(1) if which == "vga" else synth.frame_1080p(0)
thr = 70 if which == "vga" else 80
det, ext = B.BriskFeatureDetector(thr, 4), B.BriskDescriptorExtractor()
for _ in range(10):
    k = det.detect(img)
    k2, d = ext.compute(img, k)
td = tc = 0.0
for _ in range(n):
    t0 = time.perf_counter()
    k = det.detect(img)
    t1 = time.perf_counter()
    k2, d = ext.compute(img, k)
    t2 = time.perf_counter()
    td += t1 - t0
    tc += t2 - t1
print("%s: detect %.1f us, compute %.1f us, %d / %d keypoints -> %.0f frames/s" % (which, td / n * 1e6, tc / n * 1e6, len(k), len(k2), n / (td + tc)))
