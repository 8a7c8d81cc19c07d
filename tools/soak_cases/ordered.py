#!/usr/bin/env python3
"""One-off soak (GPU box) of the ordered path: thresholds 1..19, the multi-layer suppressScaleNonmaxima=false branch and
ComputeScale on random images / keypoint lists, bit-exact against the oracle (which also says where the reference has no
defined result: both sides must agree on that too)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
import synth
import ethzasl_brisk_amd as B
from test_emul_parity import banded


def same(a, b):
    return len(a) == len(b) and a.tobytes() == b.tobytes()


ctx = B.Context(0, max_candidates=400000, max_keypoints=200000)
rng = np.random.default_rng(42)
bad = n_cases = n_undefined = 0
for it in range(24):
    w, h = int(rng.integers(120, 360)), int(rng.integers(100, 260))
    img = synth.gen(w, h, 7000 + it, int(rng.integers(8, 40)))
    if it % 4 == 3:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    thr = int(rng.integers(1, 20))
    octaves = int(rng.integers(0, 4))
    ko = O.detect(img, thr, octaves)
    kg = B.BriskFeatureDetector(thr, octaves, context=ctx).detect(img, capacity=200000)
    ok = same(kg, ko)
    n_cases += 1
    bad += not ok
    print("thr %2d octaves %d %dx%d: %5d keypoints %s" % (thr, octaves, w, h, len(ko), "ok" if ok else "MISMATCH"))
for it in range(12):
    img = banded(100 + it, h=int(rng.integers(160, 260)), w=int(rng.integers(200, 360)), band=int(rng.integers(24, 60)), cell=int(rng.integers(2, 5)))
    thr, octaves = int(rng.integers(8, 70)), int(rng.integers(1, 4))
    ko = O.detect(img, thr, octaves, suppress_scale_nonmaxima=False)
    try:
        kg = B.BriskFeatureDetector(thr, octaves, suppressScaleNonmaxima=False, context=ctx).detect(img, capacity=200000)
    except B.BriskHipError as e:
        kg = None if e.code == 7 else "error %d" % e.code
    ok = (ko is None and kg is None) or (ko is not None and kg is not None and not isinstance(kg, str) and same(kg, ko))
    n_cases += 1
    n_undefined += ko is None
    bad += not ok
    print("no-scale-nms thr %2d octaves %d: %s %s" % (thr, octaves, "undefined" if ko is None else "%d keypoints" % len(ko), "ok" if ok else "MISMATCH"))
for it in range(16):
    w, h = int(rng.integers(160, 360)), int(rng.integers(140, 260))
    img = synth.gen(w, h, 9000 + it, 25)
    thr, octaves, suppress = int(rng.integers(5, 80)), int(rng.integers(0, 4)), bool(rng.integers(0, 4))
    n = int(rng.integers(1, 300))
    k = np.zeros(n, B.KEYPOINT)
    k["x"] = rng.uniform(0, w, n).astype(np.float32)
    margin = 0 if it % 4 == 0 else 75          # every fourth case may hit the undefined band at the bottom
    k["y"] = rng.uniform(0, h - margin, n).astype(np.float32)
    k["class_id"] = rng.integers(-1, 50, n)
    k["size"] = 10
    ko = O.compute_scale(img, k, thr, octaves, suppress)
    try:
        kg = B.BriskFeatureDetector(thr, octaves, suppress, context=ctx).ComputeScale(img, k)
    except B.BriskHipError as e:
        kg = None if e.code == 7 else "error %d" % e.code
    ok = (ko is None and kg is None) or (ko is not None and kg is not None and not isinstance(kg, str) and same(kg, ko))
    n_cases += 1
    n_undefined += ko is None
    bad += not ok
    print("compute_scale thr %2d octaves %d suppress %d, %3d provided: %s %s"
          % (thr, octaves, suppress, n, "undefined" if ko is None else "%d keypoints" % len(ko), "ok" if ok else "MISMATCH"))
print("ordered: %d cases (%d undefined in the reference), %d mismatches" % (n_cases, n_undefined, bad))
sys.exit(1 if bad else 0)
