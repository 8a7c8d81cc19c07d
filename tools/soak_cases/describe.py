#!/usr/bin/env python3
"""One-off soak (GPU box) of the paths tools/soak.py frames does not reach: descriptor-only calls with random keypoints (all
sizes up to the largest scale, packed against all four borders, provided angles, both patterns, invariance flags) on
padded and unpadded widths; dense / tie-heavy detection at thresholds 20..30; a 4K six-octave frame; the host-buffer
calls against the batch path.  Everything bit-exact against the oracle."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
import synth
import ethzasl_brisk_amd as B


def same(a, b):
    return len(a) == len(b) and a.tobytes() == b.tobytes()


bad = 0
ctx = B.Context(0, max_candidates=400000, max_keypoints=200000)

# ---- descriptor-only
for (w, h) in ((426, 320), (640, 480), (333, 201), (1281, 723)):
    img = synth.gen(w, h, w + h, 60)
    for version, rot, sc in ((2, True, True), (2, False, True), (2, True, False), (1, True, True)):
        ext = B.BriskDescriptorExtractor(rot, sc, version=version, context=ctx)
        oext = O.Extractor(rot, sc, version=version)
        _, size_list, _ = ext.tables()
        rng = np.random.default_rng(w * 7 + version)
        n = 3000
        k = np.zeros(n, B.KEYPOINT)
        k["size"] = np.exp(rng.uniform(np.log(6.0), np.log(260.0), n)).astype(np.float32)
        k["x"] = rng.uniform(-5, w + 5, n).astype(np.float32)
        k["y"] = rng.uniform(-5, h + 5, n).astype(np.float32)
        # a third of them exactly around the border limits of their scale
        sidx = np.array([oext.scale_index(s) if hasattr(oext, "scale_index") else 0 for s in k["size"]])
        for i in range(0, n, 3):
            b = float(size_list[sidx[i]]) if hasattr(oext, "scale_index") else 20.0
            side = rng.integers(0, 4)
            jitter = np.float32(rng.uniform(-1.5, 1.5))
            if side == 0: k["x"][i] = np.float32(b) + jitter
            elif side == 1: k["x"][i] = np.float32(w - b) + jitter
            elif side == 2: k["y"][i] = np.float32(b) + jitter
            else: k["y"][i] = np.float32(h - b) + jitter
        k["angle"] = np.where(rng.uniform(0, 1, n) < 0.5, -1.0, rng.uniform(0, 360, n)).astype(np.float32)
        ko, do = oext.compute(img, k)
        kg, dg = ext.compute(img, k)
        ok = same(kg, ko) and np.array_equal(dg, do)
        print("describe %dx%d v%d rot %d scale %d: %d of %d kept  %s" % (w, h, version, rot, sc, len(ko), n, "ok" if ok else "MISMATCH"))
        bad += not ok

# ---- dense / tie-heavy detection
for (w, h, thr, octaves, seed) in ((640, 480, 20, 4, 1), (800, 600, 25, 3, 2), (426, 320, 30, 4, 3), (1281, 723, 30, 2, 4), (640, 480, 22, 0, 5)):
    img = synth.gen(w, h, seed, 150)
    det = B.BriskFeatureDetector(thr, octaves, context=ctx)
    ko = O.detect(img, thr, octaves)
    kg = det.detect(img, capacity=200000)
    ok = same(kg, ko)
    print("detect %dx%d thr %d octaves %d: %d keypoints  %s" % (w, h, thr, octaves, len(ko), "ok" if ok else "MISMATCH"))
    bad += not ok
rng = np.random.default_rng(9)
for name, img in (("noise", rng.integers(0, 256, (300, 400), dtype=np.uint8)),
                  ("blocks", (np.kron(rng.integers(0, 2, (40, 50)), np.ones((8, 8))) * 255).astype(np.uint8)),
                  ("ramps", (np.add.outer(np.arange(240), np.arange(320)) % 256).astype(np.uint8))):
    for thr in (20, 40, 90):
        ko = O.detect(img, thr, 3)
        kg = B.BriskFeatureDetector(thr, 3, context=ctx).detect(img, capacity=200000)
        ok = same(kg, ko)
        print("detect %s thr %d: %d keypoints  %s" % (name, thr, len(ko), "ok" if ok else "MISMATCH"))
        bad += not ok

# ---- 4K, six octaves
img = synth.frame_4k(5)
ko = O.detect(img, 80, 6)
kg = B.BriskFeatureDetector(80, 6, context=ctx).detect(img, capacity=200000)
ext, oext = B.BriskDescriptorExtractor(context=ctx), O.Extractor()
k2o, do = oext.compute(img, ko)
k2g, dg = ext.compute(img, kg)
ok = same(kg, ko) and same(k2g, k2o) and np.array_equal(dg, do)
print("4K six octaves: %d / %d keypoints  %s" % (len(ko), len(k2o), "ok" if ok else "MISMATCH"))
bad += not ok
print("describe: %d mismatching cases" % bad)
sys.exit(1 if bad else 0)
