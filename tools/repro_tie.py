"""GPU box: the tie kernel on an image that is all ties (3-px blocks of four grey levels), capacities and octave counts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ethzasl_brisk_amd as B
img = np.kron(np.random.default_rng(1487 * 8).integers(0, 4, (739 // 3 + 1, 525 // 3 + 1)) * 80 + 7, np.ones((3, 3)))[:739, :525].astype(np.uint8)
for cands in (262144, 1048576):
    ctx = B.Context(0, max_candidates=cands, max_keypoints=cands // 4)
    for octaves in (0, 1, 2):
        t = time.time()
        try:
            k = B.BriskFeatureDetector(21, octaves, context=ctx).detect(img, capacity=262144)
            print(cands, octaves, "ok", len(k), round(time.time() - t, 3), ctx.debug_counters(0))
        except Exception as e:
            print(cands, octaves, "ERR", repr(e)[:200], round(time.time() - t, 3), ctx.debug_counters(0))
    ctx.close()
