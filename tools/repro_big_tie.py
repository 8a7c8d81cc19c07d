"""GPU box: the large all-tie frame of tools/soak7.py (seed 44) that made the tie kernel give up: timing and counters."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import large as soak7
from callspace import make_image
import ethzasl_brisk_amd as B
c = [soak7.big_case(i, 44) for i in range(20)]
c = [x for x in c if x[0] == 3645][0]
w, h, kind, thr, octaves, s = c
print(c)
img = make_image(kind, w, h, s)
print("grey levels", np.unique(img)[:8], "block", [int(np.argmax(img[0] != img[0, 0]))])
huge = B.Context(0, max_candidates=1 << 24, max_keypoints=1 << 22)
t = time.time()
try:
    k = B.BriskFeatureDetector(thr, octaves, context=huge).detect(img, capacity=1 << 22)
    print("ok", len(k), round(time.time() - t, 2), "s")
except Exception as e:
    print("ERR", repr(e)[:200], round(time.time() - t, 2), "s")
print(huge.debug_counters(0))
