"""GPU box: first differing rows of matcher fuzz cases (tools/soak6.py) against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import matcher as soak6
import oracle_lib as O
import ethzasl_brisk_amd as B
for i, seed in ((1479, 1), (1449, 1), (1497, 1)):
    c = soak6.make_case(i, seed)
    q, train, masks = soak6.make_data(c)
    print(c, [None if m is None else m.shape for m in (masks or [])])
    bf = B.BruteForceMatcher(); bf.add(train)
    a = bf.knnMatch(q, c[7], masks); wa = O.match_knn(q, train, c[7], masks)
    shown = 0
    for r, (x, y) in enumerate(zip(a, wa)):
        if x.tobytes() != y.tobytes() and shown < 3:
            print(" knn row", r, "gpu", x[:3], "oracle", y[:3]); shown += 1
    b = bf.radiusMatch(q, c[8], masks); wb = O.match_radius(q, train, c[8], masks)
    shown = 0
    for r, (x, y) in enumerate(zip(b, wb)):
        if x.tobytes() != y.tobytes() and shown < 3:
            print(" radius row", r, "gpu", len(x), x[:3], "oracle", len(y), y[:3]); shown += 1
