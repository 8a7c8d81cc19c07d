#!/usr/bin/env python3
"""Stress of brisk_hip_pool against the oracle with diagnostics: which result differs (detected / described / descriptors), in
which configuration (sizes mixed or not, tokens or not).  usage: probe_pool_race.py [rounds]"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O, synth
X = O.Extractor()
big = [synth.frame_vga(9100 + i) for i in range(24)]
small = [synth.gen(333, 201, 7700 + i, 40) for i in range(6)]
def ora(img, thr, octv):
    k = O.detect(img, thr, octv); k2, d = X.compute(img, k); return k.tobytes(), np.ascontiguousarray(k2).tobytes(), d.tobytes()
wb = [ora(i, 70, 4) for i in big]; ws = [ora(i, 60, 2) for i in small]
import ethzasl_brisk_amd as B
ext = B.BriskDescriptorExtractor()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for mode in ("two sizes, pattern, slow", "two sizes, no tokens, slow", "one size, pattern, slow", "two sizes, pattern", "one size, pattern", "two sizes, fabricated", "one size, no tokens", "one size, tokens", "two sizes, no tokens", "two sizes, tokens", "two sizes, detect only", "two sizes, describe only"):
    pool = B.Pool(0, max_batch=8)
    bad = []
    def worker(t):
        for it in range(16 * rounds):
            sm = mode.startswith("two") and t % 3 == 2
            imgs, want, thr, octv = (small, ws, 60, 2) if sm else (big, wb, 70, 4)
            j = (5 * t + it) % len(imgs)
            if "describe only" in mode:
                k = np.frombuffer(want[j][0], B.KEYPOINT); tok = 0
            else:
                k, tok = pool.detect(imgs[j], thr, octv)
            if k.tobytes() != want[j][0]:
                bad.append((t, it, "small" if sm else "big", "detected", len(k), len(want[j][0]) // 28))
            if "detect only" in mode:
                continue
            if "slow" in mode:
                for f in B.KEYPOINT.names:   # (what the pytest version does between its calls: slow field-wise compares)
                    _ = np.array_equal(k[f], np.frombuffer(want[j][0], B.KEYPOINT)[f])
                    _ = [x for x in range(300)]
            if "pattern" in mode:
                use = tok if it % 3 == 0 else ((tok ^ (0x5A5A << 16)) if it % 3 == 1 else 0)
            elif "fabricated" in mode:
                use = tok ^ (0x5A5A << 16)
            else:
                use = tok if ("tokens" in mode and "no tokens" not in mode) else 0
            k2, d = pool.describe(ext, imgs[j], k, use)
            if np.ascontiguousarray(k2).tobytes() != want[j][1]:
                bad.append((t, it, "small" if sm else "big", "described", len(k2), len(want[j][1]) // 28))
            elif d.tobytes() != want[j][2]:
                bad.append((t, it, "small" if sm else "big", "descriptors", int((np.frombuffer(d.tobytes(), np.uint8) != np.frombuffer(want[j][2], np.uint8)).reshape(len(k2), -1).any(1).sum()), len(k2)))
    th = [threading.Thread(target=worker, args=(t,)) for t in range(12)]
    [x.start() for x in th]; [x.join() for x in th]
    print("%-28s groups/calls %s  mismatches %d %s" % (mode, pool.stats(), len(bad), bad[:6]), flush=True)
    pool.close()
