#!/usr/bin/env python3
"""One-off fuzz (GPU box) over the options of the two classes: detection masks, suppressScaleNonmaxima = false,
uniformity enforcement / KeyPointBucketing with random parameters, extractor flags (rotation / scale invariance), both
pattern versions, pattern scales, and keypoints that come back through the host between detect() and compute() -
every case bit-equal to the oracle (or both sides agree that the reference has no defined result).
usage: python3 tools/soak.py options [cases] [seed]"""
import os
import sys
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "soak_cases"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from callspace import make_image

PATTERN_SCALES = (1.0, 0.7, 1.3, 1.0)


def make_mask(rng, w, h):
    kind = int(rng.integers(0, 3))
    if kind == 0:   # a few zero rectangles
        m = np.full((h, w), 255, np.uint8)
        for _ in range(int(rng.integers(1, 5))):
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            m[y0:y0 + int(rng.integers(1, h // 2 + 2)), x0:x0 + int(rng.integers(1, w // 2 + 2))] = 0
        return m
    if kind == 1:   # sparse: one pixel in three is zero
        return (rng.integers(0, 3, (h, w)) > 0).astype(np.uint8) * int(rng.integers(1, 256))
    m = np.zeros((h, w), np.uint8)   # only a window is allowed
    x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
    m[y0:y0 + h // 2 + 1, x0:x0 + w // 2 + 1] = 1
    return m


def make_case(i, seed):
    rng = np.random.default_rng(seed * 7919 + i)
    w, h = int(rng.integers(40, 900)), int(rng.integers(40, 640))
    kind = int(rng.integers(0, 5))
    thr = int(rng.integers(20, 120))
    if kind == 1 and w * h > 120000:
        thr = max(thr, 60)
    octaves = int(rng.integers(0, 5))
    opt = {
        "mask": bool(rng.random() < 0.4),
        "no_scale_nms": bool(rng.random() < 0.15),
        "post": int(rng.integers(0, 4)),   # 0/1 none, 2 uniformity, 3 bucketing
        "radius": float(rng.uniform(3.0, 40.0)), "max_kp": int(rng.integers(1, 2000)) if rng.random() < 0.5 else 0x7FFFFFFF,
        "nbu": int(rng.integers(1, 9)), "nbv": int(rng.integers(1, 9)),
        "rot": bool(rng.random() < 0.7), "scl": bool(rng.random() < 0.7), "version": 2 if rng.random() < 0.7 else 1,
        "pscale": PATTERN_SCALES[int(rng.integers(0, 4))],
    }
    if opt["post"] == 3 and opt["max_kp"] == 0x7FFFFFFF:
        opt["max_kp"] = int(rng.integers(1, 500))
    return (i, w, h, kind, thr, octaves, seed * 1000 + i, opt)


def oracle_case(c):
    import oracle_lib as O
    i, w, h, kind, thr, octaves, s, opt = c
    img = make_image(kind, w, h, s * 8)
    mask = make_mask(np.random.default_rng(s), w, h) if opt["mask"] else None
    k = O.detect(img, thr, octaves, mask=mask, suppress_scale_nonmaxima=not opt["no_scale_nms"])
    if k is None:
        return None
    if opt["post"] == 2:
        k = O.enforce_uniformity(k, h, w, opt["radius"], opt["max_kp"])
    elif opt["post"] == 3:
        k = O.key_point_bucketing(k, h, w, opt["max_kp"], opt["nbu"], opt["nbv"])
        if k is None:
            return "bucketing-check"
    X = O.Extractor(opt["rot"], opt["scl"], opt["version"], opt["pscale"])
    k2, d = X.compute(img, k)
    return (k.tobytes(), k2.tobytes(), d.tobytes())


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cases = [make_case(i, seed) for i in range(n)]
    import bench
    with ProcessPoolExecutor(bench.usable_cores()) as ex:   # oracle processes are forked before torch / HIP are loaded
        fut = ex.map(oracle_case, cases, chunksize=2)
        import ethzasl_brisk_amd as B
        ctx = B.Context(0, max_candidates=1 << 20, max_keypoints=1 << 18)
        exts = {}
        bad = undefined = 0
        for c, want in zip(cases, fut):
            i, w, h, kind, thr, octaves, s, opt = c
            try:
                img = make_image(kind, w, h, s * 8)
                mask = make_mask(np.random.default_rng(s), w, h) if opt["mask"] else None
                kw = {}
                if opt["post"] == 2:
                    kw = dict(uniformityRadius=opt["radius"], maxNumKpt=opt["max_kp"])
                elif opt["post"] == 3:
                    kw = dict(maxNumKpt=opt["max_kp"], numBucketsU=opt["nbu"], numBucketsV=opt["nbv"])
                det = B.BriskFeatureDetector(thr, octaves, not opt["no_scale_nms"], context=ctx, **kw)
                try:
                    k = det.detect(img, mask=mask, capacity=1 << 18)
                except B.BriskHipError as e:
                    if want is None and e.code == 7:   # BRISK_HIP_ERR_UNSUPPORTED: no defined result in the reference
                        undefined += 1
                        continue
                    if want == "bucketing-check" and e.code == 1:   # the reference CHECKs these arguments
                        undefined += 1
                        continue
                    raise
                if want is None or want == "bucketing-check":
                    bad += 1
                    print("MISMATCH", c, "the engine answered where the reference has no defined result", flush=True)
                    continue
                key = (opt["rot"], opt["scl"], opt["version"], opt["pscale"])
                if key not in exts:
                    exts[key] = B.BriskDescriptorExtractor(opt["rot"], opt["scl"], opt["version"], opt["pscale"], context=ctx)
                k2, d = exts[key].compute(img, k)
                have = (k.tobytes(), k2.tobytes(), d.tobytes())
            except Exception as e:  # noqa
                bad += 1
                print("ERROR", c, repr(e)[:300], flush=True)
                continue
            if want != have:
                bad += 1
                KP = B.KEYPOINT
                print("MISMATCH", c, "detected %d vs %d" % (len(want[0]) // KP.itemsize, len(have[0]) // KP.itemsize),
                      "described %d vs %d" % (len(want[1]) // KP.itemsize, len(have[1]) // KP.itemsize),
                      "kp equal", want[0] == have[0], want[1] == have[1], "desc equal", want[2] == have[2], flush=True)
        print("options: %d cases (seed %d), %d bad, %d without a defined result in the reference (both sides agree)" % (n, seed, bad, undefined))
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
