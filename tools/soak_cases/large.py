#!/usr/bin/env python3
"""One-off fuzz (GPU box): (a) the 16-bit image functions on random shapes and contents (incl. saturated values and the
widths the reference CHECKs against: both sides must refuse those), (b) a few large frames (2000 ... 4600 px wide, up to
6 octaves) through detect + describe, (c) ComputeScale on random keypoint lists - against the oracle.
usage: python3 tools/soak.py large [cases] [seed]"""
import os
import sys
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "soak_cases"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from callspace import make_image


def img16(i, seed):
    rng = np.random.default_rng(seed * 31337 + i)
    w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        img = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    elif kind == 1:
        img = (rng.integers(0, 2, (h, w)) * 65535).astype(np.uint16)
    else:
        img = rng.integers(65000, 65536, (h, w), dtype=np.uint16)
    return img


def big_case(i, seed):
    rng = np.random.default_rng(seed * 4241 + i)
    return (int(rng.integers(2000, 4600)), int(rng.integers(1100, 2600)), int(rng.integers(0, 5)) if i % 2 else 0,
            int(rng.integers(45, 110)), int(rng.integers(3, 7)), seed * 50 + i)


def oracle_big(c):
    import oracle_lib as O
    w, h, kind, thr, octaves, s = c
    if kind == 1:
        thr = max(thr, 90)
    img = make_image(kind, w, h, s)
    k = O.detect(img, thr, octaves)
    k2, d = O.Extractor().compute(img, k)
    return k.tobytes(), k2.tobytes(), d.tobytes()


def scale_case(i, seed):
    rng = np.random.default_rng(seed * 577 + i)
    w, h = int(rng.integers(60, 700)), int(rng.integers(60, 500))
    n = int(rng.integers(0, 60))
    pts = np.zeros(n, [("x", "<f4"), ("y", "<f4")])
    pts["x"] = rng.uniform(0, w, n)
    pts["y"] = rng.uniform(0, h, n)
    return (w, h, int(rng.integers(0, 5)), int(rng.integers(1, 100)), int(rng.integers(0, 5)), bool(rng.random() < 0.8), seed * 90 + i, pts)


def oracle_scale(c):
    import oracle_lib as O
    w, h, kind, thr, octaves, suppress, s, pts = c
    img = make_image(kind, w, h, s)
    k = np.zeros(len(pts), O.KP)
    k["x"], k["y"] = pts["x"], pts["y"]
    k["size"], k["angle"], k["class_id"] = 12.0, -1.0, -1
    out = O.compute_scale(img, k, thr, octaves, suppress)
    return None if out is None else out.tobytes()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    nbig = max(2, n // 50)
    bigs = [big_case(i, seed) for i in range(nbig)]
    scs = [scale_case(i, seed) for i in range(n)]
    import bench
    with ProcessPoolExecutor(bench.usable_cores()) as ex:   # oracle processes are forked before HIP is loaded
        fut_big = ex.map(oracle_big, bigs)
        fut_sc = ex.map(oracle_scale, scs, chunksize=4)
        import oracle_lib as O
        import ethzasl_brisk_amd as B
        ctx = B.Context(0, max_candidates=1 << 20, max_keypoints=1 << 18)
        bad = refused = 0
        for i in range(n):
            img = img16(i, seed)
            for name, fn, ofn in (("half", ctx.halfsample16, O.halfsample16), ("twothird", ctx.twothirdsample16, O.twothirdsample16),
                                  ("integral", ctx.integral_image16, O.integral16)):
                want = ofn(img)
                try:
                    got = fn(img)
                except B.BriskHipError as e:
                    if want is None:
                        refused += 1
                        continue
                    bad += 1
                    print("ERROR 16-bit", name, img.shape, repr(e)[:200], flush=True)
                    continue
                if want is None:  # the reference's loops write nothing at this shape: the engine returns OK and leaves dst untouched
                    if got.any():
                        bad += 1
                        print("MISMATCH 16-bit", name, img.shape, "written although the reference writes nothing", flush=True)
                    else:
                        refused += 1
                    continue
                if got.shape != want.shape or got.tobytes() != want.tobytes():
                    bad += 1
                    print("MISMATCH 16-bit", name, img.shape, "oracle refuses" if want is None else "", flush=True)
        ext = B.BriskDescriptorExtractor(context=ctx)
        for c, want in zip(bigs, fut_big):
            w, h, kind, thr, octaves, s = c
            if kind == 1:
                thr = max(thr, 90)
            try:
                img = make_image(kind, w, h, s)
                try:
                    k = B.BriskFeatureDetector(thr, octaves, context=ctx).detect(img, capacity=1 << 18)
                    k2, d = ext.compute(img, k)
                except B.BriskHipError as e:
                    if e.code != 4:
                        raise
                    # all-tie block images of this size: an error, not a truncated result; again on a workspace sized for it
                    huge = B.Context(0, max_candidates=1 << 24, max_keypoints=1 << 22)
                    k = B.BriskFeatureDetector(thr, octaves, context=huge).detect(img, capacity=1 << 22)
                    k2, d = B.BriskDescriptorExtractor(context=huge).compute(img, k)
                    huge.close()
                if (k.tobytes(), k2.tobytes(), d.tobytes()) != want:
                    bad += 1
                    print("MISMATCH big", c, len(k), len(want[0]) // 28, flush=True)
            except Exception as e:  # noqa
                bad += 1
                print("ERROR big", c, repr(e)[:300], flush=True)
        undefined = 0
        for c, want in zip(scs, fut_sc):
            w, h, kind, thr, octaves, suppress, s, pts = c
            try:
                img = make_image(kind, w, h, s)
                k = np.zeros(len(pts), B.KEYPOINT)
                k["x"], k["y"] = pts["x"], pts["y"]
                k["size"], k["angle"], k["class_id"] = 12.0, -1.0, -1
                try:
                    got = B.BriskFeatureDetector(thr, octaves, suppress, context=ctx).ComputeScale(img, k).tobytes()
                except B.BriskHipError as e:
                    if want is None and e.code == 7:
                        undefined += 1
                        continue
                    raise
                if want is None or got != want:
                    bad += 1
                    print("MISMATCH ComputeScale", c[:7], len(pts), "oracle: no defined result" if want is None else "", flush=True)
            except Exception as e:  # noqa
                bad += 1
                print("ERROR ComputeScale", c[:7], repr(e)[:300], flush=True)
        print("large: %d 16-bit images (%d calls refused on both sides), %d large frames, %d ComputeScale lists (%d without a defined result), %d bad"
              % (n, refused, nbig, n, undefined, bad))
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
