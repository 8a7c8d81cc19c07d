#!/usr/bin/env python3
"""One-off fuzz (GPU box) of the Hamming brute-force matcher: random query / train set sizes (empty sets and images
included), 1 ... 6 train images, descriptor lengths 16 ... 224 bytes (also lengths that are not a multiple of 16: the
reference ignores the bytes beyond the last full 128-bit word), low-entropy descriptors (plenty of equal distances),
masks of several kinds, k from 1 to beyond the train set, radii from 0 to beyond every distance - knnMatch and
radiusMatch rows compared with the oracle (brute-force-matcher.cc:80-213).
usage: python3 tools/soak.py matcher [cases] [seed]"""
import os
import sys
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def make_case(i, seed):
    rng = np.random.default_rng(seed * 6151 + i)
    dim = int(rng.choice([16, 32, 48, 48, 48, 64, 64, 40, 72, 96, 128, 200, 224]))
    nq = int(rng.integers(0, 400)) if rng.random() < 0.9 else int(rng.integers(400, 3000))
    nimg = int(rng.integers(1, 7))
    nts = [0 if rng.random() < 0.15 else int(rng.integers(1, 600)) for _ in range(nimg)]
    if rng.random() < 0.1:
        nts[0] = int(rng.integers(2000, 9000))
    levels = int(rng.choice([2, 4, 256]))
    mask_kind = int(rng.integers(0, 4)) if rng.random() < 0.5 else -1
    k = int(rng.choice([1, 1, 2, 2, 3, 5, 17, 1000]))
    radius = float(rng.choice([0.0, 0.5, 1.0, dim * 1.5, dim * 2.0 + 0.5, dim * 4.0, 1e9]))
    return (i, seed, dim, nq, nts, levels, mask_kind, k, radius)


def make_data(c):
    i, seed, dim, nq, nts, levels, mask_kind, k, radius = c
    rng = np.random.default_rng(seed * 977 + i)

    def rnd(n):
        if levels == 256:
            return rng.integers(0, 256, (n, dim), dtype=np.uint8)
        return (rng.integers(0, levels, (n, dim), dtype=np.uint8) * (255 // (levels - 1))).astype(np.uint8)
    q = rnd(nq)
    train = [rnd(n) for n in nts]
    for t in train:   # some exact duplicates of queries
        if len(t) and nq:
            for _ in range(min(3, len(t))):
                t[int(rng.integers(0, len(t)))] = q[int(rng.integers(0, nq))]
    masks = None
    if mask_kind >= 0:
        masks = []
        for t in train:
            if mask_kind == 3 and rng.random() < 0.5:
                masks.append(None)
                continue
            p = (0.5, 0.9, 0.1, 0.5)[mask_kind]
            m = (rng.random((nq, len(t))) < p).astype(np.uint8) * int(rng.integers(1, 256))
            if nq and rng.random() < 0.5:
                m[int(rng.integers(0, nq)), :] = 0
            masks.append(m)
    return q, train, masks


def oracle_case(c):
    import oracle_lib as O
    q, train, masks = make_data(c)
    k, radius = c[7], c[8]
    a = O.match_knn(q, train, k, masks)
    b = O.match_radius(q, train, radius, masks)
    return [r.tobytes() for r in a], [r.tobytes() for r in b]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cases = [make_case(i, seed) for i in range(n)]
    import bench
    with ProcessPoolExecutor(bench.usable_cores()) as ex:   # oracle processes are forked before HIP is loaded
        fut = ex.map(oracle_case, cases, chunksize=2)
        import ethzasl_brisk_amd as B
        ctx = B.Context(0)
        bad = 0
        for c, want in zip(cases, fut):
            try:
                q, train, masks = make_data(c)
                bf = B.BruteForceMatcher(context=ctx)
                bf.add(train)
                a = [r.tobytes() for r in bf.knnMatch(q, c[7], masks)]
                b = [r.tobytes() for r in bf.radiusMatch(q, c[8], masks)]
            except Exception as e:  # noqa
                bad += 1
                print("ERROR", c, repr(e)[:300], flush=True)
                continue
            if (a, b) != want:
                bad += 1
                wa, wb = want
                print("MISMATCH", c, "knn rows differ: %d" % sum(x != y for x, y in zip(a, wa)), "(%d vs %d rows)" % (len(a), len(wa)),
                      "radius rows differ: %d" % sum(x != y for x, y in zip(b, wb)), "(%d vs %d rows)" % (len(b), len(wb)), flush=True)
        print("matcher: %d cases (seed %d), %d bad" % (n, seed, bad))
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
