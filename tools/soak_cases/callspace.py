#!/usr/bin/env python3
"""One-off fuzz (GPU box): random image sizes (9 ... 1100 px a side, every residue of the tile / down-sampling column
classes), thresholds (1 ... 140: fast and ordered path), octave counts (0 ... 6), content kinds and call shapes (single
host call; device batch of 1 ... 5 frames, tight / in-place eligible / padded / unaligned layouts; host-fed batch), each
compared bit-exactly with the oracle.
usage: python3 tools/soak.py callspace [cases] [seed]"""
import os
import sys
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def make_image(kind, w, h, seed):
    import synth
    rng = np.random.default_rng(seed)
    if kind == 0 and w > 64 and h > 64:
        return synth.gen(w, h, seed, max(4, w * h // 7000))
    if kind == 1:
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == 2:   # smooth ramps + sparse impulses (isolated maxima, plateaus)
        y, x = np.mgrid[0:h, 0:w]
        img = (x * 255.0 / max(w - 1, 1) * 0.5 + y * 255.0 / max(h - 1, 1) * 0.5)
        for _ in range(max(1, w * h // 400)):
            img[rng.integers(0, h), rng.integers(0, w)] = rng.integers(0, 256)
        return np.clip(img, 0, 255).astype(np.uint8)
    if kind == 3:   # blocks of few grey levels (ties)
        b = int(rng.integers(2, 9))
        lv = rng.integers(0, 4, (h // b + 1, w // b + 1)) * 80 + 7
        return np.kron(lv, np.ones((b, b)))[:h, :w].astype(np.uint8)
    # low-amplitude noise on blobs
    img = np.kron(rng.integers(0, 2, (h // 12 + 1, w // 12 + 1)) * 150 + 40, np.ones((12, 12)))[:h, :w]
    return np.clip(img + rng.normal(0, 6, (h, w)), 0, 255).astype(np.uint8)


def make_case(i, seed):
    rng = np.random.default_rng(seed * 100003 + i)
    small = rng.random() < 0.35
    w = int(rng.integers(9, 200)) if small else int(rng.integers(64, 1100))
    h = int(rng.integers(9, 160)) if small else int(rng.integers(48, 800))
    kind = int(rng.integers(0, 5))
    thr = int(rng.integers(1, 20)) if (rng.random() < 0.12 and w * h < 90000) else int(rng.integers(20, 141))
    if kind == 1 and w * h > 120000:
        thr = max(thr, 60)   # (noise images: keep the candidate lists inside the context's capacity)
    octaves = int(rng.integers(0, 7))
    nfr = int(rng.integers(1, 6)) if rng.random() < 0.5 else 0   # 0: host call on one frame
    # memory layout of a device batch: 0 tight, 1 width a multiple of 64 (layer 0 is then read in place), 2 padded rows,
    # 3 padded rows + padded frames + a base address that is not 16-byte aligned, 4 frames in host memory (host-fed entry)
    lay = int(rng.integers(0, 5)) if nfr else 0
    if lay == 1:
        w = max(64, w // 64 * 64)
    return (i, w, h, kind, thr, octaves, nfr, seed * 1000 + i, lay)


def oracle_case(c):
    import oracle_lib as O
    i, w, h, kind, thr, octaves, nfr, s, lay = c
    out = []
    for f in range(max(nfr, 1)):
        img = make_image(kind, w, h, s * 8 + f)
        k = O.detect(img, thr, octaves)
        k2, d = O.Extractor().compute(img, k)
        out.append((k.tobytes(), k2.tobytes(), d.tobytes()))
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cases = [make_case(i, seed) for i in range(n)]
    import bench
    with ProcessPoolExecutor(bench.usable_cores()) as ex:   # oracle processes are forked before torch / HIP are loaded
        fut = ex.map(oracle_case, cases, chunksize=2)
        import torch
        import ethzasl_brisk_amd as B
        ctx = B.Context(0, max_candidates=262144, max_keypoints=65536)
        ext = B.BriskDescriptorExtractor(context=ctx)
        st = torch.cuda.current_stream().cuda_stream
        bad = skipped = 0
        big = None

        def run(c, cx, ex_):
            i, w, h, kind, thr, octaves, nfr, s, lay = c
            have = []
            if nfr == 0:
                img = make_image(kind, w, h, s * 8)
                det = B.BriskFeatureDetector(thr, octaves, context=cx)
                k = det.detect(img, capacity=1 << 20)
                k2, d = ex_.compute(img, k)
                have.append((k.tobytes(), k2.tobytes(), d.tobytes()))
            else:
                frames = np.stack([make_image(kind, w, h, s * 8 + f) for f in range(nfr)])
                if lay == 4:
                    hbuf = torch.from_numpy(frames).pin_memory()
                    cx.detect_describe_batch_host(ex_, hbuf.data_ptr(), nfr, w, h, w * h, w, thr, octaves)
                elif lay >= 2:
                    rp = w + int(1 + (s * 7 + i) % 90) if lay == 3 else (w + 63) // 64 * 64 + 64 * (i % 2)
                    fp = rp * h + (0 if lay == 2 else 16 * (i % 5) + (i % 3))
                    off = 0 if lay == 2 else 1 + i % 15
                    buf = np.full(off + fp * nfr + 64, 0xA5, np.uint8)
                    for f in range(nfr):
                        v = buf[off + f * fp: off + f * fp + rp * h].reshape(h, rp)
                        v[:, :w] = frames[f]
                    d = torch.from_numpy(buf).cuda()
                    cx.detect_describe_batch(ex_, d.data_ptr() + off, nfr, w, h, fp, rp, thr, octaves, st)
                else:
                    d = torch.from_numpy(frames).cuda()
                    cx.detect_describe_batch(ex_, d.data_ptr(), nfr, w, h, w * h, w, thr, octaves, st)
                torch.cuda.synchronize()
                assert cx.batch_status(nfr) == 0   # (raises BRISK_HIP_ERR_CAPACITY itself)
                for f in range(nfr):
                    kd, _ = cx.batch_download(f, described=False)
                    kg, dg = cx.batch_download(f, described=True)
                    have.append((kd.tobytes(), kg.tobytes(), dg.tobytes()))
            return have

        for c, want in zip(cases, fut):
            try:
                try:
                    have = run(c, ctx, ext)
                except Exception as e:  # noqa
                    if "ERR_CAPACITY" not in repr(e):
                        raise
                    # an error, never a truncated result (tie-heavy block images at low thresholds): again with a
                    # workspace sized for it, the way the drop-in classes do
                    skipped += 1
                    if big is None:
                        big = B.Context(0, max_candidates=1 << 21, max_keypoints=1 << 19)
                        big_ext = B.BriskDescriptorExtractor(context=big)
                    have = run(c, big, big_ext)
            except Exception as e:  # noqa
                bad += 1
                print("ERROR", c, repr(e)[:300], flush=True)
                continue
            if want != have:
                bad += 1
                KP = B.KEYPOINT
                for f, (a, b) in enumerate(zip(want, have)):
                    if a != b:
                        print("MISMATCH", c, "frame", f, "detected %d vs %d" % (len(a[0]) // KP.itemsize, len(b[0]) // KP.itemsize),
                              "described %d vs %d" % (len(a[1]) // KP.itemsize, len(b[1]) // KP.itemsize),
                              "kp equal", a[0] == b[0], a[1] == b[1], "desc equal", a[2] == b[2], flush=True)
        print("callspace: %d cases (seed %d), %d bad, %d of them repeated on a larger workspace after BRISK_HIP_ERR_CAPACITY" % (n, seed, bad, skipped))
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
