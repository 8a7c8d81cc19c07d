#!/usr/bin/env python3
"""Fuzzer of round 6's host-side paths against the oracle (GPU box): random image sizes / thresholds / octave counts / list
lengths through
  * brisk_hip_detect_images + brisk_hip_describe_images (the classes' multi-image overloads), with and without the
    same-images hint, pinned and pageable destinations, one list entry without keypoints now and then;
  * brisk_hip_detect_describe_batch_host_results (frames from pinned host memory, all results through
    brisk_hip_batch_download_all), row capacities that fit exactly / are one row short;
  * brisk_hip_pool from six threads (detect + describe with the detect call's token, a fabricated token, none).
usage: python3 tools/soak.py hostpaths [cases = 120] [seed = 5]"""
import os
import sys
import threading
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def make_case(i, seed):
    rng = np.random.default_rng(seed * 100003 + i)
    w = int(rng.choice([rng.integers(72, 200), rng.integers(200, 700), 640, 333, 1281, 1920]))
    h = int(rng.choice([rng.integers(72, 160), rng.integers(160, 500), 480, 201, 723, 1080]))
    if w * h > 1281 * 723:
        w, h = 1920, 1080
    thr = int(rng.choice([25, 40, 60, 70, 90, 120]))
    if w * h > 300000 and thr < 40:
        thr = 40   # (threshold 25 on a large frame exceeds the 65 536 keypoints the contexts of this suite are sized for)
    octv = int(rng.integers(0, 5))
    n = int(rng.integers(2, 10)) if w * h < 700000 else int(rng.integers(2, 4))
    nrect = max(6, int(300 * w * h / (1920 * 1080)))
    return {"i": i, "w": w, "h": h, "thr": thr, "oct": octv, "n": n, "nrect": nrect, "seed": seed * 7919 + i * 13,
            "kind": ["images", "h2h", "pool"][i % 3]}


def frames_of(c):
    import synth
    return [synth.gen(c["w"], c["h"], c["seed"] + f, c["nrect"]) for f in range(c["n"])]


def oracle_case(c):
    import oracle_lib as O
    X = O.Extractor()
    out = []
    for img in frames_of(c):
        k = O.detect(img, c["thr"], c["oct"])
        k2, d = X.compute(img, k)
        out.append((k.tobytes(), np.ascontiguousarray(k2).tobytes(), d.tobytes()))
    return out


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    cases = [make_case(i, seed) for i in range(ncases)]
    import bench
    with ProcessPoolExecutor(bench.usable_cores()) as ex:   # oracle processes are forked before torch / HIP are loaded
        fut = ex.map(oracle_case, cases, chunksize=2)
        import torch
        import ethzasl_brisk_amd as B
        KP = B.KEYPOINT
        ctx = B.Context(0, max_candidates=262144, max_keypoints=65536)
        ext = B.BriskDescriptorExtractor(context=ctx)
        pool = B.Pool(0, max_batch=8, max_keypoints=65536)
        bad = 0
        for c, want in zip(cases, fut):
            imgs = frames_of(c)
            n, w, h = c["n"], c["w"], c["h"]
            nk = [len(x[0]) // KP.itemsize for x in want]
            nd = [len(x[1]) // KP.itemsize for x in want]
            got = []
            try:
                if c["kind"] == "images":
                    pinned = bool(c["i"] & 8)
                    det = B.HostResults(n, sum(nk), 0, pinned=pinned)
                    assert ctx.batch_download_wait(ctx.detect_images(imgs, c["thr"], c["oct"], det)) == 0
                    lists = [np.ascontiguousarray(det.frame(f)[0]).copy() for f in range(n)]
                    empty = c["i"] % 5 == 0
                    if empty:
                        lists[n - 1] = lists[n - 1][:0]
                    res = B.HostResults(n, max(sum(len(k) for k in lists), 1), 48, pinned=pinned)
                    assert ctx.batch_download_wait(ctx.describe_images(ext, imgs, lists, res, same_images=bool(c["i"] & 4))) == 0
                    for f in range(n):
                        k2, d = res.frame(f, 48)
                        if empty and f == n - 1:
                            got.append((det.frame(f)[0].tobytes(), want[f][1] if len(k2) == 0 else b"x", want[f][2] if len(k2) == 0 else b"x"))
                        else:
                            got.append((np.ascontiguousarray(det.frame(f)[0]).tobytes(), np.ascontiguousarray(k2).tobytes(), np.ascontiguousarray(d).tobytes()))
                elif c["kind"] == "h2h":
                    src = torch.from_numpy(np.stack(imgs)).pin_memory()
                    short = c["i"] % 7 == 1 and sum(nd) > 1
                    res = B.HostResults(n, max(sum(nd) - (1 if short else 0), 1), 48, pinned=not (c["i"] & 8))
                    t = ctx.detect_describe_batch_host_results(ext, src.data_ptr(), n, w, h, w * h, w, c["thr"], c["oct"], res)
                    rc, flagged = ctx.batch_download_wait(t, check=False)
                    cut = [f for f in range(n) if int(res.flags[f]) & B.ROWS_CUT]
                    assert (rc == 0 and not cut) if not short else (rc == 4 and len(cut) >= 1 and flagged == len(cut)), (rc, flagged, cut, short)
                    for f in range(n):
                        kd, _ = ctx.batch_download(f, described=False)
                        if f in cut:
                            assert int(res.counts[f]) == nd[f]
                            kg, dg = ctx.batch_download(f, described=True)
                        else:
                            kg, dg = res.frame(f, 48)
                        got.append((kd.tobytes(), np.ascontiguousarray(kg).tobytes(), np.ascontiguousarray(dg).tobytes()))
                else:
                    out = [None] * n

                    def worker(f):
                        try:
                            k, tok = pool.detect(imgs[f], c["thr"], c["oct"], capacity=65536)
                            use = tok if (f + c["i"]) % 3 == 0 else ((tok ^ (0x5A5A << 16)) if (f + c["i"]) % 3 == 1 else 0)
                            k2, d = pool.describe(ext, imgs[f], k, use)
                            out[f] = (k.tobytes(), np.ascontiguousarray(k2).tobytes(), np.ascontiguousarray(d).tobytes())
                        except Exception as e:
                            out[f] = (repr(e)[:200], b"", b"")
                    th = [threading.Thread(target=worker, args=(f,)) for f in range(n)]
                    for x in th:
                        x.start()
                    for x in th:
                        x.join()
                    got = out
            except Exception as e:
                print("ERROR", c, repr(e)[:300], flush=True)
                bad += 1
                continue
            if got != want:
                bad += 1
                diff = [f for f in range(n) if got[f] != want[f]]
                print("MISMATCH", c, "frames", diff, [tuple(a == b for a, b in zip(got[f], want[f])) for f in diff[:3]], flush=True)
        rev = B.load_library().brisk_hip_kernel_revision().decode()
        print("hostpaths: %d cases (seed %d; %d frames), %d bad, kernel revision %s" % (ncases, seed, sum(c["n"] for c in cases), bad, rev))
        pool.close()
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
