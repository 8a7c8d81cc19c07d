#!/usr/bin/env python3
"""One-off soak (GPU box): many random frames of several sizes / thresholds / octave counts through the batch path,
every frame compared bit-exactly with the oracle (run with one oracle process per usable core).
usage: python3 tools/soak.py frames [frames per configuration]"""
import os
import sys
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def oracle_one(args):
    import oracle_lib as O
    import synth
    w, h, seed, nrect, thr, octaves = args
    img = synth.gen(w, h, seed, nrect)
    k = O.detect(img, thr, octaves)
    k2, d = O.Extractor().compute(img, k)
    return k.tobytes(), k2.tobytes(), d.tobytes()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    configs = [(1920, 1080, 300, 80, 4), (1920, 1080, 300, 45, 3), (640, 480, 60, 70, 4), (333, 201, 40, 60, 2),
               (1281, 723, 200, 55, 4), (426, 320, 50, 35, 3), (800, 600, 120, 100, 0)]
    jobs = [(w, h, 1000 * ci + i, nrect, thr, octaves) for ci, (w, h, nrect, thr, octaves) in enumerate(configs) for i in range(n)]
    import bench
    with ProcessPoolExecutor(bench.usable_cores()) as ex:   # oracle processes are forked before torch / HIP are loaded
        fut = ex.map(oracle_one, jobs, chunksize=2)
        import torch
        import ethzasl_brisk_amd as B
        import synth
        ctx = B.Context(0, max_candidates=262144, max_keypoints=65536)
        ext = B.BriskDescriptorExtractor(context=ctx)
        st = torch.cuda.current_stream().cuda_stream
        got = []
        for ci, (w, h, nrect, thr, octaves) in enumerate(configs):
            frames = np.stack([synth.gen(w, h, 1000 * ci + i, nrect) for i in range(n)])
            d = torch.from_numpy(frames).cuda()
            for rep in range(2):   # the second pass runs on a dirty workspace
                ctx.detect_describe_batch(ext, d.data_ptr(), n, w, h, w * h, w, thr, octaves, st)
            torch.cuda.synchronize()
            assert ctx.batch_status(n) == 0
            for f in range(n):
                kd, _ = ctx.batch_download(f, described=False)
                kg, dg = ctx.batch_download(f, described=True)
                got.append((kd.tobytes(), kg.tobytes(), dg.tobytes()))
        bad = 0
        for j, (want, have) in enumerate(zip(fut, got)):
            if want != have:
                bad += 1
                KP = B.KEYPOINT
                kw, kh = np.frombuffer(want[0], KP), np.frombuffer(have[0], KP)
                print("MISMATCH", jobs[j], "detected %d vs %d" % (len(kw), len(kh)), "described bytes equal:", want[1] == have[1],
                      "desc equal:", want[2] == have[2])
                k2w, k2h = np.frombuffer(want[1], KP), np.frombuffer(have[1], KP)
                print("   described %d vs %d" % (len(k2w), len(k2h)))
                if len(k2w) == len(k2h):
                    diff = [i for i in range(len(k2w)) if k2w[i].tobytes() != k2h[i].tobytes()]
                    print("   differing described:", len(diff), [(i, k2w[i], k2h[i]) for i in diff[:3]])
                if len(kw) == len(kh):
                    diff = [i for i in range(len(kw)) if kw[i].tobytes() != kh[i].tobytes()]
                    print("   first differing keypoints:", [(kw[i], kh[i]) for i in diff[:2]])
                else:
                    sw, sh = set(x.tobytes() for x in kw), set(x.tobytes() for x in kh)
                    print("   only oracle:", [np.frombuffer(b, KP)[0] for b in list(sw - sh)[:3]], " only gpu:", [np.frombuffer(b, KP)[0] for b in list(sh - sw)[:3]])
        print("frames: %d frames, %d mismatches" % (len(jobs), bad))
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
