"""CPU model of k_describe's line traffic on one XCD: which processing order / how many keypoints in flight keep the
gathered lines in a 4 MB L2 (LRU, 128-byte lines), with a 32 KB L1 per CU in front.  One synthetic 1080p frame, the
oracle's keypoints, the reference pattern LUT.  Usage: cache_sim.py [waves_per_cu ...]"""
import sys
from collections import OrderedDict

import numpy as np

sys.path.insert(0, "tests")
import oracle_lib as O  # noqa: E402
import synth  # noqa: E402

W, H = 1920, 1080
IST = 1936  # integral row stride (elements)
img = synth.frame_1080p(1000)
X = O.Extractor()
kps, _ = X.compute(img, O.detect(img, 80, 4))
pat = X.pattern()
n = len(kps)
sc = np.array([X.scale_index(s) for s in kps["size"]])
theta = (np.floor(1024 * kps["angle"] / 360.0 + 0.5).astype(int)) % 1024
print("keypoints", n)
INT_BASE, IMG_BASE = 0, 1 << 30


def sample_lines(k, rot):
    """per pattern point: the line ids of the 10 gathers (8 integral pairs, 2 image bytes), in issue order"""
    p = pat[sc[k], rot]  # [np][3]
    xf = p[:, 0] + kps["x"][k]
    yf = p[:, 1] + kps["y"][k]
    sg = p[:, 2]
    xl = np.floor(xf - sg + 0.5).astype(np.int64)
    xr = np.floor(xf + sg + 0.5).astype(np.int64)
    yt = np.floor(yf - sg + 0.5).astype(np.int64)
    yb = np.floor(yf + sg + 0.5).astype(np.int64)
    out = []
    for (r, c) in ((yt, xl), (yt, xr), (yt + 1, xl), (yt + 1, xr), (yb, xl), (yb, xr), (yb + 1, xl), (yb + 1, xr)):
        a = (r * IST + c) * 4
        out.append((INT_BASE + a) >> 7)
    out.append((IMG_BASE + (yb - 1) * W + xr + 1) >> 7)
    out.append((IMG_BASE + (yb - 1) * W + xl + 1) >> 7)
    return np.stack(out)  # [10][np]


def order_keys(mode):
    x, y = kps["x"].astype(int), kps["y"].astype(int)
    if mode == "band64":
        return np.lexsort((x, y >> 6))
    if mode == "band128":
        return np.lexsort((x, y >> 7))
    if mode == "scale_band64":
        return np.lexsort((x, y >> 6, sc >> 4))
    if mode == "raster":
        return np.arange(n)
    if mode == "random":
        return np.random.default_rng(0).permutation(n)
    raise ValueError(mode)


def simulate(order, waves_per_cu, ncu=32, l1_lines=256, l2_lines=32768, stages=(0, 1), fused=False):
    nw = waves_per_cu * ncu
    l1 = [OrderedDict() for _ in range(ncu)]
    l2 = OrderedDict()
    acc = req = hit = 0
    jobs = [(k, st) for st in stages for k in order]  # stage 0 of every keypoint, then stage 1 (two launches)
    if fused:
        jobs = [(k, 2) for k in order]  # one wave: orientation pass, then the rotated pass of the same keypoint
    lines_cache = {}
    pos = 0
    active = []  # (wave, lines[10][np], next instruction)
    free = list(range(nw))
    while pos < len(jobs) or active:
        while free and pos < len(jobs):
            k, st = jobs[pos]
            pos += 1
            L = np.concatenate([sample_lines(k, 0), sample_lines(k, theta[k])]) if st == 2 else sample_lines(k, theta[k] if st else 0)
            active.append([free.pop(0), L, 0])
        nxt = []
        for a in active:  # one gather instruction per wave and turn
            w, L, i = a
            cu = w % ncu
            c1 = l1[cu]
            for ln in np.unique(L[i]):
                ln = int(ln)
                acc += 1
                if ln in c1:
                    c1.move_to_end(ln)
                    continue
                c1[ln] = 1
                if len(c1) > l1_lines:
                    c1.popitem(last=False)
                req += 1
                if ln in l2:
                    hit += 1
                    l2.move_to_end(ln)
                else:
                    l2[ln] = 1
                    if len(l2) > l2_lines:
                        l2.popitem(last=False)
            a[2] += 1
            if a[2] < len(L):
                nxt.append(a)
            else:
                free.append(w)
        active = nxt
    ns = len(jobs) * pat.shape[2] * (2 if fused else 1)
    return acc / ns, req / ns, hit / max(req, 1)


if __name__ == "__main__":
    wl = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 12, 24]
    for mode, fused in (("band64", False), ("band64", True), ("raster", True), ("random", True)):
        o = order_keys(mode)
        for wpc in wl:
            a, r, h = simulate(o, wpc, fused=fused)
            print("%-13s %s %2d waves/CU (%4d keypoint passes in flight): distinct lines per sample-instruction-set %.2f, L2 requests/sample %.2f, L2 hit %.3f, misses/sample %.2f"
                  % (mode, "fused" if fused else "split", wpc, wpc * 32, a, r, h, r * (1 - h)), flush=True)
