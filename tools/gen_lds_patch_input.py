#!/usr/bin/env python3
"""Input of tools/microbench_lds_patch.hip: BASELINE config 2 keypoints with their real scale indices and rotations.

usage: gen_lds_patch_input.py <out.bin> [nframes=4]

Frames = the survey's recipe (1920 x 1080, seeds 0 ...), keypoints = the ORACLE's detect (threshold 80, 4 octaves) +
compute (border filter, angle): test infrastructure feeding a benchmark tool, never the product.
File: int32 {nframes, w, h}; per frame int32 n, then n x {f32 x, f32 y, i32 scale index, i32 theta}; then the frames
(w x h u8 each).  Keypoints are in the processing order of k_desc_prepare (64-row bands, x inside a band).
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import synth  # noqa: E402


def main():
    out = sys.argv[1]
    nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    w, h = 1920, 1080
    E = O.Extractor()
    recs, imgs = [], []
    for f in range(nframes):
        img = synth.frame_1080p(f)
        kps = O.detect(img, 80, 4)
        kps, _ = E.compute(img, kps)
        sc = np.array([E.scale_index(float(s)) for s in kps["size"]], np.int32)
        ang = kps["angle"].astype(np.float32)
        theta = (np.float32(1024.0) * ang).astype(np.float64) / 360.0 + 0.5  # brisk-descriptor-extractor.cc:734-735
        theta = theta.astype(np.int32)
        theta = np.where(theta < 0, theta + 1024, theta)
        theta = np.where(theta >= 1024, theta - 1024, theta)
        order = np.lexsort((kps["x"].astype(np.int32), kps["y"].astype(np.int32) >> 6))
        r = np.zeros(len(kps), np.dtype([("x", "<f4"), ("y", "<f4"), ("s", "<i4"), ("t", "<i4")]))
        r["x"], r["y"], r["s"], r["t"] = kps["x"], kps["y"], sc, theta
        recs.append(r[order])
        imgs.append(img)
    with open(out, "wb") as fo:
        fo.write(np.array([nframes, w, h], np.int32).tobytes())
        for r in recs:
            fo.write(np.array([len(r)], np.int32).tobytes())
            fo.write(r.tobytes())
        for img in imgs:
            fo.write(img.tobytes())
    sides = np.concatenate([2 * E.size_list()[r["s"]].astype(np.int64) + 1 for r in recs])
    print("frames %d keypoints %d  patch side <=67: %.1f %%  <=101: %.1f %%  <=151: %.1f %%  <=201: %.1f %%" % (
        nframes, len(sides), 100 * np.mean(sides <= 67), 100 * np.mean(sides <= 101), 100 * np.mean(sides <= 151),
        100 * np.mean(sides <= 201)))


if __name__ == "__main__":
    main()
