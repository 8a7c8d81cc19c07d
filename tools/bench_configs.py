#!/usr/bin/env python3
"""Times the BASELINE configurations that are not the bench line (bench.py measures config 2 / 3):
  config 1  one 640x480 frame, 4 octaves, host-buffer calls
  config 4  3840x2160 frames, 6 octaves, device-resident batch, plain and uniformity-enforced (radius 8 -> ~4 k keypoints)
  config 5  descriptor only: 100 000 provided keypoints on one 1080p frame, orientation estimated"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import ethzasl_brisk_amd as B
import synth


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


ctx = B.Context(0, max_candidates=131072, max_keypoints=131072)
ext = B.BriskDescriptorExtractor(context=ctx)
stream = torch.cuda.current_stream().cuda_stream

img = synth.frame_vga(1)
det = B.BriskFeatureDetector(70, 4, context=ctx)
k = det.detect(img)
dt = timed(lambda: ext.compute(img, det.detect(img)), 100)
print("config 1: 640x480, thr 70, 4 octaves, host-buffer detect + describe: %.3f ms / frame (%d described)" % (dt * 1e3, len(ext.compute(img, k)[0])))

frames = np.stack([synth.frame_4k(2 + i) for i in range(4)])
d = torch.from_numpy(frames).cuda()
idx = torch.arange(32, device="cuda") % 4
batch = d[idx].contiguous()
n, h, w = batch.shape
for radius in (0.0, 8.0):
    ctx.set_uniformity(radius)
    dt = timed(lambda: ctx.detect_describe_batch(ext, batch.data_ptr(), n, w, h, w * h, w, 80, 6, stream), 5)
    nk = np.mean([len(ctx.batch_download(f, True)[0]) for f in range(4)])
    print("config 4: 3840x2160, thr 80, 6 octaves, batch of %d, uniformity radius %g: %.2f ms / frame = %.0f frames/s (%.0f described / frame)"
          % (n, radius, dt / n * 1e3, n / dt, nk))
ctx.set_uniformity(0.0)

img = synth.frame_1080p(0)
rng = np.random.default_rng(7)
kp = np.zeros(100000, B.KEYPOINT)
size = np.exp(rng.uniform(np.log(8.64), np.log(200.0), 100000)).astype(np.float32)
kp["size"] = size
kp["x"] = rng.uniform(0, 1920, 100000).astype(np.float32)
kp["y"] = rng.uniform(0, 1080, 100000).astype(np.float32)
kp["angle"] = -1
k2, dd = ext.compute(img, kp)
dt = timed(lambda: ext.compute(img, kp), 10)
print("config 5: 100000 provided keypoints on one 1080p frame (%d inside the border), host-buffer describe: %.2f ms = %.1f M descriptors/s"
      % (len(k2), dt * 1e3, len(k2) / dt / 1e6))
