"""Run with BRISK_HIP_LIB=<libbrisk_hip_release.so> (tests/test_gpu_round6.py does, in a child process): the reference's goldens
and a batch with its host exit through the RELEASE library - the boundary of include/brisk_hip.h only, no brisk_hip_debug_* call."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np

import ethzasl_brisk_amd as B
from setfile import read_set

L = B.load_library()
assert "release" in B.LIB_PATH, B.LIB_PATH
assert not any(hasattr(L, s) for s in B.DEBUG_SYMBOLS), "the release library exports a debug entry point"


def same(a, b):
    return all(np.array_equal(a[f].view(np.uint32) if a[f].dtype == np.float32 else a[f],
                              b[f].view(np.uint32) if b[f].dtype == np.float32 else b[f]) for f in B.KEYPOINT.names)


ast = read_set(os.path.join(HERE, "golden", "brisk_verification_ast.set"))
har = read_set(os.path.join(HERE, "golden", "brisk_verification_harris.set"))
det = B.BriskFeatureDetector(70)
ext = B.BriskDescriptorExtractor()
for e in ast:  # test-binary-equal.cc:319-333
    k2, desc = ext.compute(e["image"], det.detect(e["image"]))
    g = e["keypoints"]
    assert len(k2) == len(g) and all(np.array_equal(k2[f].view(np.uint32), g[f].view(np.uint32)) for f in ("x", "y", "size", "angle", "response"))
    assert np.array_equal(k2["octave"], g["octave"]) and np.array_equal(k2["class_id"], g["class_id"]) and np.array_equal(desc, e["descriptors"])
for e in har:  # provided keypoints: orientation + descriptors
    g = e["keypoints"]
    k = np.zeros(len(g), B.KEYPOINT)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        k[f] = g[f]
    k["angle"] = -1
    k2, desc = ext.compute(e["image"], k)
    assert np.array_equal(k2["angle"].view(np.uint32), g["angle"].view(np.uint32)) and np.array_equal(desc, e["descriptors"])
# the batch path and its exit to host memory: both golden images (same size) as one batch
import torch
imgs = np.stack([e["image"] for e in ast])
h, w = imgs.shape[1:]
ctx = B.default_context(0)
d = torch.from_numpy(imgs).cuda()
ctx.detect_describe_batch(ext, d.data_ptr(), len(imgs), w, h, w * h, w, 70, 3, torch.cuda.current_stream().cuda_stream)
res = B.HostResults(len(imgs), 4096, 48, pinned=True)
assert ctx.batch_download_wait(ctx.batch_download_all(res, stream=torch.cuda.current_stream().cuda_stream)) == 0
for f, e in enumerate(ast):
    k, dd = res.frame(f)
    g = e["keypoints"]
    assert len(k) == len(g) and np.array_equal(k["x"].view(np.uint32), g["x"].view(np.uint32)) and np.array_equal(dd, e["descriptors"])
print("release library: goldens OK (%d + %d detected / described, %d + %d provided), batch + host exit OK"
      % (len(ast[0]["keypoints"]), len(ast[1]["keypoints"]), len(har[0]["keypoints"]), len(har[1]["keypoints"])))
