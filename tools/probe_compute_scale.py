import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import ethzasl_brisk_amd as B, synth, oracle_lib as O
img = synth.frame_1080p(5)
rng = np.random.default_rng(3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
kin = np.zeros(n, B.KEYPOINT)
kin["x"] = rng.uniform(60, 1860, n).astype(np.float32); kin["y"] = rng.uniform(60, 1020, n).astype(np.float32)
kin["size"] = 12; kin["angle"] = -1; kin["class_id"] = np.arange(n)
ctx = B.Context(0, max_candidates=262144, max_keypoints=65536)
det = B.BriskFeatureDetector(60, 4, context=ctx)
for _ in range(2): out = det.ComputeScale(img, kin)
t0 = time.perf_counter()
for _ in range(5): out = det.ComputeScale(img, kin)
dt = (time.perf_counter() - t0) / 5
print("ComputeScale %d provided points on 1080p: %.2f ms, %d keypoints out" % (n, dt * 1e3, len(out)))
want = O.compute_scale(img, kin, 60, 4)
ok = want is not None and len(want) == len(out) and all(np.array_equal(out[f].view(np.uint32) if out[f].dtype == np.float32 else out[f], want[f].view(np.uint32) if want[f].dtype == np.float32 else want[f]) for f in want.dtype.names)
print("equal to the oracle:", ok, None if want is None else len(want))
