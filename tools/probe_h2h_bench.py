import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ethzasl_brisk_amd as B, synth, bench
var = sys.argv[1]
seeds = list(range(16)) if var != "seeds100" else [100 + i for i in range(8)]
host = np.stack([synth.frame_1080p(s) for s in seeds])
ctx = B.Context(0)
ext = B.BriskDescriptorExtractor(context=ctx)
r = bench.host_fed(ctx, ext, host, 512, 48)
print(var, r["fps"], r["host_to_host_fps"], r["host_to_host_vs_h2d_only"], r["host_to_host_MB_per_batch"], flush=True)
