import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, torch
import ethzasl_brisk_amd as B, synth
ctx = B.Context(0)
ctx.set_uniformity(8.0)
img = synth.frame_4k(2)
d = torch.from_numpy(img[None]).cuda()
for rep in range(3):
    ctx.detect_batch(d.data_ptr(), 1, 3840, 2160, 3840*2160, 3840, 80, 6, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
c = ctx.debug_counters(0)
print("keypoints", c["keypoints"], "k_uf_decide phases (10 ns ticks of the 100 MHz clock): after build %d, after decisions %d, after compaction %d, end %d" % tuple(c["experiment"][:4]))
