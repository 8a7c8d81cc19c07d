"""GPU box: are the detector half and the descriptor half of a chunk complementary?  Context A loops detect-only batches,
context B loops the descriptor half alone (debug bit 27: on the keypoints of a previous full batch), each alone and both
together on streams of different priority."""
import os
import sys
import time

import numpy as np
import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import ethzasl_brisk_amd as B  # noqa: E402
import synth  # noqa: E402

W, H, CHUNK, ND = 1920, 1080, 256, 32
dev = torch.device("cuda:0")
host = np.stack([synth.frame_1080p(1000 + s) for s in range(ND)])
ring = torch.from_numpy(host).to(dev)
frames = ring[torch.arange(CHUNK, device=dev) % ND].contiguous()
del ring
flags = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
pa, pb = [int(x) for x in os.environ.get("PROBE_PRIO", "0,-1").split(",")]
A, Bc = B.Context(0), B.Context(0)
extB = B.BriskDescriptorExtractor(context=Bc)
sA = torch.cuda.Stream(device=dev, priority=pa)
sB = torch.cuda.Stream(device=dev, priority=pb)
Bc.debug_set_flags(flags)
Bc.detect_describe_batch(extB, frames.data_ptr(), CHUNK, W, H, W * H, W, 80, 4, sB.cuda_stream)   # a full batch first
torch.cuda.synchronize()
Bc.debug_set_flags(flags | (1 << 27))
A.debug_set_flags(flags)


def loop(na, nb):
    for i in range(max(na, nb)):
        if i < na:
            A.detect_batch(frames.data_ptr(), CHUNK, W, H, W * H, W, 80, 4, sA.cuda_stream)
        if i < nb:
            Bc.detect_describe_batch(extB, frames.data_ptr(), CHUNK, W, H, W * H, W, 80, 4, sB.cuda_stream)


N = 32
for na, nb, name in ((N, 0, "detector half alone"), (0, N, "descriptor half alone"), (N, N, "both, two streams")):
    loop(min(na, 2), min(nb, 2))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(na, nb)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-24s: %.3f ms per chunk(pair)" % (name, dt / N * 1e3), flush=True)
