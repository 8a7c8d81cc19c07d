"""GPU box: is k_detect ALONE (instruction-issue bound) complementary to k_describe (bound by the vector L1's line fills)?
Context A loops detect batches that end behind k_detect (BRISK_DETECT_PROBE=1: with the pyramid kernel, =2: without),
context B loops k_desc_prepare + k_describe alone (debug bits 19 + 27: on the keypoints and the integral images of a previous
full batch), each alone and both together on streams of different priority.  Timing only: the batches' results are void.
usage: BRISK_DETECT_PROBE=2 python3 tools/overlap_probe3.py [extra debug flags for the describe context]"""
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
probe = os.environ.pop("BRISK_DETECT_PROBE", "2")
Bc = B.Context(0)
extB = B.BriskDescriptorExtractor(context=Bc)
sA = torch.cuda.Stream(device=dev, priority=pa)
sB = torch.cuda.Stream(device=dev, priority=pb)
Bc.debug_set_flags(flags)
Bc.detect_describe_batch(extB, frames.data_ptr(), CHUNK, W, H, W * H, W, 80, 4, sB.cuda_stream)   # a full batch first
torch.cuda.synchronize()
Bc.debug_set_flags(flags | (1 << 27) | (1 << 19))
A = B.Context(0)
A.detect_batch(frames.data_ptr(), CHUNK, W, H, W * H, W, 80, 4, sA.cuda_stream)   # a full detect batch: a pyramid in the buffers
torch.cuda.synchronize()
os.environ["BRISK_DETECT_PROBE"] = probe   # (read once, at the first launch after this point)


def loop(na, nb):
    for i in range(max(na, nb)):
        if i < na:
            A.detect_batch(frames.data_ptr(), CHUNK, W, H, W * H, W, 80, 4, sA.cuda_stream)
        if i < nb:
            Bc.detect_describe_batch(extB, frames.data_ptr(), CHUNK, W, H, W * H, W, 80, 4, sB.cuda_stream)


N = 32
for na, nb, name in ((N, 0, "k_detect loop alone (probe %s)" % probe), (0, N, "k_describe loop alone"), (N, N, "both, two streams")):
    loop(min(na, 2), min(nb, 2))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(na, nb)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-36s: %.3f ms per chunk(pair)" % (name, dt / N * 1e3), flush=True)
