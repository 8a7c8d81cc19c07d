"""GPU box: chunks of 256 1080p frames alternating over TWO engine contexts on two streams (chunk i + 1's detector runs
while chunk i is described) against one context; k_describe launch knobs per run.
usage: overlap_probe.py [flags ...]   (each flags value: debug flags for both contexts; results printed per value)"""
import sys
import time

import numpy as np
import torch

import os as _os
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, _os.path.join(_ROOT, "tests"))
import ethzasl_brisk_amd as B  # noqa: E402
import synth  # noqa: E402

W, H, CHUNK, ND = 1920, 1080, 256, 32
dev = torch.device("cuda:0")
host = np.stack([synth.frame_1080p(1000 + s) for s in range(ND)])
ring = torch.from_numpy(host).to(dev)
frames = ring[torch.arange(CHUNK, device=dev) % ND].contiguous()
del ring
import os
flag_list = [int(x, 0) for x in sys.argv[1:]] or [0]
NCTX = [int(x) for x in os.environ.get("PROBE_NCTX", "1,2").split(",")]
NCH = int(os.environ.get("PROBE_CHUNKS", "48"))
ctxs = [B.Context(0), B.Context(0)]
exts = [B.BriskDescriptorExtractor(context=c) for c in ctxs]
PRIO = [int(x) for x in os.environ.get("PROBE_PRIO", "0,-1").split(",")]
streams = [torch.cuda.Stream(device=dev, priority=PRIO[0]), torch.cuda.Stream(device=dev, priority=PRIO[1])]


STAGGER_MS = float(os.environ.get("PROBE_STAGGER_MS", "0"))


def run(nctx, chunks):
    if nctx == 2 and STAGGER_MS > 0:
        torch.cuda.synchronize()
        with torch.cuda.stream(streams[1]):
            torch.cuda._sleep(int(STAGGER_MS * 1e-3 * 2.0e9))
    for i in range(chunks):
        j = i % nctx
        ctxs[j].detect_describe_batch(exts[j], frames.data_ptr(), CHUNK, W, H, W * H, W, 80, 4, streams[j].cuda_stream)


for flags in flag_list:
    for c in ctxs:
        c.debug_set_flags(flags)
    for nctx in NCTX:
        run(nctx, 4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = NCH
        run(nctx, n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert ctxs[0].batch_status(CHUNK) == 0
        print("flags 0x%08x contexts %d : %.3f ms per chunk, %.0f frames/s" % (flags, nctx, dt / n * 1e3, n * CHUNK / dt), flush=True)
