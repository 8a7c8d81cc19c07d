#!/usr/bin/env python3
"""Per-stage HIP-event times of ONE 1080p frame through the batch entry (where a single frame's latency goes), and the
host-buffer call latencies."""
import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import ethzasl_brisk_amd as B, synth
ctx = B.Context(0)
ext = B.BriskDescriptorExtractor(context=ctx)
img = synth.frame_1080p(0)
d = torch.from_numpy(img[None]).cuda()
st = torch.cuda.current_stream().cuda_stream
for n in (1,):
    for _ in range(5): ctx.detect_describe_batch(ext, d.data_ptr(), n, 1920, 1080, 1920*1080, 1920, 80, 4, st)
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    t0=time.perf_counter()
    for _ in range(50): ctx.detect_describe_batch(ext, d.data_ptr(), n, 1920, 1080, 1920*1080, 1920, 80, 4, st)
    torch.cuda.synchronize()
    dt=(time.perf_counter()-t0)/50
    ms,_=ctx.profile_read()
    print('device-resident 1 frame: %.3f ms per call (back to back)'%(dt*1e3), {k:round(v*1e3) for k,v in ms.items()}, 'sum us', round(sum(ms.values())*1e3))
    t0=time.perf_counter()
    for _ in range(50):
        ctx.detect_describe_batch(ext, d.data_ptr(), n, 1920, 1080, 1920*1080, 1920, 80, 4, st); torch.cuda.synchronize()
    print('with sync each: %.3f ms'%((time.perf_counter()-t0)/50*1e3))
det = B.BriskFeatureDetector(80, 4, context=ctx)
for _ in range(5): k = det.detect(img)
t0=time.perf_counter()
for _ in range(100): k = det.detect(img)
print('host detect: %.3f ms'%((time.perf_counter()-t0)/100*1e3))
t0=time.perf_counter()
for _ in range(100): k2,dd = ext.compute(img,k)
print('host describe: %.3f ms'%((time.perf_counter()-t0)/100*1e3))
