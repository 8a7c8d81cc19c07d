#!/usr/bin/env python3
"""Would a HIP graph of the one-frame call help?  The batch entry with one resident 1080p / VGA frame, launched kernel by
kernel (as shipped) and replayed as a captured graph (torch.cuda.CUDAGraph around the same C-ABI call; the engine's side
stream joins the capture through its events).  Timing experiment: the engine does not ship a graph path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ethzasl_brisk_amd as B, synth

ctx = B.Context(0)
ext = B.BriskDescriptorExtractor(context=ctx)
for name, img, thr in (("1080p", synth.frame_1080p(0), 80), ("vga", synth.frame_vga(1), 70)):
    h, w = img.shape
    d = torch.from_numpy(img[None]).cuda()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        st = s.cuda_stream
        call = lambda: ctx.detect_describe_batch(ext, d.data_ptr(), 1, w, h, w * h, w, thr, 4, st)
        for _ in range(5): call()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): call()
        s.synchronize()
        t_stream = (time.perf_counter() - t0) / 200
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                call()
            for _ in range(5): g.replay()
            s.synchronize()
            t0 = time.perf_counter()
            for _ in range(200): g.replay()
            s.synchronize()
            t_graph = (time.perf_counter() - t0) / 200
            n = ctx.batch_status(1)
            print("%s: kernel by kernel %.1f us per call, captured graph %.1f us per call (status after replay: %s)" % (name, t_stream * 1e6, t_graph * 1e6, n))
        except Exception as e:
            print("%s: kernel by kernel %.1f us per call; capture failed: %s" % (name, t_stream * 1e6, str(e)[:300]))
