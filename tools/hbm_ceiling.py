#!/usr/bin/env python3
"""Measured HBM ceiling of the box next to the 8 TB/s peak used for roofline.frac: device-to-device copy and a
read-only reduction over buffers far larger than the caches."""
import torch
n = 1 << 30  # 4 GiB of fp32
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
for name, fn, nbytes in (("copy (read + write)", lambda: b.copy_(a), 2 * a.numel() * 4),
                         ("sum (read only)", lambda: a.sum(), a.numel() * 4)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("%s: %.2f TB/s" % (name, nbytes / ms / 1e9))
