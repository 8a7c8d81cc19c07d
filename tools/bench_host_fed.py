"""Host-fed rate of brisk_hip_detect_describe_batch_host on pinned 1080p frames, beside the bare pinned H2D copy rate of
the same bytes (the ceiling of that entry point).  usage: [BRISK_HOST_SLICE=n] python3 tools/bench_host_fed.py [frames per call]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import ethzasl_brisk_amd as B, synth
ctx = B.Context(0)
ext = B.BriskDescriptorExtractor(context=ctx)
host = np.stack([synth.frame_1080p(i) for i in range(16)])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
src = torch.from_numpy(np.ascontiguousarray(host[np.arange(n) % 16])).pin_memory()
ctx.detect_describe_batch_host(ext, src.data_ptr(), n, 1920, 1080, 1920*1080, 1920, 80, 4)
torch.cuda.synchronize()
t0 = time.perf_counter(); reps = 0
while time.perf_counter() - t0 < 2.0:
    ctx.detect_describe_batch_host(ext, src.data_ptr(), n, 1920, 1080, 1920*1080, 1920, 80, 4); reps += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('slice', os.environ.get('BRISK_HOST_SLICE'), 'frames/call', n, 'fps', round(reps*n/dt,1), 'GB/s', round(reps*n*1920*1080/dt/1e9,1))
# pure H2D rate for reference
dst = torch.empty_like(src, device='cuda')
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(10): dst.copy_(src, non_blocking=True)
torch.cuda.synchronize(); dt=time.perf_counter()-t0
print('plain pinned H2D GB/s', round(10*src.numel()/dt/1e9,1))
