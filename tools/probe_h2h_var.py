import os, sys, time, json
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import ethzasl_brisk_amd as B, synth
W, H = 1920, 1080
var = sys.argv[1]
host = np.stack([synth.frame_1080p(100 + i) for i in range(8)])
dev = torch.device("cuda", 0)
if var == "stream":
    ws = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ws)
ctx = B.Context(0)
ext = B.BriskDescriptorExtractor(context=ctx)
n = 256
if var in ("resident", "resident512", "heavy", "heavy_ws"):
    m = 256 if var == "resident" else 512
    fr = torch.from_numpy(host).to(dev)[torch.arange(m, device=dev) % 8].contiguous()
    if var == "heavy_ws":
        ws = torch.cuda.Stream(device=dev); torch.cuda.set_stream(ws)
    for _ in range(320 if var.startswith("heavy") else 3):
        ctx.detect_describe_batch(ext, fr.data_ptr(), m, W, H, W * H, W, 80, 4, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
if var == "prof":
    ctx.profile_enable(True)
    fr = torch.from_numpy(host).to(dev)
    ctx.detect_describe_batch(ext, fr.data_ptr(), 8, W, H, W * H, W, 80, 4, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); ctx.profile_read(); ctx.profile_enable(False)
if var == "ceiling":
    ctx.stream_ceiling(1 << 30)
src = torch.from_numpy(np.ascontiguousarray(host[np.arange(n) % len(host)])).pin_memory()
def h2d():
    ctx.detect_describe_batch_host(ext, src.data_ptr(), n, W, H, W * H, W, 80, 4)
h2d(); torch.cuda.synchronize()
t0 = time.perf_counter(); reps = 0
while time.perf_counter() - t0 < 1.5:
    h2d(); reps += 1
torch.cuda.synchronize()
fps0 = reps * n / (time.perf_counter() - t0)
rows = n * 1400 if var != "rows" else int(1.25 * n * 990) + 4096
dsts = [B.HostResults(n, rows, 48, pinned=True) for _ in range(2)]
tk = [0, 0]
def issue(i):
    if tk[i & 1]:
        ctx.batch_download_wait(tk[i & 1])
    tk[i & 1] = ctx.detect_describe_batch_host_results(ext, src.data_ptr(), n, W, H, W * H, W, 80, 4, dsts[i & 1])
for i in range(3):
    issue(i)
t0 = time.perf_counter(); reps = 0
while time.perf_counter() - t0 < 1.5:
    issue(reps + 3); reps += 1
for i in (0, 1):
    ctx.batch_download_wait(tk[i])
fps1 = reps * n / (time.perf_counter() - t0)
print(var, round(fps0), round(fps1), round(fps1 / fps0, 4), flush=True)
