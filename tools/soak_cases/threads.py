#!/usr/bin/env python3
"""Multi-context stress (GPU box): eight host threads, each with its OWN context and its own kind of work - sizes, thresholds,
octave counts, one-frame host calls, batches, the host-to-host batch entry - all on the chip at once, every result of every
iteration compared bit-exactly with the oracle (computed once per distinct input, before the GPU is touched).

Why: the tie stage's row bands and pairs wait across workgroups through global memory; their forward-progress argument
("a workgroup only waits for tickets drawn earlier") must also hold with foreign workgroups of other contexts on the chip.

usage: python3 tools/soak.py threads [iterations per thread = 200] [watchdog seconds = 900]
The parent process only supervises: the work runs in a fresh CHILD process, which the watchdog kills (by pid) on a hang;
exit code 0 = no mismatch and no hang."""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

# (name, width, height, rects, threshold, octaves, frames per call, kind, distinct inputs)
# kind: "host" = brisk_hip_detect + brisk_hip_describe per frame (the drop-in classes' calls), "batch" = device-resident batch +
# per-frame download, "h2h" = frames from pinned host memory, results through brisk_hip_batch_download_all
WORK = [
    ("4k_one_frame_8_bands", 3840, 2160, 1200, 80, 6, 1, "host", 2),
    ("vga_64_frame_pair_batch", 640, 480, 60, 70, 4, 64, "batch", 1),
    ("1080p_dense_thr30", 1920, 1080, 300, 30, 4, 2, "batch", 1),
    ("1080p_one_frame", 1920, 1080, 300, 80, 4, 1, "host", 3),
    ("odd_333x201", 333, 201, 40, 60, 2, 8, "batch", 1),
    ("1281x723_three_frames", 1281, 723, 200, 55, 4, 3, "batch", 2),
    ("800x600_single_layer", 800, 600, 120, 100, 0, 1, "host", 3),
    ("vga_16_frames_host_to_host", 640, 480, 60, 70, 4, 16, "h2h", 2),
]


def frames_of(wi, j):
    import synth
    name, w, h, nrect, thr, octv, nf, kind, nd = WORK[wi]
    return np.stack([synth.gen(w, h, 50000 + 1000 * wi + 100 * j + f, nrect) for f in range(nf)])


def oracle_job(args):
    import oracle_lib as O
    wi, j = args
    name, w, h, nrect, thr, octv, nf, kind, nd = WORK[wi]
    X = O.Extractor()
    out = []
    for img in frames_of(wi, j):
        k = O.detect(img, thr, octv)
        k2, d = X.compute(img, k)
        out.append((k.tobytes(), k2.tobytes(), d.tobytes()))
    return wi, j, out


def child(iters):
    from concurrent.futures import ProcessPoolExecutor
    import bench
    jobs = [(wi, j) for wi in range(len(WORK)) for j in range(WORK[wi][8])]
    with ProcessPoolExecutor(min(bench.usable_cores(), len(jobs))) as ex:   # forked before torch / HIP are loaded
        want = {(wi, j): out for wi, j, out in ex.map(oracle_job, jobs)}
    import torch
    import ethzasl_brisk_amd as B
    rev = B.load_library().brisk_hip_kernel_revision().decode()
    bad = [0] * len(WORK)
    done = [0] * len(WORK)
    errors = []

    def worker(wi):
        name, w, h, nrect, thr, octv, nf, kind, nd = WORK[wi]
        try:
            ctx = B.Context(0, max_candidates=262144, max_keypoints=65536)
            ext = B.BriskDescriptorExtractor(context=ctx)
            det = B.BriskFeatureDetector(thr, octv, context=ctx)
            inputs = [frames_of(wi, j) for j in range(nd)]
            stream = torch.cuda.Stream()
            dev = [torch.from_numpy(a).cuda() for a in inputs] if kind == "batch" else None
            pin = [torch.from_numpy(a).pin_memory() for a in inputs] if kind == "h2h" else None
            dst = [B.HostResults(nf, nf * 4096, 48, pinned=(q == 0)) for q in range(2)] if kind == "h2h" else None
            torch.cuda.synchronize()
            for it in range(iters):
                j = it % nd
                got = []
                if kind == "host":
                    for img in inputs[j]:
                        k = det.detect(img, capacity=65536)
                        k2, d = ext.compute(img, k)
                        got.append((k.tobytes(), np.ascontiguousarray(k2).tobytes(), np.ascontiguousarray(d).tobytes()))
                elif kind == "batch":
                    ctx.detect_describe_batch(ext, dev[j].data_ptr(), nf, w, h, w * h, w, thr, octv, stream.cuda_stream)
                    assert ctx.batch_status(nf) == 0
                    for f in range(nf):
                        kd, _ = ctx.batch_download(f, described=False)
                        kg, dg = ctx.batch_download(f, described=True)
                        got.append((kd.tobytes(), kg.tobytes(), dg.tobytes()))
                else:
                    r = dst[it & 1]
                    t = ctx.detect_describe_batch_host_results(ext, pin[j].data_ptr(), nf, w, h, w * h, w, thr, octv, r)
                    assert ctx.batch_download_wait(t) == 0
                    for f in range(nf):
                        kd, _ = ctx.batch_download(f, described=False)
                        kg, dg = r.frame(f, 48)
                        got.append((kd.tobytes(), np.ascontiguousarray(kg).tobytes(), np.ascontiguousarray(dg).tobytes()))
                if got != want[(wi, j)]:
                    bad[wi] += 1
                    if bad[wi] <= 3:
                        print("MISMATCH", name, "iteration", it, "input", j, flush=True)
                done[wi] = it + 1
            ctx.close()
        except Exception as e:  # a thread that dies must fail the suite
            errors.append((name, repr(e)))
            print("ERROR", name, repr(e), flush=True)

    t0 = time.time()
    th = [threading.Thread(target=worker, args=(wi,)) for wi in range(len(WORK))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.time() - t0
    for wi, wk in enumerate(WORK):
        print("  %-28s %4d iterations x %2d frame(s), %d mismatches" % (wk[0], done[wi], wk[6], bad[wi]))
    total = sum(done)
    print("threads: kernel revision %s, %d threads, %d cases (%d frames), %d mismatches, %d errors, %.1f s"
          % (rev, len(WORK), total, sum(done[wi] * WORK[wi][6] for wi in range(len(WORK))), sum(bad), len(errors), dt), flush=True)
    return 1 if (sum(bad) or errors or total != iters * len(WORK)) else 0


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        sys.exit(child(int(sys.argv[2])))
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    limit = float(sys.argv[2]) if len(sys.argv) > 2 else 900.0
    # a fresh child does the work; this process never touches the GPU and only kills the child it started
    p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(iters)])
    try:
        rc = p.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        p.kill()
        p.wait()
        print("threads: HANG - the child did not finish within %.0f s and was killed" % limit, flush=True)
        rc = 3
    sys.exit(rc)


if __name__ == "__main__":
    main()
