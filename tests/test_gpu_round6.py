"""GPU tests added in round 6 (run with -m gpu): the batch path's exit to host memory - brisk_hip_batch_download_all /
_wait and the host-to-host batch entry (exact prefix-summed rows of ALL frames in one asynchronous transfer, written by the
device into pinned memory or through the context's bounce buffer into pageable memory) - against the per-frame download and
the oracle, with per-frame capacity flags, a destination that is too small, and transfers kept in flight over several
batches."""
import numpy as np
import pytest

import oracle_lib as O
import synth
from test_gpu_parity import same_kps, explain  # noqa: F401

pytestmark = pytest.mark.gpu

W, H = 640, 480
THR, OCT = 70, 4


@pytest.fixture(scope="module")
def B():
    import ethzasl_brisk_amd as B
    from ethzasl_brisk_amd import build
    build.build()
    B.load_library()
    return B


@pytest.fixture(scope="module")
def frames65():
    return np.stack([synth.frame_vga(9100 + i) for i in range(65)])


@pytest.fixture(scope="module")
def oracle65(frames65):
    X = O.Extractor()
    out = []
    for img in frames65:
        ko = O.detect(img, THR, OCT)
        out.append((ko,) + tuple(X.compute(img, ko)))
    return out


def _check_frame(res, f, want_k, want_d, strings=48):
    k, d = res.frame(f, strings)
    assert int(res.counts[f]) == len(want_k), (f, int(res.counts[f]), len(want_k))
    assert int(res.flags[f]) == 0, (f, int(res.flags[f]))
    assert same_kps(k, want_k), (f, explain(k, want_k))
    if want_d is not None:
        assert np.array_equal(d, want_d), f


@pytest.mark.parametrize("pinned", [True, False])
def test_download_all_equals_per_frame_download_and_oracle(B, frames65, oracle65, pinned):
    """65 distinct frames, device-resident batch: every frame's rows in the one transfer equal brisk_hip_batch_download of
    that frame and the oracle - described rows (which = 1) and detected keypoints (which = 0), destinations the device
    writes itself (pinned) and pageable ones (bounce buffer + host copy in the wait)."""
    import torch
    n = len(frames65)
    d = torch.from_numpy(frames65).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    ctx = B.Context(0)
    ext = B.BriskDescriptorExtractor(context=ctx)
    ctx.detect_describe_batch(ext, d.data_ptr(), n, W, H, W * H, W, THR, OCT, stream)
    total = sum(len(o[1]) for o in oracle65)
    res = B.HostResults(n + 3, total + 17, 48, pinned=pinned)       # (capacities beyond the need: nothing depends on a tight fit)
    res.kps.view(np.uint8)[:] = 0xEE
    res.desc[:] = 0xEE
    t1 = ctx.batch_download_all(res, described=True, stream=stream)
    det = B.HostResults(n, sum(len(o[0]) for o in oracle65), 0, pinned=pinned)
    t0 = ctx.batch_download_all(det, described=False, stream=stream)
    assert ctx.batch_download_wait(t1) == 0
    assert ctx.batch_download_wait(t0) == 0
    assert int(res.offsets[n]) == total and int(res.offsets[0]) == 0
    assert np.all(res.desc[total:] == 0xEE), "rows behind the stored ones were written"
    for f in range(n):
        ko, ko2, do = oracle65[f]
        _check_frame(res, f, ko2, do)
        _check_frame(det, f, ko, None)
        if f % 8 == 0:
            kg, dg = ctx.batch_download(f, described=True)
            k, dd = res.frame(f)
            assert same_kps(k, kg) and np.array_equal(dd, dg)
    assert ctx.batch_download_wait(t1) == 0                           # waiting twice: the same answer
    ctx.close()


def test_download_all_wide_rows_and_strided_prefix(B, frames65, oracle65):
    """descriptor rows at a stride beyond the descriptor (64-byte rows for 48-byte descriptors): the descriptor, then zeros"""
    import torch
    n = 9
    d = torch.from_numpy(frames65[:n]).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    ctx = B.Context(0)
    ext = B.BriskDescriptorExtractor(context=ctx)
    ctx.detect_describe_batch(ext, d.data_ptr(), n, W, H, W * H, W, THR, OCT, stream)
    total = sum(len(o[1]) for o in oracle65[:n])
    res = B.HostResults(n, total, 64, pinned=True)
    res.desc[:] = 0xEE
    assert ctx.batch_download_wait(ctx.batch_download_all(res, stream=stream)) == 0
    for f in range(n):
        _check_frame(res, f, oracle65[f][1], oracle65[f][2], strings=48)
    assert np.all(res.desc[:total, 48:] == 0)
    ctx.close()


def test_download_all_reports_capacity_per_frame(B, frames65, oracle65):
    """A keypoint capacity some frames exceed: exactly those frames are flagged (bit 2) and store no rows, the others are
    complete; the wait answers BRISK_HIP_ERR_CAPACITY with the count of flagged frames.  A destination with too few rows:
    the frames that fit are complete, the rest carry BRISK_HIP_ROWS_CUT and their true counts."""
    import torch
    n = 24
    nk = [len(o[0]) for o in oracle65[:n]]
    cap = sorted(nk)[n // 2]                                          # about half of the frames exceed it
    assert min(nk) < cap < max(nk)
    d = torch.from_numpy(frames65[:n]).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    ctx = B.Context(0, max_candidates=65536, max_keypoints=cap)
    ext = B.BriskDescriptorExtractor(context=ctx)
    ctx.detect_describe_batch(ext, d.data_ptr(), n, W, H, W * H, W, THR, OCT, stream)
    res = B.HostResults(n, sum(nk), 48, pinned=True)
    rc, flagged = ctx.batch_download_wait(ctx.batch_download_all(res, stream=stream), check=False)
    over = [f for f in range(n) if nk[f] > cap]
    assert rc == 4 and flagged == len(over) > 0, (rc, flagged, len(over))
    for f in range(n):
        if f in over:
            assert int(res.flags[f]) & 4 and res.offsets[f + 1] == res.offsets[f], (f, int(res.flags[f]))
        else:
            _check_frame(res, f, oracle65[f][1], oracle65[f][2])
    ctx.close()
    # rows_cap too small: a prefix of the frames is stored
    ctx = B.Context(0)
    ext = B.BriskDescriptorExtractor(context=ctx)
    ctx.detect_describe_batch(ext, d.data_ptr(), n, W, H, W * H, W, THR, OCT, stream)
    nd = [len(o[1]) for o in oracle65[:n]]
    fit = 10
    res = B.HostResults(n, sum(nd[:fit]) + nd[fit] - 1, 48, pinned=True)   # frame `fit` misses by one row
    rc, flagged = ctx.batch_download_wait(ctx.batch_download_all(res, stream=stream), check=False)
    assert rc == 4 and flagged == n - fit, (rc, flagged)
    assert "rows_cap" in ctx._L.brisk_hip_last_error(ctx._h).decode()
    assert int(res.offsets[n]) == sum(nd[:fit])
    for f in range(n):
        if f < fit:
            _check_frame(res, f, oracle65[f][1], oracle65[f][2])
        else:
            assert int(res.flags[f]) == B.ROWS_CUT and int(res.counts[f]) == nd[f] and res.offsets[f + 1] == res.offsets[f]
    ctx.close()


@pytest.mark.parametrize("pinned", [True, False])
def test_host_to_host_batches_in_flight(B, frames65, oracle65, pinned):
    """brisk_hip_detect_describe_batch_host_results: frames from pinned host memory, results back in host memory, two
    destination sets alternating over five batches of different content (the transfer of batch n runs beside batch n + 1; a
    third transfer completes the oldest by itself); every batch's every frame equals the oracle."""
    import torch
    n = 13
    orders = [np.roll(np.arange(65), -7 * b)[:n] for b in range(5)]
    srcs = [torch.from_numpy(np.ascontiguousarray(frames65[o])).pin_memory() for o in orders]
    ctx = B.Context(0)
    ext = B.BriskDescriptorExtractor(context=ctx)
    cap_rows = max(sum(len(oracle65[i][1]) for i in o) for o in orders)
    dsts = [B.HostResults(n, cap_rows, 48, pinned=pinned) for _ in range(3)]
    tickets = []
    for b in range(5):
        if b >= 2:                                                    # consume batch b - 2 before its destination is reused
            if b == 4:
                pass                                                  # (batch 2 is left to the engine: the third transfer completes it)
            else:
                assert ctx.batch_download_wait(tickets[b - 2]) == 0
                for f, i in enumerate(orders[b - 2]):
                    _check_frame(dsts[(b - 2) % 3], f, oracle65[i][1], oracle65[i][2])
        tickets.append(ctx.detect_describe_batch_host_results(ext, srcs[b].data_ptr(), n, W, H, W * H, W, THR, OCT, dsts[b % 3]))
    assert ctx.batch_download_wait(tickets[4]) == 0                   # completes 3 as well
    for b in (2, 3, 4):
        for f, i in enumerate(orders[b]):
            _check_frame(dsts[b % 3], f, oracle65[i][1], oracle65[i][2])
    rc, _ = ctx.batch_download_wait(tickets[0], check=False)
    assert rc == 1                                                    # retired long ago: unknown ticket
    rc, _ = ctx.batch_download_wait(12345, check=False)
    assert rc == 1
    ctx.close()


def test_download_all_argument_errors(B, frames65):
    import torch
    ctx = B.Context(0)
    ext = B.BriskDescriptorExtractor(context=ctx)
    res = B.HostResults(4, 4096, 48, pinned=False)
    with pytest.raises(B.BriskHipError) as e:
        ctx.batch_download_all(res)
    assert e.value.code == 1 and "no batch" in str(e.value)
    d = torch.from_numpy(frames65[:4]).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    ctx.detect_batch(d.data_ptr(), 4, W, H, W * H, W, THR, OCT, stream)
    with pytest.raises(B.BriskHipError) as e:
        ctx.batch_download_all(res, described=True, stream=stream)
    assert e.value.code == 1 and "described nothing" in str(e.value)
    ctx.detect_describe_batch(ext, d.data_ptr(), 4, W, H, W * H, W, THR, OCT, stream)
    for bad in (B.HostResults(3, 4096, 48, pinned=False), B.HostResults(4, 4096, 44, pinned=False), B.HostResults(4, 4096, 50, pinned=False)):
        with pytest.raises(B.BriskHipError) as e:
            ctx.batch_download_all(bad, stream=stream)
        assert e.value.code == 1
    assert ctx.batch_download_wait(ctx.batch_download_all(res, stream=stream)) == 0
    ctx.close()


def test_eight_contexts_at_once_against_the_oracle():
    """tools/soak.py threads, 32 iterations per thread = 256 cases: eight host threads with their own contexts - a one-frame 4K call
    (eight tie bands per layer), a 64-frame pair-form batch, dense frames at threshold 30, odd sizes, a single-layer pyramid, the
    host-to-host batch entry - on the chip together, every result of every iteration bit-equal to the oracle; a hang (a tie band
    waiting for a workgroup that never runs) kills the child process and fails the test."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak.py"), "threads", "32", "600"], capture_output=True, text=True)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-2000:]
    assert "256 cases" in r.stdout and ", 0 mismatches, 0 errors" in r.stdout


def test_compute_scale_points_that_share_cells(B):
    """ComputeScale's parallel form lets lanes race on cells of the score-state map (every writer stores the same value: a
    function of the image).  Lists built to collide - a 2.5-pixel grid over a textured region (neighbouring points share most
    of their 3 x 3 / 4 x 4 reads on every layer), every point four times (exact duplicates and sub-pixel neighbours), shuffled -
    against the oracle, i.e. the sequential algorithm; with and without scale NMS."""
    img = synth.frame_1080p(6)
    rng = np.random.default_rng(11)
    gx, gy = np.meshgrid(700 + 2.5 * np.arange(40), 400 + 2.5 * np.arange(40))
    base = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float32)
    pts = np.concatenate([base, base, base + np.float32(0.25), base + rng.uniform(-0.5, 0.5, base.shape).astype(np.float32)])
    pts = pts[rng.permutation(len(pts))]
    k = np.zeros(len(pts), B.KEYPOINT)
    k["x"], k["y"], k["size"], k["angle"], k["class_id"] = pts[:, 0], pts[:, 1], 12, -1, np.arange(len(pts))
    ctx = B.Context(0, max_candidates=262144, max_keypoints=65536)
    checked = 0
    for octaves, suppress in ((4, True), (2, True), (0, True)):
        want = O.compute_scale(img, k, 60, octaves, suppress)
        if want is None:
            continue
        got = B.BriskFeatureDetector(60, octaves, suppress, context=ctx).ComputeScale(img, k)
        assert len(want) > 1000 and same_kps(got, want), (octaves, suppress, explain(got, want))
        checked += 1
    assert checked >= 2
    ctx.close()


def test_reference_goldens_on_the_release_library():
    """libbrisk_hip_release.so (no BRISK_HIP_TUNING: no environment knobs, no debug bits, no brisk_hip_debug_* exports - the
    library INTEGRATION.md links) in a child process: both reference goldens bit-exact through the host calls, the batch path
    and brisk_hip_batch_download_all (tests/release_golden.py)."""
    import os
    import subprocess
    import sys
    from ethzasl_brisk_amd import build
    rel = build.build_release()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "release_golden.py")], env=dict(os.environ, BRISK_HIP_LIB=rel),
                       capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-3000:])
    assert r.returncode == 0 and "goldens OK" in r.stdout


def test_pool_combines_concurrent_calls_bit_exactly(B, frames65, oracle65):
    """brisk_hip_pool: twelve host threads call detect() then describe() on their own frames at once - two image sizes and two
    thresholds (calls of different kinds must not mix in a group), describe() with the detect call's token (device copy of the
    frame), with a token that names no retained frame and without one; every result equals the oracle, and a lone caller (a group of one) does too."""
    import threading
    pool = B.Pool(0, max_batch=8)
    ext = B.BriskDescriptorExtractor()
    small = [synth.gen(333, 201, 7700 + i, 40) for i in range(4)]
    X = O.Extractor()
    want_small = []
    for img in small:
        ko = O.detect(img, 60, 2)
        want_small.append((ko,) + tuple(X.compute(img, ko)))
    k, tok = pool.detect(frames65[0], THR, OCT)                          # alone: a group of one
    k2, d = pool.describe(ext, frames65[0], k, tok)
    assert same_kps(k, oracle65[0][0]) and same_kps(k2, oracle65[0][1]) and np.array_equal(d, oracle65[0][2])
    errors = []

    def worker(t):
        try:
            for it in range(12):
                if t % 3 == 2:                                           # another size / threshold / octave count
                    j = (t + it) % len(small)
                    img, (wk, wk2, wd), thr, octv = small[j], want_small[j], 60, 2
                else:
                    j = (5 * t + it) % 65
                    img, (wk, wk2, wd), thr, octv = frames65[j], oracle65[j], THR, OCT
                k, tok = pool.detect(img, thr, octv)
                # the detect call's token / a token no detect call ever returned (the frame is then uploaded) / none
                use = tok if it % 3 == 0 else ((tok ^ (0x5A5A << 16)) if it % 3 == 1 else 0)
                k2, d = pool.describe(ext, img, k, use)
                if not (same_kps(k, wk) and same_kps(k2, wk2) and np.array_equal(d, wd)):
                    errors.append((t, it, img.shape, "detected %s" % same_kps(k, wk), "described %s" % same_kps(k2, wk2), hex(use)))
        except Exception as e:
            errors.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(12)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:5]
    groups, calls = pool.stats()
    assert calls == 2 + 12 * 12 * 2 and groups <= calls
    pool.close()
    ext.close()


def test_pool_reports_capacity_per_caller(B, frames65, oracle65):
    """a pool whose per-frame keypoint capacity some frames exceed: exactly those callers get BRISK_HIP_ERR_CAPACITY (the
    classes then use the thread's own context), the other members of the same groups get their results"""
    import threading
    nk = [len(o[0]) for o in oracle65[:16]]
    cap = sorted(nk)[8]
    pool = B.Pool(0, max_batch=16, max_keypoints=cap)
    out = {}

    def worker(t):
        try:
            out[t] = ("ok", pool.detect(frames65[t], THR, OCT)[0])
        except B.BriskHipError as e:
            out[t] = ("err", e.code)

    for rep in range(3):
        th = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        for t in range(16):
            if nk[t] > cap:
                assert out[t] == ("err", 4), (t, out[t])
            else:
                assert out[t][0] == "ok" and same_kps(out[t][1], oracle65[t][0]), t
    pool.close()


def test_drop_in_classes_through_the_pool():
    """tests/cpp/test_threads with every eligible call pooled (--pool-threshold 1): eight threads with different thresholds,
    octaves and post-filters (those with a uniformity radius stay on their own contexts), bit-equal to the serial run; and the
    timed mode with ScopedSameImage, which presents the detect call's token to compute()."""
    import os
    import subprocess
    from test_cpp_classes import build_binary
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    b = build_binary("test_threads")
    r = subprocess.run([b, os.path.join(root, "tests", "golden"), "8", "12", "--pool-threshold", "1"], capture_output=True, text=True, timeout=600)
    print(r.stdout[-1500:], r.stderr[-1500:])
    assert r.returncode == 0 and "threads OK" in r.stdout
    r = subprocess.run([b, "--time", "8", "1", "640", "480", "--same-image", "--pool-threshold", "2"], capture_output=True, text=True, timeout=600)
    print(r.stdout[-1500:], r.stderr[-1500:])
    assert r.returncode == 0 and '"mismatches": 0' in r.stdout



def test_detect_images_and_describe_images_against_the_oracle(B, frames65, oracle65):
    """brisk_hip_detect_images / brisk_hip_describe_images (what the classes' vector overloads forward to): 70 separately
    allocated frames in one call each (two slices of the host-fed path for detection; one descriptor-only launch over all
    frames with every frame's own provided keypoint list, one of them empty), pageable destinations, against the oracle"""
    ctx = B.Context(0)
    ext = B.BriskDescriptorExtractor(context=ctx)
    idx = [(7 * i) % 65 for i in range(70)]
    imgs = [frames65[j].copy() for j in idx]
    det = B.HostResults(len(imgs), sum(len(oracle65[j][0]) for j in idx), 0, pinned=False)
    assert ctx.batch_download_wait(ctx.detect_images(imgs, THR, OCT, det)) == 0
    for f, j in enumerate(idx):
        _check_frame(det, f, oracle65[j][0], None)
    lists = [oracle65[j][0] for j in idx]
    lists[3] = lists[3][:0]                                          # an image without keypoints
    res = B.HostResults(len(imgs), sum(len(k) for k in lists), 48, pinned=False)
    assert ctx.batch_download_wait(ctx.describe_images(ext, imgs, lists, res)) == 0
    for f, j in enumerate(idx):
        if f == 3:
            assert int(res.counts[f]) == 0 and res.offsets[f + 1] == res.offsets[f]
        else:
            _check_frame(res, f, oracle65[j][1], oracle65[j][2])
    # the same list right after detect_images, stated unchanged: the frames are taken from their device copies; a list that is
    # not the detect call's list is uploaded whatever the caller says
    assert ctx.batch_download_wait(ctx.detect_images(imgs, THR, OCT, det)) == 0
    reuse0 = ctx.debug_image_reuse()
    res.desc[:] = 0
    assert ctx.batch_download_wait(ctx.describe_images(ext, imgs, lists, res, same_images=True)) == 0
    assert ctx.debug_image_reuse() == reuse0 + 1
    for f, j in enumerate(idx):
        if f != 3:
            _check_frame(res, f, oracle65[j][1], oracle65[j][2])
    other = [frames65[(j + 1) % 65].copy() for j in idx]
    lists2 = [oracle65[(j + 1) % 65][0] for j in idx]
    assert ctx.batch_download_wait(ctx.describe_images(ext, other, lists2, res, same_images=True)) == 0
    assert ctx.debug_image_reuse() == reuse0 + 1
    for f, j in enumerate(idx):
        _check_frame(res, f, oracle65[(j + 1) % 65][1], oracle65[(j + 1) % 65][2])
    ctx.close()


@pytest.mark.parametrize("w,h", [(17, 3), (9, 9), (70, 2), (333, 201)])
def test_multi_image_calls_on_small_and_odd_sizes(B, w, h):
    """brisk_hip_detect_images / _describe_images on images of a few rows (the staged copy deals quarter frames to its helper
    threads) and on an odd size, against the oracle (tiny images have no keypoints: counts 0, no rows)"""
    rng = np.random.default_rng(w * 1000 + h)
    imgs = [synth.gen(w, h, 31 + i, 8) if (w >= 70 and h >= 70) else rng.integers(0, 255, (h, w), dtype=np.uint8) for i in range(5)]
    ctx = B.Context(0)
    ext = B.BriskDescriptorExtractor(context=ctx)
    X = O.Extractor()
    want = []
    for img in imgs:
        k = O.detect(img, 40, 2)
        want.append((k,) + tuple(X.compute(img, k)))
    det = B.HostResults(len(imgs), sum(len(x[0]) for x in want) + 1, 0, pinned=False)
    assert ctx.batch_download_wait(ctx.detect_images(imgs, 40, 2, det)) == 0
    res = B.HostResults(len(imgs), sum(len(x[0]) for x in want) + 1, 48, pinned=False)
    assert ctx.batch_download_wait(ctx.describe_images(ext, imgs, [x[0] for x in want], res, same_images=True)) == 0
    for f, x in enumerate(want):
        _check_frame(det, f, x[0], None)
        _check_frame(res, f, x[1], x[2])
    ctx.close()
