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
