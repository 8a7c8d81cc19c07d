"""GPU tests added in round 4 (run with -m gpu): the bench's own launch shape (exactly 256 frames), 4K batches with deep
pyramids (persistent tie workgroups, rotating detector tile eighths), the multi-workgroup keypoint preparation
(k_dp_count / k_dp_scan / k_dp_scatter) inside batches and on provided keypoints of every count around its thresholds, the
hashed-bin uniformity filter on dense clusters, and BASELINE config 3 through the engine's own C-ABI communicator."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import synth
from test_gpu_parity import same_kps, explain  # noqa: F401
from test_gpu_round3 import _run_batch_and_compare

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    import ethzasl_brisk_amd as B
    from ethzasl_brisk_amd import build
    build.build()
    B.load_library()
    return B


def test_the_bench_launch_shape_256_frames(B):
    """exactly 256 frames of 1080p per launch (bench.py's chunk: one tie workgroup per frame, 16 / 32 blocks per frame in
    the score-block kernels, eight describe queues of 32 frames), every slot against the oracle"""
    distinct = [synth.frame_1080p(900 + s) for s in range(8)]
    _run_batch_and_compare(B, distinct, 256, 80, 4, 1920, 1080, every_slot=True)


def test_4k_six_octaves_batch_of_24(B):
    """3840 x 2160, 12 layers: 3960 detector tiles (a multiple of 8: the tile eighths only rotate because the kernel makes
    them), tie tickets of several layers per workgroup, more than 4096 keypoints per frame (multi-workgroup preparation
    with the spatial order); three distinct frames, first / middle / last slots against the oracle"""
    distinct = [synth.frame_4k(40 + s) for s in range(3)]
    _run_batch_and_compare(B, distinct, 24, 80, 6, 3840, 2160, every_slot=False)


@pytest.mark.parametrize("n", [2047, 2048, 2049, 3000, 4097, 20000])
def test_provided_keypoints_around_the_preparation_thresholds(B, n):
    """descriptor-only calls with n provided keypoints: up to 2048 the one-workgroup preparation, above it the
    multi-workgroup one (1024 inputs per block: 2049 = three blocks, the last with one input); sizes log-uniform, a third
    of the keypoints with a provided angle, many outside the border; filtered list, angles and descriptors against the oracle"""
    img = synth.frame_1080p(77)
    rng = np.random.default_rng(n)
    kp = np.zeros(n, B.KEYPOINT)
    kp["size"] = np.exp(rng.uniform(np.log(8.64), np.log(120.0), n)).astype(np.float32)
    kp["x"] = rng.uniform(-5, 1925, n).astype(np.float32)
    kp["y"] = rng.uniform(-5, 1085, n).astype(np.float32)
    kp["angle"] = np.where(rng.uniform(size=n) < 0.33, rng.uniform(0, 360, n), -1).astype(np.float32)
    kp["response"] = rng.uniform(1, 100, n).astype(np.float32)
    kp["octave"] = rng.integers(0, 8, n)
    kp["class_id"] = np.arange(n)
    ctx = B.Context(0, max_candidates=65536, max_keypoints=32768)
    ext = B.BriskDescriptorExtractor(context=ctx)
    k2, d2 = ext.compute(img, kp)
    ko2, do = O.Extractor().compute(img, kp)
    assert len(k2) == len(ko2) and 0 < len(k2) < n
    assert same_kps(k2, ko2), explain(k2, ko2)
    assert np.array_equal(d2, do)
    # packed rows (the caller's pitch = descriptor size) and a padded caller pitch give the same rows
    kbuf = kp.copy()
    dbuf = np.zeros((n, 80), np.uint8)
    nio = C.c_int(n)
    ctx.check(ctx._L.brisk_hip_describe(ctx._h, ext._h, img.ctypes.data_as(C.c_void_p), 1920, 1080, 1920, kbuf.ctypes.data_as(C.c_void_p),
                                        C.byref(nio), dbuf.ctypes.data_as(C.c_void_p), 80, 1, 1))
    assert nio.value == len(ko2) and np.array_equal(dbuf[:nio.value, :48], do) and not dbuf[:, 48:].any()
    ctx.close()


@pytest.mark.parametrize("force32", [False, True])
def test_descriptors_on_the_24_bit_and_the_32_bit_integral_image(B, force32):
    """k_describe reads the integral image modulo 2^24 from 3-byte elements (gathers at byte offsets 3 x: any alignment) where
    the pattern's boxes allow it, and from u32 elements otherwise (debug bit 18 forces that form): keypoints of every scale
    index incl. the largest boxes (scale 63), widths whose 3-byte rows end at every alignment, both patterns."""
    ctx = B.Context(0, max_candidates=65536, max_keypoints=32768)
    ctx.debug_set_flags((1 << 18) if force32 else (1 << 24))   # (a descriptor-only call takes the 32-bit form by itself)
    rng = np.random.default_rng(24)
    for (w, h), version in (((1920, 1080), 2), ((1001, 587), 2), ((1283, 722), 1), ((641, 481), 2)):
        img = synth.gen(w, h, w + h, max(8, w * h // 7000))
        n = 6000
        kp = np.zeros(n, B.KEYPOINT)
        kp["size"] = np.exp(rng.uniform(np.log(8.64), np.log(260.0), n)).astype(np.float32)
        kp["x"] = rng.uniform(0, w, n).astype(np.float32)
        kp["y"] = rng.uniform(0, h, n).astype(np.float32)
        kp["angle"] = np.where(rng.uniform(size=n) < 0.2, rng.uniform(0, 360, n), -1).astype(np.float32)
        ext = B.BriskDescriptorExtractor(version=version, context=ctx)
        k2, d2 = ext.compute(img, kp)
        assert ctx.debug_integral_bits() == (32 if force32 else 24)
        ko2, do = O.Extractor(version=version).compute(img, kp)
        assert len(ko2) > 300 and same_kps(k2, ko2), (w, h, version, explain(k2, ko2))
        assert np.array_equal(d2, do), (w, h, version)
    ctx.close()


def test_integral_format_follows_the_candidate_density_of_the_previous_batch(B):
    """detect + describe batches keep the integral image in 3-byte elements while the context's previous batch was sparse
    (at most 3 000 AGAST candidates per megapixel) and as u32 after a dense one; either way every slot equals the oracle."""
    import torch
    imgs = [synth.frame_1080p(820 + i) for i in range(4)]
    d = torch.from_numpy(np.stack(imgs)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    ctx = B.Context(0, max_candidates=262144, max_keypoints=65536)
    ext = B.BriskDescriptorExtractor(context=ctx)
    X = O.Extractor()
    want = {}
    seen = []
    for thr in (80, 80, 30, 30, 30, 80, 80):
        ctx.detect_describe_batch(ext, d.data_ptr(), 4, 1920, 1080, 1920 * 1080, 1920, thr, 4, stream)
        torch.cuda.synchronize()
        assert ctx.batch_status(4) == 0
        seen.append(ctx.debug_integral_bits(0))
        if thr not in want:
            ko = O.detect(imgs[1], thr, 4)
            want[thr] = (ko,) + X.compute(imgs[1], ko)
        kd, _ = ctx.batch_download(1, described=False)
        kg, dg = ctx.batch_download(1, described=True)
        assert same_kps(kd, want[thr][0]) and same_kps(kg, want[thr][1]) and np.array_equal(dg, want[thr][2]), (thr, seen)
    # sparse, sparse, first dense batch still on the sparse format, then u32, and back after a sparse batch has been seen
    assert seen == [24, 24, 24, 32, 32, 32, 24], seen
    ctx.close()


def test_uniformity_hashed_bins_dense_clusters(B):
    """k_uf_rank / k_uf_decide: 6 000 synthetic points in tight clusters (hundreds of points within reach of each other:
    long chains of decisions inside one wave), scores with many ties, several radii incl. the smallest (cells of 15 per
    pixel: bin coordinates beyond 16 bits at 4K) - the filter alone, through a detector whose keypoints are replaced.
    Compared with the oracle's literal occupancy-image walk."""
    rng = np.random.default_rng(5)
    img = synth.frame_1080p(3)
    ko = O.detect(img, 50, 4)
    assert 5000 < len(ko) <= 8192
    # clustered variants of the detected list: coordinates pulled towards 40 centres, scores quantised (ties)
    cx, cy = rng.uniform(100, 1820, 40), rng.uniform(100, 980, 40)
    which = rng.integers(0, 40, len(ko))
    kc = ko.copy()
    kc["x"] = (cx[which] + rng.normal(0, 14, len(ko))).astype(np.float32).clip(20, 1900)
    kc["y"] = (cy[which] + rng.normal(0, 14, len(ko))).astype(np.float32).clip(20, 1060)
    kc["response"] = np.round(kc["response"] / 4) * 4 + 1
    ctx = B.Context(0, max_candidates=262144, max_keypoints=65536)
    L = ctx._L
    import torch
    k4 = kc.copy()   # the same clusters on a 4K canvas: at radius 1 the cell columns run to 57 616 + 16 (beyond 16 bits for x > 4368 px / 15)
    k4["x"] *= 2
    k4["y"] *= 2
    for pts, h, w, cases in ((ko, 1080, 1920, ((1.0, 0x7FFFFFFF), (2.5, 0x7FFFFFFF), (9.0, 0x7FFFFFFF), (30.0, 0x7FFFFFFF), (6.0, 500))),
                             (kc, 1080, 1920, ((1.0, 0x7FFFFFFF), (2.5, 0x7FFFFFFF), (9.0, 0x7FFFFFFF), (30.0, 0x7FFFFFFF), (6.0, 500))),
                             (k4, 2160, 3840, ((1.0, 0x7FFFFFFF), (1.7, 0x7FFFFFFF)))):
        for radius, budget in cases:
            want = O.enforce_uniformity(pts, h, w, radius, budget)
            got = _filter_only(B, ctx, pts, h, w, radius, budget)
            assert same_kps(got, want), (radius, budget, len(got), len(want), explain(got, want))
    ctx.close()


def _filter_only(B, ctx, pts, h, w, radius, budget):
    """runs the engine's uniformity filter on a GIVEN keypoint list: a detect batch on a flat image (no detections), the
    list written into the detector's output buffer, then the filter kernels through the debug entry"""
    L = ctx._L
    if not hasattr(L, "brisk_hip_debug_filter_keypoints"):
        pytest.skip("debug entry missing")
    out = np.zeros(len(pts), B.KEYPOINT)
    n = C.c_int(len(pts))
    buf = np.ascontiguousarray(pts).copy()
    ctx.check(L.brisk_hip_debug_filter_keypoints(ctx._h, buf.ctypes.data_as(C.c_void_p), len(pts), h, w, C.c_double(radius), budget,
                                                 out.ctypes.data_as(C.c_void_p), C.byref(n)))
    return out[:n.value]


def test_config3_512_frames_through_the_c_abi_gather_on_one_rank(B):
    """BASELINE config 3 on a single GPU through the engine's own communicator (brisk_hip_comm_*: RCCL directly, no
    torch.distributed): one batch of 512 x 1080p frames, gathered twice (both send slabs), a sample of frames checked
    bit-exactly against the oracle through the GATHERED buffers, every slot's count against its distinct frame's."""
    import torch
    import bench
    from test_gpu_round2 import _expected
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    n, w, h, nd = 512, 1920, 1080, 8
    distinct = [synth.frame_1080p(700 + i) for i in range(nd)]
    ring = torch.from_numpy(np.stack(distinct)).to(dev)
    frames = ring[torch.arange(n, device=dev) % nd].contiguous()
    ctx = B.Context(0)
    ext = B.BriskDescriptorExtractor(context=ctx)
    stream = torch.cuda.current_stream().cuda_stream
    ctx.detect_describe_batch(ext, frames.data_ptr(), n, w, h, w * h, w, 80, 4, stream)
    torch.cuda.synchronize()
    assert ctx.batch_status(n) == 0
    g = bench.CapiGather(ctx, n, ext.descriptorSize(), dev, 0, 1, n, None, stream)
    g.check_kpad(None)
    oext = O.Extractor()
    exp = {}
    for rep in range(2):
        if rep:
            ctx.detect_describe_batch(ext, frames.data_ptr(), n, w, h, w * h, w, 80, 4, stream)
        g.run()
        g.finish()
        ac, gk, gd = g.last
        assert tuple(ac.shape) == (1, n)
        for f in (0, 1, 7, 8, 255, 256, 300, 511):
            if f % nd not in exp:
                exp[f % nd] = _expected(distinct[f % nd], 80, 4, oext)[1]
            ko2, do = exp[f % nd]
            cnt = int(ac[0, f].item())
            assert cnt == len(ko2), (f, cnt, len(ko2))
            got_k = gk[0][f, :cnt].cpu().numpy()
            want_k = np.stack([ko2[name].view(np.uint32) for name in ko2.dtype.names], 1)
            assert np.array_equal(got_k.view(np.uint32), want_k), f
            assert np.array_equal(gd[0][f, :cnt].cpu().numpy(), do), f
        cnts = ac[0].cpu().numpy()
        assert all(cnts[f] == cnts[f % nd] for f in range(n))
    g.close()   # (brisk_hip_comm_destroy)
    ctx.close()


@pytest.mark.gpu
def test_one_frame_results_beyond_the_pinned_buffer(B):
    """The results of a one-frame host call are published into pinned memory by a kernel (download_single); what does not
    fit takes the staged copies.  Debug bit 25 shrinks the buffer to 16 KB: both paths must return the same bytes."""
    import synth
    img = synth.frame_1080p(3)
    ctx = B.Context(0)
    det = B.BriskFeatureDetector(80, 4, context=ctx)
    ext = B.BriskDescriptorExtractor(context=ctx)
    k0 = det.detect(img)
    k0d, d0 = ext.compute(img, k0)
    assert len(k0) > 16384 // 28 and len(k0d) > 16384 // 76   # (more than the small buffer holds)
    ctx.debug_set_flags(1 << 25)
    try:
        k1 = det.detect(img)
        k1d, d1 = ext.compute(img, k1)
        # a few keypoints: fits the small buffer too (the publishing path with the flag set)
        k2d, d2 = ext.compute(img, k0[:100])
    finally:
        ctx.debug_set_flags(0)
    k3d, d3 = ext.compute(img, k0[:100])
    assert k0.tobytes() == k1.tobytes()
    assert k0d.tobytes() == k1d.tobytes() and np.array_equal(d0, d1)
    assert k2d.tobytes() == k3d.tobytes() and np.array_equal(d2, d3)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [8, 31, 32])
def test_1080p_batches_around_the_small_tie_kernel_threshold(B, n):
    """batches below 32 frames run k_tie_resolve_small (precomputed act / not-self masks in the static step of the cache
    replay; per-XCD tickets from 8 frames on), 32 and more the plain form: every slot against the oracle, twice (dirty
    workspace)"""
    distinct = [synth.frame_1080p(500 + s) for s in range(4)]
    total = _run_batch_and_compare(B, distinct, n, 80, 4, 1920, 1080)
    assert total > 3000


@pytest.mark.gpu
def test_bench_default_launch_shape_512_frames(B):
    """the bench's default chunk since the end of round 4: 512 1080p frames in one call (two rounds of one 12-wave tie
    workgroup per frame, more than DS_MAXQ frames per k_describe launch): every slot against the oracle, on a fresh and
    on a dirty workspace"""
    distinct = [synth.frame_1080p(700 + s) for s in range(4)]
    total = _run_batch_and_compare(B, distinct, 512, 80, 4, 1920, 1080)
    assert total > 3000
