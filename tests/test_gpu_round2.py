"""GPU tests added in round 2 (run with -m gpu): host-fed batch entry, BASELINE config 3 (512 frames + the RCCL
result gather) on one rank, ordering of calls that share the workspace but use different streams, the streaming probe."""
import os
import socket

import numpy as np
import pytest

import oracle_lib as O
import synth
from test_gpu_parity import same_kps, explain  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    import ethzasl_brisk_amd as B
    from ethzasl_brisk_amd import build
    build.build()
    B.load_library()
    return B


def _expected(img, thr, octaves, ext):
    k = O.detect(img, thr, octaves)
    return k, ext.compute(img, k)


def test_host_fed_batch_equals_oracle_and_device_batch(B):
    """brisk_hip_detect_describe_batch_host: frames in pinned host memory, sliced H2D copies overlapped with compute.
    70 VGA frames (two slices of 64 + 6), every frame bit-equal to the oracle; then a padded layout (row pitch > width,
    frame pitch > rows) that takes the per-frame copy branch; then pageable memory."""
    import torch
    ctx = B.Context(0)
    ext = B.BriskDescriptorExtractor(context=ctx)
    oext = O.Extractor()
    w, h, n = 640, 480, 70
    distinct = [synth.gen(w, h, 40 + i, 60) for i in range(5)]
    exp = [_expected(img, 60, 4, oext) for img in distinct]
    host = torch.from_numpy(np.stack([distinct[f % 5] for f in range(n)])).pin_memory()
    for rep in range(2):  # the second call reuses the staging buffers and the dirty score-state map
        ctx.detect_describe_batch_host(ext, host.data_ptr(), n, w, h, w * h, w, 60, 4)
        assert ctx.batch_status(n) == 0
        for f in range(n):
            kd, _ = ctx.batch_download(f, described=False)
            kg, dg = ctx.batch_download(f, described=True)
            ko, (ko2, do) = exp[f % 5]
            assert same_kps(kd, ko), (f, explain(kd, ko))
            assert same_kps(kg, ko2) and np.array_equal(dg, do), f
    # padded rows and frames
    pw, ph = w + 24, h + 3
    pad = np.zeros((9, ph, pw), np.uint8)
    for f in range(9):
        pad[f, :h, :w] = distinct[f % 5]
    tpad = torch.from_numpy(pad).pin_memory()
    ctx.detect_describe_batch_host(ext, tpad.data_ptr(), 9, w, h, pw * ph, pw, 60, 4)
    assert ctx.batch_status(9) == 0
    for f in range(9):
        kg, dg = ctx.batch_download(f, described=True)
        assert same_kps(kg, exp[f % 5][1][0]) and np.array_equal(dg, exp[f % 5][1][1])
    # pageable host memory (numpy): slower copies, same result
    plain = np.ascontiguousarray(np.stack([distinct[(f + 2) % 5] for f in range(3)]))
    ctx.detect_describe_batch_host(ext, plain.ctypes.data, 3, w, h, w * h, w, 60, 4)
    assert ctx.batch_status(3) == 0
    for f in range(3):
        kg, dg = ctx.batch_download(f, described=True)
        assert same_kps(kg, exp[(f + 2) % 5][1][0]) and np.array_equal(dg, exp[(f + 2) % 5][1][1])
    ctx.close()


def test_config3_512_frames_with_result_gather_on_one_rank(B):
    """BASELINE config 3 on a single GPU: one batch of 512 x 1080p frames (sharding.shard_frames at world 1), the
    asynchronous PaddedGather exchange over RCCL (backend nccl, world size 1), a sample of frames checked bit-exactly
    against the oracle through the GATHERED buffers."""
    import torch
    import torch.distributed as dist
    from ethzasl_brisk_amd import sharding
    import bench
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n, w, h = 512, 1920, 1080
        mine = sharding.shard_frames(n, 0, 1)
        assert mine == list(range(n))
        nd = 8
        distinct = [synth.frame_1080p(700 + i) for i in range(nd)]
        ring = torch.from_numpy(np.stack(distinct)).to(dev)
        frames = ring[torch.arange(n, device=dev) % nd].contiguous()
        ctx = B.Context(0)
        ext = B.BriskDescriptorExtractor(context=ctx)
        stream = torch.cuda.current_stream().cuda_stream
        ctx.detect_describe_batch(ext, frames.data_ptr(), n, w, h, w * h, w, 80, 4, stream)
        torch.cuda.synchronize()
        assert ctx.batch_status(n) == 0
        g = bench.ResultGather(ctx, n, ext.descriptorSize(), dev, 0, 1)
        g.check_kpad(None)
        g.run()
        g.finish()
        torch.cuda.synchronize()
        ac, gk, gd = g.last
        assert ac.shape == (1, n)
        oext = O.Extractor()
        exp = {}
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
        # every slot that holds the same image holds the same result (no cross-frame leakage at this batch size)
        cnts = ac[0].cpu().numpy()
        for f in range(n):
            assert cnts[f] == cnts[f % nd]
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_calls_on_different_streams_share_the_workspace_safely(B):
    """A batch queued on stream A followed at once by calls on other streams (another batch on stream B, a host-buffer
    detect on the context's own stream): the later call must wait for the earlier one before it clears / reuses the
    shared workspace (event recorded at the end of every call)."""
    import torch
    ctx = B.Context(0)
    ext = B.BriskDescriptorExtractor(context=ctx)
    oext = O.Extractor()
    w, h = 1920, 1080
    a = [synth.frame_1080p(900 + i) for i in range(4)]
    b = [synth.frame_1080p(950 + i) for i in range(2)]
    da = torch.from_numpy(np.stack(a * 16)).cuda()   # 64 frames: long enough to still run when the next call arrives
    db = torch.from_numpy(np.stack(b)).cuda()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for rep in range(3):
        ctx.detect_describe_batch(ext, da.data_ptr(), 64, w, h, w * h, w, 80, 4, sa.cuda_stream)
        ctx.detect_describe_batch(ext, db.data_ptr(), 2, w, h, w * h, w, 80, 4, sb.cuda_stream)
        assert ctx.batch_status(2) == 0
        for f in range(2):
            kg, dg = ctx.batch_download(f, described=True)
            ko2, do = _expected(b[f], 80, 4, oext)[1]
            assert same_kps(kg, ko2) and np.array_equal(dg, do), (rep, f)
        ctx.detect_describe_batch(ext, da.data_ptr(), 64, w, h, w * h, w, 80, 4, sa.cuda_stream)
        det = B.BriskFeatureDetector(80, 4, context=ctx)
        k = det.detect(b[0])
        assert same_kps(k, O.detect(b[0], 80, 4)), rep
    ctx.close()


def test_stream_ceiling_probe_and_kernel_revision(B):
    ctx = B.Context(0)
    cp, rd = ctx.stream_ceiling(1 << 29)
    assert 1000 < cp < 8000 and 1000 < rd < 8000      # GB/s: above a PCIe-class number, below the spec peak
    assert len(ctx.kernel_revision()) == 12
    ctx.close()


def test_reserve_never_shrinks(B):
    ctx = B.Context(0)
    img = synth.frame_vga(1)
    det = B.BriskFeatureDetector(70, 4, context=ctx)
    k0 = det.detect(img)
    ctx.reserve(300000, 70000)
    ctx.reserve(1000, 100)            # smaller request: nothing changes (no reallocation, no truncation)
    k1 = det.detect(img, capacity=70000)
    assert same_kps(k0, k1)
    ctx.close()


@pytest.mark.parametrize("thr", [1, 5, 12, 19])
def test_thresholds_below_20_ordered_path(B, thr):
    """AGAST thresholds 1..19 (the `> 2` cache rule of brisk-layer.cc:118-132 for stored scores <= 2): the
    ordered path (k_order_candidates + k_ordered_keypoints) against the oracle - host-buffer call, mask, batch of
    frames with description, and a fast-path call afterwards on the same context (the cache the ordered path left
    in the score-state map must be wiped completely)."""
    import torch
    ctx = B.Context(0, max_candidates=400000, max_keypoints=100000)
    ext = B.BriskDescriptorExtractor(context=ctx)
    oext = O.Extractor()
    imgs = [synth.gen(320, 240, 9, 30), synth.gen(213, 160, 4, 14)]
    for octaves in (0, 3):
        det = B.BriskFeatureDetector(thr, octaves, context=ctx)
        for img in imgs:
            ko = O.detect(img, thr, octaves)
            k = det.detect(img, capacity=100000)
            assert same_kps(k, ko), (thr, octaves, img.shape, explain(k, ko))
    # mask
    img = imgs[0]
    mask = np.zeros(img.shape, np.uint8)
    mask[40:200, 60:300] = 255
    det = B.BriskFeatureDetector(thr, 2, context=ctx)
    km = det.detect(img, mask=mask, capacity=100000)
    ko = O.detect(img, thr, 2, mask=mask)
    assert same_kps(km, ko)
    # batch of frames, detect + describe
    n = 3
    frames = np.stack([synth.gen(320, 240, 20 + i, 30) for i in range(n)])
    d = torch.from_numpy(frames).cuda()
    ctx.detect_describe_batch(ext, d.data_ptr(), n, 320, 240, 320 * 240, 320, thr, 2, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert ctx.batch_status(n) == 0
    for f in range(n):
        ko = O.detect(frames[f], thr, 2)
        ko2, do = oext.compute(frames[f], ko)
        kd, _ = ctx.batch_download(f, described=False)
        kg, dg = ctx.batch_download(f, described=True, strings=48)
        assert same_kps(kd, ko), (f, explain(kd, ko))
        assert same_kps(kg, ko2) and np.array_equal(dg, do)
    # the fast path on the same context afterwards
    det70 = B.BriskFeatureDetector(70, 3, context=ctx)
    assert same_kps(det70.detect(imgs[0]), O.detect(imgs[0], 70, 3))
    ctx.close()


def test_no_scale_nms_several_layers_at0_quirk(B):
    """suppressScaleNonmaxima=false with octaves > 0 (brisk-scale-space.cc:131-170): layer i takes its points'
    coordinates from layer 0's list (:137).  Inputs on which that stays inside the score matrices (texture only in a
    band at the top) are bit-equal to the oracle; the usual input is reported as undefined (code 7)."""
    from test_emul_parity import banded
    ctx = B.Context(0)
    for seed, cell, octaves, thr in ((0, 3, 2, 60), (1, 4, 3, 60), (2, 2, 1, 45), (3, 4, 3, 12)):
        img = banded(seed, cell=cell)
        ko = O.detect(img, thr, octaves, suppress_scale_nonmaxima=False)
        k = B.BriskFeatureDetector(thr, octaves, suppressScaleNonmaxima=False, context=ctx).detect(img)
        assert ko is not None and len(ko) > 300 and same_kps(k, ko), (seed, explain(k, ko))
    with pytest.raises(B.BriskHipError) as ei:
        B.BriskFeatureDetector(60, 2, suppressScaleNonmaxima=False, context=ctx).detect(synth.gen(320, 240, 3, 30))
    assert ei.value.code == 7
    # and the context keeps working on the fast path
    img = synth.frame_vga(1)
    assert same_kps(B.BriskFeatureDetector(70, 4, context=ctx).detect(img), O.detect(img, 70, 4))
    ctx.close()


def test_custom_ptn_pattern_and_bilinear_branch(B, golden_harris):
    """A valid custom .ptn (jittered positions, 60 points, smaller smoothing radii) through the text constructor
    (brisk-descriptor-extractor.cc:345-367), at patternScale 1.0 and at 0.45, where sigma_half < 0.5 for the small
    scales and SmoothedIntensity takes its bilinear branch (:391-408); provided-angle, rotation-estimated and
    non-scale-invariant modes; bit-equal to the oracle."""
    import ptn
    text = ptn.custom_pattern(1, sigma_factor=0.6, drop_points=6)
    e = golden_harris[0]
    g = e["keypoints"]
    k = np.zeros(len(g), B.KEYPOINT)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        k[f] = g[f]
    k["angle"] = -1
    k["size"] = np.linspace(7.0, 40.0, len(k)).astype(np.float32)
    for ps in (1.0, 0.45):
        for rot, sc in ((True, True), (False, True), (True, False)):
            ko, do = O.Extractor(rot, sc, pattern_text=text, pattern_scale=ps).compute(e["image"], k)
            ext = B.BriskDescriptorExtractor(rot, sc, pattern_text=text, patternScale=ps)
            assert ext.descriptorSize() == 48
            kg, dg = ext.compute(e["image"], k)
            assert len(ko) > 300 and same_kps(kg, ko), (ps, rot, sc, explain(kg, ko))
            assert np.array_equal(dg, do), (ps, rot, sc)


def test_compute_scale_provided_keypoints(B):
    """BriskFeatureDetector::ComputeScale (brisk-feature-detector.cc:87-92) through the C ABI: provided keypoints (detected
    ones plus random non-integral ones with their own class ids) on the lower-threshold-0 pyramid; all three branches,
    the detect-on-empty-layer case, the empty list, the undefined input (code 7), and a fast-path call afterwards."""
    from test_emul_parity import provided_keypoints
    ctx = B.Context(0)
    img = synth.gen(320, 240, 9, 30)
    for thr, octaves, suppress in ((60, 3, True), (60, 0, True), (60, 2, False), (25, 2, True), (8, 1, True)):
        k = provided_keypoints(img, max(thr, 30), 3, 70, seed=thr)
        ko = O.compute_scale(img, k, thr, octaves, suppress)
        kg = B.BriskFeatureDetector(thr, octaves, suppress, context=ctx).ComputeScale(img, k)
        assert ko is not None and len(ko) > 100 and same_kps(kg, ko), (thr, octaves, suppress, explain(kg, ko))
    few = np.zeros(3, B.KEYPOINT)
    few["x"], few["y"], few["size"] = [4, 6.5, 5], [4, 5, 7.25], 12
    det = B.BriskFeatureDetector(60, 3, context=ctx)
    assert same_kps(det.ComputeScale(img, few), O.compute_scale(img, few, 60, 3))
    assert same_kps(B.BriskFeatureDetector(60, 2, context=ctx).ComputeScale(img, few[:0]), O.compute_scale(img, few[:0], 60, 2))
    bad = np.zeros(1, B.KEYPOINT)
    bad["x"], bad["y"] = 100, 236.5
    with pytest.raises(B.BriskHipError) as ei:
        det.ComputeScale(img, bad)
    assert ei.value.code == 7
    vga = synth.frame_vga(1)
    assert same_kps(B.BriskFeatureDetector(70, 4, context=ctx).detect(vga), O.detect(vga, 70, 4))
    ctx.close()


def test_describe_box_that_ends_in_the_last_column_gpu(B):
    """the same as tests/test_emul_parity.py::test_describe_box_that_ends_in_the_last_column through the C ABI (found by
    tools/soak.py: rows padded to 64 bytes made the displaced corner of :453 read padding instead of the next row)"""
    ext = B.BriskDescriptorExtractor()
    oext = O.Extractor()
    _, size_list, size_thresh = ext.tables()
    for w, h in ((426, 320), (333, 201), (1281, 723)):
        img = synth.gen(w, h, 12, 50)
        rng = np.random.default_rng(w)
        k = np.zeros(400, B.KEYPOINT)
        k["size"] = rng.uniform(8.0, 14.0, len(k)).astype(np.float32)
        k["y"] = rng.uniform(40, h - 40, len(k)).astype(np.float32)
        k["angle"] = -1
        for i in range(len(k)):
            sc = int(np.searchsorted(size_thresh, k["size"][i], side="right")) - 1
            k["x"][i] = np.float32(w - size_list[max(sc, 0)]) - np.float32(rng.uniform(0.0, 1.2))
        ko, do = oext.compute(img, k)
        kg, dg = ext.compute(img, k)
        assert len(ko) > 100 and same_kps(kg, ko) and np.array_equal(dg, do), (w, h)


def test_uniformity_dense_clusters_and_beyond_on_chip_capacity(B):
    """k_uniformity decides the points in score order with waits only between points within 15 cells: a dense frame
    whose points form long dependency chains (radius 20 px: every point has earlier neighbours), and a frame with more
    points than the on-chip arrays hold (> 8192: the literal walk over an occupancy image, k_uniformity_seq); with and
    without a keypoint budget."""
    img = synth.frame_1080p(3)
    h, w = img.shape
    ko = O.detect(img, 50, 4)
    assert 5000 < len(ko) <= 8192
    ctx = B.Context(0, max_candidates=262144, max_keypoints=65536)
    for radius, budget in ((20.0, 0x7FFFFFFF), (3.0, 0x7FFFFFFF), (6.0, 700)):
        want = O.enforce_uniformity(ko, h, w, radius, budget)
        got = B.BriskFeatureDetector(50, 4, context=ctx, uniformityRadius=radius, maxNumKpt=budget).detect(img, capacity=65536)
        assert same_kps(got, want), (radius, budget, len(got), len(want), explain(got, want))
    ko = O.detect(img, 38, 4)
    assert len(ko) > 8192
    for radius, budget in ((5.0, 0x7FFFFFFF), (5.0, 3000)):
        want = O.enforce_uniformity(ko, h, w, radius, budget)
        got = B.BriskFeatureDetector(38, 4, context=ctx, uniformityRadius=radius, maxNumKpt=budget).detect(img, capacity=65536)
        assert same_kps(got, want), (radius, budget, len(got), len(want), explain(got, want))
    ctx.close()


def test_tie_kernel_chunked_path_on_a_small_chunk_build(B):
    """k_tie_resolve processes layers with more ties than its on-chip arrays hold in chunks of consecutive raster
    ranks.  A build of the same sources with 64-tie chunks (every layer of these images then takes many chunks, sorted
    through global scratch) in a fresh process: tie-heavy blocks, pure noise, a batch of 12 frames with one workgroup
    per (frame, layer) - all bit-equal to the oracle."""
    import subprocess
    import sys
    from ethzasl_brisk_amd import build
    lib = build.build_variant("libbrisk_test_chunk64", ["TR_CHUNK=64"])
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch, oracle_lib as O, synth, ethzasl_brisk_amd as B
from test_gpu_parity import same_kps, explain
ctx = B.Context(0, max_candidates=262144, max_keypoints=65536)
rng = np.random.default_rng(5)
blocks = np.kron((rng.integers(0, 2, (30, 40)) * 200 + 20).astype(np.uint8), np.ones((8, 8), np.uint8))
noise = np.random.default_rng(1).integers(0, 256, (240, 320), dtype=np.uint8)
for img, thr, octv in ((blocks, 60, 3), (noise, 25, 4), (synth.frame_vga(3), 30, 4)):
    img = np.ascontiguousarray(img)
    ko = O.detect(img, thr, octv)
    kg = B.BriskFeatureDetector(thr, octv, context=ctx).detect(img, capacity=65536)
    assert same_kps(kg, ko), explain(kg, ko)
frames = np.stack([synth.gen(426, 320, 70 + i, 60) for i in range(12)])
d = torch.from_numpy(frames).cuda()
n, h, w = frames.shape
ctx.detect_batch(d.data_ptr(), n, w, h, w * h, w, 35, 4, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
for f in range(n):
    kd, _ = ctx.batch_download(f, described=False)
    ko = O.detect(frames[f], 35, 4)
    assert same_kps(kd, ko), (f, explain(kd, ko))
print("chunked path ok")
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BRISK_HIP_LIB=lib)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "chunked path ok" in r.stdout
