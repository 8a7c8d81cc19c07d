"""GPU parity tests (run with -m gpu on the MI355X box).  Everything goes through the C ABI
(libbrisk_hip.so); results are compared bit-exactly with the reference's golden vectors and with the
CPU oracle on the same inputs.  Float keypoint fields are compared as bit patterns (stricter than the
1e-3 the north star asks for)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu


def same_kps(a, b):
    if len(a) != len(b):
        return False
    return all(np.array_equal(a[f].view(np.uint32) if a[f].dtype == np.float32 else a[f],
                              b[f].view(np.uint32) if b[f].dtype == np.float32 else b[f]) for f in a.dtype.names)


def explain(a, b):
    if len(a) != len(b):
        return "count %d vs %d" % (len(a), len(b))
    return {f: int((a[f] != b[f]).sum()) for f in a.dtype.names}


@pytest.fixture(scope="module")
def B():
    import ethzasl_brisk_amd as B
    from ethzasl_brisk_amd import build
    build.build()
    B.load_library()
    return B


@pytest.fixture(scope="module")
def ctx(B):
    return B.default_context(0)


def test_library_sees_gpu(B):
    assert B.load_library().brisk_hip_device_count() >= 1


@pytest.mark.parametrize("idx", [0, 1])
def test_golden_ast_detect_describe(B, golden_ast, idx):
    """The reference's own golden test (test-binary-equal.cc:319-333), through the HIP path."""
    e = golden_ast[idx]
    det = B.BriskFeatureDetector(70)
    ext = B.BriskDescriptorExtractor()
    kps = det.detect(e["image"])
    assert len(kps) == (778, 1000)[idx]
    k2, desc = ext.compute(e["image"], kps)
    g = e["keypoints"]
    assert len(k2) == len(g)
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        a, b = k2[f], g[f]
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b), f
    assert ext.descriptorSize() == 48 and np.array_equal(desc, e["descriptors"])


@pytest.mark.parametrize("idx", [0, 1])
def test_golden_harris_descriptor_only(B, golden_harris, idx):
    e = golden_harris[idx]
    g = e["keypoints"]
    k = np.zeros(len(g), B.KEYPOINT)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        k[f] = g[f]
    k["angle"] = -1
    ext = B.BriskDescriptorExtractor()
    k2, desc = ext.compute(e["image"], k)
    assert len(k2) == len(g)
    assert np.array_equal(k2["angle"].view(np.uint32), g["angle"].view(np.uint32))
    assert np.array_equal(desc, e["descriptors"])
    k["angle"] = g["angle"]
    k3, desc3 = ext.compute(e["image"], k)
    assert np.array_equal(desc3, e["descriptors"])


def test_pyramid_thrmap_and_detections_per_stage(B, ctx, golden_ast):
    """Kernel-level parity: every pyramid layer image and the detection (D) map vs the oracle."""
    img = golden_ast[0]["image"]
    B.BriskFeatureDetector(70, 3).detect(img)
    ss = O.ScaleSpace(img, 70, 3)
    L = O.lib()
    for l in range(6):
        oimg = ss.image(l)
        assert np.array_equal(ctx.debug_layer(0, l, 0), oimg), "layer image %d" % l
        thr = ss.thrmap(l)
        h, w = oimg.shape
        xy = np.zeros((100000, 2), np.int32)
        n = L.bo_oast9_16_detect(oimg.ctypes.data_as(C.c_void_p), w, h, thr.ctypes.data_as(C.c_void_p), 70, 230, 10,
                                 xy.ctypes.data_as(C.c_void_p), len(xy))
        exp = np.zeros((h, w), np.uint8)
        exp[xy[:n, 1], xy[:n, 0]] = thr[xy[:n, 1], xy[:n, 0]]
        assert np.array_equal(ctx.debug_layer(0, l, 1), exp), "D map %d" % l


CASES = [
    ("vga_cfg1_thr70_o4", lambda: synth.frame_vga(1), 70, 4),      # BASELINE config 1 (260 described)
    ("vga_thr30_o4", lambda: synth.frame_vga(2), 30, 4),
    ("vga_thr20_o3", lambda: synth.frame_vga(3), 20, 3),
    ("odd_333x217_o3", lambda: synth.gen(333, 217, 5, 40), 40, 3),
    ("tiny_101x77_o2", lambda: synth.gen(101, 77, 6, 12), 30, 2),
    ("single_layer", lambda: synth.frame_vga(4), 50, 0),
    ("one_octave", lambda: synth.frame_vga(5), 50, 1),
    ("1080p_cfg2_thr80_o4", lambda: synth.frame_1080p(0), 80, 4),  # BASELINE config 2
    ("large_5003x3001_thr80_o8", lambda: synth.gen(5003, 3001, 9, 2000), 80, 8),   # 16 layers, odd sizes, > 4K
]


@pytest.mark.parametrize("name,mk,thr,octaves", CASES, ids=[c[0] for c in CASES])
def test_detect_describe_vs_oracle(B, name, mk, thr, octaves):
    img = mk()
    ko = O.detect(img, thr, octaves)
    kg = B.BriskFeatureDetector(thr, octaves).detect(img, capacity=65536)
    assert same_kps(kg, ko), explain(kg, ko)
    ko2, do = O.Extractor().compute(img, ko)
    kg2, dg = B.BriskDescriptorExtractor().compute(img, kg)
    assert same_kps(kg2, ko2), explain(kg2, ko2)
    assert np.array_equal(dg, do)
    if name.startswith("vga_cfg1"):
        assert len(kg2) == 260
    if name.startswith("1080p"):
        assert (len(kg), len(kg2)) == (1194, 986)


def test_tie_heavy_blocks(B):
    rng = np.random.default_rng(5)
    b = (rng.integers(0, 2, (30, 40)) * 200 + 20).astype(np.uint8)
    b = np.kron(b, np.ones((8, 8), np.uint8))
    ko = O.detect(b, 60, 3)
    kg = B.BriskFeatureDetector(60, 3).detect(b, capacity=65536)
    assert same_kps(kg, ko), explain(kg, ko)


def test_direct_evaluation_safety_net(B, ctx, golden_ast):
    """k_classify_refine_direct (used only if a score block misses an access) gives the same result."""
    img = golden_ast[0]["image"]
    ko = O.detect(img, 70, 3)
    ctx.debug_set_flags(1)
    try:
        kg = B.BriskFeatureDetector(70, 3).detect(img)
    finally:
        ctx.debug_set_flags(0)
    assert same_kps(kg, ko), explain(kg, ko)


def test_mask_and_errors(B, golden_ast):
    img = golden_ast[1]["image"]
    mask = np.zeros_like(img)
    mask[:, 400:] = 255
    ko = O.detect(img, 70, 3, mask)
    kg = B.BriskFeatureDetector(70, 3).detect(img, mask)
    assert 0 < len(ko) < 1000 and same_kps(kg, ko)
    with pytest.raises(B.BriskHipError) as ei:
        B.BriskFeatureDetector(0, 3).detect(img)            # thresholds 1..255 (1..19 on the ordered path)
    assert ei.value.code == 5
    # suppressScaleNonmaxima=false with several layers: on this image the reference indexes layer 0's point list
    # (`agastPoints.at(0)[n]`, brisk-scale-space.cc:137) at rows that lie outside layer i's score matrix - no defined
    # result, reported as such (the oracle agrees); defined inputs: tests/test_gpu_round2.py
    assert O.detect(img, 70, 3, suppress_scale_nonmaxima=False) is None
    with pytest.raises(B.BriskHipError) as ei:
        B.BriskFeatureDetector(70, 3, suppressScaleNonmaxima=False).detect(img)
    assert ei.value.code == 7
    # suppressScaleNonmaxima=false with a single layer is the 2-D refinement branch (brisk-scale-space.cc:131-170)
    assert same_kps(B.BriskFeatureDetector(70, 0, suppressScaleNonmaxima=False).detect(img), O.detect(img, 70, 0))
    with pytest.raises(RuntimeError):
        B.BriskDescriptorExtractor(version=3)
    # empty keypoint list
    k, d = B.BriskDescriptorExtractor().compute(img, np.zeros(0, B.KEYPOINT))
    assert len(k) == 0 and d.shape == (0, 48)


def test_extractor_modes(B, golden_harris):
    e = golden_harris[1]
    g = e["keypoints"]
    k = np.zeros(len(g), B.KEYPOINT)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        k[f] = g[f]
    k["angle"] = -1
    for rot, sc in ((False, True), (True, False), (False, False)):
        ko, do = O.Extractor(rot, sc).compute(e["image"], k)
        kg, dg = B.BriskDescriptorExtractor(rot, sc).compute(e["image"], k)
        assert same_kps(kg, ko) and np.array_equal(dg, do)
    ko, do = O.Extractor(version=1).compute(e["image"], k)            # 512-bit BRISK 1.0 pattern
    ext1 = B.BriskDescriptorExtractor(version=1)
    kg, dg = ext1.compute(e["image"], k)
    assert ext1.descriptorSize() == 64 and same_kps(kg, ko) and np.array_equal(dg, do)
    ko, do = O.Extractor(pattern_scale=0.8).compute(e["image"], k)
    kg, dg = B.BriskDescriptorExtractor(patternScale=0.8).compute(e["image"], k)
    assert same_kps(kg, ko) and np.array_equal(dg, do)


def test_pattern_tables_on_device_match_oracle(B):
    for version in (2, 1):
        a, b, t = B.BriskDescriptorExtractor(version=version).tables()
        X = O.Extractor(version=version)
        assert np.array_equal(a, X.scale_list()) and np.array_equal(b.astype(np.uint32), X.size_list())


def _integral_equals(ctx, got, want_u64):
    """the engine keeps the integral image modulo 2^24 (3-byte elements) or as u32 (debug flag bit 18 forces that form)"""
    bits = ctx.debug_integral_bits()
    assert bits in (24, 32)
    return np.array_equal(got.astype(np.uint64), want_u64 & ((1 << bits) - 1))


@pytest.mark.parametrize("force32", [False, True])
def test_integral_kernel(B, ctx, golden_ast, force32):
    ctx.debug_set_flags((1 << 18) if force32 else (1 << 24))   # (a descriptor-only call takes the 32-bit form by itself)
    try:
        img = golden_ast[0]["image"][:333, :517].copy()          # odd sizes
        B.BriskDescriptorExtractor().compute(img, np.zeros(0, B.KEYPOINT))
        k = np.zeros(1, B.KEYPOINT)
        k["x"], k["y"], k["size"], k["angle"] = 200, 150, 12, -1
        B.BriskDescriptorExtractor().compute(img, k)
        assert ctx.debug_integral_bits() == (32 if force32 else 24)
        got = ctx.debug_integral(0, 517, 333)
        assert _integral_equals(ctx, got, O.integral(img).astype(np.int64).astype(np.uint64) & 0xFFFFFFFF)
    finally:
        ctx.debug_set_flags(0)


@pytest.mark.parametrize("force32", [False, True])
def test_integral_kernel_wide_and_tall(B, ctx, force32):
    """More than one 2048-column chunk and more than one 64-row band (4K-style row length); both element sizes."""
    ctx.debug_set_flags((1 << 18) if force32 else (1 << 24))
    try:
        rng = np.random.default_rng(11)
        for shape in ((150, 3840), (131, 2049), (65, 2047), (64, 4100)):
            img = rng.integers(0, 256, shape, dtype=np.uint8)
            k = np.zeros(1, B.KEYPOINT)
            k["x"], k["y"], k["size"], k["angle"] = shape[1] // 2, shape[0] // 2, 12, -1
            B.BriskDescriptorExtractor().compute(img, k)
            got = ctx.debug_integral(0, shape[1], shape[0])
            assert _integral_equals(ctx, got, O.integral(img).astype(np.int64).astype(np.uint64) & 0xFFFFFFFF), shape
        # the widest image the engine takes (four column chunks), tall and bright enough for the 32-bit sums to wrap
        # around like the reference's int arithmetic does
        img = rng.integers(200, 256, (2400, 8191), dtype=np.uint8)
        k = np.zeros(1, B.KEYPOINT)
        k["x"], k["y"], k["size"], k["angle"] = 4000, 1000, 12, -1
        B.BriskDescriptorExtractor().compute(img, k)
        got = ctx.debug_integral(0, 8191, 2400)
        want = np.zeros((2401, 8192), np.uint64)
        want[1:, 1:] = img.astype(np.uint64).cumsum(0).cumsum(1)
        assert want.max() > 2 ** 32 and _integral_equals(ctx, got, want)
    finally:
        ctx.debug_set_flags(0)


def test_batch_path_device_resident(B, ctx):
    """Frames resident in HBM (torch tensor), results downloaded per frame; config-3 style batch."""
    import torch
    frames = np.stack([synth.frame_vga(s) for s in range(6)])
    d = torch.from_numpy(frames).cuda()
    ext = B.BriskDescriptorExtractor()
    n, h, w = frames.shape
    ctx.detect_describe_batch(ext, d.data_ptr(), n, w, h, w * h, w, 60, 4, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert ctx.batch_status(n) == 0
    X = O.Extractor()
    for f in range(n):
        ko = O.detect(frames[f], 60, 4)
        ko2, do = X.compute(frames[f], ko)
        kd, _ = ctx.batch_download(f, described=False)
        kg, dg = ctx.batch_download(f, described=True)
        assert same_kps(kd, ko), (f, explain(kd, ko))
        assert same_kps(kg, ko2) and np.array_equal(dg, do), f


def test_batch_path_stream_slices_and_repeated_batches(B):
    """Batches sliced over internal streams (brisk_hip_set_streams) and back-to-back batches of different sizes on one
    context (buffers are reused: no state may leak from one batch into the next)."""
    import torch
    frames = np.stack([synth.frame_vga(s) for s in range(40)])
    d = torch.from_numpy(frames).cuda()
    ext_ctx = B.Context(0)
    ext = B.BriskDescriptorExtractor(context=ext_ctx)
    X = O.Extractor()
    want = {}
    for f in (0, 1, 17, 31, 39):
        ko = O.detect(frames[f], 60, 4)
        want[f] = (ko,) + X.compute(frames[f], ko)
    n, h, w = frames.shape
    stream = torch.cuda.current_stream().cuda_stream
    for nsub, nb in ((2, 40), (1, 40), (4, 40), (1, 18), (2, 33), (1, 40)):
        ext_ctx.set_streams(nsub)
        ext_ctx.detect_describe_batch(ext, d.data_ptr(), nb, w, h, w * h, w, 60, 4, stream)
        torch.cuda.synchronize()
        assert ext_ctx.batch_status(nb) == 0
        for f, (ko, ko2, do) in want.items():
            if f >= nb:
                continue
            kd, _ = ext_ctx.batch_download(f, described=False)
            kg, dg = ext_ctx.batch_download(f, described=True)
            assert same_kps(kd, ko), (nsub, nb, f, explain(kd, ko))
            assert same_kps(kg, ko2) and np.array_equal(dg, do), (nsub, nb, f)
    ext_ctx.close()


def test_full_size_properties_1080p_stream(B, ctx):
    """BASELINE config-2 sizes: properties that need no oracle run per frame - determinism across
    batch slots (same frame twice in one batch gives identical results), output ordering by
    (octave, y, x)-refined position monotonic in layer, keypoints inside the border."""
    import torch
    f0, f1 = synth.frame_1080p(0), synth.frame_1080p(1)
    frames = np.stack([f0, f1, f0, f1])
    d = torch.from_numpy(frames).cuda()
    ext = B.BriskDescriptorExtractor()
    ctx.detect_describe_batch(ext, d.data_ptr(), 4, 1920, 1080, 1920 * 1080, 1920, 80, 4,
                              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert ctx.batch_status(4) == 0
    r = [ctx.batch_download(f, True) for f in range(4)]
    assert same_kps(r[0][0], r[2][0]) and np.array_equal(r[0][1], r[2][1])
    assert same_kps(r[1][0], r[3][0]) and np.array_equal(r[1][1], r[3][1])
    k = r[0][0]
    assert len(k) == 986 and np.all(np.diff(k["octave"]) >= 0)
    assert np.all(k["x"] >= 13) and np.all(k["x"] < 1920 - 13) and np.all(k["angle"] != -1)


def test_config4_4k_six_octaves(B):
    """BASELINE config 4 geometry: 3840x2160, 6 octaves (12 layers down to 80x45), ~4k keypoints, vs the oracle."""
    img = synth.frame_4k(2)
    ko = O.detect(img, 80, 6)
    kg = B.BriskFeatureDetector(80, 6).detect(img, capacity=65536)
    assert same_kps(kg, ko), explain(kg, ko)
    ko2, do = O.Extractor().compute(img, ko)
    assert (len(ko), len(ko2)) == (4745, 4298)               # SURVEY §8(d) probe: 4298 described
    kg2, dg = B.BriskDescriptorExtractor().compute(img, kg)
    assert same_kps(kg2, ko2) and np.array_equal(dg, do)


def test_config5_descriptor_only_100k_keypoints(B):
    """BASELINE config 5: one 1080p frame + 100 000 synthetic keypoints (size log-uniform in [8.64, 200], angle -1)."""
    img = synth.frame_1080p(0)
    rng = np.random.default_rng(7)
    n = 100000
    size = np.exp(rng.uniform(np.log(8.64), np.log(200.0), n)).astype(np.float32)
    ext_o = O.Extractor()
    border = ext_o.size_list()[[ext_o.scale_index(s) for s in size[:2000]]]
    k = np.zeros(n, B.KEYPOINT)
    k["size"] = size
    k["x"] = rng.uniform(0, 1920, n).astype(np.float32)      # some fall outside the border -> filtered
    k["y"] = rng.uniform(0, 1080, n).astype(np.float32)
    k["angle"] = -1
    k["class_id"] = -1
    ctx = B.Context(0, max_candidates=65536, max_keypoints=131072)
    try:
        kg, dg = B.BriskDescriptorExtractor(context=ctx).compute(img, k)
    finally:
        pass
    ko, do = ext_o.compute(img, k)
    assert 20000 < len(ko) < n
    assert same_kps(kg, ko) and np.array_equal(dg, do)
    ctx.close()


def test_capacity_overflow_is_an_error_not_a_truncation(B, golden_ast):
    """Too small a candidate / keypoint capacity must surface as BRISK_HIP_ERR_CAPACITY (code 4)."""
    img = golden_ast[1]["image"]
    ctx = B.Context(0, max_candidates=256, max_keypoints=16384)
    with pytest.raises(B.BriskHipError) as ei:
        B.BriskFeatureDetector(70, 3, context=ctx).detect(img)
    assert ei.value.code == 4
    ctx.close()
    ctx = B.Context(0, max_candidates=65536, max_keypoints=64)
    with pytest.raises(B.BriskHipError) as ei:
        B.BriskFeatureDetector(70, 3, context=ctx).detect(img)
    assert ei.value.code == 4
    ctx.close()
    with pytest.raises(B.BriskHipError) as ei:            # caller's output array too small
        B.BriskFeatureDetector(70, 3).detect(img, capacity=100)
    assert ei.value.code == 4
    # a context must stay usable after an overflowing frame: the score-state map of that frame holds detections
    # without a candidate record, which the next batch's clean-up has to wipe completely
    ctx = B.Context(0, max_candidates=1024, max_keypoints=16384)
    small = golden_ast[0]["image"][:200, :260]
    ko = O.detect(np.ascontiguousarray(small), 70, 3)
    for _ in range(2):
        with pytest.raises(B.BriskHipError):
            B.BriskFeatureDetector(70, 3, context=ctx).detect(img)          # > 1024 candidates
        kg = B.BriskFeatureDetector(70, 3, context=ctx).detect(small)         # fits, same buffers
        assert same_kps(kg, ko), explain(kg, ko)
    ctx.close()


def test_soak_1080p_batch_vs_oracle(B, ctx):
    """BASELINE config 2 at full size: a batch of 24 distinct synthetic 1080p frames through the device-resident
    batch path, every frame compared bit-exactly with the oracle (keypoints, orientation, descriptors)."""
    import torch
    seeds = list(range(100, 124))
    frames = np.stack([synth.frame_1080p(s) for s in seeds])
    d = torch.from_numpy(frames).cuda()
    ext = B.BriskDescriptorExtractor()
    ctx.detect_describe_batch(ext, d.data_ptr(), len(seeds), 1920, 1080, 1920 * 1080, 1920, 80, 4,
                              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert ctx.batch_status(len(seeds)) == 0
    X = O.Extractor()
    total = 0
    for f in range(len(seeds)):
        ko = O.detect(frames[f], 80, 4)
        ko2, do = X.compute(frames[f], ko)
        kd, _ = ctx.batch_download(f, described=False)
        kg, dg = ctx.batch_download(f, described=True)
        assert same_kps(kd, ko), (seeds[f], explain(kd, ko))
        assert same_kps(kg, ko2), (seeds[f], explain(kg, ko2))
        assert np.array_equal(dg, do), seeds[f]
        total += len(kg)
    assert total > 20000


EDGE_CASES = [
    ("flat", lambda: np.full((120, 160), 77, np.uint8), 40, 3),
    ("noise_vga_thr20", lambda: np.random.default_rng(1).integers(0, 256, (480, 640), dtype=np.uint8), 20, 4),
    ("noise_small_thr60_o8", lambda: np.random.default_rng(2).integers(0, 256, (200, 264), dtype=np.uint8), 60, 8),
    ("checker_2px", lambda: np.kron((np.indices((60, 80)).sum(0) % 2).astype(np.uint8) * 200 + 20, np.ones((2, 2), np.uint8)), 60, 3),
    ("tiny_24x20_o2", lambda: np.random.default_rng(3).integers(0, 256, (20, 24), dtype=np.uint8), 30, 2),
    ("tiny_9x9_o1", lambda: np.random.default_rng(4).integers(0, 256, (9, 9), dtype=np.uint8), 30, 1),
    ("thin_600x12_o2", lambda: np.random.default_rng(5).integers(0, 256, (12, 600), dtype=np.uint8), 40, 2),
    ("saturated_blobs", lambda: (np.kron(np.random.default_rng(6).integers(0, 2, (24, 32)), np.ones((10, 10))) * 255).astype(np.uint8), 100, 4),
    ("max_threshold_255", lambda: synth.frame_vga(7), 255, 3),
]


@pytest.mark.parametrize("name,mk,thr,octaves", EDGE_CASES, ids=[c[0] for c in EDGE_CASES])
def test_edge_images_vs_oracle(B, name, mk, thr, octaves):
    """Degenerate inputs: no corners at all, pure noise (dense candidates, long tie chains), layers that shrink to a
    few pixels, images smaller than the descriptor pattern, saturated content, the largest threshold."""
    img = np.ascontiguousarray(mk())
    ko = O.detect(img, thr, octaves)
    big = None
    if name.startswith("noise_vga"):
        # every tenth pixel is a keypoint: the default per-frame capacities answer with BRISK_HIP_ERR_CAPACITY
        # (never a truncated result); a context sized for it reproduces the oracle
        with pytest.raises(B.BriskHipError) as ei:
            B.BriskFeatureDetector(thr, octaves).detect(img, capacity=65536)
        assert ei.value.code == 4
        big = B.Context(0, max_candidates=262144, max_keypoints=65536)
    kg = B.BriskFeatureDetector(thr, octaves, context=big).detect(img, capacity=65536)
    assert same_kps(kg, ko), explain(kg, ko)
    ko2, do = O.Extractor().compute(img, ko)
    kg2, dg = B.BriskDescriptorExtractor(context=big).compute(img, kg)
    assert same_kps(kg2, ko2), explain(kg2, ko2)
    assert np.array_equal(dg, do)
    if name == "flat":
        assert len(kg) == 0
    if big is not None:
        big.close()


def test_row_pitch_and_unaligned_buffers(B, ctx, golden_ast):
    """The C ABI takes any row pitch and any byte alignment of image, mask and descriptor rows."""
    img = golden_ast[0]["image"][:333, :517]
    ko = O.detect(np.ascontiguousarray(img), 70, 3)
    L = B.load_library()
    for pad, shift in ((0, 1), (37, 3), (123, 0)):
        buf = np.zeros(shift + img.shape[0] * (img.shape[1] + pad) + 64, np.uint8)
        view = buf[shift:shift + img.shape[0] * (img.shape[1] + pad)].reshape(img.shape[0], img.shape[1] + pad)
        view[:, :img.shape[1]] = img
        view[:, img.shape[1]:] = 255                       # padding bytes must never be read as pixels
        out = np.zeros(4096, B.KEYPOINT)
        n = C.c_int(0)
        ctx.check(L.brisk_hip_detect(ctx._h, view.ctypes.data, img.shape[1], img.shape[0], img.shape[1] + pad, 70, 3, 1, None, 0,
                                     out.ctypes.data, len(out), C.byref(n)))
        kg = out[:n.value]
        assert same_kps(kg, ko), (pad, shift, explain(kg, ko))
        # describe through a padded, shifted descriptor matrix
        ko2, do = O.Extractor().compute(np.ascontiguousarray(img), ko)
        ext = B.BriskDescriptorExtractor()
        dbuf = np.zeros(1 + len(kg) * 53 + 64, np.uint8)
        n2 = C.c_int(len(kg))
        kk = kg.copy()
        ctx.check(L.brisk_hip_describe(ctx._h, ext._h, view.ctypes.data, img.shape[1], img.shape[0], img.shape[1] + pad,
                                       kk.ctypes.data, C.byref(n2), dbuf[1:].ctypes.data, 53, 1, 1))
        assert n2.value == len(ko2) and same_kps(kk[:n2.value], ko2)
        dg = dbuf[1:1 + n2.value * 53].reshape(n2.value, 53)[:, :48]
        assert np.array_equal(dg, do)


def test_uniformity_enforcement_config4_and_batch(B):
    """BASELINE config 4 as named: 4K frame, 6 octaves, uniformity-enforced to ~4 k keypoints (radius 8 px), then
    described; GPU filter vs the oracle's restatement (itself unpinned, see oracle/brisk_oracle_uniformity.c)."""
    import torch
    img = synth.frame_4k(2)
    ko = O.detect(img, 80, 6)
    fo = O.enforce_uniformity(ko, img.shape[0], img.shape[1], 8.0)
    assert len(ko) == 4745 and len(fo) == 4012
    ctx = B.Context(0)
    det = B.BriskFeatureDetector(80, 6, context=ctx, uniformityRadius=8.0)
    kg = det.detect(img, capacity=16384)
    assert same_kps(kg, fo), explain(kg, fo)
    ko2, do = O.Extractor().compute(img, fo)
    kg2, dg = B.BriskDescriptorExtractor(context=ctx).compute(img, kg)
    assert same_kps(kg2, ko2) and np.array_equal(dg, do)
    # keypoint budget
    capped = B.BriskFeatureDetector(80, 6, context=ctx, uniformityRadius=8.0, maxNumKpt=1000).detect(img)
    assert same_kps(capped, fo[:1000])
    # off again: the plain detector
    assert same_kps(B.BriskFeatureDetector(80, 6, context=ctx).detect(img, capacity=16384), ko)
    # batch path: filter between detection and description, per frame
    frames = np.stack([synth.frame_vga(s) for s in range(5)])
    d = torch.from_numpy(frames).cuda()
    ext = B.BriskDescriptorExtractor(context=ctx)
    ctx.set_uniformity(12.0, 150)
    n, h, w = frames.shape
    ctx.detect_describe_batch(ext, d.data_ptr(), n, w, h, w * h, w, 60, 4, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    X = O.Extractor()
    for f in range(n):
        want = O.enforce_uniformity(O.detect(frames[f], 60, 4), h, w, 12.0, 150)
        kd, _ = ctx.batch_download(f, described=False)
        assert same_kps(kd, want), (f, explain(kd, want))
        ko2, do = X.compute(frames[f], want)
        kg2, dg = ctx.batch_download(f, described=True)
        assert same_kps(kg2, ko2) and np.array_equal(dg, do), f
    ctx.set_uniformity(0.0)
    ctx.close()


def test_c_abi_argument_validation(B, ctx):
    """Bad arguments come back as error codes with a message, never as a crash or a silent result."""
    L = B.load_library()
    img = np.zeros((64, 64), np.uint8)
    out = np.zeros(16, B.KEYPOINT)
    n = C.c_int(-1)

    def detect(w=64, h=64, stride=64, thr=70, octaves=3, imgp=img.ctypes.data, outp=out.ctypes.data, cap=16):
        return L.brisk_hip_detect(ctx._h, imgp, w, h, stride, thr, octaves, 1, None, 0, outp, cap, C.byref(n))

    assert detect() == 0 and n.value == 0
    assert detect(w=0) == 1 and detect(h=-3) == 1 and detect(w=8192) == 1          # BRISK_HIP_ERR_ARG
    assert detect(stride=32) == 1
    assert detect(octaves=9) == 1 and detect(octaves=-1) == 1
    assert detect(thr=0) == 5 and detect(thr=256) == 5 and detect(thr=19) == 0       # BRISK_HIP_ERR_THRESHOLD
    assert detect(imgp=None) == 1 and detect(outp=None) == 1 and detect(cap=-1) == 1
    assert L.brisk_hip_last_error(ctx._h)                                            # a message is always there
    assert L.brisk_hip_detect(None, img.ctypes.data, 64, 64, 64, 70, 3, 1, None, 0, out.ctypes.data, 16, C.byref(n)) == 1
    ext = B.BriskDescriptorExtractor()
    desc = np.zeros((4, 48), np.uint8)
    kp = np.zeros(4, B.KEYPOINT)
    m = C.c_int(4)
    assert L.brisk_hip_describe(ctx._h, ext._h, img.ctypes.data, 64, 64, 64, kp.ctypes.data, C.byref(m), desc.ctypes.data, 16, 1, 1) == 1  # pitch < 48
    m = C.c_int(-1)
    assert L.brisk_hip_describe(ctx._h, ext._h, img.ctypes.data, 64, 64, 64, kp.ctypes.data, C.byref(m), desc.ctypes.data, 48, 1, 1) == 1
    assert L.brisk_hip_set_capacity(ctx._h, 10, 10) == 1
    assert L.brisk_hip_set_uniformity(ctx._h, 0.5, 10) == 1 and L.brisk_hip_set_uniformity(ctx._h, 10.0, 0) == 1
    assert L.brisk_hip_set_uniformity(ctx._h, 0.0, 1) == 0
    h = C.c_void_p()
    assert L.brisk_hip_pattern_create(ctx._h, 3, 1.0, C.byref(h)) != 0              # only versions 1 and 2 exist
    assert L.brisk_hip_pattern_create_from_text(ctx._h, b"not a pattern", 1.0, C.byref(h)) != 0
    # the context still works
    k = B.BriskFeatureDetector(70, 3).detect(synth.frame_vga(1))
    assert len(k) > 100
