"""GPU parity tests of the Hamming brute-force matcher, through the C ABI, against the CPU oracle
(oracle/brisk_oracle_match.c) and the reference's own matching test."""
import numpy as np
import pytest

import oracle_lib as O
from test_oracle_golden import homography_outliers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    import ethzasl_brisk_amd as B
    from ethzasl_brisk_amd import build
    build.build()
    B.load_library()
    return B


def same_rows(got, want):
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        assert len(g) == len(w), "query %d: %d vs %d matches" % (i, len(g), len(w))
        for f in ("queryIdx", "trainIdx", "imgIdx"):
            assert np.array_equal(g[f], w[f]), (i, f, g, w)
        assert np.array_equal(g["distance"].view(np.uint32), w["distance"].view(np.uint32)), (i, g, w)


def test_reference_match_test_on_gpu(B, golden_ast):
    """brisk/src/test/test-match.cc:49-126 end to end on the GPU: detect (70, 2 octaves), describe, best match below
    Hamming 50, zero outliers of the known homography; and the same rows as the oracle."""
    det, ext, bf = B.BriskFeatureDetector(70, 2), B.BriskDescriptorExtractor(), B.BruteForceMatcher()
    k1, d1 = ext.compute(golden_ast[0]["image"], det.detect(golden_ast[0]["image"]))
    k2, d2 = ext.compute(golden_ast[1]["image"], det.detect(golden_ast[1]["image"]))
    bf.add(d2)
    rows = bf.knnMatch(d1, 1)
    same_rows(rows, O.match_knn(d1, [d2], 1))
    best = np.concatenate([r for r in rows if len(r) and r[0]["distance"] < 50])
    assert len(best) > 100
    assert homography_outliers(k1, k2, best) == 0
    same_rows(bf.radiusMatch(d1, 50.0), O.match_radius(d1, [d2], 50.0))
    m = bf.match(d1)
    assert len(m) == len(d1) and np.array_equal(m["trainIdx"], np.array([r[0]["trainIdx"] for r in rows]))


@pytest.mark.parametrize("dim", [48, 64, 16, 40, 96, 224])
def test_knn_and_radius_vs_oracle(B, dim):
    rng = np.random.default_rng(dim)
    # low-entropy descriptors: plenty of equal distances (tie order = (distance, image, train index))
    def rnd(n):
        return (rng.integers(0, 4, (n, dim), dtype=np.uint8) * 85).astype(np.uint8)
    q = rnd(300)
    train = [rnd(257), np.zeros((0, dim), np.uint8), rnd(64), rnd(1)]
    train[2][:5] = q[:5]                       # exact duplicates -> distance 0
    train[0][100:103] = train[2][10:13]        # the same descriptor in two images
    bf = B.BruteForceMatcher()
    bf.add(train)
    for k in (1, 2, 7):
        same_rows(bf.knnMatch(q, k), O.match_knn(q, train, k))
    for r in (0.5, float(dim * 2), float(dim * 3) + 0.5, 1e9):
        same_rows(bf.radiusMatch(q, r), O.match_radius(q, train, r))
    # a single non-empty train image and k <= 2 take the fused distance + top-2 kernel (plenty of equal distances
    # here: the (distance, train index) order must hold there too), also when empty images precede it
    for tr in ([train[0]], [train[1], train[1], train[0], train[1]], [train[3]]):
        bf1 = B.BruteForceMatcher()
        bf1.add(tr)
        for k in (1, 2):
            same_rows(bf1.knnMatch(q, k), O.match_knn(q, tr, k))


def test_masks_topup_and_empty_sets(B):
    rng = np.random.default_rng(11)
    q = rng.integers(0, 256, (40, 48), dtype=np.uint8)
    t0 = rng.integers(0, 256, (9, 48), dtype=np.uint8)
    t1 = rng.integers(0, 256, (70, 48), dtype=np.uint8)
    m0 = (rng.random((40, 9)) > 0.3).astype(np.uint8)
    m1 = (rng.random((40, 70)) > 0.5).astype(np.uint8) * 255
    m0[4, :] = 0                                # query 4 can match nothing in image 0 ...
    m1[4, :] = 0                                # ... nor in image 1: masked out (isMaskedOut: every mask has a zero row)
    m1[9, :] = 0                                # query 9 is only barred from image 1
    bf = B.BruteForceMatcher()
    bf.add([t0, t1])
    for masks in ([m0, m1], [m0, None], [None, m1]):
        same_rows(bf.knnMatch(q, 3, masks), O.match_knn(q, [t0, t1], 3, masks))
        same_rows(bf.knnMatch(q, 100, masks), O.match_knn(q, [t0, t1], 100, masks))   # k > possible matches
        same_rows(bf.radiusMatch(q, 190.0, masks), O.match_radius(q, [t0, t1], 190.0, masks))
    assert len(bf.knnMatch(q, 3, [m0, m1], compactResult=True)) == 39
    assert len(bf.knnMatch(q, 3, [m0, None], compactResult=True)) == 40   # an image without a mask keeps every query alive
    # an empty image with an (empty) mask is not counted either: nobody is masked out, query 4 gets pseudo matches only
    e = np.zeros((0, 48), np.uint8)
    bfe = B.BruteForceMatcher()
    bfe.add([t0, t1, e])
    me = [m0, m1, np.zeros((40, 0), np.uint8)]
    same_rows(bfe.knnMatch(q, 3, me), O.match_knn(q, [t0, t1, e], 3, me))
    same_rows(bfe.radiusMatch(q, 190.0, me), O.match_radius(q, [t0, t1, e], 190.0, me))
    assert len(bfe.knnMatch(q, 3, me, compactResult=True)) == 40
    # k larger than the train set, trailing empty image
    bf2 = B.BruteForceMatcher()
    bf2.add([t0[:2], t1[:1], np.zeros((0, 48), np.uint8)])
    same_rows(bf2.knnMatch(q[:3], 5), O.match_knn(q[:3], [t0[:2], t1[:1], np.zeros((0, 48), np.uint8)], 5))
    # no train data at all / no queries
    bf3 = B.BruteForceMatcher()
    assert all(len(r) == 0 for r in bf3.knnMatch(q[:3], 2))
    bf3.add(np.zeros((0, 48), np.uint8))
    assert all(len(r) == 0 for r in bf3.knnMatch(q[:3], 2))
    assert bf.knnMatch(np.zeros((0, 48), np.uint8), 2) == []
    bf4 = B.BruteForceMatcher()
    bf4.add(np.zeros((2, 8), np.uint8))
    with pytest.raises(B.BriskHipError):
        bf4.knnMatch(np.zeros((3, 8), np.uint8), 1)     # descriptor shorter than one 128-bit word


def test_large_sets_property(B):
    """20k x 20k 48-byte descriptors: first match of every query equals the brute-force argmin computed with numpy
    on a sample, distances are sorted, and a set matched against itself returns the identity at distance 0."""
    rng = np.random.default_rng(2)
    t = rng.integers(0, 256, (20000, 48), dtype=np.uint8)
    q = t.copy()
    q[::2, 0] ^= 1                                   # every other query is one bit away from its twin
    bf = B.BruteForceMatcher()
    bf.add(t)
    rows = bf.knnMatch(q, 2)
    first = np.array([r[0]["trainIdx"] for r in rows])
    assert np.array_equal(first, np.arange(20000))
    assert np.array_equal(np.array([r[0]["distance"] for r in rows]), np.tile([1.0, 0.0], 10000).astype(np.float32))
    assert all(r[0]["distance"] <= r[1]["distance"] for r in rows)
    for i in rng.integers(0, 20000, 20):
        d = np.unpackbits(q[i][None, :] ^ t, axis=1).sum(axis=1)
        d[i] = 10 ** 6
        j = int(np.argmin(d))
        assert rows[i][1]["trainIdx"] == j and rows[i][1]["distance"] == d[j]


def test_device_resident_knn(B):
    import torch
    rng = np.random.default_rng(4)
    q = rng.integers(0, 256, (500, 48), dtype=np.uint8)
    t = rng.integers(0, 256, (700, 64), dtype=np.uint8)   # rows at pitch 64, 48 bytes used
    dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
    out = torch.zeros((500, 2, 4), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(500, dtype=torch.int32, device="cuda")
    ctx = B.default_context(0)
    stream = torch.cuda.current_stream().cuda_stream
    ctx.check(ctx._L.brisk_hip_match_knn_device(ctx._h, dq.data_ptr(), 500, 48, dt.data_ptr(), 700, 64, 48, 2,
                                                out.data_ptr(), cnt.data_ptr(), stream))
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(B.DMATCH).reshape(500, 2)
    want = O.match_knn(q, [t[:, :48]], 2)
    same_rows([got[i, :c] for i, c in enumerate(cnt.cpu().numpy())], want)


def test_pipeline_in_hbm_detect_describe_match(B, golden_ast):
    """Frames in HBM -> detect + describe batch -> match frame 0 against frame 1 without the descriptors ever leaving
    the device (brisk_hip_batch_results pointers straight into brisk_hip_match_knn_device); the reference's homography
    test on the result."""
    import ctypes as C
    import torch
    frames = np.stack([golden_ast[0]["image"], golden_ast[1]["image"]])
    d = torch.from_numpy(frames).cuda()
    ctx = B.Context(0)
    ext = B.BriskDescriptorExtractor(context=ctx)
    n, h, w = frames.shape
    stream = torch.cuda.current_stream().cuda_stream
    ctx.detect_describe_batch(ext, d.data_ptr(), n, w, h, w * h, w, 70, 2, stream)
    vp = C.c_void_p
    d_det, d_n, d_kd, d_kp, d_desc = vp(), vp(), vp(), vp(), vp()
    cstride, cap, pitch = C.c_int(), C.c_int(), C.c_int()
    ctx.check(ctx._L.brisk_hip_batch_results(ctx._h, C.byref(d_det), C.byref(d_n), C.byref(cstride), C.byref(d_kd), C.byref(d_kp),
                                             C.byref(d_desc), C.byref(cap), C.byref(pitch)))
    torch.cuda.synchronize()
    k0, d0 = ctx.batch_download(0, True)
    k1, d1 = ctx.batch_download(1, True)
    out = torch.zeros((len(k0), 1, 4), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(len(k0), dtype=torch.int32, device="cuda")
    ctx.check(ctx._L.brisk_hip_match_knn_device(ctx._h, d_desc.value, len(k0), pitch.value,
                                                d_desc.value + cap.value * pitch.value, len(k1), pitch.value, 48, 1,
                                                out.data_ptr(), cnt.data_ptr(), stream))
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(B.DMATCH).reshape(len(k0))
    same_rows([got[i:i + 1] for i in range(len(k0))], O.match_knn(d0, [d1], 1))
    best = got[got["distance"] < 50]
    assert len(best) > 100 and homography_outliers(k0, k1, best) == 0
    ctx.close()
