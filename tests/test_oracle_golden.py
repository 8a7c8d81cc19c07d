"""Pins the CPU oracle against the reference's own golden vectors and unit-test restatements.

Goldens: brisk/src/test/test_data/brisk_verification_{ast,harris}.set (copied verbatim to
tests/golden/), checked the way brisk/src/test/test-binary-equal.cc:319-333 + bench-ds.h:311-430
does - except that descriptors must match exactly (reference tolerates Hamming <= 5).
"""
import numpy as np
import pytest

import oracle_lib as O


def _bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


@pytest.mark.parametrize("idx", [0, 1])
def test_ast_golden_detect_describe(golden_ast, idx):
    e = golden_ast[idx]
    assert e["image"].shape == (640, 800)
    kps = O.detect(e["image"], 70, 3)          # BriskFeatureDetector(70) -> octaves = 3
    assert len(kps) == (778, 1000)[idx]        # SURVEY F6 probe counts before border filtering
    k2, desc = O.Extractor().compute(e["image"], kps)
    g = e["keypoints"]
    assert len(k2) == len(g) == (642, 785)[idx]
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(_bits(k2[f]), _bits(g[f])), f
    assert desc.shape == e["descriptors"].shape == (len(g), 48)
    assert np.array_equal(desc, e["descriptors"])


@pytest.mark.parametrize("idx", [0, 1])
def test_harris_golden_descriptor_only(golden_harris, idx):
    """Externally provided keypoints (size 12): pins orientation + descriptor without the detector."""
    e = golden_harris[idx]
    g = e["keypoints"]
    k = np.zeros(len(g), O.KP)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        k[f] = g[f]
    k["angle"] = -1
    ext = O.Extractor()
    k2, desc = ext.compute(e["image"], k)
    assert len(k2) == len(g)
    assert np.array_equal(_bits(k2["angle"]), _bits(g["angle"]))
    assert np.array_equal(desc, e["descriptors"])
    k["angle"] = g["angle"]                     # provided-angle path (:742-752)
    k3, desc3 = ext.compute(e["image"], k)
    assert np.array_equal(desc3, e["descriptors"])


def test_extractor_tables():
    ext = O.Extractor()
    assert ext.strings == 48 and ext.points == 66
    sl, sz = ext.scale_list(), ext.size_list()
    assert sl[0] == 1.0 and np.all(np.diff(sl) > 0) and abs(sl[63] - 30.0 ** (63 / 64)) < 1e-4
    assert sz[0] == 13 and sz[63] == 316                    # border 13...316 px
    v1 = O.Extractor(version=1)
    assert v1.strings == 64 and v1.points == 60             # SURVEY F2 (briskV1 -> 512 bits)


# ---- differential unit tests mirroring the reference's own (test-downsampling.cc, test-integral-image.cc)

def plain_half(src):
    """PlainHalfSample, test-downsampling.cc:67-88."""
    s = src.astype(np.uint32)
    a, b, c, d = s[0::2, 0::2], s[0::2, 1::2], s[1::2, 0::2], s[1::2, 1::2]
    h, w = src.shape[0] // 2, src.shape[1] // 2
    a, b, c, d = a[:h, :w], b[:h, :w], c[:h, :w], d[:h, :w]
    return np.minimum(((a + 1 + c) // 2 + (b + 1 + d) // 2 + 1) // 2, 255).astype(np.uint8)


def plain_twothird(src):
    """PlainTwoThirdSample, test-downsampling.cc:90-142."""
    s = src.astype(np.uint32)
    h3, w3 = src.shape[0] // 3, src.shape[1] // 3
    s = s[:h3 * 3, :w3 * 3]
    A, B, C = s[0::3], s[1::3], s[2::3]
    D = ((A + B + 1) // 2 + A + 1) // 2
    E = ((C + B + 1) // 2 + C + 1) // 2
    out = np.zeros((2 * h3, 2 * w3), np.uint8)
    for R, r0 in ((D, 0), (E, 1)):
        p0, p1, p2 = R[:, 0::3], R[:, 1::3], R[:, 2::3]
        out[r0::2, 0::2] = ((p0 + p1 + 1) // 2 + p0 + 1) // 2
        out[r0::2, 1::2] = ((p2 + p1 + 1) // 2 + p2 + 1) // 2
    return out


def test_halfsample_vs_plain(golden_ast):
    img = golden_ast[0]["image"]                 # 800 cols: pure SIMD class, as in the reference test
    assert np.array_equal(O.halfsample(img), plain_half(img))
    rng = np.random.default_rng(0)
    for w in (32, 64, 96, 1920):                 # w % 32 == 0 -> class 1 only
        im = rng.integers(0, 256, (38, w), dtype=np.uint8)
        assert np.array_equal(O.halfsample(im), plain_half(im))


def test_halfsample_column_classes():
    """three roundings by column class (image-down-sampling.cc:296-382)."""
    rng = np.random.default_rng(1)
    for w in (240, 120, 426, 213, 532, 53, 17, 48, 31):
        im = rng.integers(0, 256, (21, w), dtype=np.uint8)
        got = O.halfsample(im)
        s = im.astype(np.int32)
        hs, end, half_end, left = w // 16, (w // 16) // 2, (w // 16) % 2, (w % 16) // 2
        assert got.shape == (10, w // 2) and 16 * end + 8 * half_end + left == w // 2
        we = (w // 2) * 2
        t, b = s[0:20:2, :we], s[1:20:2, :we]
        v = (t + b + 1) >> 1
        c1 = (v[:, 0::2] + v[:, 1::2] + 1) >> 1
        c2 = (v[:, 0::2] + v[:, 1::2]) // 2
        c3 = (t[:, 0::2] + t[:, 1::2] + b[:, 0::2] + b[:, 1::2] + 2) // 4
        exp = np.concatenate([c1[:, :16 * end], c2[:, 16 * end:16 * end + 8 * half_end],
                              c3[:, 16 * end + 8 * half_end:w // 2]], axis=1)
        assert np.array_equal(got, exp.astype(np.uint8)), w


def test_twothird_vs_plain(golden_ast):
    img = golden_ast[0]["image"]
    got, exp = O.twothirdsample(img), plain_twothird(img)
    nsimd = (800 // 15) * 10                     # SIMD columns agree with the plain restatement
    assert np.array_equal(got[:, :nsimd], exp[:, :nsimd])
    rng = np.random.default_rng(2)
    for w in (1920, 3840, 45, 15):               # no tail columns
        im = rng.integers(0, 256, (31, w), dtype=np.uint8)
        assert np.array_equal(O.twothirdsample(im), plain_twothird(im))
    im = rng.integers(0, 256, (9, 640), dtype=np.uint8)   # 630 SIMD + 9 tail px
    got = O.twothirdsample(im)
    s = im.astype(np.int32)
    for t in range(3):
        x = 630 + 3 * t
        A, B, Cc = s[0::3, x:x + 3], s[1::3, x:x + 3], s[2::3, x:x + 3]
        o = 420 + 2 * t
        assert np.array_equal(got[0::2, o], ((4 * A[:, 0] + 2 * (A[:, 1] + B[:, 0] + 1) + B[:, 1] + 1) // 9).astype(np.uint8))
        assert np.array_equal(got[0::2, o + 1], ((4 * A[:, 2] + 2 * (A[:, 1] + B[:, 2] + 1) + B[:, 1] + 1) // 9).astype(np.uint8))
        assert np.array_equal(got[1::2, o], ((4 * Cc[:, 0] + 2 * (Cc[:, 1] + B[:, 0] + 1) + B[:, 1] + 1) // 9).astype(np.uint8))
        assert np.array_equal(got[1::2, o + 1], ((4 * Cc[:, 2] + 2 * (Cc[:, 1] + B[:, 2] + 1) + B[:, 1] + 1) // 9).astype(np.uint8))


def test_integral_vs_naive(golden_ast):
    """test-integral-image.cc:48-99."""
    img = golden_ast[1]["image"]
    exp = np.zeros((641, 801), np.int64)
    exp[1:, 1:] = img.astype(np.int64).cumsum(0).cumsum(1)
    assert np.array_equal(O.integral(img).astype(np.int64), exp)
    rng = np.random.default_rng(3)
    for shape in ((1, 1), (2, 5), (7, 3), (33, 130)):
        im = rng.integers(0, 256, shape, dtype=np.uint8)
        exp = np.zeros((shape[0] + 1, shape[1] + 1), np.int64)
        exp[1:, 1:] = im.astype(np.int64).cumsum(0).cumsum(1)
        assert np.array_equal(O.integral(im).astype(np.int64), exp)


def disc_contrast(img):
    """max-min over the 37-px radius-3 disc, 0 on the 3-px border."""
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    if h < 7 or w < 7:
        return out
    offs = [(dx, dy) for dy in range(-3, 4) for dx in range(-3, 4)
            if (abs(dx) <= 1 and abs(dy) <= 3) or (abs(dy) <= 1 and abs(dx) <= 3) or (abs(dx) == 2 and abs(dy) == 2)]
    assert len(offs) == 37
    st = np.stack([img[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in offs])
    out[3:h - 3, 3:w - 3] = st.max(0) - st.min(0)
    return out


def test_threshold_map_is_disc_contrast():
    """The literal SIMD/scalar pass order of brisk-layer.cc:278-598 nets out to the plain disc
    contrast for every width (what the HIP kernel computes)."""
    rng = np.random.default_rng(4)
    for w in list(range(7, 60)) + [106, 133, 160, 213, 266, 426, 532, 640, 800]:
        im = rng.integers(0, 256, (19, w), dtype=np.uint8)
        assert np.array_equal(O.threshold_map(im), disc_contrast(im)), w
    for h in (3, 6, 7, 8):
        im = rng.integers(0, 256, (h, 40), dtype=np.uint8)
        assert np.array_equal(O.threshold_map(im), disc_contrast(im)), h


def test_detected_score_is_contrast(golden_ast):
    """every detected pixel stores the raw threshold-map value."""
    ss = O.ScaleSpace(golden_ast[0]["image"], 70, 3)
    assert ss.layers == 6
    assert [ss.image(i).shape for i in range(6)] == [(640, 800), (426, 532), (320, 400), (213, 266), (160, 200), (106, 133)]
    import ctypes as C
    L = O.lib()
    for i in range(6):
        img, thr = ss.image(i), ss.thrmap(i)
        h, w = img.shape
        xy = np.zeros((200000, 2), np.int32)
        n = L.bo_oast9_16_detect(img.ctypes.data_as(C.c_void_p), w, h, thr.ctypes.data_as(C.c_void_p), 70, 230, 10,
                                 xy.ctypes.data_as(C.c_void_p), len(xy))
        assert 0 < n < len(xy)
        for x, y in xy[:n:7]:
            p = img.ctypes.data + int(y) * w + int(x)
            assert L.bo_oast9_16_corner_score(p, w, int(thr[y, x])) == thr[y, x]


# ---- matcher: brisk::Hamming + BruteForceMatcher restatement ------------------------------------

def test_popcount_known_answer():
    """brisk/src/test/test-popcount.cc:60-105: the reference's own vectors against a bit-by-bit count."""
    d1 = np.zeros(16, np.uint8)
    d2 = np.zeros(16, np.uint8)
    for i, v in {0: 0x5, 3: 0x2, 6: 0x34, 8: 0x7, 10: 0x23, 13: 0x45, 15: 0x78}.items():
        d1[i] = v
    for i, v in {0: 0x22, 3: 0x78, 6: 0x12, 8: 0x32, 10: 0x1, 13: 0x23, 15: 0x75}.items():
        d2[i] = v
    expect = int(np.unpackbits(d1 ^ d2).sum())
    assert expect == 25
    assert O.hamming(d1, d2) == expect


def test_hamming_counts_whole_128bit_words_only():
    """hamming.h:98-112: `size / 16` words - a 40-byte descriptor is compared on its first 32 bytes."""
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, 40, dtype=np.uint8)
    b = rng.integers(0, 256, 40, dtype=np.uint8)
    assert O.hamming(a, b) == int(np.unpackbits(a[:32] ^ b[:32]).sum())
    a48, b48 = rng.integers(0, 256, 48, dtype=np.uint8), rng.integers(0, 256, 48, dtype=np.uint8)
    assert O.hamming(a48, b48) == int(np.unpackbits(a48 ^ b48).sum())


H_1TO2 = np.array([[0.8835462624646065, 0.31399802853807735, -40.079602102472926],
                   [-0.18170359412701342, 0.9417589525236417, 152.6910745330205],
                   [2.0127825613685174e-4, -1.5103648761897873e-5, 1.0]])


def homography_outliers(k1, k2, matches, thres=5.0):
    """brisk/src/test/test-match.cc:90-121"""
    out = 0
    for m in matches:
        p = H_1TO2 @ np.array([k1["x"][m["queryIdx"]], k1["y"][m["queryIdx"]], 1.0], np.float64)
        p /= p[2]
        q = np.array([k2["x"][m["trainIdx"]], k2["y"][m["trainIdx"]], 1.0], np.float64)
        out += np.linalg.norm(p - q) > thres
    return out


def test_match_homography(golden_ast):
    """The reference's matching test (test-match.cc:49-126): detector(70, 2 octaves), default extractor, best match
    below Hamming 50, every match an inlier of the known homography."""
    k1, d1 = O.Extractor().compute(golden_ast[0]["image"], O.detect(golden_ast[0]["image"], 70, 2))
    k2, d2 = O.Extractor().compute(golden_ast[1]["image"], O.detect(golden_ast[1]["image"], 70, 2))
    rows = O.match_knn(d1, [d2], 1)
    best = np.concatenate([r for r in rows if len(r) and r[0]["distance"] < 50])
    # same selection as the reference test's own loop (strict <, first index wins)
    dist = np.unpackbits(d1[:, None, :] ^ d2[None, :, :], axis=2).sum(axis=2)
    ref = [(i, int(np.argmin(dist[i]))) for i in range(len(d1)) if dist[i].min() < 50]
    assert [(int(m["queryIdx"]), int(m["trainIdx"])) for m in best] == ref
    assert len(best) > 100
    assert homography_outliers(k1, k2, best) == 0


def test_match_oracle_semantics():
    rng = np.random.default_rng(5)
    q = rng.integers(0, 256, (7, 48), dtype=np.uint8)
    t0 = rng.integers(0, 256, (5, 48), dtype=np.uint8)
    t1 = np.concatenate([t0[:2], rng.integers(0, 256, (3, 48), dtype=np.uint8)])  # duplicates across images
    rows = O.match_knn(q, [t0, t1], 3)
    for i, r in enumerate(rows):
        d = [(int(np.unpackbits(q[i] ^ t).sum()), im, j) for im, tt in enumerate((t0, t1)) for j, t in enumerate(tt)]
        d.sort()
        assert [(int(m["distance"]), int(m["imgIdx"]), int(m["trainIdx"])) for m in r] == d[:3]
    # k larger than the number of train descriptors: the reference tops the row up with INT_MAX pseudo matches on
    # train 0 of the LAST non-empty image (brute-force-matcher.cc:139-153)
    rows = O.match_knn(q[:1], [t0[:2], t1[:1], np.zeros((0, 48), np.uint8)], 5)
    r = rows[0]
    assert len(r) == 5 and list(r["distance"][3:]) == [2147483648.0, 2147483648.0]
    assert list(r["imgIdx"][3:]) == [1, 1] and list(r["trainIdx"][3:]) == [0, 0]
    # masks: masked pairs are skipped; a query is dropped only when EVERY image has a mask whose row is all zero
    # (cv::DescriptorMatcher::isMaskedOut: outCount == masks.size()) - an image without a mask keeps it alive
    m0 = np.ones((7, 5), np.uint8)
    m0[2, :] = 0
    m0[3, 1] = 0
    rows = O.match_knn(q, [t0, t1], 10, [m0, None])
    assert len(rows[2]) == 10 and all(m["imgIdx"] == 1 for m in rows[2] if m["distance"] < 1e9) and len(rows[0]) == 10
    assert not any((m["imgIdx"] == 0 and m["trainIdx"] == 1 and m["distance"] < 1e9) for m in rows[3])
    m1 = np.ones((7, t1.shape[0]), np.uint8)
    m1[2, :] = 0
    m1[4, :] = 0
    rows2 = O.match_knn(q, [t0, t1], 10, [m0, m1])
    assert len(rows2[2]) == 0 and len(rows2[4]) == 10          # query 2 can match nothing, query 4 still has image 0
    assert len(O.match_knn(q, [t0, t1, np.zeros((0, 48), np.uint8)], 10, [m0, m1, np.zeros((7, 0), np.uint8)])[2]) == 10   # an empty mask is not counted
    rr = O.match_radius(q, [t0, t1], 200.0, [m0, None])
    assert all(m["imgIdx"] == 1 for m in rr[2])
    assert len(O.match_radius(q, [t0, t1], 200.0, [m0, m1])[2]) == 0
    for i in (0, 3):
        d = [(int(np.unpackbits(q[i] ^ t).sum()), im, j) for im, tt in enumerate((t0, t1)) for j, t in enumerate(tt)
             if not (im == 0 and m0[i, j] == 0)]
        d = sorted(x for x in d if x[0] < 200.0)
        assert [(int(m["distance"]), int(m["imgIdx"]), int(m["trainIdx"])) for m in rr[i]] == d


# ---- uniformity enforcement (oracle/brisk_oracle_uniformity.c: PARITY UNPINNED, see its header) --------

def test_uniformity_oracle_properties(golden_ast):
    img = golden_ast[1]["image"]
    k = O.detect(img, 70, 3)
    f = O.enforce_uniformity(k, img.shape[0], img.shape[1], 20.0)
    assert 0 < len(f) < len(k)
    assert np.all(np.diff(f["response"]) <= 0)                         # acceptance order = descending score
    kset = {tuple(r) for r in k.view(np.uint32).reshape(len(k), 7)}
    assert all(tuple(r) in kset for r in f.view(np.uint32).reshape(len(f), 7))   # a subset, fields untouched
    assert f[0]["response"] == k["response"].max()                      # the best point always survives
    # hand-checkable cases (radius 15: occupancy at image scale).  The mask centre of an accepted point holds
    # ceil(0.99 * 255) = 253, so an equally strong point at the same place (255 >= 253) still passes ...
    two = np.zeros(2, O.KP)
    two["x"], two["y"], two["response"] = 50.0, 40.0, 100.0
    assert len(O.enforce_uniformity(two, 100, 120, 15.0)) == 2
    two["response"][1] = 90.0                                           # ... a weaker one does not ...
    assert len(O.enforce_uniformity(two, 100, 120, 15.0)) == 1
    two["x"][1] = 90.0                                                  # ... unless it is outside the 31 x 31 mask
    assert len(O.enforce_uniformity(two, 100, 120, 15.0)) == 2
    # the mask centre holds ceil(0.99 * 255) = 253: a point at the same place needs sqrt(sqrt(s / max)) * 255 >= 253
    two["x"][1] = 50.0
    two["response"][1] = 100.0 * (253.0 / 255.0) ** 4 * 0.999
    assert len(O.enforce_uniformity(two, 100, 120, 15.0)) == 1
    # max_keypoints cuts the descending list
    assert np.array_equal(O.enforce_uniformity(k, img.shape[0], img.shape[1], 20.0, 50).view(np.uint32),
                          f[:50].view(np.uint32))
    # a larger radius keeps fewer points
    assert len(O.enforce_uniformity(k, img.shape[0], img.shape[1], 40.0)) < len(f)


def test_set_file_round_trip(tmp_path):
    """The `.set` container of the reference's goldens: read -> write is byte-identical, so results of
    this engine can be handed to the reference's own verification tooling."""
    import os
    from setfile import read_set, write_set
    here = os.path.dirname(os.path.abspath(__file__))
    for name in ("brisk_verification_ast.set", "brisk_verification_harris.set"):
        src = os.path.join(here, "golden", name)
        dst = str(tmp_path / name)
        write_set(dst, read_set(src))
        assert open(dst, "rb").read() == open(src, "rb").read()


def test_key_point_bucketing_restatement():
    """KeyPointBucketing (key-point-bucketing-inl.h:40-112) as restated in the oracle, against a direct Python reading of
    the reference on random points: descending score order, per-bucket cap max / (nbu * nbv), bucket steps
    1 + (size - 1) / buckets, the single-bucket branch keeps the best max (and leaves a vector of no more than max points
    untouched, in detector order), the CHECKed argument ranges."""
    import oracle_lib as O
    rng = np.random.default_rng(7)
    rows, cols = 480, 640
    for n, mx, nbu, nbv in ((500, 100, 4, 3), (500, 100, 1, 5), (37, 1000, 8, 8), (2000, 64, 8, 8), (300, 7, 2, 2), (10, 3, 1, 1),
                            (40, 100, 1, 3), (100, 100, 5, 1)):
        k = np.zeros(n, O.KP)
        k["x"] = rng.uniform(0, cols - 0.01, n).astype(np.float32)
        k["y"] = rng.uniform(0, rows - 0.01, n).astype(np.float32)
        k["response"] = rng.integers(1, 60, n).astype(np.float32)   # many equal scores
        got = O.key_point_bucketing(k, rows, cols, mx, nbu, nbv)
        order = sorted(range(n), key=lambda i: (-k["response"][i], i))
        if nbu == 1 or nbv == 1:
            want = order[:mx] if n > mx else list(range(n))   # :87-88: not too many points -> the vector is left untouched
        else:
            cap, su, sv = mx // (nbu * nbv), 1 + (cols - 1) // nbu, 1 + (rows - 1) // nbv
            cnt, want = {}, []
            for i in order:
                b = (int(k["x"][i]) // su, int(k["y"][i]) // sv)
                assert b[0] < nbu and b[1] < nbv
                if cnt.get(b, 0) < cap:
                    cnt[b] = cnt.get(b, 0) + 1
                    want.append(i)
        assert len(got) == len(want) and all(got[j].tobytes() == k[i].tobytes() for j, i in enumerate(want)), (n, mx, nbu, nbv)
    assert O.key_point_bucketing(k, rows, cols, 10, cols, 2) is None and O.key_point_bucketing(k, rows, cols, 0, 2, 2) is None


def test_16bit_functions_against_per_pixel_restatements():
    """Halfsample16 / Twothirdsample16 / IntegralImage16 (image-down-sampling.cc:56-139, 394-548, integral-image.h:163-218):
    the oracle's block-structured restatements against per-pixel readings of the same arithmetic, the way
    test-downsampling.cc:67-142 checks the 8-bit functions against plain loops; sizes that exercise the re-done last block,
    values that exercise the saturating add and the signed pack."""
    import oracle_lib as O
    rng = np.random.default_rng(11)
    for (h, w) in ((48, 64), (37, 53), (480, 752), (19, 16), (21, 12), (40, 17)):
        img = rng.integers(0, 65536, (h, w), dtype=np.uint16)
        img[::7, ::5] = 65535
        img[1::9, 2::3] = 65534
        a = img.astype(np.int64)
        if w // 2 * 2 >= 16:
            hh, ww = h // 2, w // 2
            i00, i01 = a[0:2 * hh:2, 0:2 * ww:2], a[0:2 * hh:2, 1:2 * ww:2]
            i10, i11 = np.minimum(a[1:2 * hh:2, 0:2 * ww:2] + 2, 65535), a[1:2 * hh:2, 1:2 * ww:2]
            want = ((((i00 + i01 + 1) >> 1) + ((i10 + i11 + 1) >> 1) + 1) >> 1).astype(np.uint16)
            assert np.array_equal(O.halfsample16(img), want), (h, w)
        else:
            assert O.halfsample16(img) is None
        if w // 3 * 3 >= 12:
            hh, ww = h // 3, w // 3
            s = lambda r, c: a[r:3 * hh:3, c:3 * ww:3]  # noqa: E731
            want = np.zeros((2 * hh, 2 * ww), np.int64)
            want[0::2, 0::2] = (4 * s(0, 0) + 2 * s(0, 1) + 2 * s(1, 0) + s(1, 1)) // 9
            want[0::2, 1::2] = (4 * s(0, 2) + 2 * s(0, 1) + 2 * s(1, 2) + s(1, 1)) // 9
            want[1::2, 0::2] = (4 * s(2, 0) + 2 * s(2, 1) + 2 * s(1, 0) + s(1, 1)) // 9
            want[1::2, 1::2] = (4 * s(2, 2) + 2 * s(2, 1) + 2 * s(1, 2) + s(1, 1)) // 9
            assert np.array_equal(O.twothirdsample16(img), np.minimum(want, 32767).astype(np.uint16)), (h, w)
        else:
            assert O.twothirdsample16(img) is None
        # integral: sequential float32 row sums (value / 65536 for the columns taken four at a time, the raw value for the rest)
        want = np.zeros((h + 1, w + 1), np.float32)
        n4 = w // 4 * 4
        for y in range(h):
            s_ = np.float32(0)
            for x in range(w):
                s_ = np.float32(s_ + (np.float32(img[y, x]) * np.float32(1.0 / 65536.0) if x < n4 else np.float32(img[y, x])))
                want[y + 1, x + 1] = np.float32(want[y, x + 1] + s_)
            if h * w > 4096 and y > 40:
                break
        got = O.integral16(img)
        rows = (h + 1) if not (h * w > 4096) else 42
        assert np.array_equal(got[:rows].view(np.uint32), want[:rows].view(np.uint32)), (h, w)
