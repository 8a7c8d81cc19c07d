"""CPU checks of the device-side logic (no GPU needed).

The HIP kernels' per-item functions are `__host__ __device__`; tests/emul/brisk_emul.cpp runs them in
plain loops that mirror the kernels.  These tests pin that logic (parallel IsMax2D classification,
history-free refinement, order-faithful tie replay, factorised pattern tables, box sampling) against
the oracle, so that GPU minutes are only spent on GPU-specific behaviour.  The emulation is test
infrastructure, never loaded by the product.
"""
import ctypes as C
import numpy as np
import pytest

import emul_lib as E
import oracle_lib as O
import synth


def same_kps(a, b):
    if len(a) != len(b):
        return False
    return all(np.array_equal(a[f].view(np.uint32) if a[f].dtype == np.float32 else a[f],
                              b[f].view(np.uint32) if b[f].dtype == np.float32 else b[f]) for f in a.dtype.names)


def test_closed_form_scores_match_bisection():
    """K' = clamp(M-1, 0, 254) equals cornerScore(b=0) of the oracle, 9_16 and 5_8."""
    import ctypes as C
    rng = np.random.default_rng(0)
    L, Lo = E.lib(), O.lib()
    for trial in range(6):
        if trial < 3:
            img = rng.integers(0, 256, (64, 64), dtype=np.uint8)
        else:  # smooth-ish patches with strong corners
            img = np.clip(rng.integers(0, 2, (8, 8)).repeat(8, 0).repeat(8, 1) * 150 + rng.integers(0, 40, (64, 64)), 0, 255).astype(np.uint8)
        for y in range(3, 61):
            for x in range(3, 61):
                p = img.ctypes.data + y * 64 + x
                assert L.emul_oast_Kp(p, 64) == Lo.bo_oast9_16_corner_score(p, 64, 0)
                assert L.emul_agast58_Kp(p, 64) == Lo.bo_agast5_8_corner_score(p, 64, 0)


@pytest.mark.parametrize("idx", [0, 1])
def test_detect_matches_oracle_golden_images(golden_ast, idx):
    img = golden_ast[idx]["image"]
    ko = O.detect(img, 70, 3)
    # mode bits 0-1: tie scheme (0 Gauss-Seidel sweeps, 1 Jacobi sweeps, 2 raster-sorted in order = kernel main
    # path); bit 2: lane-parallel score-block caches.  Candidate order / scheme must not matter.
    for seed, mode in ((0, 0), (7, 4 | 2), (9, 4 | 1), (3, 2)):
        ke, stats = E.detect(img, 70, 3, seed, mode)
        assert same_kps(ke, ko)
        assert stats[4] == 0                         # the 3x3 / 4x4 score blocks cover every access
    assert stats[1] > 100                            # tie candidates exist


CASES = [
    ("vga_thr70_o4", lambda: synth.frame_vga(1), 70, 4),
    ("vga_thr30_o4", lambda: synth.frame_vga(2), 30, 4),
    ("vga_thr20_o3", lambda: synth.frame_vga(3), 20, 3),
    ("odd_333x217_o3", lambda: synth.gen(333, 217, 5, 40), 40, 3),
    ("tiny_101x77_o2", lambda: synth.gen(101, 77, 6, 12), 30, 2),
    ("single_layer", lambda: synth.frame_vga(4), 50, 0),
    ("one_octave", lambda: synth.frame_vga(5), 50, 1),
]


@pytest.mark.parametrize("name,mk,thr,octaves", CASES, ids=[c[0] for c in CASES])
def test_detect_matches_oracle_synthetic(name, mk, thr, octaves):
    img = mk()
    ko = O.detect(img, thr, octaves)
    for mode in (0, 4 | 2):
        ke, st = E.detect(img, thr, octaves, 3, mode)
        assert len(ko) > 0 and same_kps(ke, ko) and st[4] == 0


def test_detect_tie_heavy_blocks():
    """Blocky image: almost every 2D maximum ties with a neighbour; relaxation chains > 50 deep."""
    rng = np.random.default_rng(5)
    b = (rng.integers(0, 2, (30, 40)) * 200 + 20).astype(np.uint8)
    b = np.kron(b, np.ones((8, 8), np.uint8))
    ko = O.detect(b, 60, 3)
    ke, stats = E.detect(b, 60, 3, 11, 1)
    assert same_kps(ke, ko)
    assert stats[1] > 0.8 * stats[0] and stats[3] > 50
    ke, stats = E.detect(b, 60, 3, 12, 4 | 2)
    assert same_kps(ke, ko) and stats[4] == 0


def test_detect_1080p_config2():
    img = synth.frame_1080p(0)
    ko = O.detect(img, 80, 4)
    assert len(ko) == 1194                           # SURVEY §8(d) config 2 probe
    ke, st = E.detect(img, 80, 4, 1, 4 | 2)
    assert same_kps(ke, ko) and st[4] == 0


@pytest.mark.parametrize("version", [2, 1])
def test_pattern_tables_match_oracle(version):
    P, X = E.Pattern(version=version), O.Extractor(version=version)
    assert (P.strings, P.points) == (X.strings, X.points)
    sl, szl, thr = P.tables()
    assert np.array_equal(sl, X.scale_list()) and np.array_equal(szl.astype(np.uint32), X.size_list())
    lut = X.pattern()                                 # the reference's 51.9 MB LUT
    for s in (0, 1, 17, 63):
        for r in range(0, 1024, 41):
            got = np.array([P.point(s, r, i) for i in range(P.points)])
            assert np.array_equal(got.view(np.uint32), lut[s, r].view(np.uint32))
    rng = np.random.default_rng(1)
    sizes = np.concatenate([rng.uniform(1, 400, 4000).astype(np.float32), thr[1:], np.nextafter(thr[1:], np.float32(0)),
                            np.nextafter(thr[1:], np.float32(1e9))])
    for s in sizes:
        assert P.scale_index(s) == X.scale_index(s)


@pytest.mark.parametrize("version", [2, 1])
def test_describe_matches_oracle(golden_ast, version):
    P, X = E.Pattern(version=version), O.Extractor(version=version)
    for e, thr, octv in ((golden_ast[0], 70, 3), (golden_ast[1], 40, 4)):
        k = O.detect(e["image"], thr, octv)
        ko, do = X.compute(e["image"], k)
        ke, de = P.describe(e["image"], k)
        assert same_kps(ke, ko) and np.array_equal(de, do)


@pytest.mark.parametrize("scale", [0.7, 1.3, 0.45, 2.5])
def test_generated_kernel_pattern_scales_change_the_descriptor_length(golden_ast, scale):
    """generateKernel's pair thresholds are not scaled with the pattern (brisk-descriptor-extractor.cc:338: dMax 5.85,
    dMin 8.2 whatever patternScale is), so a briskV1 extractor at another pattern scale has another number of short pairs
    and another descriptor length (:176): 128 bytes at 0.7, 48 at 1.3, 192 at 0.45"""
    P, X = E.Pattern(version=1, pattern_scale=scale), O.Extractor(version=1, pattern_scale=scale)
    assert (P.strings, P.points) == (X.strings, X.points) and P.strings != 64
    e = golden_ast[0]
    k = O.detect(e["image"], 70, 3)
    ko, do = X.compute(e["image"], k)
    ke, de = P.describe(e["image"], k)
    assert len(ko) > (100 if scale < 2 else 20) and do.shape[1] == X.strings
    assert same_kps(ke, ko) and np.array_equal(de, do)


def test_describe_flags_and_custom_pattern(golden_harris):
    e = golden_harris[0]
    g = e["keypoints"]
    k = np.zeros(len(g), O.KP)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        k[f] = g[f]
    k["angle"] = -1
    ke, de = E.Pattern().describe(e["image"], k)
    assert np.array_equal(ke["angle"].view(np.uint32), g["angle"].view(np.uint32))
    assert np.array_equal(de, e["descriptors"])
    for rot, sc in ((False, True), (True, False), (False, False)):
        ko, do = O.Extractor(rot, sc).compute(e["image"], k)
        ke, de = E.Pattern().describe(e["image"], k, rot, sc)
        assert same_kps(ke, ko) and np.array_equal(de, do)
    # pattern scale 0.8 (and, via the text path, the same default pattern re-serialised)
    ko, do = O.Extractor(pattern_scale=0.8).compute(e["image"], k)
    ke, de = E.Pattern(pattern_scale=0.8).describe(e["image"], k)
    assert same_kps(ke, ko) and np.array_equal(de, do)


def test_b2_fast_exact():
    """k_detect scales the contrast with an fma instead of (tc * thr) / 100: must be the same integer everywhere."""
    assert E.lib().emul_b2_fast_mismatches() == 0


def test_tie_decision_on_eight_lanes_equals_serial():
    """k_tie_resolve checks the eight tie-list neighbours on eight lanes; same decision as the serial walk."""
    L = E.lib()
    L.emul_tie_decide_mismatches.argtypes = [__import__("ctypes").c_uint, __import__("ctypes").c_int]
    assert L.emul_tie_decide_mismatches(7, 200000) == 0


def test_two_step_tie_replay_equals_one_step():
    """the tie kernel replays the cache in two steps (everything static before it waits for pending ties, the open
    touch events after): identical to brisk_state_at on random windows, every slot, every touch geometry"""
    import ctypes as C
    L = E.lib()
    L.emul_state_split_mismatches.argtypes = [C.c_uint, C.c_int]
    assert L.emul_state_split_mismatches(11, 60000) == 0


def test_tie_pipeline_row_bound_covers_every_touch_footprint():
    """k_tie_resolve's layer pipeline: a tie may start once the layer below is past brisk_tie_rows_needed(row); every
    footprint the refinement code records (and the footprint formula for every row) must respect that bound"""
    import ctypes as C
    L = E.lib()
    L.emul_tie_rows_needed_violations.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    for img, thr, octv in ((synth.frame_vga(1), 40, 4), (synth.gen(333, 201, 3, 40), 30, 3)):
        img = np.ascontiguousarray(img)
        h, w = img.shape
        assert L.emul_tie_rows_needed_violations(img.ctypes.data, w, h, thr, octv) == 0


def test_block_form_of_score_max_above_equals_the_generic_one():
    """k_classify_refine's GetScoreMaxAbove on the 4 x 4 block (brisk_score_max_above_blk) vs brisk_score_max_other<0>."""
    L = E.lib()
    L.emul_score_max_above_blk_mismatches.argtypes = [C.c_uint, C.c_int, C.c_void_p]
    stats = np.zeros(4, np.int32)
    assert L.emul_score_max_above_blk_mismatches(5, 3000000, stats.ctypes.data_as(C.c_void_p)) == 0
    # the cases cover both outcomes and recorded touches; the block always covers the accesses
    assert stats[0] > 500000 and stats[1] > 500000 and stats[2] == 0 and stats[3] > 1000000, stats


def test_block_form_of_score_max_below_equals_the_generic_one():
    """the same for GetScoreMaxBelow (incl. its tie rule between equal inner samples)"""
    L = E.lib()
    L.emul_score_max_below_blk_mismatches.argtypes = [C.c_uint, C.c_int, C.c_void_p]
    stats = np.zeros(4, np.int32)
    assert L.emul_score_max_below_blk_mismatches(7, 3000000, stats.ctypes.data_as(C.c_void_p)) == 0
    assert stats[0] > 500000 and stats[1] > 500000 and stats[2] == 0, stats


def test_packed_oast_score_equals_the_scalar_one():
    """k_score_blocks evaluates the 9-of-16 arcs on packed 16-bit lanes (brisk_oast9_16_M_from_pk)"""
    L = E.lib()
    L.emul_oast_pk_mismatches.argtypes = [C.c_uint, C.c_int]
    assert L.emul_oast_pk_mismatches(3, 2000000) == 0


def test_block_anchor_integer_quotients_equal_the_float_ones():
    assert E.lib().emul_block_anchor_mismatches() == 0


def test_pregate_is_a_necessary_condition():
    """k_detect phase A (packed 16-bit pre-gate on the compass pixels) must never drop a pixel that
    brisk_detect_px (the exact per-pixel detection) accepts: synthetic frames, pure noise, saturated blocks, all
    threshold regimes; and its threshold bound must hold for every (threshold, contrast)."""
    import ctypes as C
    L = E.lib()
    L.emul_pregate_missed.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_long)]
    assert L.emul_pregate_bound_violations() == 0
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (240, 320), dtype=np.uint8)
    blocks = (np.kron(rng.integers(0, 2, (30, 40)), np.ones((8, 8))) * 255).astype(np.uint8)
    soft = np.clip(blocks.astype(np.int32) // 2 + rng.integers(-20, 21, blocks.shape), 0, 255).astype(np.uint8)
    for img in (synth.frame_vga(1), synth.gen(333, 201, 3, 40), noise, blocks, soft):
        img = np.ascontiguousarray(img)
        h, w = img.shape
        for thr in (1, 5, 20, 40, 80, 130, 255):
            n = C.c_long(0)
            assert L.emul_pregate_missed(img.ctypes.data, w, h, w, thr, C.byref(n)) == 0, (img.shape, thr)
    n = C.c_long(0)
    img = np.ascontiguousarray(synth.frame_vga(1))
    L.emul_pregate_missed(img.ctypes.data, 640, 480, 640, 80, C.byref(n))
    assert n.value < 0.05 * 640 * 480          # the gate is selective (a few % of the pixels pass)


@pytest.mark.parametrize("thr", [1, 3, 7, 12, 19, 20, 45])
def test_ordered_path_literal_cache_equals_oracle(thr):
    """k_ordered_keypoints (AGAST thresholds 1..19): IsMax2D and the refinement on the literal lazy cache, candidates
    in (layer, y, x) order.  Same device functions as the kernel, run in the harness; also valid (and checked) at
    thresholds the fast path covers."""
    rng = np.random.default_rng(11)
    imgs = [synth.gen(160, 120, 5, 12), synth.gen(213, 107, 6, 20),
            np.clip(synth.gen(96, 96, 7, 6).astype(np.int32) // 3 + rng.integers(0, 6, (96, 96)), 0, 255).astype(np.uint8)]
    for img in imgs:
        for octaves in (0, 1, 3):
            ko = O.detect(img, thr, octaves)
            ke, _ = E.detect(img, thr, octaves, shuffle_seed=3, jacobi=8)
            assert same_kps(ke, ko), (img.shape, thr, octaves, len(ke), len(ko))


def test_fast_path_below_threshold_20_is_exact_unless_a_detection_stores_a_score_of_2_or_less():
    """What separates thresholds below 20 from the fast path is ONE thing: a detection may store a score <= 2, which the
    reference's lazy cache treats as "not cached" (brisk-layer.cc:118-132).  A frame without such a detection - thresholds
    10 ... 19 on ordinary images - is bit-equal to the oracle on the fast path (the engine decides per frame on the device:
    BriskFrameCounters::low_score); frames with one differ there and are exact on the ordered path."""
    rng = np.random.default_rng(2)
    lowc = (120 + rng.integers(0, 4, (100, 140))).astype(np.uint8)
    seen_clean, seen_low = 0, 0
    for img in (synth.gen(400, 240, 100, 30), synth.gen(400, 240, 103, 30), synth.gen(333, 201, 3, 40), lowc):
        for thr in (3, 8, 10, 14, 19):
            ko = O.detect(img, thr, 3)
            kf, st = E.detect(img, thr, 3, 3, 4 | 2)   # fast path, as the kernels run it
            if st[5] == 0:
                seen_clean += 1
                assert same_kps(kf, ko), (img.shape, thr)
            else:
                seen_low += 1
                kord, _ = E.detect(img, thr, 3, 3, 8)  # ordered path
                assert same_kps(kord, ko), (img.shape, thr)
    assert seen_clean >= 6 and seen_low >= 4


def banded(seed, h=240, w=320, band=48, cell=3):
    """texture only in a band at the top: every layer's AGAST points then lie in its first rows, which is what makes
    the `agastPoints.at(0)[n]` indexing of the suppressScaleNonmaxima=false branch stay inside the score matrices"""
    rng = np.random.default_rng(seed)
    img = np.full((h, w), 128, np.uint8)
    blocks = (np.kron(rng.integers(0, 2, (band // cell + 1, w // cell + 1)), np.ones((cell, cell))) * 180 + 30).astype(np.uint8)
    img[4:4 + band, :] = blocks[:band, :w]
    return img


def test_ordered_path_no_scale_nms_branch_equals_oracle():
    """suppressScaleNonmaxima=false with several layers (brisk-scale-space.cc:131-170, `at(0)` quirk): defined inputs
    bit-equal to the oracle, undefined inputs (at() would throw / reads outside a matrix) reported as such by both."""
    for seed, cell, octaves, thr in ((0, 3, 2, 60), (1, 4, 3, 60), (2, 2, 1, 45), (3, 4, 3, 12)):
        img = banded(seed, cell=cell)
        ko = O.detect(img, thr, octaves, suppress_scale_nonmaxima=False)
        ke, _ = E.detect(img, thr, octaves, shuffle_seed=5, jacobi=16)
        assert ko is not None and len(ko) > 300 and len(set(ko["size"])) >= 2
        assert same_kps(ke, ko), (seed, cell, octaves, thr)
    img = synth.gen(320, 240, 3, 30)
    assert O.detect(img, 60, 2, suppress_scale_nonmaxima=False) is None
    assert E.detect(img, 60, 2, jacobi=16)[0] is None


def provided_keypoints(img, thr, octaves, margin, seed=0, n_extra=40):
    """a provided-keypoint list for ComputeScale: detected keypoints plus random ones (non-integral coordinates, other
    class ids), all at least `margin` pixels above the bottom border so that the reference's linear map addressing
    (brisk-layer.cc:110-115) stays inside every layer"""
    rng = np.random.default_rng(seed)
    h, w = img.shape
    k = O.detect(img, thr, octaves)
    extra = np.zeros(n_extra, O.KP)
    extra["x"] = rng.uniform(0, w, n_extra).astype(np.float32)
    extra["y"] = rng.uniform(0, h - margin, n_extra).astype(np.float32)
    extra["size"] = 9
    extra["angle"] = 33
    extra["class_id"] = rng.integers(0, 1000, n_extra)
    k = np.concatenate([k[k["y"] < h - margin], extra])
    k["class_id"][::3] = 77
    return k


def test_compute_scale_walk_equals_oracle():
    """BriskFeatureDetector::ComputeScale (provided keypoints): the sequential walk of the engine (same function as
    the kernel's lane) against the oracle - all three branches (suppress / one layer / no scale NMS), the
    detect-on-empty-layer case, the empty list, and an input the reference has no defined result for."""
    img = synth.gen(320, 240, 9, 30)
    for thr, octaves, suppress in ((60, 3, True), (60, 0, True), (60, 2, False), (25, 2, True), (8, 1, True)):
        k = provided_keypoints(img, max(thr, 30), 3, 70, seed=thr)
        ko = O.compute_scale(img, k, thr, octaves, suppress)
        ke = E.compute_scale(img, k, thr, octaves, suppress)
        assert ko is not None and len(ko) > 100, (thr, octaves, suppress)
        assert same_kps(ke, ko), (thr, octaves, suppress, len(ke), len(ko))
        assert set(ko["class_id"]) >= {77}
    # only points near the top-left corner: the upper layers admit none of them and detect instead (lower threshold 0)
    few = np.zeros(3, O.KP)
    few["x"], few["y"], few["size"] = [4, 6.5, 5], [4, 5, 7.25], 12
    ko = O.compute_scale(img, few, 60, 3)
    assert ko is not None and len(ko) > 50 and same_kps(E.compute_scale(img, few, 60, 3), ko)
    # empty list = detection with lower threshold 0
    ko = O.compute_scale(img, few[:0], 60, 2)
    assert len(ko) > 100 and same_kps(E.compute_scale(img, few[:0], 60, 2), ko)
    # a point in the last admitted rows of a layer: the reference reads beyond the image
    bad = np.zeros(1, O.KP)
    bad["x"], bad["y"] = 100, 236.5
    assert O.compute_scale(img, bad, 60, 3) is None and E.compute_scale(img, bad, 60, 3) is None


def test_compute_scale_phases_are_order_free():
    """What lets k_cs_admit / _scores / _refine run ComputeScale one lane per (layer, provided point): the walk's three phases -
    threshold-0 touches, the provided lists' scores, the per-point refinement - executed as separate passes with the items of
    every pass in a SHUFFLED order (three seeds) give the oracle's result: same keypoints, same order, same bits; the inputs
    without a defined result are recognised, a layer without admitted points is handed to the walk."""
    img = synth.gen(320, 240, 9, 30)
    rng = np.random.default_rng(12)
    for thr, octaves, suppress in ((60, 3, True), (60, 0, True), (60, 2, False), (25, 2, True), (8, 1, True), (40, 3, True)):
        k = provided_keypoints(img, max(thr, 30), 3, 70, seed=thr)
        # many points on few pixels: touches and refinements of different points meet on the same cache entries
        extra = np.zeros(300, O.KP)
        extra["x"] = rng.uniform(60, 110, 300).astype(np.float32)
        extra["y"] = rng.uniform(60, 100, 300).astype(np.float32)
        extra["size"], extra["angle"], extra["class_id"] = 12, -1, 5
        k = np.concatenate([k, extra])
        ko = O.compute_scale(img, k, thr, octaves, suppress)
        for seed in (1, 2, 3):
            ke = E.compute_scale_phased(img, k, thr, octaves, suppress, seed)
            if ko is None:
                assert ke is None, (thr, octaves, suppress, seed)
            else:
                assert not isinstance(ke, str) and same_kps(ke, ko), (thr, octaves, suppress, seed, None if ke is None else len(ke), len(ko))
    few = np.zeros(3, O.KP)
    few["x"], few["y"], few["size"] = [4, 6.5, 5], [4, 5, 7.25], 12
    assert E.compute_scale_phased(img, few, 60, 3) == "walk"
    bad = np.zeros(1, O.KP)   # one layer, a point in its last admitted row: the ring of the linear offset leaves the image
    bad["x"], bad["y"] = 100, 237
    assert O.compute_scale(img, bad, 60, 0) is None and E.compute_scale_phased(img, bad, 60, 0) is None


def test_describe_box_that_ends_in_the_last_column():
    """SmoothedIntensity's displaced bottom corner (brisk-descriptor-extractor.cc:453) sits one column right of the box;
    for a keypoint on the border limit that is column `cols`, which the reference's linear address turns into the first
    pixel of the next row.  Widths whose rows the engine pads (426, 333): keypoints packed against the right border."""
    oext = O.Extractor()
    pat = E.Pattern()
    _, size_list, _ = pat.tables()
    for w, h in ((426, 320), (333, 201)):
        img = synth.gen(w, h, 12, 50)
        k = np.zeros(400, O.KP)
        rng = np.random.default_rng(w)
        k["size"] = rng.uniform(8.0, 14.0, len(k)).astype(np.float32)
        k["y"] = rng.uniform(40, h - 40, len(k)).astype(np.float32)
        k["angle"] = -1
        for i in range(len(k)):
            border = size_list[pat.scale_index(k["size"][i])]
            k["x"][i] = np.float32(w - border) - np.float32(rng.uniform(0.0, 1.2))
        ko, do = oext.compute(img, k)
        ke, de = pat.describe(img, k)
        assert len(ko) > 100 and same_kps(ke, ko) and np.array_equal(de, do)


def test_division_by_multiplication_is_exact():
    """k_describe divides the weighted box sum by scaling2 with a host-computed magic multiplier (brisk_div_magic /
    brisk_div_by_magic, Hacker's Delight 10-1): equal to C's truncating division for every scaling2 of both built-in
    patterns (and of small / large pattern scales) on random and edge numerators, and for assorted divisors."""
    import ctypes as C
    rng = np.random.default_rng(3)
    nums = np.concatenate([rng.integers(-2**31, 2**31, 20000, dtype=np.int64), rng.integers(0, 1 << 30, 20000, dtype=np.int64)]).astype(np.int32)
    L = E.lib()
    for version, scale in ((2, 1.0), (1, 1.0), (2, 0.4), (2, 3.0)):
        p = E.Pattern(version, scale)
        assert L.emul_div_magic_mismatches(p._h, nums.ctypes.data_as(C.c_void_p), len(nums)) == 0, (version, scale)
    for d in (2, 3, 7, 4095, 4096, 4097, 65535, 65536, 1 << 30, (1 << 30) + 1, 2**31 - 1, 12345677):
        assert L.emul_div_magic_mismatches_d(d, nums.ctypes.data_as(C.c_void_p), len(nums)) == 0, d


def test_two_sided_box_sum_equals_the_one_sided_one(golden_ast):
    """k_describe's lane pairs (round 5) compute a sample's weighted box sum from its two sides - six numbers each
    (brisk_box_side), combined on the lane that owns the sample (brisk_box_acc_pair, as the even and as the odd lane).  Every
    sample the emulator describes goes through both forms, on 32-bit and on 24-bit integral samples: a golden image, a
    1080p frame, and frames whose boxes end in the last column (displaced corner pixels from the frame)."""
    L = E.lib()
    L.emul_box_pair_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]
    L.emul_box_pair_stats.restype = None
    c0, m0 = C.c_long(), C.c_long()
    L.emul_box_pair_stats(C.byref(c0), C.byref(m0))
    pat = E.Pattern()
    for img in (golden_ast[0]["image"], synth.frame_1080p(11)):
        k = O.detect(img, 60, 4)
        assert len(k) > 50
        pat.describe(img, k)
    _, size_list, _ = pat.tables()
    for w, h in ((426, 320), (333, 201)):
        img = synth.gen(w, h, 12, 50)
        k = np.zeros(400, O.KP)
        rng = np.random.default_rng(w)
        k["size"] = rng.uniform(8.0, 14.0, len(k)).astype(np.float32)
        k["y"] = rng.uniform(40, h - 40, len(k)).astype(np.float32)
        k["angle"] = -1
        for i in range(len(k)):
            border = size_list[pat.scale_index(k["size"][i])]
            k["x"][i] = np.float32(w - border) - np.float32(rng.uniform(0.0, 1.2))
        pat.describe(img, k)
    c1, m1 = C.c_long(), C.c_long()
    L.emul_box_pair_stats(C.byref(c1), C.byref(m1))
    assert c1.value - c0.value > 500000, c1.value - c0.value
    assert m1.value == 0 and m0.value == 0
