"""GPU tests added in round 5 (run with -m gpu): the pattern / context device guard, the explicit integral-format knob of
the C ABI, the descriptor pitch brisk_hip_batch_results reports after a packed host describe call, and the restructured
run loop of k_describe (tickets and next-run records read behind the run's last gathers) on batch sizes that leave waves
with zero, one and many runs."""
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


def test_pattern_of_another_device_is_refused(B):
    """An extractor built for GPU A used with a context of GPU B would make B's kernels dereference A's tables: every entry
    that takes a pattern refuses the pair (BRISK_HIP_ERR_ARG, a message that says what to do).  One GPU per box: the
    handle's device field is forged (brisk_hip_debug_forge_pattern_device), restored, and the same calls succeed."""
    import torch
    ctx = B.Context(0)
    ext = B.BriskDescriptorExtractor(context=ctx)
    img = synth.frame_vga(3)
    kps = B.BriskFeatureDetector(70, 4, context=ctx).detect(img)
    d = torch.from_numpy(np.stack([img, img])).cuda()
    pinned = torch.from_numpy(np.stack([img, img])).pin_memory()
    stream = torch.cuda.current_stream().cuda_stream
    L = ctx._L
    assert L.brisk_hip_debug_forge_pattern_device(ext._h, 5) == 0
    for call in (lambda: ext.compute(img, kps),
                 lambda: ctx.detect_describe_batch(ext, d.data_ptr(), 2, 640, 480, 640 * 480, 640, 70, 4, stream),
                 lambda: ctx.detect_describe_batch_host(ext, pinned.data_ptr(), 2, 640, 480, 640 * 480, 640, 70, 4)):
        with pytest.raises(B.BriskHipError) as e:
            call()
        assert e.value.code == 1 and "device 5" in str(e.value) and "device 0" in str(e.value), str(e.value)
    assert L.brisk_hip_debug_forge_pattern_device(ext._h, -1) == 0
    k2, desc = ext.compute(img, kps)
    ko = O.detect(img, 70, 4)
    ko2, do = O.Extractor().compute(img, ko)
    assert same_kps(k2, ko2) and np.array_equal(desc, do)
    ctx.detect_describe_batch(ext, d.data_ptr(), 2, 640, 480, 640 * 480, 640, 70, 4, stream)
    torch.cuda.synchronize()
    assert ctx.batch_status(2) == 0
    kg, dg = ctx.batch_download(1, described=True)
    assert same_kps(kg, ko2) and np.array_equal(dg, do)
    ctx.close()


def test_explicit_integral_format(B):
    """brisk_hip_set_integral_format: U32 / U24 hold for every call whatever the previous batch looked like (AUTO follows
    its candidate density: test_gpu_round4), descriptor-only host calls included; every result equals the oracle."""
    import torch
    imgs = [synth.frame_1080p(840 + i) for i in range(2)]
    d = torch.from_numpy(np.stack(imgs)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    ctx = B.Context(0, max_candidates=262144, max_keypoints=65536)
    ext = B.BriskDescriptorExtractor(context=ctx)
    X = O.Extractor()
    want = {}
    for fmt, thrs in ((32, (80, 80)), (24, (30, 30, 80)), (0, (80, 80))):
        ctx.set_integral_format(fmt)
        for thr in thrs:
            ctx.detect_describe_batch(ext, d.data_ptr(), 2, 1920, 1080, 1920 * 1080, 1920, thr, 4, stream)
            torch.cuda.synchronize()
            assert ctx.batch_status(2) == 0
            bits = ctx.debug_integral_bits(0)
            assert bits == (fmt if fmt else 24), (fmt, thr, bits)
            if thr not in want:
                ko = O.detect(imgs[1], thr, 4)
                want[thr] = X.compute(imgs[1], ko)
            kg, dg = ctx.batch_download(1, described=True)
            assert same_kps(kg, want[thr][0]) and np.array_equal(dg, want[thr][1]), (fmt, thr)
        # a descriptor-only host call: u32 unless 24 is asked for
        ko = O.detect(imgs[0], 80, 4)
        k2, desc = ext.compute(imgs[0], ko)
        assert ctx.debug_integral_bits(0) == (24 if fmt == 24 else 32)
        k2o, do = X.compute(imgs[0], ko)
        assert same_kps(k2, k2o) and np.array_equal(desc, do)
    with pytest.raises(B.BriskHipError):
        ctx.set_integral_format(16)
    ctx.close()


def test_batch_results_pitch_after_a_packed_host_describe(B):
    """brisk_hip_describe with packed destination rows writes slot 0's descriptor rows at that pitch (48): what
    brisk_hip_batch_results reports must be THAT pitch (round-4 advisor finding: it reported the workspace's 64), so that
    the device rows can go straight into brisk_hip_match_knn_device; after a batch it is the workspace pitch again."""
    import torch
    img = synth.frame_vga(11)
    ctx = B.Context(0)
    ext = B.BriskDescriptorExtractor(context=ctx)
    kps = B.BriskFeatureDetector(60, 4, context=ctx).detect(img)
    k2, desc = ext.compute(img, kps)
    assert desc.shape[1] == 48 and len(k2) > 200
    vp = C.c_void_p
    d_det, d_n, d_kd, d_kp, d_desc = vp(), vp(), vp(), vp(), vp()
    cstride, cap, pitch = C.c_int(), C.c_int(), C.c_int()
    ctx.check(ctx._L.brisk_hip_batch_results(ctx._h, C.byref(d_det), C.byref(d_n), C.byref(cstride), C.byref(d_kd), C.byref(d_kp),
                                             C.byref(d_desc), C.byref(cap), C.byref(pitch)))
    assert pitch.value == 48, pitch.value
    # the device rows against the host copy of the same descriptors (shifted by one row: no trivial zero distances)
    train = torch.from_numpy(np.roll(desc, 1, axis=0).copy()).cuda()
    nq = len(k2)
    out = torch.zeros((nq, 2, 4), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(nq, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ctx.check(ctx._L.brisk_hip_match_knn_device(ctx._h, d_desc.value, nq, pitch.value, train.data_ptr(), nq, 48, 48, 2,
                                                out.data_ptr(), cnt.data_ptr(), stream))
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(B.DMATCH).reshape(nq, 2)
    want = O.match_knn(desc, [np.roll(desc, 1, axis=0).copy()], 2)
    for q in range(nq):
        assert [(m["trainIdx"], m["distance"]) for m in got[q]] == [(m["trainIdx"], m["distance"]) for m in want[q]], q
    assert all(got[q][0]["trainIdx"] == (q + 1) % nq and got[q][0]["distance"] == 0 for q in range(nq))
    # a batch afterwards: rows at the workspace pitch again
    d = torch.from_numpy(img[None]).cuda()
    ctx.detect_describe_batch(ext, d.data_ptr(), 1, 640, 480, 640 * 480, 640, 60, 4, stream)
    torch.cuda.synchronize()
    ctx.check(ctx._L.brisk_hip_batch_results(ctx._h, C.byref(d_det), C.byref(d_n), C.byref(cstride), C.byref(d_kd), C.byref(d_kp),
                                             C.byref(d_desc), C.byref(cap), C.byref(pitch)))
    assert pitch.value == 64
    ctx.close()


@pytest.mark.parametrize("nframes", [1, 2, 7, 9, 40])
def test_describe_run_loop_small_and_odd_batches(B, nframes):
    """k_describe's run loop reads a ticket two runs after it was taken and keeps two record buffers: batches in which most
    waves get no run, exactly one, or an odd number (static dealing below 8 frames, ticket queues from 8 on; 9 and 40
    frames leave queues of different lengths), every slot against the oracle."""
    distinct = [synth.frame_vga(200 + s) for s in range(3)]
    _run_batch_and_compare(B, distinct, nframes, 70, 4, 640, 480, every_slot=True)


def test_describe_runs_of_eight_and_provided_angles(B):
    """dense frames take runs of 8 keypoints per ticket (threshold 30), and keypoints that arrive WITH an angle skip the
    orientation pass: the run's ticket / record requests then travel with the rotated pass's first burst instead."""
    img = synth.frame_1080p(77)
    ko = O.detect(img, 30, 4)
    assert len(ko) > 20000
    X = O.Extractor()
    ctx = B.Context(0, max_candidates=262144, max_keypoints=65536)
    ext = B.BriskDescriptorExtractor(context=ctx)
    for mode in range(3):
        kin = ko.copy()
        if mode == 1:
            kin["angle"] = (np.arange(len(kin)) * 7.3 % 360).astype(np.float32)   # every keypoint brings its angle
        if mode == 2:
            kin["angle"][::3] = 123.5                                               # a third of them do
        kg, dg = ext.compute(img, kin)
        kw, dw = X.compute(img, kin)
        assert same_kps(kg, kw), explain(kg, kw)
        assert np.array_equal(dg, dw), mode
    ctx.close()


def test_compute_scale_one_lane_per_layer_and_point(B):
    """BriskFeatureDetector::ComputeScale (brisk-feature-detector.cc:87-92) off the one-lane walk: the provided points are
    touched, scored and refined one lane per (layer, point) and emitted in (layer, provided) order (k_cs_*).  3 000 and 16 000
    random points on a 1080p frame in all three branches (scale NMS, several layers without it, one layer), class ids kept;
    a list whose points sit in the outer 30 pixels only, so that the small layers admit none of them and DETECT instead (the
    call then stays on the one-lane walk, decided on the device); and the undefined input - each against the oracle, and the
    3 000-point call against the clock (0.45 s on the one-lane walk; the review asked for 5 ms)."""
    import time
    img = synth.frame_1080p(5)
    rng = np.random.default_rng(3)
    ctx = B.Context(0, max_candidates=262144, max_keypoints=65536)

    def points(n, lo_x, hi_x, lo_y, hi_y):
        k = np.zeros(n, B.KEYPOINT)
        k["x"] = rng.uniform(lo_x, hi_x, n).astype(np.float32)
        k["y"] = rng.uniform(lo_y, hi_y, n).astype(np.float32)
        k["x"][::7] = np.floor(k["x"][::7])          # some integral coordinates
        k["size"], k["angle"], k["class_id"] = 12, -1, np.arange(n) * 3 + 1
        return k

    for n, octaves, suppress in ((3000, 4, True), (16000, 4, True), (3000, 3, False), (3000, 0, True), (700, 2, True)):
        k = points(n, 60, 1860, 60, 1020)
        want = O.compute_scale(img, k, 60, octaves, suppress)
        det = B.BriskFeatureDetector(60, octaves, suppress, context=ctx)
        got = det.ComputeScale(img, k)
        if want is None:   # (the no-scale-NMS branch indexes layer i with layer 0's list: undefined on most inputs)
            continue
        assert len(want) > n // 2 and same_kps(got, want), (n, octaves, suppress, explain(got, want))
    k = points(3000, 60, 1860, 60, 1020)
    det = B.BriskFeatureDetector(60, 4, context=ctx)
    det.ComputeScale(img, k)
    t0 = time.perf_counter()
    for _ in range(3):
        det.ComputeScale(img, k)
    dt = (time.perf_counter() - t0) / 3
    assert dt < 0.02, "ComputeScale of 3 000 points took %.1f ms" % (dt * 1e3)
    # points in the outer frame only: layers of scale >= 6 admit none (x / scale - offset < 3) and detect instead
    edge = points(400, 4, 30, 4, 1076)
    want = O.compute_scale(img, edge, 60, 4)
    if want is not None:
        assert same_kps(det.ComputeScale(img, edge), want)
    else:
        with pytest.raises(B.BriskHipError) as ei:
            det.ComputeScale(img, edge)
        assert ei.value.code == 7
    small = synth.gen(320, 240, 9, 30)
    edge2 = points(60, 3.5, 12, 3.5, 200)
    want = O.compute_scale(small, edge2, 40, 3)
    d2 = B.BriskFeatureDetector(40, 3, context=ctx)
    if want is not None:
        assert len(want) > 20 and same_kps(d2.ComputeScale(small, edge2), want), explain(d2.ComputeScale(small, edge2), want)
    else:
        with pytest.raises(B.BriskHipError):
            d2.ComputeScale(small, edge2)
    ctx.close()


@pytest.mark.parametrize("n,thr", [(33, 80), (64, 80), (40, 30)])
def test_two_ties_per_wave_batches_up_to_64_frames(B, n, thr):
    """k_tie_resolve_pair (two raster-adjacent ties per wave, round 5) serves every call of up to 64 frames: 33 and 64
    frames at BASELINE config-2 content (64 x 8 layers = 512 tickets on 256 persistent workgroups), 40 dense frames at
    threshold 30 (~30 k ties per frame: layers beyond the on-chip chunk of 3 072 ranks, so pairs straddle chunk ends and the
    last pair of a chunk has one member) - every slot against the oracle, on a fresh and on a dirty workspace."""
    distinct = [synth.frame_1080p(900 + s) for s in range(3)]
    total = _run_batch_and_compare(B, distinct, n, thr, 4, 1920, 1080, every_slot=(thr != 30))
    assert total > (30000 if thr == 30 else 3000)


def test_two_ties_per_wave_agrees_with_one_tie_per_wave(B):
    """the A / B knob of the tie kernel's forms (BRISK_TR_PAIR is read once per process: both forms run in child processes):
    one 4K frame with 6 octaves (layers of 23 ... 1 026 ties) and an all-tie image give the same bytes either way."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, hashlib, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import ethzasl_brisk_amd as B, synth
h = hashlib.sha256()
det = B.BriskFeatureDetector(80, 6)
k = det.detect(synth.frame_4k(2)); h.update(k.tobytes())
flat = np.full((240, 320), 90, np.uint8); flat[::7, ::5] = 200
det2 = B.BriskFeatureDetector(30, 3)
k2 = det2.detect(flat, capacity=65536); h.update(k2.tobytes())
print(len(k), len(k2), h.hexdigest())
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for knob in ("1", "0"):
        env = dict(os.environ, BRISK_TR_PAIR=knob)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1], outs
    assert int(outs[0].split()[0]) > 3000


@pytest.mark.parametrize("drop", [1, 7])
def test_pair_dealt_gathers_with_odd_pattern_sizes_and_both_image_formats(B, drop):
    """k_describe deals a sample's gathers to lane pairs (round 5): custom patterns of 65 and 59 points make the run's sample
    count odd (the last pair has one member, keypoint boundaries fall on odd lanes), few and many keypoints give runs of 1 ... 8,
    and the integral format is forced to 24 bits (pair form) and to 32 bits (one sample per lane) - the host compute() call on a
    1080p frame, descriptors and kept keypoints bit-equal to the oracle."""
    import ptn
    text = ptn.custom_pattern(3, sigma_factor=1.0, drop_points=drop)
    img = synth.frame_1080p(17)
    kall = O.detect(img, 60, 4)
    assert len(kall) > 1500
    oext = O.Extractor(True, True, pattern_text=text, pattern_scale=1.0)
    ctx = B.Context(0)
    ext = B.BriskDescriptorExtractor(True, True, pattern_text=text, patternScale=1.0, context=ctx)
    for fmt in (24, 32):
        ctx.set_integral_format(fmt)
        for n in (1, 2, 3, 5, 64, 777, len(kall)):
            k = np.ascontiguousarray(kall[:n])
            ko, do = oext.compute(img, k)
            kg, dg = ext.compute(img, k)
            assert same_kps(kg, ko), (drop, fmt, n, explain(kg, ko))
            assert np.array_equal(dg, do), (drop, fmt, n)
    ctx.close()


def test_tie_layers_cut_into_row_bands_agree_with_one_workgroup_per_layer(B):
    """One- and two-frame calls deal every layer's ties to up to eight workgroups by image row (BRISK_TR_BANDS, read once per
    process: child processes).  A band's first rows wait for pending ties of the band above through the score-state map: an
    image of nothing but ties (every boundary is crossed by chains), a 4K frame with six octaves, a VGA frame, a 1080p frame at
    threshold 60 and one at threshold 30 (layers beyond the on-chip chunk are not cut, the ones above them are) give the same bytes with 1, 2, 3, 4 and 8 bands - and the suite's oracle comparisons run at the default (8 for one frame)."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, hashlib, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import ethzasl_brisk_amd as B, synth
h = hashlib.sha256()
flat = np.full((240, 320), 90, np.uint8); flat[::7, ::5] = 200
k0 = B.BriskFeatureDetector(30, 3).detect(flat, capacity=65536); h.update(k0.tobytes())
k1 = B.BriskFeatureDetector(80, 6).detect(synth.frame_4k(2)); h.update(k1.tobytes())
k2 = B.BriskFeatureDetector(70, 4).detect(synth.frame_vga(1)); h.update(k2.tobytes())
k3 = B.BriskFeatureDetector(60, 4).detect(synth.frame_1080p(3), capacity=65536); h.update(k3.tobytes())
# threshold 30: the lower layers hold more ties than the on-chip chunk (one workgroup takes all of them), the upper ones are cut
big = B.Context(0, max_candidates=262144, max_keypoints=65536)
k4 = B.BriskFeatureDetector(30, 4, context=big).detect(synth.frame_1080p(4), capacity=65536); h.update(k4.tobytes())
print(len(k0), len(k1), len(k2), len(k3), len(k4), h.hexdigest())
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for bands in ("1", "2", "3", "4", "8"):
        env = dict(os.environ, BRISK_TR_BANDS=bands)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert len(set(outs)) == 1, outs
    counts = [int(v) for v in outs[0].split()[:5]]
    assert counts[1] > 3000 and counts[3] > 2500 and counts[4] > 20000, counts
