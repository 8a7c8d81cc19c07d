"""GPU tests added in round 3 (run with -m gpu): launch configurations of the batch path that the earlier suites never
reached at 1080p - tie workgroups that own 2..7 layers (batches of 65..255 frames), the 32-blocks-per-frame branch of
the score-block / classification kernels - and the re-built descriptor kernel (tail points of the pattern batched per
wave) on every pattern size."""
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


def _run_batch_and_compare(B, frames_distinct, n, thr, octaves, w, h, every_slot=True):
    """n slots filled round-robin from the distinct frames; every slot (or the first, middle and last ones of each
    distinct frame) is compared with the oracle: keypoints as detected, keypoints as described, descriptors."""
    import torch
    nd = len(frames_distinct)
    X = O.Extractor()
    want = []
    for img in frames_distinct:
        ko = O.detect(img, thr, octaves)
        want.append((ko,) + X.compute(img, ko))
    stack = np.stack([frames_distinct[f % nd] for f in range(n)])
    d = torch.from_numpy(stack).cuda()
    cap = max(len(k[0]) for k in want)
    ctx = B.Context(0, 4 * cap, 2 * cap) if cap > 12000 else B.Context(0)
    ext = B.BriskDescriptorExtractor(context=ctx)
    for rep in range(2):  # the second batch runs on the dirty workspace of the first
        ctx.detect_describe_batch(ext, d.data_ptr(), n, w, h, w * h, w, thr, octaves, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert ctx.batch_status(n) == 0
        slots = range(n) if every_slot else sorted(set(list(range(nd)) + list(range(n // 2, n // 2 + nd)) + list(range(n - nd, n))))
        for f in slots:
            ko, ko2, do = want[f % nd]
            kd, _ = ctx.batch_download(f, described=False)
            kg, dg = ctx.batch_download(f, described=True)
            assert same_kps(kd, ko), (n, rep, f, explain(kd, ko))
            assert same_kps(kg, ko2), (n, rep, f, explain(kg, ko2))
            assert np.array_equal(dg, do), (n, rep, f)
    ctx.close()
    return sum(len(k[1]) for k in want)


@pytest.mark.parametrize("n", [96, 128, 200, 255, 65])
def test_1080p_batches_between_65_and_255_frames(B, n):
    """lpw (layers per tie workgroup) = 4, 4, 8, 8 and 3 at BASELINE config-2 content (round 4's rule: the fewest layers per
    workgroup whose ticket count fits 256; 12-wave workgroups from 192 frames on); eight distinct frames, every
    slot of the batch - the last one included - bit-equal to the oracle."""
    distinct = [synth.frame_1080p(300 + s) for s in range(8)]
    total = _run_batch_and_compare(B, distinct, n, 80, 4, 1920, 1080)
    assert total > 6000


def test_1080p_dense_threshold_30_at_128_frames(B):
    """A dense regime (threshold 30: ~30 k detections and as many ties per frame) at 128 frames: the tie workgroups own
    four layers each and take the chunked sort."""
    distinct = [synth.frame_1080p(400 + s) for s in range(3)]
    total = _run_batch_and_compare(B, distinct, 128, 30, 4, 1920, 1080, every_slot=False)
    assert total > 30000


@pytest.mark.parametrize("version,points", [(2, 66), (1, 60)])
def test_describe_many_keypoints_per_wave_patterns(B, version, points):
    """k_describe on runs of keypoints per wave: V2 (66 points: the two tail points go through the batched tail pass)
    and V1 (60 points: no tail); provided keypoints without and with orientation, all flags."""
    img = synth.frame_1080p(77)
    ko = O.detect(img, 60, 4)
    assert len(ko) > 1500
    for rot, scl in ((True, True), (False, True), (True, False)):
        ext = B.BriskDescriptorExtractor(rot, scl, version)
        assert ext.points == points
        X = O.Extractor(rot, scl, version)
        k2, d2 = ext.compute(img, ko)
        ko2, do = X.compute(img, ko)
        assert same_kps(k2, ko2), explain(k2, ko2)
        assert np.array_equal(d2, do)
        # again with the estimated angles provided (angle != -1 skips the orientation pass)
        k3, d3 = ext.compute(img, ko2)
        ko3, do3 = X.compute(img, ko2)
        assert same_kps(k3, ko3) and np.array_equal(d3, do3)


def test_padded_pitch_with_odd_width_is_not_read_in_place(B):
    """Width 1000 at a caller pitch of 1024 with garbage in the pad columns, in a buffer that ends with the last pixel of
    the last frame: the engine must take its private layer-0 copy (reading in place would use the pad bytes and read up to
    63 bytes past the buffer) and match the oracle."""
    import torch
    w, h, pitch, n = 1000, 600, 1024, 3
    imgs = [synth.gen(w, h, 900 + i, 120) for i in range(n)]
    nbytes = (n - 1) * pitch * h + (h - 1) * pitch + w
    buf = np.random.default_rng(5).integers(0, 256, nbytes, dtype=np.uint8)
    for f, img in enumerate(imgs):
        for y in range(h):
            o = f * pitch * h + y * pitch
            buf[o:o + w] = img[y]
    d = torch.from_numpy(buf).cuda()
    assert d.data_ptr() % 16 == 0
    ctx = B.Context(0)
    ext = B.BriskDescriptorExtractor(context=ctx)
    ctx.detect_describe_batch(ext, d.data_ptr(), n, w, h, pitch * h, pitch, 70, 4, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert ctx.batch_status(n) == 0
    X = O.Extractor()
    for f, img in enumerate(imgs):
        ko = O.detect(img, 70, 4)
        ko2, do = X.compute(img, ko)
        kd, _ = ctx.batch_download(f, described=False)
        kg, dg = ctx.batch_download(f, described=True)
        assert same_kps(kd, ko), (f, explain(kd, ko))
        assert same_kps(kg, ko2) and np.array_equal(dg, do), f
    ctx.close()


def test_detect_then_compute_on_the_same_buffer_uploads_once_only_on_the_callers_word(B):
    """The drop-in classes call detect() and compute() on the same image.  compute() always uses the pixels it is given
    (the reference's does): an image changed in place between the two calls yields the descriptors of the CHANGED image.
    Only a caller that states "same buffer, unchanged" (same_image / brisk_hip_describe_same_image) skips the second
    upload; the statement is ignored for another buffer.  Results equal the oracle's throughout."""
    ctx = B.Context(0)
    det = B.BriskFeatureDetector(70, 4, context=ctx)
    ext = B.BriskDescriptorExtractor(context=ctx)
    X = O.Extractor()
    for name, img in (("vga", synth.frame_vga(3)), ("odd_width", synth.gen(1001, 587, 12, 150)), ("1080p", synth.frame_1080p(9)),
                      ("narrow", synth.gen(100, 90, 4, 20))):
        img = np.ascontiguousarray(img)
        h0 = ctx.debug_image_reuse()
        k = det.detect(img)
        k2, d2 = ext.compute(img, k)
        assert ctx.debug_image_reuse() == h0, name           # no assumption without the caller's word
        ko = O.detect(img, 70, 4)
        ko2, do = X.compute(img, ko)
        assert same_kps(k, ko) and same_kps(k2, ko2) and np.array_equal(d2, do), name
        k = det.detect(img)
        k3, d3 = ext.compute(img, k, same_image=True)        # stated: the device copy is used
        assert ctx.debug_image_reuse() == h0 + 1 and same_kps(k3, ko2) and np.array_equal(d3, do), name
        k3, d3 = ext.compute(img, k, same_image=True)        # again: still on the device
        assert ctx.debug_image_reuse() == h0 + 2 and same_kps(k3, ko2) and np.array_equal(d3, do), name
        # the image changes in place between detect and compute - by a few pixels only (what a sampled hash would miss)
        k = det.detect(img)
        img[40:43, 60:63] = 255 - img[40:43, 60:63]
        k4, d4 = ext.compute(img, k)
        assert ctx.debug_image_reuse() == h0 + 2, name
        ko4, do4 = X.compute(img, k)
        assert same_kps(k4, ko4) and np.array_equal(d4, do4), name
        # the statement about another buffer (same content): not the detect call's image, uploaded
        k = det.detect(img)
        other = img.copy()
        k5, d5 = ext.compute(other, k, same_image=True)
        ko5, do5 = X.compute(other, k)
        assert ctx.debug_image_reuse() == h0 + 2 and same_kps(k5, ko5) and np.array_equal(d5, do5), name
    ctx.close()


@pytest.mark.parametrize("nbu,nbv,mx", [(8, 6, 480), (1, 4, 300), (16, 16, 256), (3, 2, 100000), (1, 2, 100000)])
def test_key_point_bucketing_vs_oracle(B, nbu, nbv, mx):
    """KeyPointBucketing (key-point-bucketing-inl.h:40-112) as a post-filter of the detector: host-buffer call and a batch
    (bucketing inside, descriptors of the kept keypoints), against the oracle's restatement (parity unpinned)."""
    import torch
    imgs = [synth.frame_1080p(500 + i) for i in range(3)]
    ctx = B.Context(0)
    det = B.BriskFeatureDetector(60, 4, context=ctx, maxNumKpt=mx, numBucketsU=nbu, numBucketsV=nbv)
    X = O.Extractor()
    want = []
    for img in imgs:
        ko = O.detect(img, 60, 4)
        kb = O.key_point_bucketing(ko, 1080, 1920, mx, nbu, nbv)
        assert kb is not None and (len(kb) < len(ko) or mx > len(ko))
        want.append((kb,) + X.compute(img, kb))
    kg = det.detect(imgs[0])
    assert same_kps(kg, want[0][0]), explain(kg, want[0][0])
    # the object's settings travelled with its call: the context's own settings are untouched (still off)
    assert same_kps(B.BriskFeatureDetector(60, 4, context=ctx).detect(imgs[0]), O.detect(imgs[0], 60, 4))
    ctx.set_bucketing(nbu, nbv, mx)  # context setting: what the batch path uses
    assert same_kps(det.detect(imgs[0]), want[0][0])
    ext = B.BriskDescriptorExtractor(context=ctx)
    d = torch.from_numpy(np.stack(imgs)).cuda()
    ctx.detect_describe_batch(ext, d.data_ptr(), 3, 1920, 1080, 1920 * 1080, 1920, 60, 4, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert ctx.batch_status(3) == 0
    for f in range(3):
        kd, _ = ctx.batch_download(f, described=False)
        k2, d2 = ctx.batch_download(f, described=True)
        assert same_kps(kd, want[f][0]), (f, explain(kd, want[f][0]))
        assert same_kps(k2, want[f][1]) and np.array_equal(d2, want[f][2]), f
    # off again: plain detection
    ctx.set_bucketing(0, 0, 0)
    assert same_kps(B.BriskFeatureDetector(60, 4, context=ctx).detect(imgs[0]), O.detect(imgs[0], 60, 4))
    with pytest.raises(B.BriskHipError):
        B.BriskFeatureDetector(60, 4, context=ctx, maxNumKpt=10, numBucketsU=4000, numBucketsV=2).detect(imgs[0])
    ctx.close()


def test_16bit_image_functions_vs_oracle(B):
    """Halfsample16 / Twothirdsample16 / IntegralImage16 on the device, bit-equal to the oracle's restatements (which are
    pinned by per-pixel readings of the reference arithmetic); sizes with a re-done last block, saturating values, the
    unscaled tail columns of the integral, and the sizes the reference cannot handle."""
    ctx = B.Context(0)
    rng = np.random.default_rng(21)
    for (h, w) in ((480, 752), (37, 53), (1080, 1920), (19, 16), (21, 13), (65, 130)):
        img = rng.integers(0, 65536, (h, w), dtype=np.uint16)
        img[::5, ::3] = 65535
        for fn, ofn in ((ctx.halfsample16, O.halfsample16), (ctx.twothirdsample16, O.twothirdsample16)):
            want = ofn(img)
            if want is None:   # the reference's loop writes nothing at this width: neither does the engine (dst untouched)
                assert not fn(img).any(), (h, w)
            else:
                assert np.array_equal(fn(img), want), (h, w)
        assert np.array_equal(ctx.integral_image16(img).view(np.uint32), O.integral16(img).view(np.uint32)), (h, w)
    small = rng.integers(0, 65536, (20, 11), dtype=np.uint16)
    assert O.halfsample16(small) is None and O.twothirdsample16(small) is None
    for fn in (ctx.halfsample16, ctx.twothirdsample16):
        assert not fn(small).any()   # nothing written, no error (as the reference)
    assert np.array_equal(ctx.integral_image16(small).view(np.uint32), O.integral16(small).view(np.uint32))
    ctx.close()


def test_describe_work_queues_beyond_2048_frames(B):
    """More than 8 x 256 frames in one launch: k_describe's frames are then dealt to more than eight queues (one queue
    holds at most 256 frames).  2100 small frames, a sample of slots - first, last, around the queue boundaries - against the
    oracle; then 5 and 9 frames on the same context (one queue / eight queues with almost empty groups)."""
    import torch
    w, h = 192, 128
    distinct = [synth.gen(w, h, 700 + s, 24) for s in range(7)]
    X = O.Extractor()
    want = []
    for img in distinct:
        ko = O.detect(img, 50, 2)
        want.append((ko,) + X.compute(img, ko))
    assert sum(len(k[1]) for k in want) > 50
    ctx = B.Context(0, 4096, 1024)
    ext = B.BriskDescriptorExtractor(context=ctx)
    for n in (2100, 5, 9):
        stack = np.stack([distinct[f % 7] for f in range(n)])
        d = torch.from_numpy(stack).cuda()
        ctx.detect_describe_batch(ext, d.data_ptr(), n, w, h, w * h, w, 50, 2, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert ctx.batch_status(n) == 0
        slots = sorted(set([f for f in (0, 1, 7, 8, 9, 255, 256, 257, 1023, 1024, 2047, 2048, 2049, 2090, 2098, 2099) if f < n] + list(range(min(n, 12)))))
        for f in slots:
            ko, ko2, do = want[f % 7]
            kd, _ = ctx.batch_download(f, described=False)
            kg, dg = ctx.batch_download(f, described=True)
            assert same_kps(kd, ko), (n, f, explain(kd, ko))
            assert same_kps(kg, ko2) and np.array_equal(dg, do), (n, f)
    ctx.close()


def test_random_sizes_thresholds_octaves_fuzz():
    """tools/soak.py callspace in a process of its own (it forks its oracle workers before HIP is initialised): 250 random cases -
    image sides 9 ... 1100, thresholds 1 ... 140 (ordered and fast path), 0 ... 6 octaves, five content kinds incl. all-tie
    block images, host calls and small device batches - each bit-equal to the oracle (keypoints before and after
    compute(), descriptors).  Cases the default workspace answers with BRISK_HIP_ERR_CAPACITY are repeated on a larger one."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak.py"), "callspace", "250", "11"], capture_output=True, text=True, timeout=900)
    tail = [ln for ln in r.stdout.splitlines() if ln.startswith(("callspace", "ERROR", "MISMATCH"))]
    print("\n".join(tail[-20:]), r.stderr[-2000:])
    assert r.returncode == 0 and tail and tail[-1].startswith("callspace: 250 cases") and " 0 bad" in tail[-1]


@pytest.mark.parametrize("tool,cases", [("options", 200), ("matcher", 200)])
def test_option_and_matcher_fuzz(tool, cases):
    """tools/soak.py options (options of the two classes: masks, suppressScaleNonmaxima = false incl. the inputs without a defined
    result, uniformity / bucketing parameters, invariance flags, both pattern versions at three pattern scales) and
    tools/soak.py matcher (the matcher: set sizes incl. empty ones, 1 ... 6 train images, descriptor lengths 16 ... 224, masks,
    k, radii), each in a process of its own, every case equal to the oracle"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak.py"), tool, str(cases), "13"], capture_output=True, text=True, timeout=900)
    tail = [ln for ln in r.stdout.splitlines() if ln.startswith((tool, "ERROR", "MISMATCH"))]
    print("\n".join(tail[-20:]), r.stderr[-2000:])
    assert r.returncode == 0 and tail and tail[-1].startswith("%s: %d cases" % (tool, cases)) and " 0 bad" in tail[-1]


@pytest.mark.parametrize("scale", [0.7, 1.3, 2.5])
def test_generated_kernel_at_other_pattern_scales_has_other_descriptor_lengths(B, scale):
    """briskV1 at patternScale != 1: generateKernel's pair thresholds are not scaled (brisk-descriptor-extractor.cc:338),
    so the extractor has 128-byte (0.7) or 48-byte (1.3) descriptors; host call and device batch, then the default
    extractor again on the same context (the descriptor rows of the workspace only ever grow)"""
    import torch
    img = synth.gen(640, 480, 31, 60)
    ko = O.detect(img, 60, 3)
    X = O.Extractor(version=1, pattern_scale=scale)
    ko2, do = X.compute(img, ko)
    assert X.strings == {0.7: 128, 1.3: 48, 2.5: 16}[scale] and len(ko2) > (100 if scale < 2 else 20)   # (2.5: > 1024 long pairs, read from global memory)
    ctx = B.Context(0)
    ext = B.BriskDescriptorExtractor(version=1, patternScale=scale, context=ctx)
    assert ext.descriptorSize() == X.strings
    kg = B.BriskFeatureDetector(60, 3, context=ctx).detect(img)
    kg2, dg = ext.compute(img, kg)
    assert same_kps(kg2, ko2) and dg.shape == do.shape and np.array_equal(dg, do)
    d = torch.from_numpy(np.stack([img, img[::-1].copy()])).cuda()
    ctx.detect_describe_batch(ext, d.data_ptr(), 2, 640, 480, 640 * 480, 640, 60, 3, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert ctx.batch_status(2) == 0
    kb, db = ctx.batch_download(0, described=True, strings=X.strings)
    assert same_kps(kb, ko2) and np.array_equal(db, do)
    ko3, do3 = O.Extractor().compute(img, ko)
    kg3, dg3 = B.BriskDescriptorExtractor(context=ctx).compute(img, kg)
    assert same_kps(kg3, ko3) and np.array_equal(dg3, do3)
    # the matcher takes these lengths too
    m = B.BruteForceMatcher(context=ctx)
    m.add([dg])
    got = m.knnMatch(dg[:50], 2)
    want = O.match_knn(do[:50], [do], 2)
    assert [[(x["trainIdx"], x["distance"]) for x in r] for r in got] == [[(x["trainIdx"], x["distance"]) for x in r] for r in want]
    ctx.close()
