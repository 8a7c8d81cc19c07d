"""ethzasl_brisk_amd - MI355X-native BRISK detect+describe engine.

Python-side mirror of the two reference classes over the C ABI (include/brisk_hip.h).  It exists for
the test-suite and bench.py; the product is libbrisk_hip.so plus the C++ host classes in
include/brisk/.  There is no CPU fallback: every compute call needs the HIP library and a GPU.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# BRISK_HIP_LIB: tuning experiments only (an alternative build of the same library, e.g. with another -D knob)
LIB_PATH = os.environ.get("BRISK_HIP_LIB") or os.path.join(HERE, "libbrisk_hip.so")

KEYPOINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])

ERRORS = {1: "BRISK_HIP_ERR_ARG", 2: "BRISK_HIP_ERR_NO_DEVICE", 3: "BRISK_HIP_ERR_HIP", 4: "BRISK_HIP_ERR_CAPACITY",
          5: "BRISK_HIP_ERR_THRESHOLD", 6: "BRISK_HIP_ERR_PATTERN", 7: "BRISK_HIP_ERR_UNSUPPORTED"}

# every symbol include/brisk_hip.h declares
ABI_SYMBOLS = [
    "brisk_hip_create", "brisk_hip_destroy", "brisk_hip_last_error", "brisk_hip_set_capacity",
    "brisk_hip_device_count", "brisk_hip_pattern_create", "brisk_hip_pattern_create_from_text",
    "brisk_hip_pattern_destroy", "brisk_hip_pattern_descriptor_size", "brisk_hip_pattern_points",
    "brisk_hip_pattern_tables", "brisk_hip_detect", "brisk_hip_describe", "brisk_hip_detect_describe_batch",
    "brisk_hip_detect_batch", "brisk_hip_batch_results", "brisk_hip_batch_download", "brisk_hip_batch_status",
    "brisk_hip_profile_enable", "brisk_hip_profile_stages",
    "brisk_hip_profile_stage_name", "brisk_hip_profile_read", "brisk_hip_set_bucketing", "brisk_hip_halfsample16", "brisk_hip_twothirdsample16",
    "brisk_hip_integral_image16",
    "brisk_hip_set_streams", "brisk_hip_profile_frames_per_launch",
    "brisk_hip_match_knn", "brisk_hip_match_radius", "brisk_hip_match_knn_device", "brisk_hip_set_uniformity",
    "brisk_hip_reserve", "brisk_hip_detect_uniform", "brisk_hip_detect_describe_batch_host", "brisk_hip_stream_ceiling",
    "brisk_hip_kernel_revision", "brisk_hip_compute_scale", "brisk_hip_describe_same_image", "brisk_hip_detect_filtered",
    "brisk_hip_comm_unique_id", "brisk_hip_comm_create", "brisk_hip_comm_destroy", "brisk_hip_comm_rank", "brisk_hip_comm_world",
    "brisk_hip_comm_gather_results", "brisk_hip_comm_wait",
    "brisk_hip_set_integral_format",
    "brisk_hip_host_register", "brisk_hip_host_unregister", "brisk_hip_usable_cpus",
    "brisk_hip_detect_images", "brisk_hip_describe_images",
    "brisk_hip_pool_create", "brisk_hip_pool_destroy", "brisk_hip_pool_last_error", "brisk_hip_pool_detect", "brisk_hip_pool_describe", "brisk_hip_pool_stats",
    "brisk_hip_batch_download_all", "brisk_hip_batch_download_wait", "brisk_hip_detect_describe_batch_host_results",
]
# every symbol include/brisk_hip_debug.h declares: test / tuning builds (BRISK_HIP_TUNING) only
DEBUG_SYMBOLS = [
    "brisk_hip_debug_layer", "brisk_hip_debug_integral", "brisk_hip_debug_counters", "brisk_hip_debug_counters_raw",
    "brisk_hip_debug_set_flags", "brisk_hip_debug_image_reuse", "brisk_hip_debug_filter_keypoints", "brisk_hip_debug_integral_bits",
    "brisk_hip_debug_forge_pattern_device", "brisk_hip_debug_pool_phases",
]


class PostFilter(C.Structure):
    """brisk_hip_postfilter: the optional post-filters of one detect call"""
    _fields_ = [("uniformity_radius", C.c_double), ("uniformity_max_keypoints", C.c_int), ("num_buckets_u", C.c_int),
                ("num_buckets_v", C.c_int), ("bucket_max_keypoints", C.c_int)]


class BatchHostResults(C.Structure):
    """brisk_hip_batch_host_results: capacities + the five destination arrays of a whole batch in host memory"""
    _fields_ = [("frames_cap", C.c_int), ("desc_stride", C.c_int), ("rows_cap", C.c_longlong), ("counts", C.c_void_p),
                ("flags", C.c_void_p), ("offsets", C.c_void_p), ("kps", C.c_void_p), ("desc", C.c_void_p)]


ROWS_CUT = 0x100


class HostResults:
    """Destination arrays of brisk_hip_batch_download_all: `frames` frames, `rows` rows in total, descriptor rows of
    `desc_stride` bytes (0 = keypoints only).  pinned=True takes them from torch's pinned allocator (the device then writes
    them directly); otherwise plain NumPy arrays (pageable: the engine goes through its own pinned bounce buffer)."""

    def __init__(self, frames, rows, desc_stride=48, pinned=True):
        self.frames, self.rows, self.desc_stride = int(frames), int(rows), int(desc_stride)
        self._keep = []

        def arr(n, dtype):
            n = max(int(n), 1)
            if pinned:
                import torch
                t = torch.empty(n * np.dtype(dtype).itemsize, dtype=torch.uint8).pin_memory()
                self._keep.append(t)
                return t.numpy().view(dtype)
            return np.empty(n, dtype)
        self.counts = arr(frames, np.int32)
        self.flags = arr(frames, np.int32)
        self.offsets = arr(frames + 1, np.int64)
        self.kps = arr(rows, KEYPOINT)
        self.desc = arr(rows * max(desc_stride, 1), np.uint8).reshape(max(rows, 1), max(desc_stride, 1)) if desc_stride else None
        self.struct = BatchHostResults(self.frames, self.desc_stride or 4, self.rows, self.counts.ctypes.data, self.flags.ctypes.data,
                                       self.offsets.ctypes.data, self.kps.ctypes.data,
                                       self.desc.ctypes.data if self.desc is not None else None)

    def frame(self, f, strings=None):
        """(keypoints, descriptors) of frame f: views of the rows [offsets[f], offsets[f + 1])"""
        a, b = int(self.offsets[f]), int(self.offsets[f + 1])
        d = None if self.desc is None else self.desc[a:b, :strings or self.desc_stride]
        return self.kps[a:b], d


class BriskHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (ERRORS.get(code, code), msg))
        self.code = code


_lib = None


def load_library():
    """Loads libbrisk_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch wheels bundle their own HIP runtime under the same SONAME (libamdhip64.so.7).  If torch is
        # going to be used in this process (tests, bench.py) it must be loaded first so that this library
        # binds to the runtime already in the process: two HIP runtimes in one process cannot both own the GPU.
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError("libbrisk_hip.so is missing: run `python -m ethzasl_brisk_amd.build` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, ip = C.c_void_p, C.POINTER(C.c_int)
    L.brisk_hip_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.brisk_hip_destroy.argtypes = [vp]
    L.brisk_hip_destroy.restype = None
    L.brisk_hip_last_error.argtypes = [vp]
    L.brisk_hip_last_error.restype = C.c_char_p
    L.brisk_hip_set_capacity.argtypes = [vp, C.c_int, C.c_int]
    L.brisk_hip_pattern_create.argtypes = [vp, C.c_int, C.c_float, C.POINTER(vp)]
    L.brisk_hip_pattern_create_from_text.argtypes = [vp, C.c_char_p, C.c_float, C.POINTER(vp)]
    L.brisk_hip_pattern_destroy.argtypes = [vp]
    L.brisk_hip_pattern_destroy.restype = None
    L.brisk_hip_pattern_descriptor_size.argtypes = [vp]
    L.brisk_hip_pattern_points.argtypes = [vp]
    L.brisk_hip_pattern_tables.argtypes = [vp, vp, vp, vp]
    L.brisk_hip_detect.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp,
                                   C.c_int, ip]
    L.brisk_hip_describe.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, ip, vp, C.c_int, C.c_int, C.c_int]
    L.brisk_hip_describe_same_image.argtypes = L.brisk_hip_describe.argtypes
    L.brisk_hip_detect_describe_batch.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int,
                                                  C.c_int, vp]
    L.brisk_hip_detect_batch.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, vp]
    L.brisk_hip_batch_results.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), ip, C.POINTER(vp), C.POINTER(vp),
                                          C.POINTER(vp), ip, ip]
    L.brisk_hip_batch_download.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, ip, vp, C.c_int]
    L.brisk_hip_batch_status.argtypes = [vp, C.c_int, ip]
    if hasattr(L, "brisk_hip_debug_layer"):
        L.brisk_hip_debug_layer.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, ip, ip]
    if hasattr(L, "brisk_hip_debug_integral"):
        L.brisk_hip_debug_integral.argtypes = [vp, C.c_int, vp]
    if hasattr(L, "brisk_hip_debug_integral_bits"):
        L.brisk_hip_debug_integral_bits.argtypes = [vp, C.c_int]
    if hasattr(L, "brisk_hip_debug_counters"):
        L.brisk_hip_debug_counters.argtypes = [vp, C.c_int, vp, ip]
    if hasattr(L, "brisk_hip_debug_counters_raw"):
        L.brisk_hip_debug_counters_raw.argtypes = [vp, C.c_int, vp, C.c_int]
    L.brisk_hip_profile_enable.argtypes = [vp, C.c_int]
    if hasattr(L, "brisk_hip_debug_set_flags"):
        L.brisk_hip_debug_set_flags.argtypes = [vp, C.c_int]
    L.brisk_hip_set_integral_format.argtypes = [vp, C.c_int]
    if hasattr(L, "brisk_hip_debug_forge_pattern_device"):
        L.brisk_hip_debug_forge_pattern_device.argtypes = [vp, C.c_int]
    if hasattr(L, "brisk_hip_debug_image_reuse"):
        L.brisk_hip_debug_image_reuse.argtypes = [vp]
    L.brisk_hip_set_bucketing.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    for f in (L.brisk_hip_halfsample16, L.brisk_hip_twothirdsample16, L.brisk_hip_integral_image16):
        f.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    L.brisk_hip_set_streams.argtypes = [vp, C.c_int]
    L.brisk_hip_profile_frames_per_launch.argtypes = [vp]
    L.brisk_hip_profile_stage_name.argtypes = [C.c_int]
    L.brisk_hip_profile_stage_name.restype = C.c_char_p
    L.brisk_hip_profile_read.argtypes = [vp, vp, ip]
    L.brisk_hip_set_uniformity.argtypes = [vp, C.c_double, C.c_int]
    L.brisk_hip_match_knn.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp, vp]
    L.brisk_hip_match_radius.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_float,
                                         C.c_int, vp, vp]
    L.brisk_hip_match_knn_device.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.brisk_hip_reserve.argtypes = [vp, C.c_int, C.c_int]
    L.brisk_hip_detect_uniform.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int,
                                           C.c_double, C.c_int, vp, C.c_int, ip]
    L.brisk_hip_detect_filtered.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int,
                                            C.POINTER(PostFilter), vp, C.c_int, ip]
    if hasattr(L, "brisk_hip_debug_filter_keypoints"):
        L.brisk_hip_debug_filter_keypoints.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp, ip]
    L.brisk_hip_comm_unique_id.argtypes = [vp]
    L.brisk_hip_comm_create.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.brisk_hip_comm_destroy.argtypes = [vp]
    L.brisk_hip_comm_destroy.restype = None
    L.brisk_hip_comm_rank.argtypes = [vp]
    L.brisk_hip_comm_world.argtypes = [vp]
    L.brisk_hip_comm_gather_results.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    L.brisk_hip_comm_wait.argtypes = [vp, vp]
    L.brisk_hip_compute_scale.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp,
                                          C.c_int, ip]
    L.brisk_hip_detect_describe_batch_host.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int,
                                                       C.c_int]
    L.brisk_hip_pool_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.brisk_hip_pool_destroy.argtypes = [vp]
    L.brisk_hip_pool_destroy.restype = None
    L.brisk_hip_pool_last_error.argtypes = [vp]
    L.brisk_hip_pool_last_error.restype = C.c_char_p
    L.brisk_hip_pool_stats.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    L.brisk_hip_pool_detect.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, ip, C.POINTER(C.c_ulonglong)]
    L.brisk_hip_pool_describe.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, ip, vp, C.c_int, C.c_int, C.c_int, C.c_ulonglong]
    L.brisk_hip_detect_images.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(BatchHostResults),
                                          C.POINTER(C.c_uint)]
    L.brisk_hip_describe_images.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(BatchHostResults), C.POINTER(C.c_uint)]
    L.brisk_hip_host_register.argtypes = [vp, C.c_size_t]
    L.brisk_hip_host_unregister.argtypes = [vp]
    L.brisk_hip_batch_download_all.argtypes = [vp, C.c_int, C.POINTER(BatchHostResults), vp, C.POINTER(C.c_uint)]
    L.brisk_hip_batch_download_wait.argtypes = [vp, C.c_uint, ip]
    L.brisk_hip_detect_describe_batch_host_results.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int,
                                                               C.c_int, C.POINTER(BatchHostResults), C.POINTER(C.c_uint)]
    L.brisk_hip_stream_ceiling.argtypes = [vp, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.brisk_hip_kernel_revision.argtypes = []
    L.brisk_hip_kernel_revision.restype = C.c_char_p
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _kcopy(a, n=None):
    """copy of the first n records of a contiguous KEYPOINT array through a byte view (NumPy copies structured arrays
    field by field: 2.8 ms per 100 000 keypoints instead of 0.1 ms)"""
    n = len(a) if n is None else n
    out = np.empty(n, KEYPOINT)
    out.view(np.uint8)[:] = a.view(np.uint8)[:n * KEYPOINT.itemsize]
    return out


class Context:
    """Device workspace (one per GPU).  Shared by detector and extractor objects."""

    def __init__(self, device=0, max_candidates=None, max_keypoints=None):
        self._L = load_library()
        h = C.c_void_p()
        rc = self._L.brisk_hip_create(device, C.byref(h))
        if rc:
            raise BriskHipError(rc, "brisk_hip_create(device=%d) failed" % device)
        self._h = h
        self.device = device
        if max_candidates or max_keypoints:
            self.check(self._L.brisk_hip_set_capacity(h, max_candidates or 65536, max_keypoints or 16384))

    def check(self, rc):
        if rc:
            raise BriskHipError(rc, self._L.brisk_hip_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.brisk_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- per-stage access (parity tests) --
    def debug_layer(self, frame, layer, which=0):
        w, h = C.c_int(), C.c_int()
        self.check(self._L.brisk_hip_debug_layer(self._h, frame, layer, which, None, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        self.check(self._L.brisk_hip_debug_layer(self._h, frame, layer, which, _ptr(out), C.byref(w), C.byref(h)))
        return out

    def debug_counters(self, frame):
        """work counts of one frame of the last batch: candidates, keypoints, described, flags, ties per layer"""
        out = np.zeros(28, np.int32)
        nl = C.c_int()
        self.check(self._L.brisk_hip_debug_counters(self._h, frame, _ptr(out), C.byref(nl)))
        return {"candidates": int(out[0]), "keypoints": int(out[1]), "described": int(out[2]), "flags": int(out[3]),
                "ties": [int(v) for v in out[4:4 + nl.value]], "experiment": [int(v) for v in out[20:28]]}

    def debug_counters_raw(self, frame):
        """the frame's counter record as int32 words (instrumented build variants append fields)"""
        out = np.zeros(1024, np.int32)
        n = self._L.brisk_hip_debug_counters_raw(self._h, frame, _ptr(out), out.nbytes)
        if n < 0:
            raise BriskHipError(-1, "brisk_hip_debug_counters_raw failed")
        return out[:n // 4]

    def debug_integral(self, frame, w, h):
        """(h + 1) x (w + 1) u32: the integral image of the last describe, modulo 2 ** debug_integral_bits()"""
        out = np.zeros((h + 1, w + 1), np.uint32)
        self.check(self._L.brisk_hip_debug_integral(self._h, frame, _ptr(out)))
        return out

    def debug_integral_bits(self, frame=0):
        return self._L.brisk_hip_debug_integral_bits(self._h, frame)

    def set_uniformity(self, radius, max_keypoints=0x7FFFFFFF):
        """Optional uniformity enforcement after the detector (0 = off); see brisk_hip_set_uniformity."""
        self.check(self._L.brisk_hip_set_uniformity(self._h, float(radius), int(max_keypoints)))

    def set_bucketing(self, num_buckets_u, num_buckets_v, max_keypoints):
        """Optional KeyPointBucketing after the detector while uniformity is off ((0, 0, *) = off); see brisk_hip_set_bucketing."""
        self.check(self._L.brisk_hip_set_bucketing(self._h, int(num_buckets_u), int(num_buckets_v), int(max_keypoints)))

    def _image16(self, fn, image, out):
        img = np.ascontiguousarray(image, np.uint16)
        h, w = img.shape
        self.check(fn(self._h, _ptr(img), w, h, w, _ptr(out), out.shape[1]))
        return out

    def halfsample16(self, image):
        """brisk::Halfsample16 (image-down-sampling.cc:56-139)"""
        h, w = np.shape(image)
        return self._image16(self._L.brisk_hip_halfsample16, image, np.zeros((h // 2, w // 2), np.uint16))

    def twothirdsample16(self, image):
        """brisk::Twothirdsample16 (image-down-sampling.cc:394-548)"""
        h, w = np.shape(image)
        return self._image16(self._L.brisk_hip_twothirdsample16, image, np.zeros((h // 3 * 2, w // 3 * 2), np.uint16))

    def integral_image16(self, image):
        """brisk::IntegralImage16 (internal/integral-image.h:163-218): (h + 1) x (w + 1) float32"""
        h, w = np.shape(image)
        return self._image16(self._L.brisk_hip_integral_image16, image, np.zeros((h + 1, w + 1), np.float32))

    def set_streams(self, n):
        self.check(self._L.brisk_hip_set_streams(self._h, n))

    def profile_frames_per_launch(self):
        return self._L.brisk_hip_profile_frames_per_launch(self._h)

    def debug_image_reuse(self):
        """describe calls that reused the device copy of the image a detect call had uploaded"""
        return self._L.brisk_hip_debug_image_reuse(self._h)

    def set_integral_format(self, fmt):
        """0 = automatic (from the previous batch's candidate density), 24 / 32 = that element size for every call"""
        self.check(self._L.brisk_hip_set_integral_format(self._h, int(fmt)))

    def debug_set_flags(self, flags):
        self.check(self._L.brisk_hip_debug_set_flags(self._h, flags))

    # -- per-stage HIP-event timing of the batch path --
    def profile_enable(self, on=True):
        self.check(self._L.brisk_hip_profile_enable(self._h, int(on)))

    def profile_read(self):
        """{stage name: average ms per call}, number of calls averaged."""
        n = self._L.brisk_hip_profile_stages()
        ms = np.zeros(n, np.float32)
        calls = C.c_int()
        self.check(self._L.brisk_hip_profile_read(self._h, _ptr(ms), C.byref(calls)))
        return {self._L.brisk_hip_profile_stage_name(i).decode(): float(ms[i]) for i in range(n)}, calls.value

    # -- device-resident batch path --
    def detect_describe_batch(self, pattern, d_frames_ptr, nframes, w, h, frame_pitch, row_pitch, threshold, octaves,
                              stream=None):
        self.check(self._L.brisk_hip_detect_describe_batch(self._h, pattern._h, C.c_void_p(d_frames_ptr), nframes, w, h,
                                                           frame_pitch, row_pitch, threshold, octaves,
                                                           C.c_void_p(stream) if stream else None))

    def detect_describe_batch_host(self, pattern, h_frames_ptr, nframes, w, h, frame_pitch, row_pitch, threshold, octaves):
        """frames in (pinned) HOST memory: sliced H2D copies on a copy stream overlapped with compute"""
        self.check(self._L.brisk_hip_detect_describe_batch_host(self._h, pattern._h, C.c_void_p(h_frames_ptr), nframes, w,
                                                                h, frame_pitch, row_pitch, threshold, octaves))

    def batch_download_all(self, dst, described=True, stream=None):
        """queues the transfer of the last batch's results into `dst` (HostResults); returns the ticket"""
        t = C.c_uint()
        self.check(self._L.brisk_hip_batch_download_all(self._h, int(described), C.byref(dst.struct),
                                                        C.c_void_p(stream) if stream else None, C.byref(t)))
        return t.value

    def batch_download_wait(self, ticket, check=True):
        """completes transfer `ticket`; returns the number of flagged frames (check=False: (rc, flagged) instead of raising)"""
        n = C.c_int()
        rc = self._L.brisk_hip_batch_download_wait(self._h, ticket, C.byref(n))
        if not check:
            return rc, n.value
        self.check(rc)
        return n.value

    def detect_describe_batch_host_results(self, pattern, h_frames_ptr, nframes, w, h, frame_pitch, row_pitch, threshold, octaves,
                                           dst):
        """host frames in, host results out (everything queued on return); returns the ticket"""
        t = C.c_uint()
        self.check(self._L.brisk_hip_detect_describe_batch_host_results(self._h, pattern._h, C.c_void_p(h_frames_ptr), nframes, w, h,
                                                                        frame_pitch, row_pitch, threshold, octaves,
                                                                        C.byref(dst.struct), C.byref(t)))
        return t.value

    def detect_images(self, images, threshold, octaves, dst):
        """cv::FeatureDetector::detect(vector<Mat>) as one batch: `images` = equally sized 2-D uint8 arrays; returns the ticket"""
        imgs = [np.ascontiguousarray(a, np.uint8) for a in images]
        h, w = imgs[0].shape
        ptrs = (C.c_void_p * len(imgs))(*[a.ctypes.data for a in imgs])
        t = C.c_uint()
        self.check(self._L.brisk_hip_detect_images(self._h, ptrs, len(imgs), w, h, w, threshold, octaves, C.byref(dst.struct), C.byref(t)))
        self._keep_images = imgs
        return t.value

    def describe_images(self, pattern, images, keypoints, dst, rotation_invariant=True, scale_invariant=True, same_images=False):
        """cv::DescriptorExtractor::compute(vector<Mat>, vector<vector<KeyPoint>>) as one batch; returns the ticket.
        same_images: `images` are the arrays the last detect_images call was given, unchanged (no second upload)"""
        imgs = [np.ascontiguousarray(a, np.uint8) for a in images]
        ks = [np.ascontiguousarray(k, KEYPOINT) for k in keypoints]
        h, w = imgs[0].shape
        ptrs = (C.c_void_p * len(imgs))(*[a.ctypes.data for a in imgs])
        kptrs = (C.c_void_p * len(ks))(*[k.ctypes.data if len(k) else None for k in ks])
        nk = np.array([len(k) for k in ks], np.int32)
        t = C.c_uint()
        self.check(self._L.brisk_hip_describe_images(self._h, pattern._h, ptrs, len(imgs), w, h, w, kptrs, _ptr(nk), int(rotation_invariant),
                                                     int(scale_invariant), int(same_images), C.byref(dst.struct), C.byref(t)))
        self._keep_images = (imgs, ks, nk)
        return t.value

    def reserve(self, min_candidates, min_keypoints):
        self.check(self._L.brisk_hip_reserve(self._h, int(min_candidates), int(min_keypoints)))

    def stream_ceiling(self, nbytes=1 << 30):
        """(copy GB/s counting read + write, read-only GB/s) of the engine's float4 streaming kernels on this box"""
        a, b = C.c_double(), C.c_double()
        self.check(self._L.brisk_hip_stream_ceiling(self._h, nbytes, C.byref(a), C.byref(b)))
        return a.value, b.value

    def kernel_revision(self):
        return self._L.brisk_hip_kernel_revision().decode()

    def detect_batch(self, d_frames_ptr, nframes, w, h, frame_pitch, row_pitch, threshold, octaves, stream=None):
        self.check(self._L.brisk_hip_detect_batch(self._h, C.c_void_p(d_frames_ptr), nframes, w, h, frame_pitch,
                                                  row_pitch, threshold, octaves, C.c_void_p(stream) if stream else None))

    def batch_status(self, nframes):
        f = C.c_int()
        self.check(self._L.brisk_hip_batch_status(self._h, nframes, C.byref(f)))
        return f.value

    def batch_download(self, frame, described=True, strings=48):
        n = C.c_int()
        self.check(self._L.brisk_hip_batch_download(self._h, frame, int(described), None, 0, C.byref(n), None, 0))
        kps = np.zeros(max(n.value, 1), KEYPOINT)
        desc = np.zeros((max(n.value, 1), strings), np.uint8)
        self.check(self._L.brisk_hip_batch_download(self._h, frame, int(described), _ptr(kps), len(kps), C.byref(n),
                                                    _ptr(desc) if described else None, strings))
        return _kcopy(kps, n.value), (desc[:n.value].copy() if described else None)


class Pool:
    """brisk_hip_pool: the one-frame calls of many threads combined into batches (thread-safe, blocking calls)"""

    def __init__(self, device=0, max_batch=16, max_keypoints=16384):
        self._L = load_library()
        h = C.c_void_p()
        rc = self._L.brisk_hip_pool_create(device, max_batch, max_keypoints, C.byref(h))
        if rc:
            raise BriskHipError(rc, "brisk_hip_pool_create failed")
        self._h = h

    def _check(self, rc):
        if rc:
            raise BriskHipError(rc, self._L.brisk_hip_pool_last_error(self._h).decode())

    def detect(self, image, threshold, octaves, capacity=16384):
        """-> (keypoints, token of the frame's device copy)"""
        img = np.ascontiguousarray(image, np.uint8)
        h, w = img.shape
        out = np.empty(capacity, KEYPOINT)
        n, tok = C.c_int(), C.c_ulonglong()
        self._check(self._L.brisk_hip_pool_detect(self._h, _ptr(img), w, h, w, threshold, octaves, _ptr(out), capacity, C.byref(n), C.byref(tok)))
        return _kcopy(out, n.value), tok.value

    def describe(self, pattern, image, keypoints, token=0, rotation_invariant=True, scale_invariant=True):
        img = np.ascontiguousarray(image, np.uint8)
        h, w = img.shape
        k = _kcopy(np.ascontiguousarray(keypoints, KEYPOINT))
        n = C.c_int(len(k))
        s = pattern.descriptorSize()
        desc = np.empty((max(len(k), 1), s), np.uint8)
        if len(k) == 0:
            k = np.zeros(1, KEYPOINT)
        self._check(self._L.brisk_hip_pool_describe(self._h, pattern._h, _ptr(img), w, h, w, _ptr(k), C.byref(n), _ptr(desc), s,
                                                    int(rotation_invariant), int(scale_invariant), token))
        return k[:n.value], desc[:n.value]

    def stats(self):
        """(groups run, calls they carried)"""
        g, c = C.c_ulonglong(), C.c_ulonglong()
        self._L.brisk_hip_pool_stats(self._h, C.byref(g), C.byref(c))
        return g.value, c.value

    def close(self):
        if getattr(self, "_h", None):
            self._L.brisk_hip_pool_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


class BriskFeatureDetector:
    """Mirror of brisk::BriskFeatureDetector (brisk/include/brisk/brisk-feature-detector.h:51-83)."""

    def __init__(self, thresh, octaves=3, suppressScaleNonmaxima=True, context=None, uniformityRadius=0.0,
                 maxNumKpt=0x7FFFFFFF, numBucketsU=0, numBucketsV=0):
        self.threshold = int(thresh)
        self.octaves = int(octaves)
        self.m_suppressScaleNonmaxima = bool(suppressScaleNonmaxima)
        # engine option (not reference behaviour of this class): EnforceKeyPointUniformity as a post-filter
        self.uniformityRadius, self.maxNumKpt = float(uniformityRadius), int(maxNumKpt)
        # engine option as well: KeyPointBucketing (what the reference's ScaleSpaceLayer uses while uniformity is off)
        self.numBucketsU, self.numBucketsV = int(numBucketsU), int(numBucketsV)
        self._ctx = context or default_context()

    def detect(self, image, mask=None, capacity=16384):
        """detectImpl (brisk-feature-detector.cc:77-85): returns the keypoints (KEYPOINT array)."""
        img = np.ascontiguousarray(image)
        if img.dtype != np.uint8 or img.ndim != 2:
            raise ValueError("image must be a 2-D uint8 array (CV_8UC1)")
        h, w = img.shape
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, np.uint8)
            if m.shape != img.shape:
                raise ValueError("mask must have the image's shape")
        out = np.empty(capacity, KEYPOINT)   # (records [0, n) are written by the call)
        n = C.c_int()
        c = self._ctx
        # the object's post-filter settings travel with the call (the context's own settings are neither used nor changed)
        pf = PostFilter(self.uniformityRadius, self.maxNumKpt, self.numBucketsU, self.numBucketsV, self.maxNumKpt)
        c.check(c._L.brisk_hip_detect_filtered(c._h, _ptr(img), w, h, w, self.threshold, self.octaves,
                                               int(self.m_suppressScaleNonmaxima), _ptr(m), w if m is not None else 0,
                                               C.byref(pf), _ptr(out), capacity, C.byref(n)))
        return _kcopy(out, n.value)


    def ComputeScale(self, image, keypoints, capacity=None):
        """ComputeScale (brisk-feature-detector.cc:87-92): returns the keypoints GetKeypoints produces for the provided
        ones (the reference replaces the vector's contents)."""
        img = np.ascontiguousarray(image)
        if img.dtype != np.uint8 or img.ndim != 2:
            raise ValueError("image must be a 2-D uint8 array (CV_8UC1)")
        h, w = img.shape
        k = np.ascontiguousarray(keypoints, KEYPOINT)
        layers = 1 if self.octaves == 0 else 2 * self.octaves
        cap = capacity or (len(k) * layers + 65536)
        c = self._ctx
        c.reserve(65536, max(len(k), cap))
        out = np.zeros(cap, KEYPOINT)
        n = C.c_int()
        c.check(c._L.brisk_hip_compute_scale(c._h, _ptr(img), w, h, w, self.threshold, self.octaves,
                                             int(self.m_suppressScaleNonmaxima), _ptr(k) if len(k) else None, len(k),
                                             _ptr(out), cap, C.byref(n)))
        return _kcopy(out, n.value)


class BriskDescriptorExtractor:
    """Mirror of brisk::BriskDescriptorExtractor (brisk/include/brisk/brisk-descriptor-extractor.h:54-202)."""
    briskV1 = 1
    briskV2 = 2
    kDescriptorLength = 384

    def __init__(self, rotationInvariant=True, scaleInvariant=True, version=2, patternScale=1.0, fname=None,
                 pattern_text=None, context=None):
        self.rotationInvariance = bool(rotationInvariant)
        self.scaleInvariance = bool(scaleInvariant)
        self._ctx = c = context or default_context()
        h = C.c_void_p()
        if fname is not None:
            pattern_text = open(fname).read()
        if pattern_text is not None:
            c.check(c._L.brisk_hip_pattern_create_from_text(c._h, pattern_text.encode(), patternScale, C.byref(h)))
        else:
            if version not in (1, 2):
                raise RuntimeError("only Version::briskV1 or Version::briskV2 supported!")
            c.check(c._L.brisk_hip_pattern_create(c._h, version, patternScale, C.byref(h)))
        self._h = h

    def descriptorSize(self):
        return self._ctx._L.brisk_hip_pattern_descriptor_size(self._h)

    @property
    def points(self):
        """number of pattern points (60 for briskV1, 66 for briskV2)"""
        return self._ctx._L.brisk_hip_pattern_points(self._h)

    def descriptorType(self):
        return 0  # CV_8U

    def tables(self):
        a, b, t = np.zeros(64, np.float32), np.zeros(64, np.int32), np.zeros(64, np.float32)
        self._ctx.check(self._ctx._L.brisk_hip_pattern_tables(self._h, _ptr(a), _ptr(b), _ptr(t)))
        return a, b, t

    def compute(self, image, keypoints, same_image=False):
        """compute() (brisk-descriptor-extractor.cc:612-778): returns (filtered keypoints, descriptors).
        same_image: the caller states that `image` is the very buffer the context's last detect() call was given and
        that it has not changed (brisk_hip_describe_same_image: no second upload)."""
        img = np.ascontiguousarray(image)
        if img.dtype != np.uint8 or img.ndim != 2:
            raise RuntimeError("Unsupported image format. Must be CV_16UC1 or CV_8UC1.")  # :678 (8-bit only here)
        h, w = img.shape
        k = _kcopy(np.ascontiguousarray(keypoints, KEYPOINT))
        n = C.c_int(len(k))
        s = self.descriptorSize()
        desc = np.empty((max(len(k), 1), s), np.uint8)   # (rows [0, n) are written by the call, the others dropped below)
        if len(k) == 0:
            k = np.zeros(1, KEYPOINT)
        c = self._ctx
        fn = c._L.brisk_hip_describe_same_image if same_image else c._L.brisk_hip_describe
        c.check(fn(c._h, self._h, _ptr(img), w, h, w, _ptr(k), C.byref(n), _ptr(desc), s,
                   int(self.rotationInvariance), int(self.scaleInvariance)))
        return k[:n.value], desc[:n.value]

    def close(self):
        if getattr(self, "_h", None) and getattr(self._ctx, "_h", None):
            self._ctx._L.brisk_hip_pattern_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


DMATCH = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])  # cv::DMatch


class BruteForceMatcher:
    """Mirror of brisk::BruteForceMatcher with brisk::Hamming (brisk/include/brisk/brute-force-matcher.h:52-93,
    brisk/src/brute-force-matcher.cc:80-213): add() train descriptor sets, then knnMatch / radiusMatch / match.
    Matches come back as one structured array (DMATCH) per query, ordered by (distance, imgIdx, trainIdx)."""

    def __init__(self, context=None):
        self.ctx = context or default_context()
        self.trainDescCollection = []

    def add(self, descriptors):
        for d in (descriptors if isinstance(descriptors, (list, tuple)) else [descriptors]):
            self.trainDescCollection.append(np.ascontiguousarray(d, np.uint8))

    def clear(self):
        self.trainDescCollection = []

    def empty(self):
        return not self.trainDescCollection

    def isMaskSupported(self):
        return True

    def _args(self, query, masks):
        query = np.ascontiguousarray(query, np.uint8)
        if query.ndim != 2:
            raise ValueError("descriptors must be a 2-D uint8 array")
        train = self.trainDescCollection
        nimg = len(train)
        for t in train:
            if t.ndim != 2 or (t.shape[0] and t.shape[1] != query.shape[1]):
                raise ValueError("train descriptors must have the query's descriptor size")
        tptr = (C.c_void_p * max(nimg, 1))(*[t.ctypes.data for t in train])
        ntr = np.array([t.shape[0] for t in train] + [0], np.int32)
        tpitch = np.array([t.strides[0] if t.shape[0] else query.shape[1] for t in train] + [0], np.int32)
        mptr, mpitch, keep = None, None, []
        if masks is not None:
            if len(masks) != nimg:
                raise ValueError("one mask (or None) per train image")
            ms = [None if m is None else np.ascontiguousarray(m, np.uint8) for m in masks]
            for m, t in zip(ms, train):
                if m is not None and m.shape != (query.shape[0], t.shape[0]):
                    raise ValueError("mask must be queries x train descriptors")
            keep = ms
            mptr = (C.c_void_p * max(nimg, 1))(*[None if m is None else m.ctypes.data for m in ms])
            mpitch = np.array([0 if m is None else m.strides[0] for m in ms] + [0], np.int32)
        return query, nimg, tptr, ntr, tpitch, mptr, mpitch, keep

    def knnMatch(self, queryDescriptors, k, masks=None, compactResult=False):
        q, nimg, tptr, ntr, tpitch, mptr, mpitch, keep = self._args(queryDescriptors, masks)
        nq = q.shape[0]
        out = np.zeros((nq, max(k, 1)), DMATCH)
        cnt = np.zeros(max(nq, 1), np.int32)
        L = self.ctx._L
        self.ctx.check(L.brisk_hip_match_knn(self.ctx._h, _ptr(q), nq, q.strides[0] if nq else q.shape[1], q.shape[1], nimg,
                                             tptr, _ptr(ntr), _ptr(tpitch), mptr, None if mpitch is None else _ptr(mpitch),
                                             k, _ptr(out), _ptr(cnt)))
        rows = [out[i, :cnt[i]].copy() for i in range(nq)]
        return [r for r in rows if len(r)] if compactResult else rows

    def radiusMatch(self, queryDescriptors, maxDistance, masks=None, compactResult=False):
        q, nimg, tptr, ntr, tpitch, mptr, mpitch, keep = self._args(queryDescriptors, masks)
        nq = q.shape[0]
        L = self.ctx._L
        cap = 64
        while True:
            out = np.zeros((nq, cap), DMATCH)
            cnt = np.zeros(max(nq, 1), np.int32)
            self.ctx.check(L.brisk_hip_match_radius(self.ctx._h, _ptr(q), nq, q.strides[0] if nq else q.shape[1], q.shape[1],
                                                    nimg, tptr, _ptr(ntr), _ptr(tpitch), mptr,
                                                    None if mpitch is None else _ptr(mpitch), float(maxDistance), cap,
                                                    _ptr(out), _ptr(cnt)))
            need = int(cnt[:nq].max()) if nq else 0
            if need <= cap:
                break
            cap = need
        rows = [out[i, :cnt[i]].copy() for i in range(nq)]
        return [r for r in rows if len(r)] if compactResult else rows

    def match(self, queryDescriptors, masks=None):
        """cv::DescriptorMatcher::match: the best match of every query (queries without one are dropped)."""
        rows = self.knnMatch(queryDescriptors, 1, masks, compactResult=True)
        return np.concatenate(rows) if rows else np.zeros(0, DMATCH)
