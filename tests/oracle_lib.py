"""ctypes binding of the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
               ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h", ".inc"))]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
        L = C.CDLL(so)
        i32p, vp = C.POINTER(C.c_int32), C.c_void_p
        L.bo_halfsample8.argtypes = [vp, C.c_int, C.c_int, vp]
        L.bo_twothirdsample8.argtypes = [vp, C.c_int, C.c_int, vp]
        L.bo_threshold_map.argtypes = [vp, C.c_int, C.c_int, vp]
        L.bo_integral_image8.argtypes = [vp, C.c_int, C.c_int, vp]
        L.bo_oast9_16_corner_score.argtypes = [vp, C.c_int, C.c_int]
        L.bo_agast5_8_corner_score.argtypes = [vp, C.c_int, C.c_int]
        L.bo_oast9_16_detect.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int]
        L.bo_scale_space_create.restype = vp
        L.bo_scale_space_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.bo_scale_space_destroy.argtypes = [vp]
        L.bo_scale_space_layers.argtypes = [vp]
        L.bo_scale_space_map.restype = vp
        L.bo_scale_space_map.argtypes = [vp, C.c_int, C.c_int, i32p, i32p]
        L.bo_scale_space_get_keypoints.argtypes = [vp, C.POINTER(vp)]
        L.bo_detect.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]
        L.bo_detect_ex.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]
        L.bo_free.argtypes = [vp]
        L.bo_compute_scale.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.POINTER(vp)]
        L.bo_extractor_create.restype = vp
        L.bo_extractor_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_char_p]
        L.bo_extractor_destroy.argtypes = [vp]
        L.bo_extractor_descriptor_size.argtypes = [vp]
        L.bo_extractor_points.argtypes = [vp]
        L.bo_extractor_scale_list.restype = vp
        L.bo_extractor_scale_list.argtypes = [vp]
        L.bo_extractor_size_list.restype = vp
        L.bo_extractor_size_list.argtypes = [vp]
        L.bo_extractor_pattern.restype = vp
        L.bo_extractor_pattern.argtypes = [vp]
        L.bo_extractor_scale_index.argtypes = [vp, C.c_float]
        L.bo_extractor_compute.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp]
        L.bo_hamming.argtypes = [vp, vp, C.c_int]
        L.bo_enforce_uniformity.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp]
        L.bo_halfsample16.argtypes = [vp, C.c_int, C.c_int, vp]
        L.bo_twothirdsample16.argtypes = [vp, C.c_int, C.c_int, vp]
        L.bo_integral_image16.argtypes = [vp, C.c_int, C.c_int, vp]
        L.bo_key_point_bucketing.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
        L.bo_match_knn.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp, vp]
        L.bo_match_knn.restype = None
        L.bo_match_radius.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_float, vp]
        L.bo_match_radius.restype = vp
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def halfsample(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h // 2, w // 2), np.uint8)
    lib().bo_halfsample8(_p(img), w, h, _p(out))
    return out


def twothirdsample(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((2 * (h // 3), 2 * (w // 3)), np.uint8)
    lib().bo_twothirdsample8(_p(img), w, h, _p(out))
    return out


def threshold_map(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().bo_threshold_map(_p(img), w, h, _p(out))
    return out


def integral(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h + 1, w + 1), np.int32)
    lib().bo_integral_image8(_p(img), w, h, _p(out))
    return out


def _take_kps(out, n):
    kps = np.frombuffer(C.string_at(out.value, n * KP.itemsize), dtype=KP).copy() if n else np.zeros(0, KP)
    lib().bo_free(out)
    return kps


def detect(img, threshold, octaves, mask=None, suppress_scale_nonmaxima=True):
    """detectImpl; returns None where the reference has no defined result (suppressScaleNonmaxima=False with several
    layers on an input that makes agastPoints.at(0) throw or IsMax2D read outside a matrix)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = C.c_void_p()
    m = None if mask is None else _p(np.ascontiguousarray(mask, np.uint8))
    n = lib().bo_detect_ex(_p(img), w, h, threshold, octaves, int(bool(suppress_scale_nonmaxima)), m, C.byref(out))
    if n < 0:
        return None
    return _take_kps(out, n)


def compute_scale(img, keypoints, threshold, octaves, suppress_scale_nonmaxima=True):
    """BriskFeatureDetector::ComputeScale; None where the reference has no defined result."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    k = np.ascontiguousarray(keypoints, KP)
    out = C.c_void_p()
    n = lib().bo_compute_scale(_p(img), w, h, threshold, octaves, int(bool(suppress_scale_nonmaxima)),
                               _p(k) if len(k) else None, len(k), C.byref(out))
    if n < 0:
        return None
    return _take_kps(out, n)


class ScaleSpace:
    """Stage-level access: pyramid images, threshold maps, lazy score maps, keypoints."""

    def __init__(self, img, threshold, octaves):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        self._h = lib().bo_scale_space_create(_p(img), w, h, threshold, octaves)
        self.layers = lib().bo_scale_space_layers(self._h)

    def map(self, layer, which):
        w, h = C.c_int32(), C.c_int32()
        p = lib().bo_scale_space_map(self._h, layer, which, C.byref(w), C.byref(h))
        return np.frombuffer(C.string_at(p, w.value * h.value), np.uint8).reshape(h.value, w.value).copy()

    def image(self, layer):
        return self.map(layer, 0)

    def scores(self, layer):
        return self.map(layer, 1)

    def thrmap(self, layer):
        return self.map(layer, 2)

    def keypoints(self):
        out = C.c_void_p()
        n = lib().bo_scale_space_get_keypoints(self._h, C.byref(out))
        return _take_kps(out, n)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().bo_scale_space_destroy(self._h)
            self._h = None


class Extractor:
    def __init__(self, rotation_invariant=True, scale_invariant=True, version=2, pattern_scale=1.0,
                 pattern_text=None):
        t = None if pattern_text is None else pattern_text.encode()
        self._h = lib().bo_extractor_create(int(rotation_invariant), int(scale_invariant), version,
                                            pattern_scale, t)
        assert self._h
        self.strings = lib().bo_extractor_descriptor_size(self._h)
        self.points = lib().bo_extractor_points(self._h)

    def scale_list(self):
        return np.frombuffer(C.string_at(lib().bo_extractor_scale_list(self._h), 64 * 4), np.float32).copy()

    def size_list(self):
        return np.frombuffer(C.string_at(lib().bo_extractor_size_list(self._h), 64 * 4), np.uint32).copy()

    def pattern(self):
        n = 64 * 1024 * self.points * 3
        return np.frombuffer(C.string_at(lib().bo_extractor_pattern(self._h), n * 4),
                             np.float32).reshape(64, 1024, self.points, 3)

    def scale_index(self, size):
        return lib().bo_extractor_scale_index(self._h, float(size))

    def compute(self, img, kps):
        """Returns (filtered keypoints with angle, descriptors)."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        k = np.ascontiguousarray(kps, KP).copy()
        desc = np.zeros((max(len(k), 1), self.strings), np.uint8)
        n = lib().bo_extractor_compute(self._h, _p(img), w, h, _p(k), len(k), _p(desc))
        return k[:n].copy(), desc[:n].copy()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().bo_extractor_destroy(self._h)
            self._h = None


DMATCH = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])


def hamming(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().bo_hamming(_p(a), _p(b), a.size)


def _match_args(query, train, masks):
    query = np.ascontiguousarray(query, np.uint8)
    train = [np.ascontiguousarray(t, np.uint8) for t in train]
    nimg = len(train)
    tptr = (C.c_void_p * max(nimg, 1))(*[t.ctypes.data for t in train])
    ntr = np.array([t.shape[0] for t in train] + [0], np.int32)
    tpitch = np.array([t.strides[0] if t.shape[0] else query.shape[1] for t in train] + [0], np.int32)
    mptr, mpitch, keep = None, None, None
    if masks is not None:
        keep = [None if m is None else np.ascontiguousarray(m, np.uint8) for m in masks]
        mptr = (C.c_void_p * max(nimg, 1))(*[None if m is None else m.ctypes.data for m in keep])
        mpitch = np.array([0 if m is None else m.strides[0] for m in keep] + [0], np.int32)
    return query, train, nimg, tptr, ntr, tpitch, mptr, mpitch, keep


def match_knn(query, train, k, masks=None):
    query, train, nimg, tptr, ntr, tpitch, mptr, mpitch, keep = _match_args(query, train, masks)
    nq = query.shape[0]
    out = np.zeros((nq, max(k, 1)), DMATCH)
    cnt = np.zeros(max(nq, 1), np.int32)
    lib().bo_match_knn(_p(query), nq, query.strides[0] if nq else query.shape[1], query.shape[1], nimg, tptr, _p(ntr),
                       _p(tpitch), mptr, None if mpitch is None else _p(mpitch), k, _p(out), _p(cnt))
    return [out[i, :cnt[i]].copy() for i in range(nq)]


def match_radius(query, train, max_distance, masks=None):
    query, train, nimg, tptr, ntr, tpitch, mptr, mpitch, keep = _match_args(query, train, masks)
    nq = query.shape[0]
    cnt = np.zeros(max(nq, 1), np.int32)
    ptr = lib().bo_match_radius(_p(query), nq, query.strides[0] if nq else query.shape[1], query.shape[1], nimg, tptr,
                                _p(ntr), _p(tpitch), mptr, None if mpitch is None else _p(mpitch), float(max_distance),
                                _p(cnt))
    total = int(cnt[:nq].sum())
    flat = np.frombuffer(C.string_at(ptr, total * DMATCH.itemsize), dtype=DMATCH).copy() if total else np.zeros(0, DMATCH)
    lib().bo_free(ptr)
    rows, o = [], 0
    for i in range(nq):
        rows.append(flat[o:o + cnt[i]])
        o += cnt[i]
    return rows


def enforce_uniformity(kps, rows, cols, radius, max_keypoints=0x7FFFFFFF):
    kps = np.ascontiguousarray(kps, KP)
    out = np.zeros(max(len(kps), 1), KP)
    n = lib().bo_enforce_uniformity(_p(kps), len(kps), rows, cols, float(radius), int(max_keypoints), _p(out))
    return out[:n].copy()


def key_point_bucketing(kps, rows, cols, max_keypoints, nbu, nbv):
    """KeyPointBucketing (key-point-bucketing-inl.h:40-112) on (x, y, response); None for arguments the reference CHECKs"""
    kps = np.ascontiguousarray(kps, KP)
    out = np.zeros(max(len(kps), 1), KP)
    n = lib().bo_key_point_bucketing(_p(kps), len(kps), rows, cols, int(max_keypoints), int(nbu), int(nbv), _p(out))
    return None if n < 0 else out[:n].copy()


def halfsample16(img):
    img = np.ascontiguousarray(img, np.uint16)
    h, w = img.shape
    out = np.zeros((h // 2, w // 2), np.uint16)
    return out if lib().bo_halfsample16(_p(img), w, h, _p(out)) == 0 else None


def twothirdsample16(img):
    img = np.ascontiguousarray(img, np.uint16)
    h, w = img.shape
    out = np.zeros((h // 3 * 2, w // 3 * 2), np.uint16)
    return out if lib().bo_twothirdsample16(_p(img), w, h, _p(out)) == 0 else None


def integral16(img):
    img = np.ascontiguousarray(img, np.uint16)
    h, w = img.shape
    out = np.zeros((h + 1, w + 1), np.float32)
    lib().bo_integral_image16(_p(img), w, h, _p(out))
    return out
