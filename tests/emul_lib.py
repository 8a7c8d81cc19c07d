"""ctypes binding of the TEST-ONLY host emulation of the HIP kernels (tests/emul/brisk_emul.cpp)."""
import ctypes as C
import os
import subprocess
import numpy as np

from oracle_lib import KP, ROOT

EMUL_DIR = os.path.join(ROOT, "tests", "emul")
CSRC = os.path.join(ROOT, "ethzasl_brisk_amd", "csrc")
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(EMUL_DIR, "libbrisk_emul.so")
        srcs = [os.path.join(EMUL_DIR, "brisk_emul.cpp")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC)
                                                               if f.endswith((".h", ".cpp", ".inc"))]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-w", "-o", so,
                                   os.path.join(EMUL_DIR, "brisk_emul.cpp"), os.path.join(CSRC, "brisk_pattern.cpp"), "-lm"])
        L = C.CDLL(so)
        vp = C.c_void_p
        L.emul_detect.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, C.POINTER(vp), vp]
        L.emul_free.argtypes = [vp]
        L.emul_compute_scale.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.POINTER(vp)]
        L.emul_compute_scale_phased.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_uint, C.POINTER(vp)]
        L.emul_oast_Kp.argtypes = [vp, C.c_int]
        L.emul_agast58_Kp.argtypes = [vp, C.c_int]
        L.emul_detect_px.argtypes = [vp, C.c_int, C.c_int]
        L.emul_pattern_create.restype = vp
        L.emul_pattern_create.argtypes = [C.c_int, C.c_float, C.c_char_p]
        L.emul_pattern_destroy.argtypes = [vp]
        L.emul_pattern_strings.argtypes = [vp]
        L.emul_pattern_points.argtypes = [vp]
        L.emul_pattern_tables.argtypes = [vp, vp, vp, vp]
        L.emul_pattern_point.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
        L.emul_scale_index.argtypes = [vp, C.c_float, C.c_int]
        L.emul_scale_index_host.argtypes = [C.c_float]
        L.emul_div_magic_mismatches.argtypes = [vp, vp, C.c_int]
        L.emul_div_magic_mismatches.restype = C.c_long
        L.emul_div_magic_mismatches_d.argtypes = [C.c_int, vp, C.c_int]
        L.emul_div_magic_mismatches_d.restype = C.c_long
        L.emul_describe.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def detect(img, threshold, octaves, shuffle_seed=0, jacobi=0):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = C.c_void_p()
    stats = np.zeros(6, np.int32)
    n = lib().emul_detect(_p(img), w, h, threshold, octaves, shuffle_seed, jacobi, C.byref(out), _p(stats))
    if n < 0:
        lib().emul_free(out)
        return None, stats
    kps = np.frombuffer(C.string_at(out.value, n * KP.itemsize), dtype=KP).copy() if n else np.zeros(0, KP)
    lib().emul_free(out)
    return kps, stats


def compute_scale(img, keypoints, threshold, octaves, suppress_scale_nonmaxima=True):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    k = np.ascontiguousarray(keypoints, KP)
    out = C.c_void_p()
    n = lib().emul_compute_scale(_p(img), w, h, threshold, octaves, int(bool(suppress_scale_nonmaxima)),
                                 _p(k) if len(k) else None, len(k), C.byref(out))
    if n < 0:
        return None
    kps = np.frombuffer(C.string_at(out.value, n * KP.itemsize), dtype=KP).copy() if n else np.zeros(0, KP)
    lib().emul_free(out)
    return kps


def compute_scale_phased(img, keypoints, threshold, octaves, suppress_scale_nonmaxima=True, seed=1):
    """the same call in the pieces the k_cs_* kernels run in parallel, every phase's items in a shuffled order; None = no
    defined result in the reference, 'walk' = a layer admits no provided point (the engine runs the one-lane walk)"""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    k = np.ascontiguousarray(keypoints, KP)
    out = C.c_void_p()
    n = lib().emul_compute_scale_phased(_p(img), w, h, threshold, octaves, int(bool(suppress_scale_nonmaxima)), _p(k), len(k),
                                        seed, C.byref(out))
    if n == -2:
        return "walk"
    if n < 0:
        return None
    kps = np.frombuffer(C.string_at(out.value, n * KP.itemsize), dtype=KP).copy() if n else np.zeros(0, KP)
    lib().emul_free(out)
    return kps


class Pattern:
    def __init__(self, version=2, pattern_scale=1.0, text=None):
        self._h = lib().emul_pattern_create(version, pattern_scale, None if text is None else text.encode())
        assert self._h
        self.strings = lib().emul_pattern_strings(self._h)
        self.points = lib().emul_pattern_points(self._h)

    def tables(self):
        a, b, c = np.zeros(64, np.float32), np.zeros(64, np.int32), np.zeros(64, np.float32)
        lib().emul_pattern_tables(self._h, _p(a), _p(b), _p(c))
        return a, b, c

    def point(self, scale, rot, i):
        o = np.zeros(3, np.float32)
        lib().emul_pattern_point(self._h, scale, rot, i, _p(o))
        return o

    def scale_index(self, size, scale_invariant=1):
        return lib().emul_scale_index(self._h, float(size), scale_invariant)

    def describe(self, img, kps, rotation_invariant=True, scale_invariant=True):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        k = np.ascontiguousarray(kps, KP).copy()
        pitch = max(64, self.strings)
        desc = np.zeros((max(len(k), 1), pitch), np.uint8)
        n = lib().emul_describe(self._h, _p(img), w, h, _p(k), len(k), _p(desc), pitch, int(rotation_invariant),
                                int(scale_invariant))
        return k[:n].copy(), desc[:n, :self.strings].copy()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().emul_pattern_destroy(self._h)
            self._h = None
