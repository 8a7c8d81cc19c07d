"""Reader for the reference's `.set` golden files (test-only helper).

Format restated from /root/reference/brisk/src/test/serialization.cc:46-149 and
bench-ds.cc:57-94 (little-endian): u32 n_entries; per entry: string path
(u32 len + bytes); Mat image (i32 rows, cols, type, elemSize + raw bytes);
u32 n_kp x {f32 angle, i32 class_id, i32 octave, f32 x, f32 y, f32 response,
f32 size}; Mat descriptors; map<string,Blob> (u32 n x {string key, u32 size,
bytes}).
"""
import struct
import numpy as np

KP_DTYPE = np.dtype([("angle", "<f4"), ("class_id", "<i4"), ("octave", "<i4"),
                     ("x", "<f4"), ("y", "<f4"), ("response", "<f4"), ("size", "<f4")])


class _R:
    def __init__(self, b):
        self.b = b
        self.o = 0

    def u32(self):
        v = struct.unpack_from("<I", self.b, self.o)[0]
        self.o += 4
        return v

    def i32(self):
        v = struct.unpack_from("<i", self.b, self.o)[0]
        self.o += 4
        return v

    def raw(self, n):
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def string(self):
        return self.raw(self.u32()).decode("latin1")

    def mat(self):
        rows, cols, typ, esz = self.i32(), self.i32(), self.i32(), self.i32()
        data = np.frombuffer(self.raw(rows * cols * esz), dtype=np.uint8)
        return data.reshape(rows, cols * esz).copy(), typ, esz


def read_set(path):
    """Returns a list of dicts: path, image (HxW u8), keypoints (KP_DTYPE), descriptors (KxS u8)."""
    r = _R(open(path, "rb").read())
    out = []
    for _ in range(r.u32()):
        e = {"path": r.string()}
        img, typ, esz = r.mat()
        assert typ == 0 and esz == 1
        e["image"] = img
        nk = r.u32()
        e["keypoints"] = np.frombuffer(r.raw(nk * KP_DTYPE.itemsize), dtype=KP_DTYPE).copy()
        desc, typ, esz = r.mat()
        e["descriptors"] = desc
        e["descriptor_type"] = (typ, esz)
        blobs = {}
        for _ in range(r.u32()):
            k = r.string()
            blobs[k] = r.raw(r.u32())
        e["blobs"] = blobs
        out.append(e)
    assert r.o == len(r.b), (r.o, len(r.b))
    return out


def write_set(path, entries):
    """Writer for the same format (serialization.cc:46-149, bench-ds.cc:57-94): entries as read_set returns them
    (path, image, keypoints, descriptors, blobs).  Reading a reference golden file and writing it back is
    byte-identical (tests/test_oracle_golden.py::test_set_file_round_trip)."""
    out = bytearray()

    def string(t):
        b = t.encode("latin1")
        out.extend(struct.pack("<I", len(b)))
        out.extend(b)

    def mat(a, typ, esz):
        a = np.ascontiguousarray(a, np.uint8)
        rows, cols = (a.shape[0], a.shape[1] // esz) if a.ndim == 2 else (0, 0)
        out.extend(struct.pack("<iiii", rows, cols, typ, esz))
        out.extend(a.tobytes())

    out.extend(struct.pack("<I", len(entries)))
    for e in entries:
        string(e["path"])
        mat(e["image"], 0, 1)
        k = np.ascontiguousarray(e["keypoints"], KP_DTYPE)
        out.extend(struct.pack("<I", len(k)))
        out.extend(k.tobytes())
        typ, esz = e.get("descriptor_type", (0, 1))
        mat(e["descriptors"], typ, esz)
        blobs = e.get("blobs", {})
        out.extend(struct.pack("<I", len(blobs)))
        for key, val in blobs.items():
            string(key)
            out.extend(struct.pack("<I", len(val)))
            out.extend(val)
    with open(path, "wb") as f:
        f.write(bytes(out))
