// set_serialization.h - reader / writer of the reference's `.set` containers for the C++ test programs.
// Format: brisk/src/test/serialization.cc:46-149 (scalars little-endian as stored, std::string and std::vector with a
// uint32 length prefix, Mat = rows, cols, type, element size, data; KeyPoint = angle, class_id, octave, x, y, response,
// size) and brisk/src/test/bench-ds.cc:57-94 (DatasetEntry = path, image, keypoints, descriptors, blobs).  Written
// against the drop-in agast::Mat / agast::KeyPoint types; the 16-bit image path of the reference is not covered
// (its describe branch is broken).  A file that is read and written back is byte-identical
// (tests/cpp/test_serialization.cc).
#ifndef TESTS_CPP_SET_SERIALIZATION_H_
#define TESTS_CPP_SET_SERIALIZATION_H_

#include <agast/wrap-opencv.h>

#include <cstdint>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace setio {

template <typename T> inline void Put(const T& v, std::ostream* out) { out->write(reinterpret_cast<const char*>(&v), sizeof(T)); }
template <typename T> inline void Get(T* v, std::istream* in) {
  in->read(reinterpret_cast<char*>(v), sizeof(T));
  if (!in->good()) throw std::runtime_error("unexpected end of .set data");
}

inline void Serialize(const std::string& s, std::ostream* out) {
  Put<uint32_t>((uint32_t)s.size(), out);
  out->write(s.data(), (std::streamsize)s.size());
}
inline void DeSerialize(std::string* s, std::istream* in) {
  uint32_t n;
  Get(&n, in);
  s->resize(n);
  if (n) in->read(&(*s)[0], n);
  if (!in->good()) throw std::runtime_error("unexpected end of .set data");
}

// Mat header as stored: rows, cols, type, element size.  The drop-in Mat is 8-bit only: a row holds cols * esz bytes.
struct StoredMat {
  agast::Mat mat;
  int cols = 0, type = 0, elem_size = 1;
};
inline void Serialize(const StoredMat& m, std::ostream* out) {
  Put<int>(m.mat.rows, out);
  Put<int>(m.cols, out);
  Put<int>(m.type, out);
  Put<int>(m.elem_size, out);
  for (int r = 0; r < m.mat.rows; ++r)
    out->write(reinterpret_cast<const char*>(m.mat.data + (size_t)r * m.mat.step), (std::streamsize)m.cols * m.elem_size);
}
inline void DeSerialize(StoredMat* m, std::istream* in) {
  int rows;
  Get(&rows, in);
  Get(&m->cols, in);
  Get(&m->type, in);
  Get(&m->elem_size, in);
  if (rows < 0 || m->cols < 0 || m->elem_size < 1) throw std::runtime_error("bad matrix header in .set data");
  m->mat = agast::Mat(rows, m->cols * m->elem_size, CV_8UC1);
  for (int r = 0; r < rows; ++r) {
    in->read(reinterpret_cast<char*>(m->mat.data + (size_t)r * m->mat.step), (std::streamsize)m->cols * m->elem_size);
    if (!in->good()) throw std::runtime_error("unexpected end of .set data");
  }
}

inline void Serialize(const agast::KeyPoint& k, std::ostream* out) {
  Put<float>(k.angle, out);
  Put<int>(k.class_id, out);
  Put<int>(k.octave, out);
  Put<float>(k.pt.x, out);
  Put<float>(k.pt.y, out);
  Put<float>(k.response, out);
  Put<float>(k.size, out);
}
inline void DeSerialize(agast::KeyPoint* k, std::istream* in) {
  Get(&k->angle, in);
  Get(&k->class_id, in);
  Get(&k->octave, in);
  Get(&k->pt.x, in);
  Get(&k->pt.y, in);
  Get(&k->response, in);
  Get(&k->size, in);
}

struct DatasetEntry {  // bench-ds.h: path, image, keypoints, descriptors, named blobs
  std::string path;
  StoredMat image;
  std::vector<agast::KeyPoint> keypoints;
  StoredMat descriptors;
  std::vector<std::pair<std::string, std::string> > blobs;  // in file order
};
inline void Serialize(const DatasetEntry& e, std::ostream* out) {
  Serialize(e.path, out);
  Serialize(e.image, out);
  Put<uint32_t>((uint32_t)e.keypoints.size(), out);
  for (const agast::KeyPoint& k : e.keypoints) Serialize(k, out);
  Serialize(e.descriptors, out);
  Put<uint32_t>((uint32_t)e.blobs.size(), out);
  for (const auto& b : e.blobs) {
    Serialize(b.first, out);
    Serialize(b.second, out);
  }
}
inline void DeSerialize(DatasetEntry* e, std::istream* in) {
  DeSerialize(&e->path, in);
  DeSerialize(&e->image, in);
  uint32_t n;
  Get(&n, in);
  e->keypoints.resize(n);
  for (agast::KeyPoint& k : e->keypoints) DeSerialize(&k, in);
  DeSerialize(&e->descriptors, in);
  Get(&n, in);
  e->blobs.resize(n);
  for (auto& b : e->blobs) {
    DeSerialize(&b.first, in);
    DeSerialize(&b.second, in);
  }
}

inline std::vector<DatasetEntry> ReadSet(const std::string& fn) {
  std::ifstream in(fn.c_str(), std::ios::binary);
  if (!in.good()) throw std::runtime_error("cannot open " + fn);
  uint32_t n;
  Get(&n, &in);
  std::vector<DatasetEntry> out(n);
  for (DatasetEntry& e : out) DeSerialize(&e, &in);
  return out;
}
inline void WriteSet(const std::string& fn, const std::vector<DatasetEntry>& set) {
  std::ofstream out(fn.c_str(), std::ios::binary);
  if (!out.good()) throw std::runtime_error("cannot create " + fn);
  Put<uint32_t>((uint32_t)set.size(), &out);
  for (const DatasetEntry& e : set) Serialize(e, &out);
}

}  // namespace setio
#endif  // TESTS_CPP_SET_SERIALIZATION_H_
