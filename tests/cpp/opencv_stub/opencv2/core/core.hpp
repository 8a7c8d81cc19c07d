// opencv2/core/core.hpp - TEST DOUBLE, not OpenCV.
//
// The drop-in headers (include/brisk/*.h, include/agast/wrap-opencv.h) have a -DBRISK_HAVE_OPENCV branch in which the
// classes derive from cv::Feature2D / cv::DescriptorMatcher and take cv::InputArray / cv::OutputArray, like the reference
// (brisk/include/brisk/brisk.h:56-59, brisk-feature-detector.h:51-83).  This image has no OpenCV, so that branch could
// never be compiled; this double declares just enough of the PUBLIC OpenCV 3 / 4 interface - names, signatures and
// conversion rules (an explicit cv::Ptr constructor, MatStep, _InputArray / _OutputArray proxies) - for the branch to be
// compiled and run by tests/test_cpp_classes.py when pkg-config finds no real OpenCV.  It pins nothing about OpenCV's
// behaviour; it catches typos and signature drift in three #ifdef branches before a maintainer does.
#ifndef BRISK_TEST_OPENCV_STUB_CORE_HPP_
#define BRISK_TEST_OPENCV_STUB_CORE_HPP_

#include <stddef.h>
#include <string.h>

#include <memory>
#include <vector>

#define BRISK_TEST_OPENCV_STUB 1

#define CV_8U 0
#define CV_8S 1
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_CN_SHIFT 3
#define CV_MAT_DEPTH_MASK 7
#define CV_MAT_DEPTH(flags) ((flags) & CV_MAT_DEPTH_MASK)
#define CV_MAKETYPE(depth, cn) (CV_MAT_DEPTH(depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_16UC1 CV_MAKETYPE(CV_16U, 1)
#define CV_32SC1 CV_MAKETYPE(CV_32S, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)

namespace cv {

typedef unsigned char uchar;

template <typename T>
class Point_ {
 public:
  Point_() : x(0), y(0) {}
  Point_(T _x, T _y) : x(_x), y(_y) {}
  T x, y;
};
typedef Point_<float> Point2f;
typedef Point_<int> Point;

// cv::Ptr: the raw-pointer constructor is explicit in OpenCV 3 (cvstd.hpp) - the stricter of the two versions
template <typename T>
struct Ptr : public std::shared_ptr<T> {
  Ptr() {}
  template <typename Y> explicit Ptr(Y* p) : std::shared_ptr<T>(p) {}
  template <typename Y> Ptr(const Ptr<Y>& o) : std::shared_ptr<T>(o) {}
  bool empty() const { return !this->get(); }
};

struct MatStep {
  MatStep() : v(0) {}
  MatStep(size_t s) : v(s) {}
  operator size_t() const { return v; }
  MatStep& operator=(size_t s) { v = s; return *this; }
  size_t v;
};

class Mat {
 public:
  enum { AUTO_STEP = 0 };
  Mat() : flags(0), dims(2), rows(0), cols(0), data(nullptr) {}
  Mat(int r, int c, int type) : flags(0), dims(2), rows(0), cols(0), data(nullptr) { create(r, c, type); }
  Mat(int r, int c, int type, void* user, size_t user_step = AUTO_STEP)
      : flags(type), dims(2), rows(r), cols(c), data(static_cast<uchar*>(user)) {
    step = user_step ? user_step : (size_t)c * elemSize();
  }
  void create(int r, int c, int type) {
    if (data && buf_ && r == rows && c == cols && type == this->type()) return;
    flags = type; rows = r; cols = c;
    step = (size_t)c * elemSize();
    buf_.reset(new uchar[(size_t)step * (size_t)(r > 0 ? r : 0) + 64], std::default_delete<uchar[]>());
    data = buf_.get();
  }
  static Mat zeros(int r, int c, int type) {
    Mat m(r, c, type);
    if (m.data && r > 0) memset(m.data, 0, (size_t)m.step * (size_t)r);
    return m;
  }
  Mat clone() const {
    Mat m(rows, cols, type());
    for (int y = 0; y < rows; ++y) memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, (size_t)cols * elemSize());
    return m;
  }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  int type() const { return flags & 0xFFF; }
  int depth() const { return CV_MAT_DEPTH(flags); }
  size_t elemSize() const {
    static const int sz[8] = {1, 1, 2, 2, 4, 4, 8, 2};
    return (size_t)sz[depth()] * (size_t)(((flags >> CV_CN_SHIFT) & 511) + 1);
  }
  bool isContinuous() const { return (size_t)step == (size_t)cols * elemSize(); }
  template <typename T> T& at(int r, int c) { return *reinterpret_cast<T*>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
  template <typename T> const T& at(int r, int c) const {
    return *reinterpret_cast<const T*>(data + (size_t)r * step + (size_t)c * sizeof(T));
  }
  int flags, dims, rows, cols;
  uchar* data;
  MatStep step;

 private:
  std::shared_ptr<uchar> buf_;
};

// proxy classes of function arguments (core/mat.hpp): a Mat or a vector of Mat on the caller's side
class _InputArray {
 public:
  _InputArray() : m_(nullptr), v_(nullptr) {}
  _InputArray(const Mat& m) : m_(const_cast<Mat*>(&m)), v_(nullptr) {}
  _InputArray(const std::vector<Mat>& v) : m_(nullptr), v_(const_cast<std::vector<Mat>*>(&v)) {}
  Mat getMat(int i = -1) const {
    if (m_) return *m_;
    if (v_ && i >= 0 && i < (int)v_->size()) return (*v_)[(size_t)i];
    return Mat();
  }
  void getMatVector(std::vector<Mat>& mv) const {
    mv.clear();
    if (v_) mv = *v_;
    else if (m_ && !m_->empty()) mv.push_back(*m_);
  }
  bool empty() const { return m_ ? m_->empty() : (v_ ? v_->empty() : true); }
  bool isMatVector() const { return v_ != nullptr; }
  void* getObj() const { return v_ ? static_cast<void*>(v_) : static_cast<void*>(m_); }

 protected:
  Mat* m_;
  std::vector<Mat>* v_;
};
class _OutputArray : public _InputArray {
 public:
  _OutputArray() {}
  _OutputArray(Mat& m) : _InputArray(m) {}
  _OutputArray(std::vector<Mat>& v) : _InputArray(v) {}
  Mat& getMatRef(int i = -1) const { return (m_ || i < 0) ? *m_ : (*v_)[(size_t)i]; }
  bool needed() const { return m_ || v_; }
};
class _InputOutputArray : public _OutputArray {
 public:
  _InputOutputArray() {}
  _InputOutputArray(Mat& m) : _OutputArray(m) {}
};
typedef const _InputArray& InputArray;
typedef InputArray InputArrayOfArrays;
typedef const _OutputArray& OutputArray;
typedef OutputArray OutputArrayOfArrays;
typedef const _InputOutputArray& InputOutputArray;
inline InputOutputArray noArray() {
  static _InputOutputArray none;
  return none;
}

class Algorithm {
 public:
  virtual ~Algorithm() {}
  virtual void clear() {}
  virtual bool empty() const { return false; }
};

// core/types.hpp
class KeyPoint {
 public:
  KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
  KeyPoint(Point2f _pt, float _size, float _angle = -1, float _response = 0, int _octave = 0, int _class_id = -1)
      : pt(_pt), size(_size), angle(_angle), response(_response), octave(_octave), class_id(_class_id) {}
  KeyPoint(float x, float y, float _size, float _angle = -1, float _response = 0, int _octave = 0, int _class_id = -1)
      : pt(x, y), size(_size), angle(_angle), response(_response), octave(_octave), class_id(_class_id) {}
  Point2f pt;
  float size, angle, response;
  int octave, class_id;
};
class DMatch {
 public:
  DMatch() : queryIdx(-1), trainIdx(-1), imgIdx(-1), distance(3.402823466e+38f) {}
  DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(-1), distance(d) {}
  DMatch(int q, int t, int i, float d) : queryIdx(q), trainIdx(t), imgIdx(i), distance(d) {}
  int queryIdx, trainIdx, imgIdx;
  float distance;
  bool operator<(const DMatch& m) const { return distance < m.distance; }
};

}  // namespace cv
#endif  // BRISK_TEST_OPENCV_STUB_CORE_HPP_
