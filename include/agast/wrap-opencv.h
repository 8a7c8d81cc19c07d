// agast/wrap-opencv.h - container types used by the BRISK host classes of the MI355X engine.
//
// Same role as the reference's agast/include/agast/wrap-opencv.h:41-99: with -DBRISK_HAVE_OPENCV the
// containers are cv::Mat / cv::KeyPoint (so the classes drop into OpenCV pipelines); without OpenCV a
// minimal self-contained Mat / KeyPoint with the members the BRISK API touches is provided.
// The KeyPoint layout is binary-identical to cv::KeyPoint (and to brisk_hip_keypoint of the C ABI).
#ifndef AGAST_WRAP_OPENCV_H_
#define AGAST_WRAP_OPENCV_H_

#include <stdint.h>
#include <string.h>

#include <memory>
#include <stdexcept>
#include <vector>

#ifdef BRISK_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
namespace agast {
using cv::KeyPoint;
using cv::Mat;
using cv::Point2f;
}  // namespace agast
#else

#ifndef CV_8U
#define CV_8U 0
#define CV_16U 2
#define CV_32S 4
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_16UC1 CV_MAKETYPE(CV_16U, 1)
#define CV_32SC1 CV_MAKETYPE(CV_32S, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#endif

namespace agast {

struct Point2f {
  float x = 0.f, y = 0.f;
};

// cv::KeyPoint-compatible (pt, size, angle, response, octave, class_id; angle / class_id default -1)
struct KeyPoint {
  Point2f pt;
  float size = 0.f;
  float angle = -1.f;
  float response = 0.f;
  int octave = 0;
  int class_id = -1;
  KeyPoint() {}
  KeyPoint(float x, float y, float _size, float _angle = -1.f, float _response = 0.f, int _octave = 0,
           int _class_id = -1)
      : size(_size), angle(_angle), response(_response), octave(_octave), class_id(_class_id) {
    pt.x = x;
    pt.y = y;
  }
};

// Dense single-channel matrix with shared ownership of its buffer (or a non-owning view).
class Mat {
 public:
  int rows = 0, cols = 0;
  unsigned char* data = nullptr;
  size_t step = 0;  // bytes per row

  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, void* user_data, size_t user_step = 0)  // view, no ownership
      : rows(r), cols(c), data(static_cast<unsigned char*>(user_data)), type_(type) {
    step = user_step ? user_step : (size_t)c * elemSize();
  }
  void create(int r, int c, int type) {
    rows = r;
    cols = c;
    type_ = type;
    step = (size_t)c * elemSize();
    buf_.reset(new unsigned char[step * (size_t)(r > 0 ? r : 0) + 64], std::default_delete<unsigned char[]>());
    data = buf_.get();
  }
  static Mat zeros(int r, int c, int type) {
    Mat m(r, c, type);
    if (m.data) memset(m.data, 0, m.step * (size_t)r);
    return m;
  }
  Mat clone() const {
    Mat m(rows, cols, type_);
    for (int y = 0; y < rows; ++y) memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, (size_t)cols * elemSize());
    return m;
  }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  int type() const { return type_; }
  size_t elemSize() const {
    switch (type_ & 7) {
      case CV_8U: case 1: return 1;
      case CV_16U: case 3: return 2;
      default: return 4;
    }
  }
  bool isContinuous() const { return step == (size_t)cols * elemSize(); }
  template <typename T> T& at(int r, int c) { return *reinterpret_cast<T*>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
  template <typename T> const T& at(int r, int c) const {
    return *reinterpret_cast<const T*>(data + (size_t)r * step + (size_t)c * sizeof(T));
  }

 private:
  int type_ = CV_8UC1;
  std::shared_ptr<unsigned char> buf_;
};

}  // namespace agast
#endif  // BRISK_HAVE_OPENCV

namespace agast {
inline float& KeyPointX(KeyPoint& k) { return k.pt.x; }
inline const float& KeyPointX(const KeyPoint& k) { return k.pt.x; }
inline float& KeyPointY(KeyPoint& k) { return k.pt.y; }
inline const float& KeyPointY(const KeyPoint& k) { return k.pt.y; }
inline float& KeyPointSize(KeyPoint& k) { return k.size; }
inline const float& KeyPointSize(const KeyPoint& k) { return k.size; }
inline float& KeyPointAngle(KeyPoint& k) { return k.angle; }
inline const float& KeyPointAngle(const KeyPoint& k) { return k.angle; }
inline float& KeyPointResponse(KeyPoint& k) { return k.response; }
inline const float& KeyPointResponse(const KeyPoint& k) { return k.response; }
inline int& KeyPointOctave(KeyPoint& k) { return k.octave; }
inline const int& KeyPointOctave(const KeyPoint& k) { return k.octave; }
}  // namespace agast

static_assert(sizeof(agast::KeyPoint) == 28, "KeyPoint must be binary-identical to cv::KeyPoint");

#endif  // AGAST_WRAP_OPENCV_H_
