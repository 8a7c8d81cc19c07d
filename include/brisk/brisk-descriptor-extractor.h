// brisk/brisk-descriptor-extractor.h - BriskDescriptorExtractor of the MI355X engine.
//
// Drop-in for the reference class (brisk/include/brisk/brisk-descriptor-extractor.h:54-202): the eight
// constructors, Version enum, kDescriptorLength, public rotationInvariance / scaleInvariance, descriptorSize(),
// descriptorType() and both compute() overloads.  Tables are built by the engine (host libm, as the reference
// does) and kept on the GPU; compute() forwards to brisk_hip_describe.
#ifndef BRISK_BRISK_DESCRIPTOR_EXTRACTOR_H_
#define BRISK_BRISK_DESCRIPTOR_EXTRACTOR_H_

#include <agast/wrap-opencv.h>
#include <brisk/hip-context.h>

#include <bitset>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace brisk {

#ifdef BRISK_HAVE_OPENCV
class BriskDescriptorExtractor : public cv::Feature2D {
#else
class BriskDescriptorExtractor {
#endif
 public:
  static const unsigned int kDescriptorLength = 384;
  enum Version { briskV1 = 1, briskV2 = 2 };

  explicit BriskDescriptorExtractor() : BriskDescriptorExtractor(true, true) {}
  explicit BriskDescriptorExtractor(bool rotationInvariant, bool scaleInvariant)
      : BriskDescriptorExtractor(rotationInvariant, scaleInvariant, briskV2, 1.0f) {}
  explicit BriskDescriptorExtractor(bool rotationInvariant, bool scaleInvariant, int version)
      : BriskDescriptorExtractor(rotationInvariant, scaleInvariant, version, 1.0f) {}
  explicit BriskDescriptorExtractor(bool rotationInvariant, bool scaleInvariant, int version, float patternScale)
      : rotationInvariance(rotationInvariant), scaleInvariance(scaleInvariant) {
    if (version != briskV1 && version != briskV2)
      throw std::runtime_error("only Version::briskV1 or Version::briskV2 supported!");
    brisk_hip_ctx* ctx = hip::DefaultContext();
    hip::Check(ctx, brisk_hip_pattern_create(ctx, version, patternScale, &pattern_), "brisk_hip_pattern_create");
  }
  explicit BriskDescriptorExtractor(const std::string& fname) : BriskDescriptorExtractor(fname, true) {}
  explicit BriskDescriptorExtractor(const std::string& fname, bool rotationInvariant)
      : BriskDescriptorExtractor(fname, rotationInvariant, true) {}
  explicit BriskDescriptorExtractor(const std::string& fname, bool rotationInvariant, bool scaleInvariant)
      : BriskDescriptorExtractor(fname, rotationInvariant, scaleInvariant, 1.0f) {}
  explicit BriskDescriptorExtractor(const std::string& fname, bool rotationInvariant, bool scaleInvariant,
                                    float patternScale)
      : rotationInvariance(rotationInvariant), scaleInvariance(scaleInvariant) {
    std::ifstream f(fname.c_str());
    if (!f.is_open()) throw std::runtime_error("BriskDescriptorExtractor: cannot open pattern file " + fname);
    std::stringstream ss;
    ss << f.rdbuf();
    brisk_hip_ctx* ctx = hip::DefaultContext();
    hip::Check(ctx, brisk_hip_pattern_create_from_text(ctx, ss.str().c_str(), patternScale, &pattern_),
               "brisk_hip_pattern_create_from_text");
  }
  BriskDescriptorExtractor(const BriskDescriptorExtractor&) = delete;
  BriskDescriptorExtractor& operator=(const BriskDescriptorExtractor&) = delete;
  virtual ~BriskDescriptorExtractor() { brisk_hip_pattern_destroy(pattern_); }

  int descriptorSize() const { return brisk_hip_pattern_descriptor_size(pattern_); }
  int descriptorType() const { return CV_8U; }

  bool rotationInvariance;
  bool scaleInvariance;

  // compute(): filters `keypoints` (border test), writes their orientation, allocates `descriptors`
  // as keypoints.size() x descriptorSize() CV_8UC1 (brisk-descriptor-extractor.cc:601-604, 612-778).
  virtual void compute(const agast::Mat& image, std::vector<agast::KeyPoint>& keypoints, agast::Mat& descriptors) const {
    computeImpl(image, keypoints, descriptors);
  }
  virtual void compute(const agast::Mat& image, std::vector<agast::KeyPoint>& keypoints,
                       std::vector<std::bitset<kDescriptorLength> >& descriptors) const {
    computeImpl(image, keypoints, descriptors);
  }
  // The multi-image overload the reference's class inherits from its OpenCV base (cv::DescriptorExtractor::compute(const vector<Mat>&,
  // vector<vector<KeyPoint>>&, vector<Mat>& descriptors), whose default loops over computeImpl): images of one size and layout run as
  // ONE batch on the device (brisk_hip_describe_images); differing sizes image by image.  Same results either way.
#ifndef BRISK_HAVE_OPENCV
  virtual void compute(const std::vector<agast::Mat>& images, std::vector<std::vector<agast::KeyPoint> >& keypoints,
                       std::vector<agast::Mat>& descriptors) const {
    computeBatch(images, keypoints, descriptors);
  }
#endif
#ifdef BRISK_HAVE_OPENCV
  virtual void compute(cv::InputArrayOfArrays images, std::vector<std::vector<cv::KeyPoint> >& keypoints, cv::OutputArrayOfArrays descriptors) {
    std::vector<cv::Mat> imgs;
    images.getMatVector(imgs);
    if (descriptors.isMatVector()) {
      computeBatch(imgs, keypoints, *static_cast<std::vector<cv::Mat>*>(descriptors.getObj()));
    } else {
      std::vector<cv::Mat> d;
      computeBatch(imgs, keypoints, d);
    }
  }
  virtual void detectAndCompute(cv::InputArray image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& keypoints,
                                cv::OutputArray descriptors, bool /*useProvidedKeypoints*/ = false) {
    computeImpl(image.getMat(), keypoints, descriptors.getMatRef());
  }
#else
  virtual void detectAndCompute(const agast::Mat& image, const agast::Mat& /*mask*/, std::vector<agast::KeyPoint>& keypoints,
                                agast::Mat& descriptors, bool /*useProvidedKeypoints*/ = false) {
    computeImpl(image, keypoints, descriptors);
  }
#endif

 protected:
  void computeBatch(const std::vector<agast::Mat>& images, std::vector<std::vector<agast::KeyPoint> >& keypoints,
                    std::vector<agast::Mat>& descriptors) const {
    const size_t n = images.size();
    if (keypoints.size() != n) throw std::runtime_error("BriskDescriptorExtractor::compute: one keypoint list per image");
    descriptors.assign(n, agast::Mat());
    bool batch = n >= 2;
    size_t total = 0, most = 0;
    for (size_t i = 0; i < n; ++i) {
      if (images[i].type() != CV_8UC1) throw std::runtime_error("Unsupported image format. Must be CV_16UC1 or CV_8UC1.");
      batch = batch && !images[i].empty() && images[i].rows == images[0].rows && images[i].cols == images[0].cols && images[i].step == images[0].step;
      total += keypoints[i].size();
      most = keypoints[i].size() > most ? keypoints[i].size() : most;
    }
    if (!batch) {
      for (size_t i = 0; i < n; ++i) computeImpl(images[i], keypoints[i], descriptors[i]);
      return;
    }
    const int strings = descriptorSize();
    brisk_hip_ctx* ctx = hip::DefaultContext();
    brisk_hip_reserve(ctx, 4 * (int)most, (int)most);
    std::vector<const uint8_t*> ptrs(n);
    std::vector<const brisk_hip_keypoint*> kin(n);
    std::vector<int> nin(n);
    for (size_t i = 0; i < n; ++i) {
      ptrs[i] = images[i].data;
      kin[i] = reinterpret_cast<const brisk_hip_keypoint*>(keypoints[i].data());
      nin[i] = (int)keypoints[i].size();
    }
    // (the border filter only removes keypoints: the provided ones bound the rows; the thread's page-locked result arrays)
    brisk_hip_batch_host_results* dst = hip::ThreadResultScratch().Prepare((int)n, (long long)(total ? total : 1), strings);
    unsigned ticket = 0;
    int flagged = 0;
    int rc = brisk_hip_describe_images(ctx, pattern_, ptrs.data(), (int)n, images[0].cols, images[0].rows, (int)images[0].step, kin.data(), nin.data(),
                                       rotationInvariance ? 1 : 0, scaleInvariance ? 1 : 0, hip::SameImageHint() ? 1 : 0, dst, &ticket);
    if (rc == BRISK_HIP_OK) rc = brisk_hip_batch_download_wait(ctx, ticket, &flagged);
    if (rc != BRISK_HIP_OK && rc != BRISK_HIP_ERR_CAPACITY) hip::Check(ctx, rc, "brisk_hip_describe_images");
    const agast::KeyPoint* rows = reinterpret_cast<const agast::KeyPoint*>(dst->kps);
    for (size_t i = 0; i < n; ++i) {
      if (rc != BRISK_HIP_OK && dst->flags[i]) { computeImpl(images[i], keypoints[i], descriptors[i]); continue; }  // (an engine capacity: the single-image call grows the workspace)
      const long long a = dst->offsets[i], cnt = dst->offsets[i + 1] - a;
      keypoints[i].assign(rows + a, rows + a + cnt);
      descriptors[i] = agast::Mat((int)cnt, strings, CV_8UC1);
      if (cnt > 0) {
        if ((size_t)descriptors[i].step == (size_t)strings) memcpy(descriptors[i].data, dst->desc + (size_t)a * strings, (size_t)cnt * strings);
        else for (long long r = 0; r < cnt; ++r) memcpy(descriptors[i].data + (size_t)r * descriptors[i].step, dst->desc + (size_t)(a + r) * strings, (size_t)strings);
      }
    }
  }
  virtual void computeImpl(const agast::Mat& image, std::vector<agast::KeyPoint>& keypoints,
                           agast::Mat& descriptors) const {
    if (image.type() != CV_8UC1)  // the reference's 16-bit branch is broken; 8-bit only
      throw std::runtime_error("Unsupported image format. Must be CV_16UC1 or CV_8UC1.");
    const int strings = descriptorSize();
    int n = (int)keypoints.size();
    agast::Mat tmp = agast::Mat::zeros(n > 0 ? n : 1, strings, CV_8UC1);
    // several threads inside the classes right now: the call joins the batch the device's shared pool is forming
    const hip::CallScope scope;
    if (hip::PoolThreshold() > 0 && scope.n >= hip::PoolThreshold() && n <= 16384) {
      if (brisk_hip_pool* pool = hip::SharedPool(hip::ThisThread().device)) {
        const hip::PooledImage& li = hip::LastPooledImage();
        const unsigned long long token =
            (hip::SameImageHint() && li.token && li.data == image.data && li.rows == image.rows && li.cols == image.cols) ? li.token : 0ull;
        std::vector<agast::KeyPoint> kin = keypoints;  // (filtered in place: the thread's own context gets the original list if the pool declines)
        int np = n;
        const int rc = brisk_hip_pool_describe(pool, pattern_, image.data, image.cols, image.rows, (int)image.step,
                                               reinterpret_cast<brisk_hip_keypoint*>(kin.data()), &np, tmp.data, (int)tmp.step,
                                               rotationInvariance ? 1 : 0, scaleInvariance ? 1 : 0, token);
        if (rc == BRISK_HIP_OK) {
          kin.resize((size_t)np);
          keypoints.swap(kin);
          descriptors = agast::Mat::zeros(np, strings, CV_8UC1);
          for (int i = 0; i < np; ++i) memcpy(descriptors.data + (size_t)i * descriptors.step, tmp.data + (size_t)i * tmp.step, strings);
          return;
        }
        if (rc != BRISK_HIP_ERR_CAPACITY) throw std::runtime_error(std::string("brisk_hip_pool_describe failed (code ") + std::to_string(rc) + "): " + brisk_hip_pool_last_error(pool));
      }
    }
    brisk_hip_ctx* ctx = hip::DefaultContext();
    brisk_hip_reserve(ctx, 4 * n, n);  // grows the workspace when needed, never shrinks it
    // (hip::ScopedSameImage: the caller's word that this is the unchanged buffer of the thread's last detect() call)
    hip::Check(ctx,
               (hip::SameImageHint() ? brisk_hip_describe_same_image : brisk_hip_describe)(
                   ctx, pattern_, image.data, image.cols, image.rows, (int)image.step,
                   reinterpret_cast<brisk_hip_keypoint*>(keypoints.data()), &n, tmp.data, (int)tmp.step,
                   rotationInvariance ? 1 : 0, scaleInvariance ? 1 : 0),
               "brisk_hip_describe");
    keypoints.resize((size_t)n);
    descriptors = agast::Mat::zeros(n, strings, CV_8UC1);
    for (int i = 0; i < n; ++i) memcpy(descriptors.data + (size_t)i * descriptors.step, tmp.data + (size_t)i * tmp.step, strings);
  }
  virtual void computeImpl(const agast::Mat& image, std::vector<agast::KeyPoint>& keypoints,
                           std::vector<std::bitset<kDescriptorLength> >& descriptors) const {
    agast::Mat d;
    computeImpl(image, keypoints, d);
    descriptors.assign(keypoints.size(), std::bitset<kDescriptorLength>());
    for (size_t k = 0; k < keypoints.size(); ++k)
      for (unsigned b = 0; b < kDescriptorLength && b < (unsigned)d.cols * 8; ++b)
        if (d.data[k * d.step + (b >> 3)] & (1u << (b & 7))) descriptors[k].set(b, true);
  }

  brisk_hip_pattern* pattern_ = nullptr;
};

}  // namespace brisk
#endif  // BRISK_BRISK_DESCRIPTOR_EXTRACTOR_H_
