// brisk/brisk-feature-detector.h - BriskFeatureDetector of the MI355X engine.
//
// Drop-in for the reference class (brisk/include/brisk/brisk-feature-detector.h:51-83): same constructor,
// public members and methods; detectImpl forwards to brisk_hip_detect (pyramid, contrast-adaptive AGAST,
// 2-D/3-D non-maximum suppression and refinement all run on the GPU).
#ifndef BRISK_BRISK_FEATURE_DETECTOR_H_
#define BRISK_BRISK_FEATURE_DETECTOR_H_

#include <agast/wrap-opencv.h>
#include <brisk/hip-context.h>

#include <string>
#include <vector>

namespace brisk {

#ifdef BRISK_HAVE_OPENCV
class BriskFeatureDetector : public cv::Feature2D {
#else
class BriskFeatureDetector {
#endif
 public:
  BriskFeatureDetector(int thresh, int octaves = 3, bool suppressScaleNonmaxima = true)
      : threshold(thresh), octaves(octaves), m_suppressScaleNonmaxima(suppressScaleNonmaxima) {}
  virtual ~BriskFeatureDetector() {}
  int threshold;
  int octaves;

  // The multi-image overload the reference's class inherits from its OpenCV base
  // (cv::FeatureDetector::detect(const vector<Mat>&, vector<vector<KeyPoint>>&, const vector<Mat>& masks), whose default loops over
  // detectImpl): images of one size and layout without masks run as ONE batch on the device (brisk_hip_detect_images); anything
  // else - differing sizes, masks, post-filters, suppressScaleNonmaxima = false - image by image.  Same results either way.
  // (with OpenCV: the override of cv::Feature2D::detect(InputArrayOfArrays, ...) below)
#ifndef BRISK_HAVE_OPENCV
  void detect(const std::vector<agast::Mat>& images, std::vector<std::vector<agast::KeyPoint> >& keypoints,
              const std::vector<agast::Mat>& masks = std::vector<agast::Mat>()) const {
    detectBatch(images, keypoints, masks);
  }
  void detect(const agast::Mat& image, std::vector<agast::KeyPoint>& keypoints,
              const agast::Mat& mask = agast::Mat()) const {
    detectImpl(image, keypoints, mask);
  }
  // cv::Feature2D-style entry (descriptors are not produced by the detector)
  virtual void detectAndCompute(const agast::Mat& image, const agast::Mat& mask, std::vector<agast::KeyPoint>& keypoints,
                                agast::Mat& /*descriptors*/, bool /*useProvidedKeypoints*/ = false) {
    detectImpl(image, keypoints, mask);
  }
#else
  using cv::Feature2D::detect;  // (the single-image overloads of the base class stay visible)
  virtual void detect(cv::InputArrayOfArrays images, std::vector<std::vector<cv::KeyPoint> >& keypoints,
                      cv::InputArrayOfArrays masks = cv::noArray()) {
    std::vector<cv::Mat> imgs, ms;
    images.getMatVector(imgs);
    if (!masks.empty()) masks.getMatVector(ms);
    detectBatch(imgs, keypoints, ms);
  }
  virtual void detectAndCompute(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints,
                                cv::OutputArray /*descriptors*/, bool /*useProvidedKeypoints*/ = false) {
    detectImpl(image.getMat(), keypoints, mask.getMat());
  }
#endif

  // Engine option, not reference behaviour of this class: EnforceKeyPointUniformity
  // (brisk/include/brisk/internal/uniformity-enforcement-inl.h:44-194; in the reference only the Harris
  // ScaleSpaceFeatureDetector applies it) as a post-filter of the detected keypoints.  radius = 0 switches it off.
  void SetUniformityRadius(double radius, size_t maxNumKpt = 0x7FFFFFFF) {
    m_uniformityRadius = radius;
    m_maxNumKpt = maxNumKpt > 0x7FFFFFFFu ? 0x7FFFFFFF : (int)maxNumKpt;
  }

  // Engine option as well: KeyPointBucketing (brisk/include/brisk/internal/key-point-bucketing-inl.h:40-112), what the
  // reference's ScaleSpaceLayer applies when uniformity enforcement is off (scale-space-layer-inl.h:376-378).  Takes
  // effect while the uniformity radius is 0; (0, 0) switches it off.
  void SetKeyPointBucketing(size_t numBucketsU, size_t numBucketsV, size_t maxNumKpt) {
    m_bucketsU = (int)numBucketsU; m_bucketsV = (int)numBucketsV;
    m_bucketMax = maxNumKpt > 0x7FFFFFFFu ? 0x7FFFFFFF : (int)maxNumKpt;
  }

  // brisk-feature-detector.cc:87-92: scores and scales for provided keypoints (`keypoints` is replaced by the result,
  // up to one entry per layer that admits a point).  One lane per (layer, point) on the device (the walk's phases are
  // order-free among themselves; a call in which a layer admits no point runs the reference's sequential walk); throws
  // where the reference has no defined result (a point within a few rows of a layer's bottom border makes it read
  // beyond the image, brisk-layer.cc:110-115; see brisk_hip_compute_scale).
  void ComputeScale(const agast::Mat& image, std::vector<agast::KeyPoint>& keypoints) const {
    if (image.empty() || image.type() != CV_8UC1) throw std::runtime_error("BriskFeatureDetector: image must be CV_8UC1");
    brisk_hip_ctx* ctx = hip::DefaultContext();
    const std::vector<agast::KeyPoint> in = keypoints;
    brisk_hip_reserve(ctx, 65536, (int)in.size());
    size_t cap = in.size() * (size_t)(octaves == 0 ? 1 : 2 * octaves) + 16384;
    for (;;) {
      keypoints.assign(cap, agast::KeyPoint());
      int n = 0;
      const int rc = brisk_hip_compute_scale(ctx, image.data, image.cols, image.rows, (int)image.step, threshold, octaves,
                                             m_suppressScaleNonmaxima ? 1 : 0,
                                             reinterpret_cast<const brisk_hip_keypoint*>(in.data()), (int)in.size(),
                                             reinterpret_cast<brisk_hip_keypoint*>(keypoints.data()), (int)cap, &n);
      if (rc == BRISK_HIP_ERR_CAPACITY && cap < (1u << 22)) {
        cap *= 4;
        brisk_hip_reserve(ctx, (int)(cap * 4), (int)cap);
        continue;
      }
      if (rc != BRISK_HIP_OK) keypoints.clear();
      hip::Check(ctx, rc, "brisk_hip_compute_scale");
      keypoints.resize((size_t)n);
      return;
    }
  }

 protected:
  void detectBatch(const std::vector<agast::Mat>& images, std::vector<std::vector<agast::KeyPoint> >& keypoints,
                   const std::vector<agast::Mat>& masks) const {
    const size_t n = images.size();
    keypoints.assign(n, std::vector<agast::KeyPoint>());
    bool batch = n >= 2 && masks.empty() && m_suppressScaleNonmaxima && m_uniformityRadius == 0.0 && m_bucketsU == 0 && m_bucketsV == 0;
    for (size_t i = 0; i < n && batch; ++i)
      batch = !images[i].empty() && images[i].type() == CV_8UC1 && images[i].rows == images[0].rows && images[i].cols == images[0].cols &&
              images[i].step == images[0].step;
    if (!batch) {
      for (size_t i = 0; i < n; ++i) detectImpl(images[i], keypoints[i], i < masks.size() ? masks[i] : agast::Mat());
      return;
    }
    brisk_hip_ctx* ctx = hip::DefaultContext();
    std::vector<const uint8_t*> ptrs(n);
    for (size_t i = 0; i < n; ++i) ptrs[i] = images[i].data;
    long long rows_cap = (long long)n * 2048;
    for (int attempt = 0;; ++attempt) {
      // (the thread's page-locked result arrays: the device writes the rows straight into them)
      brisk_hip_batch_host_results* dst = hip::ThreadResultScratch().Prepare((int)n, rows_cap, 0);
      unsigned ticket = 0;
      int flagged = 0;
      int rc = brisk_hip_detect_images(ctx, ptrs.data(), (int)n, images[0].cols, images[0].rows, (int)images[0].step, threshold, octaves, dst,
                                       &ticket);
      if (rc == BRISK_HIP_OK) rc = brisk_hip_batch_download_wait(ctx, ticket, &flagged);
      if (rc != BRISK_HIP_OK && rc != BRISK_HIP_ERR_CAPACITY) hip::Check(ctx, rc, "brisk_hip_detect_images");
      // frames cut for want of rows: once more with what they need (their counts came back); frames that hit an engine capacity
      // go through the single-image call below, which grows the workspace
      long long need = 0;
      bool cut = false;
      for (size_t i = 0; i < n; ++i) { need += dst->counts[i]; cut = cut || (dst->flags[i] & BRISK_HIP_ROWS_CUT); }
      if (cut && attempt == 0) { rows_cap = need + 64; continue; }
      const agast::KeyPoint* rows = reinterpret_cast<const agast::KeyPoint*>(dst->kps);
      for (size_t i = 0; i < n; ++i) {
        if (dst->flags[i]) { detectImpl(images[i], keypoints[i], agast::Mat()); continue; }
        keypoints[i].assign(rows + dst->offsets[i], rows + dst->offsets[i + 1]);
      }
      return;
    }
  }
  // brisk-feature-detector.cc:77-85
  virtual void detectImpl(const agast::Mat& image, std::vector<agast::KeyPoint>& keypoints,
                          const agast::Mat& mask = agast::Mat()) const {
    keypoints.clear();
    if (image.empty()) throw std::runtime_error("BriskFeatureDetector: empty image");
    if (image.type() != CV_8UC1) throw std::runtime_error("BriskFeatureDetector: image must be CV_8UC1");
    // several threads inside the classes right now: the call joins the batch the device's shared pool is forming (plain
    // detection only; whatever the pool cannot serve - a capacity it does not have - takes the thread's own context below)
    const hip::CallScope scope;
    hip::LastPooledImage().token = 0;
    if (hip::PoolThreshold() > 0 && scope.n >= hip::PoolThreshold() && mask.empty() && m_suppressScaleNonmaxima && m_uniformityRadius == 0.0 &&
        m_bucketsU == 0 && m_bucketsV == 0) {
      if (brisk_hip_pool* pool = hip::SharedPool(hip::ThisThread().device)) {
        keypoints.resize(16384);
        int n = 0;
        unsigned long long token = 0;
        const int rc = brisk_hip_pool_detect(pool, image.data, image.cols, image.rows, (int)image.step, threshold, octaves,
                                             reinterpret_cast<brisk_hip_keypoint*>(keypoints.data()), (int)keypoints.size(), &n, &token);
        if (rc == BRISK_HIP_OK) {
          keypoints.resize((size_t)n);
          hip::PooledImage& li = hip::LastPooledImage();
          li.data = image.data; li.rows = image.rows; li.cols = image.cols; li.token = token;
          return;
        }
        keypoints.clear();
        if (rc != BRISK_HIP_ERR_CAPACITY) throw std::runtime_error(std::string("brisk_hip_pool_detect failed (code ") + std::to_string(rc) + "): " + brisk_hip_pool_last_error(pool));
      }
    }
    brisk_hip_ctx* ctx = hip::DefaultContext();  // this thread's workspace
    // the object's post-filter settings travel with the call: the thread's context keeps whatever a user set on it
    brisk_hip_postfilter pf;
    pf.uniformity_radius = m_uniformityRadius; pf.uniformity_max_keypoints = m_maxNumKpt;
    pf.num_buckets_u = m_bucketsU; pf.num_buckets_v = m_bucketsV; pf.bucket_max_keypoints = m_bucketMax;
    size_t cap = 16384;
    for (;;) {
      keypoints.resize(cap);
      int n = 0;
      const int rc = brisk_hip_detect_filtered(ctx, image.data, image.cols, image.rows, (int)image.step, threshold, octaves,
                                               m_suppressScaleNonmaxima ? 1 : 0, mask.empty() ? nullptr : mask.data,
                                               mask.empty() ? 0 : (int)mask.step, &pf,
                                               reinterpret_cast<brisk_hip_keypoint*>(keypoints.data()), (int)cap, &n);
      if (rc == BRISK_HIP_ERR_CAPACITY && cap < (1u << 22)) {  // output buffer too small: retry larger
        keypoints.clear();
        cap *= 4;
        brisk_hip_reserve(ctx, (int)(cap * 4), (int)cap);
        continue;
      }
      if (rc != BRISK_HIP_OK) keypoints.clear();
      hip::Check(ctx, rc, "brisk_hip_detect");
      keypoints.resize((size_t)n);
      return;
    }
  }
  bool m_suppressScaleNonmaxima;
  double m_uniformityRadius = 0.0;
  int m_maxNumKpt = 0x7FFFFFFF;
  int m_bucketsU = 0, m_bucketsV = 0, m_bucketMax = 0;
};

}  // namespace brisk
#endif  // BRISK_BRISK_FEATURE_DETECTOR_H_
