// brisk/brute-force-matcher.h - BruteForceMatcher (Hamming) of the MI355X engine.
//
// Drop-in for the reference class (brisk/include/brisk/brute-force-matcher.h:52-93, brisk/src/brute-force-matcher.cc:
// 43-213): knnMatchImpl / radiusMatchImpl forward to brisk_hip_match_knn / brisk_hip_match_radius, where the
// distances (brisk::Hamming, brisk/include/brisk/internal/hamming.h:98-112) and the selection run on the GPU.
// With BRISK_HAVE_OPENCV the class derives from cv::DescriptorMatcher exactly like the reference; without OpenCV a
// self-contained class with the cv::DescriptorMatcher methods the reference's users call is provided.
// Equal distances come back in (distance, imgIdx, trainIdx) order (the reference's std::sort leaves that order
// unspecified for rows of more than 16 matches).
#ifndef BRISK_BRUTE_FORCE_MATCHER_H_
#define BRISK_BRUTE_FORCE_MATCHER_H_

#include <agast/wrap-opencv.h>
#include <brisk/hip-context.h>

#include <vector>

namespace brisk {

#ifndef BRISK_HAVE_OPENCV
// cv::DMatch-compatible (binary-identical to brisk_hip_dmatch)
struct DMatch {
  int queryIdx = -1, trainIdx = -1, imgIdx = -1;
  float distance = 3.402823466e+38f;
  DMatch() {}
  DMatch(int q, int t, int i, float d) : queryIdx(q), trainIdx(t), imgIdx(i), distance(d) {}
  bool operator<(const DMatch& m) const { return distance < m.distance; }
};
#else
typedef cv::DMatch DMatch;
#endif
static_assert(sizeof(DMatch) == sizeof(brisk_hip_dmatch), "DMatch must be binary-identical to cv::DMatch");

// Same functor the reference exposes (hamming.h:53-113); evaluated on the host only by user code that calls it
// directly - the matcher itself computes distances on the GPU.
class Hamming {
 public:
  typedef unsigned char ValueType;
  typedef int ResultType;
  ResultType operator()(const unsigned char* a, const unsigned char* b, const int size) const {
    int r = 0;
    for (int i = 0; i < (size / 16) * 16; ++i) r += __builtin_popcount((unsigned)(a[i] ^ b[i]));
    return r;
  }
};

namespace internal {
// shared by both flavours of the class
inline void MatchOnDevice(const agast::Mat& query, const std::vector<agast::Mat>& train, const std::vector<agast::Mat>& masks,
                          bool radius, int k, float maxDistance, bool compactResult,
                          std::vector<std::vector<DMatch> >& matches) {
  matches.clear();
  if (query.empty()) return;
  if (query.type() != CV_8UC1) throw std::runtime_error("BruteForceMatcher: descriptors must be CV_8UC1");
  brisk_hip_ctx* ctx = hip::DefaultContext();
  const int nimg = (int)train.size();
  std::vector<const uint8_t*> tptr(nimg + 1, nullptr), mptr(nimg + 1, nullptr);
  std::vector<int> ntrain(nimg + 1, 0), tpitch(nimg + 1, 0), mpitch(nimg + 1, 0);
  for (int i = 0; i < nimg; ++i) {
    if (train[i].empty()) { tpitch[i] = query.cols; continue; }
    if (train[i].type() != CV_8UC1 || train[i].cols != query.cols)
      throw std::runtime_error("BruteForceMatcher: train descriptors must have the query's type and size");
    tptr[i] = train[i].data; ntrain[i] = train[i].rows; tpitch[i] = (int)train[i].step;
  }
  const bool use_masks = !masks.empty();
  if (use_masks) {
    if ((int)masks.size() != nimg) throw std::runtime_error("BruteForceMatcher: one mask per train image expected");
    for (int i = 0; i < nimg; ++i) {
      if (masks[i].empty()) continue;
      if (masks[i].rows != query.rows || masks[i].cols != ntrain[i] || masks[i].type() != CV_8UC1)
        throw std::runtime_error("BruteForceMatcher: mask must be CV_8UC1, queries x train descriptors");
      mptr[i] = masks[i].data; mpitch[i] = (int)masks[i].step;
    }
  }
  const int nq = query.rows;
  std::vector<int> count((size_t)nq, 0);
  std::vector<DMatch> flat;
  int per = radius ? 64 : k;
  for (;;) {
    flat.assign((size_t)nq * (size_t)(per > 0 ? per : 1), DMatch());
    brisk_hip_dmatch* out = reinterpret_cast<brisk_hip_dmatch*>(flat.data());
    int rc;
    if (radius)
      rc = brisk_hip_match_radius(ctx, query.data, nq, (int)query.step, query.cols, nimg, tptr.data(), ntrain.data(),
                                  tpitch.data(), use_masks ? mptr.data() : nullptr, mpitch.data(), maxDistance, per, out,
                                  count.data());
    else
      rc = brisk_hip_match_knn(ctx, query.data, nq, (int)query.step, query.cols, nimg, tptr.data(), ntrain.data(),
                               tpitch.data(), use_masks ? mptr.data() : nullptr, mpitch.data(), per, out, count.data());
    hip::Check(ctx, rc, radius ? "brisk_hip_match_radius" : "brisk_hip_match_knn");
    int need = 0;
    for (int q = 0; q < nq; ++q) need = count[q] > need ? count[q] : need;
    if (!radius || need <= per) break;
    per = need;  // some query has more matches inside the radius than the row held: once more with room for all
  }
  matches.reserve((size_t)nq);
  for (int q = 0; q < nq; ++q) {
    if (count[q] == 0 && compactResult) continue;  // (a masked-out query; brute-force-matcher.cc:96-99,180-183)
    matches.push_back(std::vector<DMatch>(flat.begin() + (size_t)q * per, flat.begin() + (size_t)q * per + count[q]));
  }
}
}  // namespace internal

#ifdef BRISK_HAVE_OPENCV
class BruteForceMatcher : public cv::DescriptorMatcher {
 public:
  BruteForceMatcher(const brisk::Hamming& distance = brisk::Hamming()) : distance_(distance) {}
  virtual ~BruteForceMatcher() {}
  virtual bool isMaskSupported() const { return true; }
  virtual cv::Ptr<cv::DescriptorMatcher> clone(bool emptyTrainData = false) const {
    BruteForceMatcher* matcher = new BruteForceMatcher(distance_);
    if (!emptyTrainData)
      for (size_t i = 0; i < trainDescCollection.size(); ++i) matcher->trainDescCollection.push_back(trainDescCollection[i].clone());
    return cv::Ptr<cv::DescriptorMatcher>(matcher);  // (cv::Ptr's raw-pointer constructor is explicit in OpenCV 3)
  }

 protected:
  virtual void knnMatchImpl(cv::InputArray queryDescriptors, std::vector<std::vector<cv::DMatch> >& matches, int k,
                            cv::InputArrayOfArrays masks = cv::noArray(), bool compactResult = false) {
    std::vector<cv::Mat> m;
    masks.getMatVector(m);
    internal::MatchOnDevice(queryDescriptors.getMat(), trainDescCollection, m, false, k, 0.f, compactResult, matches);
  }
  virtual void radiusMatchImpl(cv::InputArray queryDescriptors, std::vector<std::vector<cv::DMatch> >& matches,
                               float maxDistance, cv::InputArrayOfArrays masks = cv::noArray(), bool compactResult = false) {
    std::vector<cv::Mat> m;
    masks.getMatVector(m);
    internal::MatchOnDevice(queryDescriptors.getMat(), trainDescCollection, m, true, 0, maxDistance, compactResult, matches);
  }
  brisk::Hamming distance_;
};
#else
class BruteForceMatcher {
 public:
  BruteForceMatcher(const brisk::Hamming& distance = brisk::Hamming()) : distance_(distance) {}
  virtual ~BruteForceMatcher() {}
  virtual bool isMaskSupported() const { return true; }
  // cv::DescriptorMatcher surface
  virtual void add(const std::vector<agast::Mat>& descriptors) {
    trainDescCollection.insert(trainDescCollection.end(), descriptors.begin(), descriptors.end());
  }
  void add(const agast::Mat& descriptors) { trainDescCollection.push_back(descriptors); }
  const std::vector<agast::Mat>& getTrainDescriptors() const { return trainDescCollection; }
  virtual void clear() { trainDescCollection.clear(); }
  virtual bool empty() const { return trainDescCollection.empty(); }
  virtual void train() {}
  virtual BruteForceMatcher* clone(bool emptyTrainData = false) const {
    BruteForceMatcher* matcher = new BruteForceMatcher(distance_);
    if (!emptyTrainData)
      for (size_t i = 0; i < trainDescCollection.size(); ++i) matcher->trainDescCollection.push_back(trainDescCollection[i].clone());
    return matcher;
  }
  void knnMatch(const agast::Mat& queryDescriptors, std::vector<std::vector<DMatch> >& matches, int k,
                const std::vector<agast::Mat>& masks = std::vector<agast::Mat>(), bool compactResult = false) {
    knnMatchImpl(queryDescriptors, matches, k, masks, compactResult);
  }
  void radiusMatch(const agast::Mat& queryDescriptors, std::vector<std::vector<DMatch> >& matches, float maxDistance,
                   const std::vector<agast::Mat>& masks = std::vector<agast::Mat>(), bool compactResult = false) {
    radiusMatchImpl(queryDescriptors, matches, maxDistance, masks, compactResult);
  }
  // best match of every query against the train collection (queries without a match are skipped)
  void match(const agast::Mat& queryDescriptors, std::vector<DMatch>& matches,
             const std::vector<agast::Mat>& masks = std::vector<agast::Mat>()) {
    std::vector<std::vector<DMatch> > knn;
    knnMatchImpl(queryDescriptors, knn, 1, masks, true);
    matches.clear();
    for (size_t i = 0; i < knn.size(); ++i)
      if (!knn[i].empty()) matches.push_back(knn[i][0]);
  }
  // two-set convenience forms (cv::DescriptorMatcher::match / knnMatch with explicit train descriptors)
  void match(const agast::Mat& queryDescriptors, const agast::Mat& trainDescriptors, std::vector<DMatch>& matches) {
    BruteForceMatcher tmp(distance_);
    tmp.add(trainDescriptors);
    tmp.match(queryDescriptors, matches);
  }
  void knnMatch(const agast::Mat& queryDescriptors, const agast::Mat& trainDescriptors,
                std::vector<std::vector<DMatch> >& matches, int k) {
    BruteForceMatcher tmp(distance_);
    tmp.add(trainDescriptors);
    tmp.knnMatch(queryDescriptors, matches, k);
  }

 protected:
  virtual void knnMatchImpl(const agast::Mat& queryDescriptors, std::vector<std::vector<DMatch> >& matches, int k,
                            const std::vector<agast::Mat>& masks, bool compactResult) {
    internal::MatchOnDevice(queryDescriptors, trainDescCollection, masks, false, k, 0.f, compactResult, matches);
  }
  virtual void radiusMatchImpl(const agast::Mat& queryDescriptors, std::vector<std::vector<DMatch> >& matches,
                               float maxDistance, const std::vector<agast::Mat>& masks, bool compactResult) {
    internal::MatchOnDevice(queryDescriptors, trainDescCollection, masks, true, 0, maxDistance, compactResult, matches);
  }
  std::vector<agast::Mat> trainDescCollection;
  brisk::Hamming distance_;
};
#endif  // BRISK_HAVE_OPENCV

}  // namespace brisk
#endif  // BRISK_BRUTE_FORCE_MATCHER_H_
