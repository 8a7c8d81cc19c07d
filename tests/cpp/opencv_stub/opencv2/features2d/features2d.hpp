// opencv2/features2d/features2d.hpp - TEST DOUBLE, not OpenCV (see opencv2/core/core.hpp of this directory).
// cv::Feature2D and cv::DescriptorMatcher as far as the drop-in headers derive from them: the public, non-virtual entry
// points forward to the virtual ones the way OpenCV 3 / 4 do (detect / compute -> detectAndCompute; knnMatch / radiusMatch
// / match -> knnMatchImpl / radiusMatchImpl on the train collection).
#ifndef BRISK_TEST_OPENCV_STUB_FEATURES2D_HPP_
#define BRISK_TEST_OPENCV_STUB_FEATURES2D_HPP_

#include <opencv2/core/core.hpp>

namespace cv {

class Feature2D : public virtual Algorithm {
 public:
  virtual ~Feature2D() {}
  virtual void detect(InputArray image, std::vector<KeyPoint>& keypoints, InputArray mask = noArray()) {
    Mat none;
    detectAndCompute(image, mask, keypoints, _OutputArray(none), false);
  }
  virtual void compute(InputArray image, std::vector<KeyPoint>& keypoints, OutputArray descriptors) {
    detectAndCompute(image, noArray(), keypoints, descriptors, true);
  }
  virtual void detect(InputArrayOfArrays images, std::vector<std::vector<KeyPoint> >& keypoints, InputArrayOfArrays masks = noArray()) {
    std::vector<Mat> imgs, ms;
    images.getMatVector(imgs);
    if (!masks.empty()) masks.getMatVector(ms);
    keypoints.resize(imgs.size());
    for (size_t i = 0; i < imgs.size(); ++i) {
      if (i < ms.size()) detect(_InputArray(imgs[i]), keypoints[i], _InputArray(ms[i]));
      else detect(_InputArray(imgs[i]), keypoints[i], noArray());
    }
  }
  virtual void compute(InputArrayOfArrays images, std::vector<std::vector<KeyPoint> >& keypoints, OutputArrayOfArrays descriptors) {
    std::vector<Mat> imgs;
    images.getMatVector(imgs);
    if (!descriptors.isMatVector()) return;
    std::vector<Mat>& d = *static_cast<std::vector<Mat>*>(descriptors.getObj());
    d.resize(imgs.size());
    for (size_t i = 0; i < imgs.size(); ++i) compute(imgs[i], keypoints[i], d[i]);
  }
  virtual void detectAndCompute(InputArray, InputArray, std::vector<KeyPoint>&, OutputArray, bool = false) {}
  virtual int descriptorSize() const { return 0; }
  virtual int descriptorType() const { return CV_32F; }
  virtual int defaultNorm() const { return 4; }
};
typedef Feature2D FeatureDetector;
typedef Feature2D DescriptorExtractor;

class DescriptorMatcher : public Algorithm {
 public:
  virtual ~DescriptorMatcher() {}
  virtual void add(InputArrayOfArrays descriptors) {
    std::vector<Mat> v;
    descriptors.getMatVector(v);
    trainDescCollection.insert(trainDescCollection.end(), v.begin(), v.end());
  }
  const std::vector<Mat>& getTrainDescriptors() const { return trainDescCollection; }
  virtual void clear() { trainDescCollection.clear(); }
  virtual bool empty() const { return trainDescCollection.empty(); }
  virtual bool isMaskSupported() const = 0;
  virtual void train() {}
  void match(InputArray queryDescriptors, std::vector<DMatch>& matches, InputArrayOfArrays masks = noArray()) {
    std::vector<std::vector<DMatch> > knn;
    knnMatch(queryDescriptors, knn, 1, masks, true);
    matches.clear();
    for (size_t i = 0; i < knn.size(); ++i)
      if (!knn[i].empty()) matches.push_back(knn[i][0]);
  }
  void knnMatch(InputArray queryDescriptors, std::vector<std::vector<DMatch> >& matches, int k, InputArrayOfArrays masks = noArray(),
                bool compactResult = false) {
    train();
    knnMatchImpl(queryDescriptors, matches, k, masks, compactResult);
  }
  void radiusMatch(InputArray queryDescriptors, std::vector<std::vector<DMatch> >& matches, float maxDistance,
                   InputArrayOfArrays masks = noArray(), bool compactResult = false) {
    train();
    radiusMatchImpl(queryDescriptors, matches, maxDistance, masks, compactResult);
  }
  virtual Ptr<DescriptorMatcher> clone(bool emptyTrainData = false) const = 0;

 protected:
  virtual void knnMatchImpl(InputArray queryDescriptors, std::vector<std::vector<DMatch> >& matches, int k,
                            InputArrayOfArrays masks = noArray(), bool compactResult = false) = 0;
  virtual void radiusMatchImpl(InputArray queryDescriptors, std::vector<std::vector<DMatch> >& matches, float maxDistance,
                               InputArrayOfArrays masks = noArray(), bool compactResult = false) = 0;
  std::vector<Mat> trainDescCollection;
};

}  // namespace cv
#endif  // BRISK_TEST_OPENCV_STUB_FEATURES2D_HPP_
