// test_multi_image.cc - the multi-image overloads the reference's classes inherit from their OpenCV bases
// (cv::FeatureDetector::detect(const vector<Mat>&, vector<vector<KeyPoint>>&, const vector<Mat>& masks),
// cv::DescriptorExtractor::compute(const vector<Mat>&, vector<vector<KeyPoint>>&, vector<Mat>&); brisk-feature-detector.h:51,
// brisk-descriptor-extractor.h:54): images of one size run as ONE batch on the device, and must give exactly what the
// single-image calls give - which in turn are pinned on the reference's goldens (test_binary_equal.cc).  Also the lists that
// cannot be batched (differing sizes, masks) and a list with an image whose keypoint list is empty.
// Usage: test_multi_image <golden dir>; exit code 0 = every image of every list bit-equal to its single-image call.
#include <brisk/brisk.h>

#include "set_serialization.h"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

typedef std::vector<agast::KeyPoint> Kps;

static bool same_kps(const Kps& a, const Kps& b) {
  return a.size() == b.size() && (a.empty() || memcmp(a.data(), b.data(), a.size() * sizeof(agast::KeyPoint)) == 0);
}
static bool same_mat(const agast::Mat& a, const agast::Mat& b) {
  if (a.rows != b.rows || a.cols != b.cols) return false;
  for (int r = 0; r < a.rows; ++r)
    if (memcmp(a.data + (size_t)r * a.step, b.data + (size_t)r * b.step, (size_t)a.cols) != 0) return false;
  return true;
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : ".";
  try {
    const std::vector<setio::DatasetEntry> set = setio::ReadSet(dir + "/brisk_verification_ast.set");
    std::vector<agast::Mat> two;
    for (size_t i = 0; i < set.size() && i < 2; ++i) two.push_back(set.at(i).image.mat);
    if (two.size() < 2 || two[0].rows != two[1].rows || two[0].cols != two[1].cols) { printf("golden set: need two images of one size\n"); return 2; }
    brisk::BriskFeatureDetector det(70);
    brisk::BriskDescriptorExtractor ext;
    int bad = 0;
    // 1. seven images of one size (the goldens, flipped copies, a blank one): one batch
    std::vector<agast::Mat> imgs;
    for (int i = 0; i < 7; ++i) {
      agast::Mat m = two[i & 1].clone();
      if (i >= 2 && i < 6)  // (content of its own: rows mirrored top to bottom, every second one also left to right)
        for (int y = 0; y < m.rows; ++y)
          for (int x = 0; x < m.cols; ++x)
            m.data[(size_t)y * m.step + x] = two[i & 1].data[(size_t)(m.rows - 1 - y) * two[i & 1].step + ((i & 2) ? m.cols - 1 - x : x)];
      if (i == 6) memset(m.data, 128, m.step * (size_t)m.rows);  // no keypoints at all
      imgs.push_back(m);
    }
    std::vector<Kps> kps;
    det.detect(imgs, kps);
    std::vector<Kps> single(imgs.size());
    for (size_t i = 0; i < imgs.size(); ++i) {
      det.detect(imgs[i], single[i]);
      if (!same_kps(kps[i], single[i])) { printf("detect: image %zu differs (%zu vs %zu keypoints)\n", i, kps[i].size(), single[i].size()); ++bad; }
    }
    std::vector<agast::Mat> desc;
    std::vector<Kps> kd = kps;
    ext.compute(imgs, kd, desc);
    for (size_t i = 0; i < imgs.size(); ++i) {
      Kps k1 = single[i];
      agast::Mat d1;
      ext.compute(imgs[i], k1, d1);
      if (!same_kps(kd[i], k1) || !same_mat(desc[i], d1)) { printf("compute: image %zu differs (%zu vs %zu rows)\n", i, kd[i].size(), k1.size()); ++bad; }
    }
    printf("batch of %zu images: %zu + %zu + ... keypoints, blank image %zu\n", imgs.size(), kps[0].size(), kps[1].size(), kps[6].size());
    // 2. lists that are not batched: a smaller image among them; masks
    std::vector<agast::Mat> mixed = {imgs[0], agast::Mat(imgs[1].rows - 64, imgs[1].cols - 64, CV_8UC1, imgs[1].data, imgs[1].step).clone(), imgs[2]};
    std::vector<Kps> km;
    det.detect(mixed, km);
    std::vector<agast::Mat> dm;
    std::vector<Kps> kmd = km;
    ext.compute(mixed, kmd, dm);
    for (size_t i = 0; i < mixed.size(); ++i) {
      Kps k1;
      agast::Mat d1;
      det.detect(mixed[i], k1);
      if (!same_kps(km[i], k1)) { printf("mixed sizes: detect of image %zu differs\n", i); ++bad; }
      ext.compute(mixed[i], k1, d1);
      if (!same_kps(kmd[i], k1) || !same_mat(dm[i], d1)) { printf("mixed sizes: compute of image %zu differs\n", i); ++bad; }
    }
    std::vector<agast::Mat> masks(2);
    for (int i = 0; i < 2; ++i) {
      masks[i] = agast::Mat::zeros(two[i].rows, two[i].cols, CV_8UC1);
      for (int y = 100; y < 400; ++y) memset(masks[i].data + (size_t)y * masks[i].step + 50 + 100 * i, 255, 500);
    }
    std::vector<Kps> kmask;
    det.detect(two, kmask, masks);
    for (int i = 0; i < 2; ++i) {
      Kps k1;
      det.detect(two[i], k1, masks[i]);
      if (!same_kps(kmask[i], k1) || k1.empty()) { printf("masks: image %d differs\n", i); ++bad; }
    }
    if (bad) { printf("FAILED: %d difference(s)\n", bad); return 1; }
    printf("multi-image OK\n");
    return 0;
  } catch (const std::exception& e) {
    printf("%s\n", e.what());
    return 2;
  }
}
