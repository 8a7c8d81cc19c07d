// test_multi_image.cc - the multi-image overloads the reference's classes inherit from their OpenCV bases
// (cv::FeatureDetector::detect(const vector<Mat>&, vector<vector<KeyPoint>>&, const vector<Mat>& masks),
// cv::DescriptorExtractor::compute(const vector<Mat>&, vector<vector<KeyPoint>>&, vector<Mat>&); brisk-feature-detector.h:51,
// brisk-descriptor-extractor.h:54): images of one size run as ONE batch on the device, and must give exactly what the
// single-image calls give - which in turn are pinned on the reference's goldens (test_binary_equal.cc).  Also the lists that
// cannot be batched (differing sizes, masks) and a list with an image whose keypoint list is empty.
// Usage: test_multi_image <golden dir>; exit code 0 = every image of every list bit-equal to its single-image call.
//        test_multi_image --time <images per call> <seconds> [width height]: detect(vector) + compute(vector) on synthetic frames
//        (1080p by default, pageable buffers) for <seconds>; one JSON line with the frames/s of the call pair; --same-image: compute()
//        under brisk::hip::ScopedSameImage (the frames of detect(vector) are taken from their device copies).
#include <brisk/brisk.h>

#include "set_serialization.h"
#include "synthetic_frame.h"

#include <chrono>

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

static int time_mode(int n, double seconds, int w, int h, bool same_image) {
  std::vector<std::vector<uint8_t> > pix(n < 16 ? n : 16);
  for (size_t i = 0; i < pix.size(); ++i) pix[i] = synthetic_frame(w, h, 3000u + (unsigned)i);
  std::vector<std::vector<uint8_t> > own((size_t)n);  // (every image its own buffer, as a caller's cv::Mat objects are)
  std::vector<agast::Mat> imgs;
  for (int i = 0; i < n; ++i) {
    own[(size_t)i] = pix[(size_t)i % pix.size()];
    imgs.push_back(agast::Mat(h, w, CV_8UC1, own[(size_t)i].data(), (size_t)w));
  }
  brisk::BriskFeatureDetector det(80, 4);
  brisk::BriskDescriptorExtractor ext;
  std::vector<Kps> kps;
  std::vector<agast::Mat> desc;
  det.detect(imgs, kps);
  ext.compute(imgs, kps, desc);
  size_t rows = 0;
  for (const agast::Mat& d : desc) rows += (size_t)d.rows;
  // the first image against its single-image calls
  Kps k1;
  agast::Mat d1;
  det.detect(imgs[0], k1);
  ext.compute(imgs[0], k1, d1);
  const bool ok = same_kps(kps[0], k1) && same_mat(desc[0], d1);
  const auto t0 = std::chrono::steady_clock::now();
  long calls = 0;
  double dt = 0;
  do {
    det.detect(imgs, kps);
    if (same_image) {
      brisk::hip::ScopedSameImage hint;  // the caller's word: compute() gets detect()'s unchanged buffers - no second upload
      ext.compute(imgs, kps, desc);
    } else {
      ext.compute(imgs, kps, desc);
    }
    ++calls;
    dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  } while (dt < seconds);
  printf("{\"images_per_call\": %d, \"same_image\": %d, \"width\": %d, \"height\": %d, \"frames_per_s\": %.1f, \"call_pairs\": %ld, \"ms_per_call_pair\": %.3f, "
         "\"mean_described\": %.1f, \"first_image_equals_single_calls\": %s}\n",
         n, (int)same_image, w, h, calls * (double)n / dt, calls, 1e3 * dt / calls, (double)rows / n, ok ? "true" : "false");
  return ok ? 0 : 1;
}

int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]) == "--time") {
    try {
      bool same_image = false;
      std::vector<std::string> pos;
      for (int i = 2; i < argc; ++i) {
        if (std::string(argv[i]) == "--same-image") same_image = true;
        else pos.push_back(argv[i]);
      }
      return time_mode(pos.size() > 0 ? atoi(pos[0].c_str()) : 64, pos.size() > 1 ? atof(pos[1].c_str()) : 2.0, pos.size() > 3 ? atoi(pos[2].c_str()) : 1920,
                       pos.size() > 3 ? atoi(pos[3].c_str()) : 1080, same_image);
    } catch (const std::exception& e) {
      printf("%s\n", e.what());
      return 2;
    }
  }
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
    {  // the same list under ScopedSameImage right after detect(vector): no upload, the same rows; then with one image CHANGED in
       // between and no hint: the new pixels count
      std::vector<Kps> k2;
      det.detect(imgs, k2);
      std::vector<agast::Mat> d2;
      std::vector<Kps> kd2 = k2;
      {
        brisk::hip::ScopedSameImage hint;
        ext.compute(imgs, kd2, d2);
      }
      for (size_t i = 0; i < imgs.size(); ++i)
        if (!same_kps(kd2[i], kd[i]) || !same_mat(d2[i], desc[i])) { printf("same-image compute: image %zu differs\n", i); ++bad; }
      det.detect(imgs, k2);
      agast::Mat keep = imgs[2].clone();
      memcpy(imgs[2].data, imgs[0].data, imgs[2].step * (size_t)imgs[2].rows);  // image 2 := image 0's pixels
      std::vector<Kps> kd3(imgs.size());
      kd3[2] = kps[0];
      for (size_t i = 0; i < imgs.size(); ++i) if (i != 2) kd3[i] = kps[i];
      std::vector<agast::Mat> d3;
      ext.compute(imgs, kd3, d3);
      if (!same_kps(kd3[2], kd[0]) || !same_mat(d3[2], desc[0])) { printf("changed image without the hint: stale pixels were described\n"); ++bad; }
      memcpy(imgs[2].data, keep.data, imgs[2].step * (size_t)imgs[2].rows);
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
