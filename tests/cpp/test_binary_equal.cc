// test_binary_equal.cc - the reference's golden test (brisk/src/test/test-binary-equal.cc:319-333,
// bench-ds.h:311-430) written against the drop-in host classes of the MI355X engine.
// Reads brisk_verification_ast.set, runs BriskFeatureDetector(70) + BriskDescriptorExtractor() on each stored
// image and requires every keypoint field to be exactly equal and every descriptor row identical (the reference
// tolerates a Hamming distance of 5; here 0 is required).  Also checks the Harris set's descriptors on the
// externally provided keypoints.  Exit code 0 = verification success.
#include <brisk/brisk.h>
#include <brisk/brute-force-matcher.h>

#include "set_serialization.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

// .set layout: brisk/src/test/serialization.cc:46-149, bench-ds.cc:57-94 (tests/cpp/set_serialization.h)
struct Entry {
  std::string path;
  agast::Mat image;
  std::vector<agast::KeyPoint> keypoints;
  agast::Mat descriptors;
};
static std::vector<Entry> read_set(const std::string& fn) {
  std::vector<Entry> out;
  for (setio::DatasetEntry& d : setio::ReadSet(fn)) {
    Entry e;
    e.path = d.path;
    e.image = d.image.mat;
    e.keypoints = d.keypoints;
    e.descriptors = d.descriptors.mat;
    out.push_back(e);
  }
  return out;
}

static bool same_kp(const agast::KeyPoint& a, const agast::KeyPoint& b) { return memcmp(&a, &b, sizeof(a)) == 0; }

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "tests/golden";
  int failures = 0;
  try {
    {  // TEST(Brisk, ValidationAST)
      std::vector<Entry> ds = read_set(dir + "/brisk_verification_ast.set");
      brisk::BriskFeatureDetector detector(70);
      brisk::BriskDescriptorExtractor extractor;
      for (Entry& e : ds) {
        std::vector<agast::KeyPoint> kps;
        agast::Mat desc;
        detector.detect(e.image, kps);
        extractor.compute(e.image, kps, desc);
        bool ok = kps.size() == e.keypoints.size() && desc.rows == e.descriptors.rows && desc.cols == e.descriptors.cols;
        for (size_t i = 0; ok && i < kps.size(); ++i) ok = same_kp(kps[i], e.keypoints[i]);
        for (int r = 0; ok && r < desc.rows; ++r)
          ok = memcmp(desc.data + (size_t)r * desc.step, e.descriptors.data + (size_t)r * e.descriptors.step, desc.cols) == 0;
        std::printf("AST    %-24s keypoints %zu / %zu  %s\n", e.path.c_str(), kps.size(), e.keypoints.size(), ok ? "OK" : "MISMATCH");
        failures += ok ? 0 : 1;
      }
    }
    {  // descriptors of the Harris golden set on its stored keypoints (orientation re-estimated)
      std::vector<Entry> ds = read_set(dir + "/brisk_verification_harris.set");
      brisk::BriskDescriptorExtractor extractor(true, true);
      for (Entry& e : ds) {
        std::vector<agast::KeyPoint> kps = e.keypoints;
        for (agast::KeyPoint& k : kps) k.angle = -1;
        std::vector<std::bitset<384> > bits;
        extractor.compute(e.image, kps, bits);
        bool ok = kps.size() == e.keypoints.size();
        for (size_t i = 0; ok && i < kps.size(); ++i) {
          ok = same_kp(kps[i], e.keypoints[i]);
          for (unsigned b = 0; ok && b < 384; ++b)
            ok = bits[i][b] == ((e.descriptors.data[i * e.descriptors.step + (b >> 3)] >> (b & 7)) & 1);
        }
        std::printf("HARRIS %-24s keypoints %zu / %zu  %s\n", e.path.c_str(), kps.size(), e.keypoints.size(), ok ? "OK" : "MISMATCH");
        failures += ok ? 0 : 1;
      }
    }
    {  // TEST(Brisk, MatchBitset) (brisk/src/test/test-match.cc:49-126) through brisk::BruteForceMatcher
      std::vector<Entry> ds = read_set(dir + "/brisk_verification_ast.set");
      brisk::BriskFeatureDetector detector(70, 2);
      brisk::BriskDescriptorExtractor extractor;
      std::vector<agast::KeyPoint> k1, k2;
      agast::Mat d1, d2;
      detector.detect(ds[0].image, k1);
      detector.detect(ds[1].image, k2);
      extractor.compute(ds[0].image, k1, d1);
      extractor.compute(ds[1].image, k2, d2);
      brisk::BruteForceMatcher matcher;
      matcher.add(d2);
      std::vector<std::vector<brisk::DMatch> > knn;
      matcher.knnMatch(d1, knn, 1);
      const double H[3][3] = {{0.8835462624646065, 0.31399802853807735, -40.079602102472926},
                              {-0.18170359412701342, 0.9417589525236417, 152.6910745330205},
                              {2.0127825613685174e-4, -1.5103648761897873e-5, 1.0}};
      unsigned matches = 0, outliers = 0;
      bool ok = knn.size() == k1.size();
      for (size_t i = 0; ok && i < knn.size(); ++i) {
        if (knn[i].empty() || !(knn[i][0].distance < 50)) continue;
        // the matcher's best match is the reference test's own brute-force choice
        int best = -1, best_score = 50;
        for (int j = 0; j < d2.rows; ++j) {
          const int sc = brisk::Hamming()(d1.data + i * d1.step, d2.data + (size_t)j * d2.step, d1.cols);
          if (sc < best_score) { best = j; best_score = sc; }
        }
        ok = ok && best == knn[i][0].trainIdx && (float)best_score == knn[i][0].distance;
        const agast::KeyPoint &a = k1[i], &b = k2[knn[i][0].trainIdx];
        const double w = H[2][0] * a.pt.x + H[2][1] * a.pt.y + H[2][2];
        const double ex = (H[0][0] * a.pt.x + H[0][1] * a.pt.y + H[0][2]) / w - b.pt.x;
        const double ey = (H[1][0] * a.pt.x + H[1][1] * a.pt.y + H[1][2]) / w - b.pt.y;
        ++matches;
        if (std::sqrt(ex * ex + ey * ey) > 5) ++outliers;
      }
      ok = ok && matches > 100 && outliers == 0;
      std::printf("MATCH  %zu / %zu keypoints, %u matches below 50, %u outliers  %s\n", k1.size(), k2.size(), matches, outliers,
                  ok ? "OK" : "MISMATCH");
      failures += ok ? 0 : 1;
    }
  } catch (const std::exception& ex) {
    std::printf("exception: %s\n", ex.what());
    return 2;
  }
  std::printf(failures ? "******* Verification failed *******\n" : "******* Verification success *******\n");
  return failures ? 1 : 0;
}
