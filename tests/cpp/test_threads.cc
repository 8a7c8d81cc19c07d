// test_threads.cc - the drop-in host classes used from several threads at once.
// The reference's classes are re-entrant (BriskFeatureDetector::detectImpl is const and builds its state per call,
// brisk-feature-detector.cc:77-85; the extractor's tables are immutable after construction), so N threads with their
// own detector objects and ONE shared extractor must produce exactly what a single thread produces.  Every thread uses
// different parameters (threshold, octaves, uniformity radius) so that any state shared between calls would show.
// Usage: test_threads <golden dir> [threads] [iterations]; exit code 0 = all threads bit-equal to the serial run.
#include <brisk/brisk.h>

#include "set_serialization.h"

#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

static agast::Mat first_image(const std::string& fn) { return setio::ReadSet(fn).at(0).image.mat; }

struct Result {
  std::vector<agast::KeyPoint> kps;
  agast::Mat desc;
};

static bool equal(const Result& a, const Result& b) {
  if (a.kps.size() != b.kps.size() || a.desc.rows != b.desc.rows || a.desc.cols != b.desc.cols) return false;
  if (!a.kps.empty() && memcmp(a.kps.data(), b.kps.data(), a.kps.size() * sizeof(agast::KeyPoint)) != 0) return false;
  for (int r = 0; r < a.desc.rows; ++r)
    if (memcmp(a.desc.data + (size_t)r * a.desc.step, b.desc.data + (size_t)r * b.desc.step, a.desc.cols) != 0) return false;
  return true;
}

static Result run(const agast::Mat& img, int thr, int octaves, double radius, const brisk::BriskDescriptorExtractor& ext) {
  brisk::BriskFeatureDetector det(thr, octaves);
  det.SetUniformityRadius(radius);
  Result r;
  det.detect(img, r.kps);
  ext.compute(img, r.kps, r.desc);
  return r;
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : ".";
  const int nthreads = argc > 2 ? atoi(argv[2]) : 4;
  const int iters = argc > 3 ? atoi(argv[3]) : 12;
  try {
    const agast::Mat img = first_image(dir + "/brisk_verification_ast.set");
    brisk::BriskDescriptorExtractor ext;  // shared by all threads
    const int thr[8] = {70, 45, 90, 60, 30, 110, 55, 80};
    const int oct[8] = {3, 4, 2, 0, 3, 1, 4, 2};
    const double rad[8] = {0.0, 6.0, 0.0, 12.0, 0.0, 0.0, 9.0, 0.0};
    std::vector<Result> serial;
    for (int t = 0; t < nthreads; ++t) serial.push_back(run(img, thr[t % 8], oct[t % 8], rad[t % 8], ext));
    std::atomic<int> bad(0);
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; ++t)
      pool.emplace_back([&, t] {
        try {
          for (int i = 0; i < iters; ++i)
            if (!equal(run(img, thr[t % 8], oct[t % 8], rad[t % 8], ext), serial[t])) bad++;
        } catch (const std::exception& e) {
          printf("thread %d: %s\n", t, e.what());
          bad++;
        }
      });
    for (std::thread& th : pool) th.join();
    for (int t = 0; t < nthreads; ++t) printf("thread %d: thr %d octaves %d radius %.0f -> %zu keypoints\n", t, thr[t % 8], oct[t % 8], rad[t % 8], serial[t].kps.size());
    if (bad) {
      printf("FAILED: %d result(s) differ from the serial run\n", (int)bad);
      return 1;
    }
    printf("threads OK: %d threads x %d iterations bit-equal to the serial run\n", nthreads, iters);
    return 0;
  } catch (const std::exception& e) {
    printf("%s\n", e.what());
    return 2;
  }
}
