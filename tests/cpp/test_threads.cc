// test_threads.cc - the drop-in host classes used from several threads at once.
// The reference's classes are re-entrant (BriskFeatureDetector::detectImpl is const and builds its state per call,
// brisk-feature-detector.cc:77-85; the extractor's tables are immutable after construction), so N threads with their
// own detector objects and ONE shared extractor must produce exactly what a single thread produces.  Every thread uses
// different parameters (threshold, octaves, uniformity radius) so that any state shared between calls would show.
// Usage: test_threads <golden dir> [threads] [iterations]; exit code 0 = all threads bit-equal to the serial run.
//        test_threads --time <threads> <seconds> [width height] [--pinned] [--same-image]: throughput of the classes under N host threads - every
//        thread has its own four synthetic frames (1080p by default) and calls detect() + compute() on them for <seconds>;
//        every result is compared with the serial run's.  Prints one JSON line (aggregate frames/s, per-call latency, host
//        CPU time per call) that bench.py collects into config.other_configs["threads"].  --pinned: the frames are registered
//        with brisk_hip_host_register first (DMA straight from the caller's buffer); --same-image: compute() runs under
//        brisk::hip::ScopedSameImage (the caller's word that it is detect()'s unchanged buffer: no second upload);
//        --pool-threshold K: concurrent callers from which the classes hand calls to the device's shared pool (0 = never,
//        default 4: brisk::hip::SetPoolThreshold).
//        test_threads <golden dir> [threads] [iterations] --pool-threshold 1: the bit-equality run with EVERY eligible call pooled.
#include <brisk/brisk.h>
#include <brisk_hip_debug.h>

#include "set_serialization.h"

#include <time.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <random>
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

// a textured frame in the manner of the benchmark's stream (coarse blocks, rectangles, 3x3 blur, noise);
// the pixels need not match tests/synth.py - only a comparable keypoint load (about a thousand at threshold 80)
static std::vector<uint8_t> synthetic_frame(int w, int h, unsigned seed) {
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> u01(0.f, 1.f);
  std::vector<float> a((size_t)w * h), b((size_t)w * h);
  const int cw = w / 40 + 2;
  std::vector<float> coarse((size_t)cw * (h / 40 + 2));
  for (float& c : coarse) c = 60.f + 130.f * u01(rng);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) a[(size_t)y * w + x] = coarse[(size_t)(y / 40) * cw + x / 40];
  const int nrect = (int)(300.0 * w * h / (1920.0 * 1080.0)) + 8;
  for (int r = 0; r < nrect; ++r) {
    const int rw = 6 + (int)(54 * u01(rng)), rh = 6 + (int)(54 * u01(rng));
    const int x0 = (int)((w - rw) * u01(rng)), y0 = (int)((h - rh) * u01(rng));
    const float v = 255.f * u01(rng);
    for (int y = y0; y < y0 + rh; ++y)
      for (int x = x0; x < x0 + rw; ++x) a[(size_t)y * w + x] = v;
  }
  std::normal_distribution<float> noise(0.f, 2.f);
  std::vector<uint8_t> out((size_t)w * h);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      float sum = 0.f;
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int yy = y + dy < 0 ? 0 : (y + dy >= h ? h - 1 : y + dy), xx = x + dx < 0 ? 0 : (x + dx >= w ? w - 1 : x + dx);
          sum += a[(size_t)yy * w + xx];
        }
      const float v = sum / 9.f + noise(rng) + 0.5f;
      out[(size_t)y * w + x] = (uint8_t)(v < 0.f ? 0.f : (v > 255.f ? 255.f : v));
    }
  return out;
}

static double thread_cpu_seconds() {
  timespec ts;
  clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static int time_mode(int nthreads, double seconds, int w, int h, bool pinned, bool same_image) {
  const int per_thread = 4, thr = 80, octaves = 4;
  std::vector<std::vector<uint8_t>> pix((size_t)nthreads * per_thread);
  {  // (frames are generated by all host threads at once: 16 x 4 1080p frames take seconds otherwise)
    std::vector<std::thread> gen;
    for (int t = 0; t < nthreads; ++t)
      gen.emplace_back([&, t] { for (int i = 0; i < per_thread; ++i) pix[(size_t)t * per_thread + i] = synthetic_frame(w, h, 1000u + 17u * t + i); });
    for (std::thread& g : gen) g.join();
  }
  auto mat = [&](size_t i) { return agast::Mat(h, w, CV_8UC1, pix[i].data(), (size_t)w); };
  brisk::BriskDescriptorExtractor ext;  // shared by all threads
  if (pinned)
    for (auto& v : pix)
      if (brisk_hip_host_register(v.data(), v.size()) != BRISK_HIP_OK) { printf("brisk_hip_host_register failed\n"); return 2; }
  std::vector<Result> serial;
  size_t kp_total = 0;
  for (size_t i = 0; i < pix.size(); ++i) {
    serial.push_back(run(mat(i), thr, octaves, 0.0, ext));
    kp_total += serial.back().kps.size();
  }
  std::atomic<int> bad(0), ready(0);
  std::atomic<bool> go(false), stop(false);
  std::vector<long> calls(nthreads, 0);
  std::vector<double> cpu(nthreads, 0.0), lat(nthreads, 0.0), lat_max(nthreads, 0.0);
  std::vector<std::thread> pool;
  for (int t = 0; t < nthreads; ++t)
    pool.emplace_back([&, t] {
      try {
        brisk::BriskFeatureDetector det(thr, octaves);
        Result r;
        for (int i = 0; i < per_thread; ++i) {  // warm-up: the thread's context and its buffers exist before the clock starts
          det.detect(mat((size_t)t * per_thread + i), r.kps);
          ext.compute(mat((size_t)t * per_thread + i), r.kps, r.desc);
        }
        ready++;
        while (!go.load()) std::this_thread::yield();
        const double c0 = thread_cpu_seconds();
        for (long n = 0; !stop.load(); ++n) {
          const size_t i = (size_t)t * per_thread + (size_t)(n % per_thread);
          const auto t0 = std::chrono::steady_clock::now();
          const agast::Mat img = mat(i);
          det.detect(img, r.kps);
          if (same_image) {
            brisk::hip::ScopedSameImage hint;
            ext.compute(img, r.kps, r.desc);
          } else {
            ext.compute(img, r.kps, r.desc);
          }
          const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
          lat[t] += ms;
          if (ms > lat_max[t]) lat_max[t] = ms;
          calls[t] = n + 1;
          if (!equal(r, serial[i])) bad++;
        }
        cpu[t] = thread_cpu_seconds() - c0;
      } catch (const std::exception& e) {
        printf("thread %d: %s\n", t, e.what());
        bad++;
        ready++;
      }
    });
  while (ready.load() < nthreads) std::this_thread::yield();
  const auto t0 = std::chrono::steady_clock::now();
  go = true;
  std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
  stop = true;
  for (std::thread& th : pool) th.join();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  long total = 0;
  double cpu_sum = 0, lat_sum = 0, lmax = 0;
  for (int t = 0; t < nthreads; ++t) { total += calls[t]; cpu_sum += cpu[t]; lat_sum += lat[t]; if (lat_max[t] > lmax) lmax = lat_max[t]; }
  if (pinned)
    for (auto& v : pix) (void)brisk_hip_host_unregister(v.data());
  unsigned long long pg = 0, pc = 0;
  double ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (brisk::hip::PoolThreshold() > 0 && nthreads >= brisk::hip::PoolThreshold())
    if (brisk_hip_pool* pool = brisk::hip::SharedPool(brisk::hip::ThisThread().device)) {
      (void)brisk_hip_pool_stats(pool, &pg, &pc);
      (void)brisk_hip_debug_pool_phases(pool, ph);
    }
  if (pc)  // microseconds per pooled call (leader phases: per group)
    printf("pool phases: join %.0f us, staging %.0f, whole call %.0f | per group: wait for context + members %.0f, queue copies %.0f, queue batch %.0f, "
           "device + transfer %.0f | member wait %.0f per member\n", 1e6 * ph[0] / pc, 1e6 * ph[1] / pc, 1e6 * ph[7] / pc, 1e6 * ph[2] / pg, 1e6 * ph[3] / pg,
           1e6 * ph[4] / pg, 1e6 * ph[5] / pg, pc > pg ? 1e6 * ph[6] / (pc - pg) : 0.0);
  printf("{\"threads\": %d, \"pool_threshold\": %d, \"pool_groups\": %llu, \"pool_calls\": %llu, \"pinned\": %d, \"same_image\": %d, \"width\": %d, \"height\": %d, \"frames_per_s\": %.1f, \"calls\": %ld, \"seconds\": %.3f, "
         "\"latency_ms_mean\": %.4f, \"latency_ms_max\": %.3f, \"host_cpu_ms_per_call\": %.4f, \"mean_keypoints\": %.1f, "
         "\"mismatches\": %d}\n",
         nthreads, brisk::hip::PoolThreshold(), pg, pc, (int)pinned, (int)same_image, w, h, total / dt, total, dt, total ? lat_sum / total : 0.0, lmax, total ? 1e3 * cpu_sum / total : 0.0,
         (double)kp_total / pix.size(), (int)bad);
  return bad ? 1 : 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]) == "--time") {
    try {
      bool pinned = false, same_image = false;
      std::vector<std::string> pos;
      for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--pool-threshold" && i + 1 < argc) { brisk::hip::SetPoolThreshold(atoi(argv[++i])); continue; }
        if (a == "--pinned") pinned = true;
        else if (a == "--same-image") same_image = true;
        else pos.push_back(a);
      }
      return time_mode(pos.size() > 0 ? atoi(pos[0].c_str()) : 4, pos.size() > 1 ? atof(pos[1].c_str()) : 2.0,
                       pos.size() > 3 ? atoi(pos[2].c_str()) : 1920, pos.size() > 3 ? atoi(pos[3].c_str()) : 1080, pinned, same_image);
    } catch (const std::exception& e) {
      printf("%s\n", e.what());
      return 2;
    }
  }
  std::vector<std::string> pos;
  for (int i = 1; i < argc; ++i) {
    if (std::string(argv[i]) == "--pool-threshold" && i + 1 < argc) { brisk::hip::SetPoolThreshold(atoi(argv[++i])); continue; }
    pos.push_back(argv[i]);
  }
  const std::string dir = pos.size() > 0 ? pos[0] : ".";
  const int nthreads = pos.size() > 1 ? atoi(pos[1].c_str()) : 4;
  const int iters = pos.size() > 2 ? atoi(pos[2].c_str()) : 12;
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
