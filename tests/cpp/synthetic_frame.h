// synthetic_frame.h - a textured test frame for the timing modes of the C++ tests (test infrastructure)
#ifndef TESTS_CPP_SYNTHETIC_FRAME_H_
#define TESTS_CPP_SYNTHETIC_FRAME_H_
#include <cstdint>
#include <random>
#include <vector>

// a textured frame in the manner of the benchmark's stream (coarse blocks, rectangles, 3x3 blur, noise);
// the pixels need not match tests/synth.py - only a comparable keypoint load (about a thousand at threshold 80)
inline std::vector<uint8_t> synthetic_frame(int w, int h, unsigned seed) {
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

#endif  // TESTS_CPP_SYNTHETIC_FRAME_H_
