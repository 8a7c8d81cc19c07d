// The reference's 16-bit image functions through the drop-in header (include/brisk/internal/image-functions-16.h),
// checked against per-pixel readings of the reference arithmetic (image-down-sampling.cc:56-139, 394-548,
// integral-image.h:163-218) the way the reference's test-downsampling.cc:67-142 checks the 8-bit functions.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include <brisk/internal/image-functions-16.h>

int main() {
  const int h = 123, w = 254;
  agast::Mat src(h, w, CV_16UC1);
  unsigned x = 12345u;
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      x = x * 1664525u + 1013904223u;
      src.at<uint16_t>(r, c) = (r % 7 == 0 && c % 5 == 0) ? 65535 : (uint16_t)(x >> 16);
    }
  int bad = 0;
  try {
    agast::Mat half(h / 2, w / 2, CV_16UC1);
    brisk::Halfsample16(src, half);
    for (int r = 0; r < h / 2; ++r)
      for (int c = 0; c < w / 2; ++c) {
        const unsigned a = src.at<uint16_t>(2 * r, 2 * c), b = src.at<uint16_t>(2 * r, 2 * c + 1);
        const unsigned cc = std::min(src.at<uint16_t>(2 * r + 1, 2 * c) + 2u, 65535u), d = src.at<uint16_t>(2 * r + 1, 2 * c + 1);
        bad += half.at<uint16_t>(r, c) != (uint16_t)((((a + b + 1) >> 1) + ((cc + d + 1) >> 1) + 1) >> 1);
      }
    printf("Halfsample16 %s\n", bad ? "MISMATCH" : "OK");
    agast::Mat tt(h / 3 * 2, w / 3 * 2, CV_16UC1);
    brisk::Twothirdsample16(src, tt);
    int bad2 = 0;
    for (int r = 0; r < h / 3; ++r)
      for (int c = 0; c < w / 3; ++c) {
        auto s = [&](int dr, int dc) { return (int)src.at<uint16_t>(3 * r + dr, 3 * c + dc); };
        const int e[4] = {(4 * s(0, 0) + 2 * s(0, 1) + 2 * s(1, 0) + s(1, 1)) / 9, (4 * s(0, 2) + 2 * s(0, 1) + 2 * s(1, 2) + s(1, 1)) / 9,
                          (4 * s(2, 0) + 2 * s(2, 1) + 2 * s(1, 0) + s(1, 1)) / 9, (4 * s(2, 2) + 2 * s(2, 1) + 2 * s(1, 2) + s(1, 1)) / 9};
        for (int k = 0; k < 4; ++k) bad2 += tt.at<uint16_t>(2 * r + (k >> 1), 2 * c + (k & 1)) != (uint16_t)std::min(e[k], 32767);
      }
    printf("Twothirdsample16 %s\n", bad2 ? "MISMATCH" : "OK");
    agast::Mat integ;
    brisk::IntegralImage16(src, &integ);
    int bad3 = (integ.rows != h + 1 || integ.cols != w + 1 || integ.type() != CV_32FC1);
    const int n4 = w / 4 * 4;
    for (int r = 0; r < h && !bad3; ++r) {
      float s = 0.f;
      for (int c = 0; c < w; ++c) {
        s = s + (c < n4 ? src.at<uint16_t>(r, c) * (float)(1.0 / 65536.0) : (float)src.at<uint16_t>(r, c));
        const float want = integ.at<float>(r, c + 1) + s;
        bad3 += memcmp(&want, &integ.at<float>(r + 1, c + 1), 4) != 0;
      }
    }
    printf("IntegralImage16 %s\n", bad3 ? "MISMATCH" : "OK");
    bad += bad2 + bad3;
  } catch (const std::exception& e) {
    printf("exception: %s\n", e.what());
    return 2;
  }
  return bad ? 1 : 0;
}
