/* brisk_oracle_16bit.c - CPU restatement of the reference's 16-bit image functions.
 *
 * TEST INFRASTRUCTURE ONLY (see brisk_oracle.h): never linked into or loaded by the product library.
 *
 *   bo_halfsample16       Halfsample16      brisk/src/image-down-sampling.cc:56-139
 *   bo_twothirdsample16   Twothirdsample16  brisk/src/image-down-sampling.cc:394-548
 *   bo_integral_image16   IntegralImage16   brisk/include/brisk/internal/integral-image.h:163-218
 *
 * The block structure of the SSE code is kept (blocks of 16 / 12 source columns, the last block re-done at the right
 * border), so that the written region and every rounding step are the reference's:
 *   Halfsample16: avg(avg(a, b), avg(sat(sat(c + 1) + 1), d)) with _mm_avg_epu16 = (x + y + 1) >> 1 and saturating adds
 *     on the lower left pixel (:113-116);
 *   Twothirdsample16: (4 corner + 2 edge + 2 edge + centre) / 9 in 32-bit integers, then _mm_packs_epi32, i.e. SIGNED
 *     saturation: results above 32767 are stored as 32767 (:518-519);
 *   IntegralImage16: float sums; a row's running sum adds value / 65536 for the columns handled four at a time and the
 *     RAW value for the 0..3 remaining columns (:213-216).
 * Pinned by scalar per-pixel restatements in tests/test_oracle_golden.py (what test-downsampling.cc:67-142 does for the
 * 8-bit functions; the reference has no test of the 16-bit ones). */
#include <stdint.h>
#include <string.h>

#include "brisk_oracle.h"

static uint16_t sat_add16(uint16_t a, uint16_t b) { const unsigned s = (unsigned)a + b; return (uint16_t)(s > 65535u ? 65535u : s); }
static uint16_t avg16(uint16_t a, uint16_t b) { return (uint16_t)(((unsigned)a + b + 1u) >> 1); }

/* src: w x h (row stride w), dst: (w / 2) x (h / 2).  Returns 0, or -1 where the reference's loop writes nothing (w / 2 * 2 < 16) */
int bo_halfsample16(const uint16_t* src, int w, int h, uint16_t* dst) {
  const int dw = w / 2;
  const int colsMax = (w / 2) * 2 - 16;
  const int rows = (h / 2) * 2 - 1;
  if (colsMax < 0) return -1;
  for (int y = 0; y < rows; y += 2) {
    int end = 0, x_store = 0;
    for (int x = 0; x <= colsMax; x += 16) {
      for (int k = 0; k < 8; ++k) {
        const uint16_t i00 = src[(size_t)y * w + x + 2 * k], i01 = src[(size_t)y * w + x + 2 * k + 1];
        uint16_t i10 = src[(size_t)(y + 1) * w + x + 2 * k];
        const uint16_t i11 = src[(size_t)(y + 1) * w + x + 2 * k + 1];
        i10 = sat_add16(i10, 1);
        const uint16_t r1 = avg16(i00, i01);
        i10 = sat_add16(i10, 1);
        const uint16_t r2 = avg16(i10, i11);
        dst[(size_t)(y / 2) * dw + x_store + k] = avg16(r1, r2);
      }
      x_store += 8;
      if (end) break;
      if (x + 16 >= colsMax) { x = colsMax - 16; x_store = dw - 8; end = 1; }
    }
  }
  return 0;
}

static uint16_t packs32(int v) { return (uint16_t)(int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v)); }

/* src: w x h, dst: (w / 3 * 2) x (h / 3 * 2).  Returns -1 where the reference's loop writes nothing (w / 3 * 3 < 12) */
int bo_twothirdsample16(const uint16_t* src, int w, int h, uint16_t* dst) {
  const int dw = (w / 3) * 2;
  const int colsMax = (w / 3) * 3 - 12;
  const int rows = (h / 3) * 3 - 2;
  if (colsMax < 0) return -1;
  for (int y = 0; y < rows; y += 3) {
    int end = 0, x_store = 0;
    for (int x = 0; x <= colsMax; x += 12) {
      for (int b = 0; b < 4; ++b) {  /* four 3x3 blocks -> 2x2 outputs each */
        const uint16_t* p0 = src + (size_t)y * w + x + 3 * b;
        const uint16_t* p1 = p0 + w;
        const uint16_t* p2 = p1 + w;
        const int mid = (int)p1[1];
        const int r1l = mid + 2 * (int)p1[0], r1r = mid + 2 * (int)p1[2];
        const int t0 = 4 * (int)p0[0] + 2 * (int)p0[1] + r1l, t1 = 4 * (int)p0[2] + 2 * (int)p0[1] + r1r;
        const int b0 = 4 * (int)p2[0] + 2 * (int)p2[1] + r1l, b1 = 4 * (int)p2[2] + 2 * (int)p2[1] + r1r;
        uint16_t* d0 = dst + (size_t)(y / 3 * 2) * dw + x_store + 2 * b;
        d0[0] = packs32(t0 / 9); d0[1] = packs32(t1 / 9);
        d0[dw] = packs32(b0 / 9); d0[dw + 1] = packs32(b1 / 9);
      }
      x_store += 8;
      if (end) break;
      if (x + 12 >= colsMax) { x = colsMax - 12; x_store = dw - 8; end = 1; }
    }
  }
  return 0;
}

/* src: w x h u16, out: (h + 1) x (w + 1) floats */
void bo_integral_image16(const uint16_t* src, int w, int h, float* out) {
  const int sumstep = w + 1;
  memset(out, 0, (size_t)(w + 1) * sizeof(float));
  float* sum = out + sumstep + 1;
  const int maxCols = w - 4;
  const float cvtScale = (float)(1.0 / 65536.0);
  for (int y = 0; y < h; ++y, src += w, sum += sumstep) {
    float s = sum[-1] = 0.0f;
    int x;
    for (x = 0; x <= maxCols; x += 4) {
      const float s0 = s + src[x] * cvtScale;
      const float s1 = s0 + src[x + 1] * cvtScale;
      const float s2 = s1 + src[x + 2] * cvtScale;
      const float s3 = s2 + src[x + 3] * cvtScale;
      s = s3;
      sum[x] = s0 + sum[x - sumstep];
      sum[x + 1] = s1 + sum[x + 1 - sumstep];
      sum[x + 2] = s2 + sum[x + 2 - sumstep];
      sum[x + 3] = s3 + sum[x + 3 - sumstep];
    }
    for (int _x = x; _x < w; ++_x) {  /* (:213-216: the raw value, not value / 65536) */
      s += src[_x];
      sum[_x] = sum[_x - sumstep] + s;
    }
  }
}
