// brisk_image16.hip - the reference's 16-bit image functions for gfx950:
//   Halfsample16      brisk/src/image-down-sampling.cc:56-139
//   Twothirdsample16  brisk/src/image-down-sampling.cc:394-548
//   IntegralImage16   brisk/include/brisk/internal/integral-image.h:163-218
// They are not on the 8-bit detect + describe path (the reference's own 16-bit describe branch is broken,
// brisk-descriptor-extractor.cc:672-674); offered as stand-alone device functions behind the C ABI with the reference's
// arithmetic: the saturating "+ 2" on the lower left pixel of Halfsample16, the signed pack of Twothirdsample16 (results
// above 32767 become 32767), the float sums of IntegralImage16 in the reference's order (a row's running sum is a chain of
// float additions: one lane per row; the column sums are a chain too: one lane per column).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "brisk_kernels.h"

__global__ void __launch_bounds__(256) k_halfsample16(const uint16_t* __restrict__ src, int sstride, uint16_t* __restrict__ dst, int dstride,
                                                      int dw, int dh) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= dw || y >= dh) return;
  const unsigned top = *reinterpret_cast<const unsigned*>(src + (long)(2 * y) * sstride + 2 * x);
  const unsigned bot = *reinterpret_cast<const unsigned*>(src + (long)(2 * y + 1) * sstride + 2 * x);
  const unsigned i00 = top & 0xFFFFu, i01 = top >> 16, i11 = bot >> 16;
  const unsigned i10 = min((bot & 0xFFFFu) + 2u, 65535u);  // two saturating "+ 1" (:113-115)
  const unsigned r1 = (i00 + i01 + 1u) >> 1, r2 = (i10 + i11 + 1u) >> 1;
  dst[(long)y * dstride + x] = (uint16_t)((r1 + r2 + 1u) >> 1);
}

// one thread per 3 x 3 source block -> 2 x 2 outputs
__global__ void __launch_bounds__(256) k_twothirdsample16(const uint16_t* __restrict__ src, int sstride, uint16_t* __restrict__ dst,
                                                          int dstride, int bw, int bh) {
  const int bx = blockIdx.x * 256 + threadIdx.x, by = blockIdx.y;
  if (bx >= bw || by >= bh) return;
  const uint16_t* p0 = src + (long)(3 * by) * sstride + 3 * bx;
  const uint16_t* p1 = p0 + sstride;
  const uint16_t* p2 = p1 + sstride;
  const int mid = p1[1];
  const int r1l = mid + 2 * (int)p1[0], r1r = mid + 2 * (int)p1[2];
  const int t0 = (4 * (int)p0[0] + 2 * (int)p0[1] + r1l) / 9, t1 = (4 * (int)p0[2] + 2 * (int)p0[1] + r1r) / 9;
  const int b0 = (4 * (int)p2[0] + 2 * (int)p2[1] + r1l) / 9, b1 = (4 * (int)p2[2] + 2 * (int)p2[1] + r1r) / 9;
  uint16_t* d = dst + (long)(2 * by) * dstride + 2 * bx;
  d[0] = (uint16_t)min(t0, 32767); d[1] = (uint16_t)min(t1, 32767);            // _mm_packs_epi32 (:518-519)
  d[dstride] = (uint16_t)min(b0, 32767); d[dstride + 1] = (uint16_t)min(b1, 32767);
}

// row sums: lane = row, a chain of float additions along the row (value / 65536 for the columns the reference takes four
// at a time, the raw value for the remaining 0..3 columns, :213-216); 64 rows x 64 columns tiles go through LDS so that
// global loads and stores run along rows
__global__ void __launch_bounds__(64) k_integral16_rows(const uint16_t* __restrict__ src, int sstride, float* __restrict__ rowsum, int w, int h) {
  __shared__ float tile[64][65];
  const int lane = threadIdx.x, y0 = blockIdx.x * 64;
  const int n4 = w / 4 * 4;
  float s = 0.0f;
  for (int x0 = 0; x0 < w; x0 += 64) {
    for (int r = 0; r < 64; ++r) {
      const int y = y0 + r, x = x0 + lane;
      tile[r][lane] = (y < h && x < w) ? (float)src[(long)y * sstride + x] : 0.0f;
    }
    __syncthreads();
    const int m = min(64, w - x0);
    for (int c = 0; c < m; ++c) {
      const float v = tile[lane][c];
      s = s + ((x0 + c < n4) ? v * (float)(1.0 / 65536.0) : v);
      tile[lane][c] = s;
    }
    __syncthreads();
    for (int r = 0; r < 64; ++r) {
      const int y = y0 + r, x = x0 + lane;
      if (y < h && x < w) rowsum[(long)y * w + x] = tile[r][lane];
    }
    __syncthreads();
  }
}
// column sums: lane = column
__global__ void __launch_bounds__(256) k_integral16_cols(const float* __restrict__ rowsum, float* __restrict__ out, int ostride, int w, int h) {
  const int x = blockIdx.x * 256 + threadIdx.x;  // output column x (0 .. w)
  if (x > w) return;
  out[x] = 0.0f;
  float acc = 0.0f;
  for (int y = 0; y < h; ++y) {
    acc = (x == 0) ? 0.0f : acc + rowsum[(long)y * w + x - 1];
    out[(long)(y + 1) * ostride + x] = acc;
  }
}

void brisk_launch_halfsample16(const uint16_t* src, int sstride, int w, int h, uint16_t* dst, int dstride, hipStream_t s) {
  const int dw = w / 2, dh = h / 2;
  if (dw <= 0 || dh <= 0) return;
  hipLaunchKernelGGL(k_halfsample16, dim3((dw + 255) / 256, dh), dim3(256), 0, s, src, sstride, dst, dstride, dw, dh);
}
void brisk_launch_twothirdsample16(const uint16_t* src, int sstride, int w, int h, uint16_t* dst, int dstride, hipStream_t s) {
  const int bw = w / 3, bh = h / 3;
  if (bw <= 0 || bh <= 0) return;
  hipLaunchKernelGGL(k_twothirdsample16, dim3((bw + 255) / 256, bh), dim3(256), 0, s, src, sstride, dst, dstride, bw, bh);
}
void brisk_launch_integral16(const uint16_t* src, int sstride, int w, int h, float* rowsum, float* out, int ostride, hipStream_t s) {
  hipLaunchKernelGGL(k_integral16_rows, dim3((h + 63) / 64), dim3(64), 0, s, src, sstride, rowsum, w, h);
  hipLaunchKernelGGL(k_integral16_cols, dim3((w + 1 + 255) / 256), dim3(256), 0, s, rowsum, out, ostride, w, h);
}
