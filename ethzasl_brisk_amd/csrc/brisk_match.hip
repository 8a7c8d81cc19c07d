// brisk_match.hip - Hamming brute-force matcher kernels (the step after the detect + describe path).
//
// Replaces brisk::BruteForceMatcher::commonKnnMatchImpl / commonRadiusMatchImpl
// (brisk/src/brute-force-matcher.cc:80-213) with brisk::Hamming (brisk/include/brisk/internal/hamming.h:98-112)
// as the distance: popcount of a ^ b over size / 16 128-bit words.  Integer work, HBM/L2-bound, no MFMA:
//   k_match_dist    u16 distance matrix of a block of queries against all train descriptors (all train images
//                   concatenated in image order); 0xFFFF = the reference's INT_MAX (masked pair)
//   k_match_knn     one wave per query: k rounds of "first minimum" selection, as the reference does
//   k_match_radius  one wave per query: distance histogram in LDS, then stable placement by (distance, image,
//                   train index)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "brisk_kernels.h"

#define MT_THREADS 256
#define MT_QTILE 32
#define MT_MAXWORDS_LONG 28  // descriptors up to 224 bytes (generateKernel at small pattern scales); BRISK's own are 48 and 64 (8 x u64)

// grid: (ceil(nt / 256), ceil(nqb / MT_QTILE)); thread = one train descriptor, loop over the query tile in LDS
template <int MT_MAXWORDS>
__global__ void __launch_bounds__(MT_THREADS) k_match_dist(const uint8_t* __restrict__ query, int q_pitch, int q0, int nqb,
                                                           const uint8_t* __restrict__ train, int t_pitch, int nt,
                                                           int words /* u64 per descriptor */,
                                                           const uint8_t* __restrict__ mask, long mask_pitch,
                                                           uint16_t* __restrict__ dist, long dist_pitch) {
  __shared__ unsigned long long qs[MT_QTILE][MT_MAXWORDS];
  const int t = blockIdx.x * MT_THREADS + threadIdx.x;
  const int qt0 = blockIdx.y * MT_QTILE;
  for (int i = threadIdx.x; i < MT_QTILE * words; i += MT_THREADS) {
    const int q = i / words, w = i % words;
    unsigned long long v = 0;
    if (qt0 + q < nqb) {
      const uint8_t* p = query + (long)(q0 + qt0 + q) * q_pitch + w * 8;
      for (int b = 0; b < 8; ++b) v |= (unsigned long long)p[b] << (8 * b);  // (no alignment assumption on the rows)
    }
    qs[q][w] = v;
  }
  __syncthreads();
  if (t >= nt) return;
  unsigned long long tv[MT_MAXWORDS];
#pragma unroll
  for (int w = 0; w < MT_MAXWORDS; ++w) {
    tv[w] = 0;
    if (w < words) {
      const uint8_t* p = train + (long)t * t_pitch + w * 8;
      unsigned long long v = 0;
      for (int b = 0; b < 8; ++b) v |= (unsigned long long)p[b] << (8 * b);
      tv[w] = v;
    }
  }
  const int nq = min(MT_QTILE, nqb - qt0);
  for (int q = 0; q < nq; ++q) {
    int d = 0;
#pragma unroll
    for (int w = 0; w < MT_MAXWORDS; ++w)
      if (w < words) d += __popcll(tv[w] ^ qs[q][w]);
    if (mask && mask[(long)(q0 + qt0 + q) * mask_pitch + t] == 0) d = 0xFFFF;
    dist[(long)(qt0 + q) * dist_pitch + t] = (uint16_t)d;
  }
}

// masked-out queries (OpenCV DescriptorMatcher::isMaskedOut: `outCount == masks.size()`): EVERY image has a non-empty
// mask whose row for this query is all zero - the query can match nothing anywhere.
// grid: nqb blocks of 64 threads; img_start[nimg + 1] are the offsets of the images in the concatenated train set,
// has_mask[i] != 0 if image i has a mask.
__global__ void __launch_bounds__(64) k_match_masked_out(const uint8_t* __restrict__ mask, long mask_pitch, int q0,
                                                         const int* __restrict__ img_start, const int* __restrict__ has_mask,
                                                         int nimg, int* __restrict__ masked) {
  const int q = blockIdx.x, lane = threadIdx.x;
  int out = nimg > 0 ? 1 : 0;
  for (int i = 0; i < nimg && out; ++i) {
    if (!has_mask[i]) { out = 0; break; }  // no mask, or no train descriptors (an empty cv::Mat): not counted
    bool any = false;
    for (int t = img_start[i] + lane; t < img_start[i + 1]; t += 64) any |= mask[(long)(q0 + q) * mask_pitch + t] != 0;
    if (__any(any)) out = 0;
  }
  if (lane == 0) masked[q] = out;
}

// (img_start == nullptr: a single train image [0, nt))
__device__ __forceinline__ int mt_image_of(const int* img_start, int nimg, int t) {
  if (!img_start) return 0;
  int i = 0;
  while (i + 1 < nimg && t >= img_start[i + 1]) ++i;
  return i;
}

// one wave per query; out row = (q0 + q) * k
__global__ void __launch_bounds__(64) k_match_knn(const uint16_t* __restrict__ dist, long dist_pitch, int q0, int nt,
                                                  const int* __restrict__ img_start, int nimg, const int* __restrict__ masked,
                                                  int k, BriskDMatch* __restrict__ out, int* __restrict__ out_count) {
  const int q = blockIdx.x, lane = threadIdx.x;
  const uint16_t* row = dist + (long)q * dist_pitch;
  BriskDMatch* orow = out + (long)(q0 + q) * k;
  if (masked && masked[q]) {
    if (lane == 0) out_count[q0 + q] = 0;
    return;
  }
  int last_nonempty = -1;
  if (!img_start) last_nonempty = nt > 0 ? 0 : -1;
  else
    for (int i = 0; i < nimg; ++i)
      if (img_start[i + 1] > img_start[i]) last_nonempty = i;
  int count = 0;
  unsigned long long last = 0;
  bool have_last = false;
  for (int kk = 0; kk < k; ++kk) {
    // first minimum over (distance, concatenated index) = the reference's minMaxLoc per image + strict '<' across
    // images.  The reference overwrites a selected entry with INT_MAX; selections come in strictly increasing
    // (distance, index) order, so "the smallest key above the previous selection" is the same thing without a write.
    unsigned long long best = ~0ull;
    for (int t = lane; t < nt; t += 64) {
      const unsigned long long key = ((unsigned long long)row[t] << 32) | (unsigned)t;
      if ((!have_last || key > last) && key < best) best = key;
    }
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = __shfl_xor(best, off, 64);
      best = o < best ? o : best;
    }
    const unsigned d = (unsigned)(best >> 32);
    if (nt == 0 || d >= 0xFFFFu) break;  // nothing real left
    const int t = (int)(best & 0xFFFFFFFFu);
    if (lane == 0) {
      const int img = mt_image_of(img_start, nimg, t);
      BriskDMatch m;
      m.queryIdx = q0 + q; m.trainIdx = t - (img_start ? img_start[img] : 0); m.imgIdx = img; m.distance = (float)d;
      orow[count] = m;
    }
    last = best;
    have_last = true;
    ++count;
  }
  // reference quirk (brute-force-matcher.cc:139-153): with every entry at INT_MAX the comparison
  // `minVal < bestMatch.distance` still succeeds (2147483647.0 < FLT_MAX, and again against float(INT_MAX) =
  // 2147483648), so the remaining k - count slots are filled with {train 0 of the LAST non-empty image, 2147483648.f}
  if (last_nonempty >= 0 && lane == 0) {
    for (int c = count; c < k; ++c) {
      BriskDMatch m;
      m.queryIdx = q0 + q; m.trainIdx = 0; m.imgIdx = last_nonempty; m.distance = 2147483648.0f;
      orow[c] = m;
    }
  }
  if (last_nonempty >= 0) count = k;
  if (lane == 0) out_count[q0 + q] = count;
}

// one wave per query.  out row = (q0 + q) * cap, at most cap matches are stored, out_count = matches found.
#define MR_BINS 1793  // distances 0 ... 8 x 224 bytes (BRISK's own descriptors: 0 ... 512; the prefix below only walks the first 513 bins then)
__global__ void __launch_bounds__(64) k_match_radius(const uint16_t* __restrict__ dist, long dist_pitch, int q0, int nt,
                                                     const int* __restrict__ img_start, int nimg,
                                                     const int* __restrict__ masked, float max_distance, int cap,
                                                     BriskDMatch* __restrict__ out, int* __restrict__ out_count, int nbins) {
  __shared__ int bins[MR_BINS + 1];
  const int q = blockIdx.x, lane = threadIdx.x;
  const uint16_t* row = dist + (long)q * dist_pitch;
  BriskDMatch* orow = out + (long)(q0 + q) * cap;
  if (masked && masked[q]) {
    if (lane == 0) out_count[q0 + q] = 0;
    return;
  }
  for (int b = lane; b <= nbins; b += 64) bins[b] = 0;
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  for (int t = lane; t < nt; t += 64) {
    const unsigned d = row[t];
    if (d != 0xFFFFu && (float)d < max_distance) atomicAdd(&bins[d], 1);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {  // exclusive prefix over the nbins (513 for a 64-byte descriptor) distance values
    int acc = 0;
    for (int b = 0; b < nbins; ++b) { const int c = bins[b]; bins[b] = acc; acc += c; }
    bins[nbins] = acc;
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  const int total = bins[nbins];
  // stable placement: chunks of 64 train entries in order; inside a chunk equal distances keep lane order
  for (int t0 = 0; t0 < nt; t0 += 64) {
    const int t = t0 + lane;
    unsigned d = 0xFFFFu;
    if (t < nt) d = row[t];
    const bool hit = d != 0xFFFFu && (float)d < max_distance;
    unsigned long long todo = __ballot(hit);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const unsigned dsel = __shfl(d, leader, 64);
      const unsigned long long same = __ballot(hit && d == dsel);
      const int base = bins[dsel];
      if (hit && d == dsel) {
        const int pos = base + __popcll(same & ((1ull << lane) - 1ull));
        if (pos < cap) {
          const int img = mt_image_of(img_start, nimg, t);
          BriskDMatch m;
          m.queryIdx = q0 + q; m.trainIdx = t - (img_start ? img_start[img] : 0); m.imgIdx = img; m.distance = (float)d;
          orow[pos] = m;
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (lane == leader) bins[dsel] = base + __popcll(same);
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
      todo &= ~same;
    }
  }
  if (lane == 0) out_count[q0 + q] = total;
}

// ------------------------------------------------------------------------------------------------
// Fused k-NN for k <= 2, one train set, no masks (the frame-to-frame / frame-to-map case): no distance matrix.
// Workgroup = 64 queries (one per lane, descriptor in registers) x MF_WAVES waves; wave w scans the w-th slice of
// the train set, whose descriptors are wave-uniform (scalar loads, XOR against SGPRs); per pair 2 x W32 VALU
// (v_xor + v_bcnt with accumulate) and a 3-instruction top-2 update on packed keys (distance << 22 | train
// index).  The MF_WAVES partial top-2 lists of a query are merged through LDS.
// ------------------------------------------------------------------------------------------------
#define MF_WAVES 16
#define MF_IDX_BITS 22
template <int W32>
__global__ void __launch_bounds__(MF_WAVES * 64) k_match_knn_fused(const uint8_t* __restrict__ query, int q_pitch, int nq,
                                                                    const uint8_t* __restrict__ train, int t_pitch, int nt,
                                                                    int k, BriskDMatch* __restrict__ out,
                                                                    int* __restrict__ out_count) {
  __shared__ unsigned part[MF_WAVES][2][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = blockIdx.x * 64 + lane;
  unsigned qv[W32];
  {
    const uint8_t* p = query + (long)min(q, nq - 1) * q_pitch;
#pragma unroll
    for (int w = 0; w < W32; ++w)  // (byte loads: no alignment assumption on caller rows)
      qv[w] = (unsigned)p[4 * w] | ((unsigned)p[4 * w + 1] << 8) | ((unsigned)p[4 * w + 2] << 16) | ((unsigned)p[4 * w + 3] << 24);
  }
  const int per = (nt + MF_WAVES - 1) / MF_WAVES;
  const int t0 = __builtin_amdgcn_readfirstlane(wave * per), t1 = min(nt, t0 + per);
  unsigned b1 = 0xFFFFFFFFu, b2 = 0xFFFFFFFFu;
  const bool aligned = (((uintptr_t)train | (unsigned)t_pitch) & 3) == 0;
  if (aligned) {
    for (int t = t0; t < t1; ++t) {
      const unsigned* tp = reinterpret_cast<const unsigned*>(train + (long)t * t_pitch);  // wave-uniform address
      unsigned d = 0;
#pragma unroll
      for (int w = 0; w < W32; ++w) d += __popc(qv[w] ^ tp[w]);
      const unsigned key = (d << MF_IDX_BITS) | (unsigned)t;
      b2 = min(b2, max(b1, key));
      b1 = min(b1, key);
    }
  } else {
    for (int t = t0; t < t1; ++t) {
      const uint8_t* tp = train + (long)t * t_pitch;
      unsigned d = 0;
#pragma unroll
      for (int w = 0; w < W32; ++w) {
        const unsigned tv = (unsigned)tp[4 * w] | ((unsigned)tp[4 * w + 1] << 8) | ((unsigned)tp[4 * w + 2] << 16) | ((unsigned)tp[4 * w + 3] << 24);
        d += __popc(qv[w] ^ tv);
      }
      const unsigned key = (d << MF_IDX_BITS) | (unsigned)t;
      b2 = min(b2, max(b1, key));
      b1 = min(b1, key);
    }
  }
  part[wave][0][lane] = b1;
  part[wave][1][lane] = b2;
  __syncthreads();
  if (wave == 0 && q < nq) {
    unsigned m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu;
#pragma unroll
    for (int w = 0; w < MF_WAVES; ++w)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned key = part[w][i][lane];
        m2 = min(m2, max(m1, key));
        m1 = min(m1, key);
      }
    BriskDMatch* orow = out + (long)q * k;
    BriskDMatch m;
    m.queryIdx = q; m.imgIdx = 0;
    m.trainIdx = (int)(m1 & ((1u << MF_IDX_BITS) - 1)); m.distance = (float)(m1 >> MF_IDX_BITS);
    orow[0] = m;
    if (k > 1) {
      m.trainIdx = (int)(m2 & ((1u << MF_IDX_BITS) - 1)); m.distance = (float)(m2 >> MF_IDX_BITS);
      orow[1] = m;
    }
    out_count[q] = k;
  }
}

// returns false when the case is not covered (the caller uses the distance-matrix path)
bool brisk_launch_match_knn_fused(const uint8_t* query, int q_pitch, int nq, const uint8_t* train, int t_pitch, int nt,
                                  int words32, int k, BriskDMatch* out, int* out_count, hipStream_t s) {
  if (k < 1 || k > 2 || nt < k || nt >= (1 << MF_IDX_BITS) || nq <= 0) return false;
  const dim3 grid((nq + 63) / 64), block(MF_WAVES * 64);
  switch (words32) {
    case 4: hipLaunchKernelGGL(k_match_knn_fused<4>, grid, block, 0, s, query, q_pitch, nq, train, t_pitch, nt, k, out, out_count); break;
    case 8: hipLaunchKernelGGL(k_match_knn_fused<8>, grid, block, 0, s, query, q_pitch, nq, train, t_pitch, nt, k, out, out_count); break;
    case 12: hipLaunchKernelGGL(k_match_knn_fused<12>, grid, block, 0, s, query, q_pitch, nq, train, t_pitch, nt, k, out, out_count); break;
    case 16: hipLaunchKernelGGL(k_match_knn_fused<16>, grid, block, 0, s, query, q_pitch, nq, train, t_pitch, nt, k, out, out_count); break;
    default: return false;
  }
  return true;
}

void brisk_launch_match_dist(const uint8_t* query, int q_pitch, int q0, int nqb, const uint8_t* train, int t_pitch, int nt,
                             int words, const uint8_t* mask, long mask_pitch, uint16_t* dist, long dist_pitch,
                             hipStream_t s) {
  if (nt <= 0 || nqb <= 0) return;
  const dim3 grid((nt + MT_THREADS - 1) / MT_THREADS, (nqb + MT_QTILE - 1) / MT_QTILE);
  if (words <= 8)
    hipLaunchKernelGGL(k_match_dist<8>, grid, dim3(MT_THREADS), 0, s, query, q_pitch, q0, nqb, train, t_pitch, nt, words, mask,
                       mask_pitch, dist, dist_pitch);
  else
    hipLaunchKernelGGL(k_match_dist<MT_MAXWORDS_LONG>, grid, dim3(MT_THREADS), 0, s, query, q_pitch, q0, nqb, train, t_pitch, nt,
                       words, mask, mask_pitch, dist, dist_pitch);
}
void brisk_launch_match_masked_out(const uint8_t* mask, long mask_pitch, int q0, int nqb, const int* img_start,
                                   const int* has_mask, int nimg, int* masked, hipStream_t s) {
  if (nqb <= 0) return;
  hipLaunchKernelGGL(k_match_masked_out, dim3(nqb), dim3(64), 0, s, mask, mask_pitch, q0, img_start, has_mask, nimg, masked);
}
void brisk_launch_match_knn(const uint16_t* dist, long dist_pitch, int q0, int nqb, int nt, const int* img_start, int nimg,
                            const int* masked, int k, BriskDMatch* out, int* out_count, hipStream_t s) {
  if (nqb <= 0) return;
  hipLaunchKernelGGL(k_match_knn, dim3(nqb), dim3(64), 0, s, dist, dist_pitch, q0, nt, img_start, nimg, masked, k, out,
                     out_count);
}
void brisk_launch_match_radius(const uint16_t* dist, long dist_pitch, int q0, int nqb, int nt, const int* img_start,
                               int nimg, const int* masked, float max_distance, int cap, BriskDMatch* out, int* out_count,
                               int dim_bytes, hipStream_t s) {
  if (nqb <= 0) return;
  hipLaunchKernelGGL(k_match_radius, dim3(nqb), dim3(64), 0, s, dist, dist_pitch, q0, nt, img_start, nimg, masked,
                     max_distance, cap, out, out_count, min(dim_bytes * 8 + 1, MR_BINS));
}
