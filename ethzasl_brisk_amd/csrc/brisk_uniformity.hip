// brisk_uniformity.hip - keypoint uniformity enforcement as an optional post-filter of the detector.
//
// EnforceKeyPointUniformity (brisk/include/brisk/internal/uniformity-enforcement-inl.h:44-194) with the occupancy mask
// of ScaleSpaceLayer (brisk/include/brisk/internal/scale-space-layer-inl.h:88-97).  In the reference the filter is
// only wired into the Harris ScaleSpaceFeatureDetector (scale-space-layer-inl.h:372-375); the engine offers it behind
// BriskFeatureDetector for BASELINE config 4 ("uniformity-enforced").  The algorithm is a greedy pass over the
// keypoints in descending score order against an occupancy image.  What a point reads from that image is
//   min(255, sum of the mask values that the ACCEPTED earlier points within 15 cells added at its cell)
// (saturating adds of non-negative values commute), so no image is needed: k_uniformity decides the points in score
// order with one wave per point, a wave waits (in LDS) only for the decisions of the earlier points within 15 cells of
// its own and sums their contributions directly.  Frames with more points than the on-chip arrays hold take the
// literal walk over an occupancy image in global memory (k_uniformity_seq: one workgroup per frame, the 31 x 31 update
// of an accepted point spread over the workgroup).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "brisk_common.h"
#include "brisk_kernels.h"

#define UF_THREADS 1024
#define UF_LDS_POINTS 8192

// L1-bypassing byte read (the workgroup's own earlier stores are at L2)
__device__ __forceinline__ unsigned uf_load_fresh(const uint8_t* p) {
  const uintptr_t a = (uintptr_t)p;
  const unsigned v = __hip_atomic_load(reinterpret_cast<const unsigned*>(a & ~(uintptr_t)3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (v >> (8 * (a & 3))) & 0xFFu;
}

// kp: [frame][kp_cap] keypoints of the detector (rewritten: kept keypoints in descending score order);
// order: [frame][kp_cap] scratch; tmp: [frame][kp_cap] scratch; occ: [frame][oh * ow] zeroed occupancy images
__global__ void __launch_bounds__(UF_THREADS) k_uniformity_seq(BriskKeyPoint* __restrict__ kp, BriskFrameCounters* __restrict__ counters,
                                                           int* __restrict__ order, BriskKeyPoint* __restrict__ tmp,
                                                           uint8_t* __restrict__ occ, long occ_frame, int ow, int kp_cap,
                                                           float scaling, int max_keypoints) {
  __shared__ float tile[UF_THREADS];
  __shared__ int accept_s, kept_s;
  __shared__ float nsc_s;
  __shared__ int cy_s, cx_s;
  const int frame = blockIdx.x, tid = threadIdx.x;
  const int n = min(counters[frame].nkp, kp_cap);
  BriskKeyPoint* K = kp + (long)frame * kp_cap;
  BriskKeyPoint* T = tmp + (long)frame * kp_cap;
  int* ord = order + (long)frame * kp_cap;
  uint8_t* O = occ + (long)frame * occ_frame;
  if (n <= UF_LDS_POINTS) return;  // k_uniformity has done this frame
  for (long i = tid; i < occ_frame / 16; i += UF_THREADS) reinterpret_cast<uint4*>(O)[i] = make_uint4(0, 0, 0, 0);
  __threadfence();
  __syncthreads();
  // rank by (score descending, input index ascending)
  for (int j0 = 0; j0 < n; j0 += UF_THREADS) {
    const int j = j0 + tid;
    const float mine = (j < n) ? K[j].response : 0.f;
    int rank = 0;
    for (int t0 = 0; t0 < n; t0 += UF_THREADS) {
      __syncthreads();
      tile[tid] = (t0 + tid < n) ? K[t0 + tid].response : 0.f;
      __syncthreads();
      const int m = min(UF_THREADS, n - t0);
      if (j < n)
        for (int q = 0; q < m; ++q) {
          const float s = tile[q];
          rank += (s > mine || (s == mine && t0 + q < j)) ? 1 : 0;
        }
    }
    if (j < n) ord[rank] = j;
  }
  __threadfence();
  __syncthreads();
  const float maxScore = K[__hip_atomic_load(&ord[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)].response;
  if (tid == 0) kept_s = 0;
  // occupancy cell and normalised score of the sorted points, computed in parallel and kept on chip: the sequential
  // walk below then waits for nothing but the occupancy image itself
  __shared__ int pcell[UF_LDS_POINTS];     // cy << 16 | cx
  __shared__ float pnsc[UF_LDS_POINTS];
  const int nl = min(n, UF_LDS_POINTS);
  for (int r = tid; r < nl; r += UF_THREADS) {
    const BriskKeyPoint p = K[__hip_atomic_load(&ord[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)];
    pcell[r] = ((int)(p.y * scaling + 16) << 16) | (int)(p.x * scaling + 16);
    pnsc[r] = sqrtf(sqrtf(p.response / maxScore)) * 255.0f;
  }
  __syncthreads();
  // mask value of this thread's cell of the 31 x 31 neighbourhood (scale-space-layer-inl.h:89-97)
  const int my = tid / 31, mx = tid % 31;
  float lut = 0.f;
  if (tid < 31 * 31) {
    const double v = 1 - (double)((15 - mx) * (15 - mx) + (15 - my) * (15 - my)) / (double)(15 * 15);
    lut = (float)(v > 0.0 ? v : 0.0);
  }
  for (int r = 0; r < n; ++r) {
    if (tid == 0) {
      int cy, cx;
      float nsc1;
      if (r < nl) {
        cy = pcell[r] >> 16; cx = pcell[r] & 0xFFFF; nsc1 = pnsc[r];
      } else {
        const BriskKeyPoint p = K[__hip_atomic_load(&ord[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)];
        cy = (int)(p.y * scaling + 16); cx = (int)(p.x * scaling + 16);
        nsc1 = sqrtf(sqrtf(p.response / maxScore)) * 255.0f;
      }
      const double s0 = (double)uf_load_fresh(&O[(long)cy * ow + cx]);
      const int acc = !(nsc1 < s0);
      accept_s = acc;
      if (acc) {
        nsc_s = 0.99f * nsc1;
        cy_s = cy; cx_s = cx;
        ord[kept_s] = ord[r];   // (kept_s <= r: the accepted prefix of the order array is compacted in place)
        kept_s = kept_s + 1;
      }
    }
    __syncthreads();
    const bool acc = accept_s != 0;
    if (acc && tid < 31 * 31) {
      uint8_t* c = &O[(long)(cy_s + my - 15) * ow + (cx_s + mx - 15)];
      const int add = (int)(uint8_t)(int)ceilf(lut * nsc_s);
      const int s = (int)uf_load_fresh(c) + add;
      *c = (uint8_t)(s > 255 ? 255 : s);
    }
    if (acc) __threadfence();
    const bool done = acc && kept_s >= max_keypoints;
    __syncthreads();
    if (done) break;
  }
  __syncthreads();
  const int kept = kept_s;
  __threadfence();
  __syncthreads();
  for (int i = tid; i < kept; i += UF_THREADS) T[i] = K[__hip_atomic_load(&ord[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)];
  __threadfence();
  __syncthreads();
  for (int i = tid; i < kept; i += UF_THREADS) K[i] = T[i];
  if (tid == 0) counters[frame].nkp = kept;
}

// mask value a point adds at offset (dx, dy) = (cell - its own cell) (scale-space-layer-inl.h:89-97,
// uniformity-enforcement-inl.h:150-170): ceil(lut * 0.99 * nsc) as u8
__device__ __forceinline__ int uf_contribution(int dx, int dy, float nsc) {
  const double v = 1 - (double)(dx * dx + dy * dy) / (double)(15 * 15);
  const float lut = (float)(v > 0.0 ? v : 0.0);
  const float nsc99 = 0.99f * nsc;
  return (int)(uint8_t)(int)ceilf(lut * nsc99);
}

__global__ void __launch_bounds__(UF_THREADS) k_uniformity(BriskKeyPoint* __restrict__ kp, BriskFrameCounters* __restrict__ counters,
                                                           int* __restrict__ order, BriskKeyPoint* __restrict__ tmp, int kp_cap,
                                                           float scaling, int max_keypoints) {
  __shared__ float tile[UF_THREADS];
  __shared__ int pcell[UF_LDS_POINTS];     // cy << 16 | cx of the point with score rank r
  __shared__ float pnsc[UF_LDS_POINTS];    // its normalised score
  __shared__ int dec[UF_LDS_POINTS];       // 0 pending, 1 accepted, 2 rejected
  __shared__ int wsum[UF_THREADS / 64];
  const int frame = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = min(counters[frame].nkp, kp_cap);
  BriskKeyPoint* K = kp + (long)frame * kp_cap;
  BriskKeyPoint* T = tmp + (long)frame * kp_cap;
  int* ord = order + (long)frame * kp_cap;
  if (n == 0 || n > UF_LDS_POINTS) return;  // (the larger frames: k_uniformity_seq)
  // rank by (score descending, input index ascending)
  for (int j0 = 0; j0 < n; j0 += UF_THREADS) {
    const int j = j0 + tid;
    const float mine = (j < n) ? K[j].response : 0.f;
    int rank = 0;
    for (int t0 = 0; t0 < n; t0 += UF_THREADS) {
      __syncthreads();
      tile[tid] = (t0 + tid < n) ? K[t0 + tid].response : 0.f;
      __syncthreads();
      const int m = min(UF_THREADS, n - t0);
      if (j < n)
        for (int q = 0; q < m; ++q) {
          const float s = tile[q];
          rank += (s > mine || (s == mine && t0 + q < j)) ? 1 : 0;
        }
    }
    if (j < n) ord[rank] = j;
  }
  __threadfence();
  __syncthreads();
  const float maxScore = K[__hip_atomic_load(&ord[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)].response;
  for (int r = tid; r < n; r += UF_THREADS) {
    const BriskKeyPoint p = K[__hip_atomic_load(&ord[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)];
    pcell[r] = ((int)(p.y * scaling + 16) << 16) | (int)(p.x * scaling + 16);
    pnsc[r] = sqrtf(sqrtf(p.response / maxScore)) * 255.0f;
    dec[r] = 0;
  }
  __syncthreads();
  // decisions in score order, one wave per point: the wave's lanes scan the earlier points, a lane that finds one
  // within reach waits for its decision (the earliest undecided point never waits) and adds what it contributed
  for (int r = wave; r < n; r += UF_THREADS / 64) {
    const int cyr = pcell[r] >> 16, cxr = pcell[r] & 0xFFFF;
    int sum = 0;
    for (int j0 = 0; j0 < r; j0 += 64) {
      const int j = j0 + lane;
      if (j < r) {
        const int c = pcell[j];
        const int dy = cyr - (c >> 16), dx = cxr - (c & 0xFFFF);
        if (dx >= -15 && dx <= 15 && dy >= -15 && dy <= 15) {
          int d = 0;
          for (int spin = 0; spin < (1 << 24); ++spin) {
            d = __hip_atomic_load(&dec[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (d) break;
            __builtin_amdgcn_s_sleep(1);
          }
          if (d == 0) atomicOr(&counters[frame].overflow, 8);  // (never observed) reported as an internal error
          if (d == 1) sum += uf_contribution(dx, dy, pnsc[j]);
        }
      }
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const double s0 = (double)(sum > 255 ? 255 : sum);
    const bool acc = !(pnsc[r] < s0);
    if (lane == 0) __hip_atomic_store(&dec[r], acc ? 1 : 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  // the accepted points in score order, at most max_keypoints of them
  int run = 0;
  __shared__ int kept_s;
  for (int r0 = 0; r0 < n; r0 += UF_THREADS) {
    const int r = r0 + tid;
    const bool a = (r < n) && dec[r] == 1;
    const unsigned long long bal = __ballot(a);
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int before = run, total = 0;
    for (int q = 0; q < UF_THREADS / 64; ++q) {
      const int t = wsum[q];
      before += (q < wave) ? t : 0;
      total += t;
    }
    const int pos = before + __popcll(bal & ((1ull << lane) - 1ull));
    if (a && pos < max_keypoints) T[pos] = K[__hip_atomic_load(&ord[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)];
    run += total;
    __syncthreads();
  }
  if (tid == 0) kept_s = min(run, max_keypoints);
  __threadfence();
  __syncthreads();
  const int kept = kept_s;
  for (int i = tid; i < kept; i += UF_THREADS) K[i] = T[i];
  if (tid == 0) counters[frame].nkp = kept;
}

void brisk_launch_uniformity(BriskKeyPoint* kp, BriskFrameCounters* counters, int* order, BriskKeyPoint* tmp, uint8_t* occ,
                             long occ_frame, int ow, int kp_cap, float scaling, int max_keypoints, int nframes, hipStream_t s) {
  if (nframes <= 0) return;
  hipLaunchKernelGGL(k_uniformity, dim3(nframes), dim3(UF_THREADS), 0, s, kp, counters, order, tmp, kp_cap, scaling,
                     max_keypoints);
  // frames with more points than k_uniformity's on-chip arrays hold (it left them untouched; clears its own image)
  hipLaunchKernelGGL(k_uniformity_seq, dim3(nframes), dim3(UF_THREADS), 0, s, kp, counters, order, tmp, occ, occ_frame, ow,
                     kp_cap, scaling, max_keypoints);
}

// ------------------------------------------------------------------------------------------------
// KeyPointBucketing (brisk/include/brisk/internal/key-point-bucketing-inl.h:40-112, key-point-bucketing.h:50-67): the
// reference's other way to limit and spread the keypoints of a layer (scale-space-layer-inl.h:372-378: used when
// uniformity enforcement is off).  Points sorted by score, descending; with one bucket in either direction the best
// `max_keypoints` are kept (:87-98), otherwise the image is divided into nbu x nbv buckets of
// (1 + (cols - 1) / nbu) x (1 + (rows - 1) / nbv) pixels and a point is kept while its bucket holds fewer than
// max_keypoints / (nbu * nbv) points (:48-63).  A point's fate depends only on how many better points share its bucket,
// so there is no sequential pass: rank by counting, rank inside the bucket by counting, stable compaction.
// Output order = descending score (equal scores keep the detector's (layer, y, x) order; the reference's std::sort leaves
// it open).  Like the uniformity filter this is offered as a post-filter of BriskFeatureDetector: parity unpinned.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(UF_THREADS) k_bucketing(BriskKeyPoint* __restrict__ kp, BriskFrameCounters* __restrict__ counters,
                                                          int* __restrict__ order, BriskKeyPoint* __restrict__ tmp, int kp_cap,
                                                          int rows, int cols, int nbu, int nbv, int max_keypoints) {
  __shared__ float tile[UF_THREADS];
  __shared__ int btile[UF_THREADS];
  __shared__ int wsum[UF_THREADS / 64];
  __shared__ int kept_s;
  const int frame = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = min(counters[frame].nkp, kp_cap);
  BriskKeyPoint* K = kp + (long)frame * kp_cap;
  BriskKeyPoint* T = tmp + (long)frame * kp_cap;
  int* ord = order + (long)frame * kp_cap;  // ord[i]: bit 31 = kept, low bits = output position of keypoint i among the kept
  if (n == 0) return;
  const bool single = (nbu == 1 || nbv == 1);
  const int cap = single ? max_keypoints : max_keypoints / (nbu * nbv);
  const unsigned step_u = 1u + (unsigned)(cols - 1) / (unsigned)nbu, step_v = 1u + (unsigned)(rows - 1) / (unsigned)nbv;
  // pass 1: is keypoint j kept?  (rank among the points of its bucket - of all points with one bucket - below the cap)
  for (int j0 = 0; j0 < n; j0 += UF_THREADS) {
    const int j = j0 + tid;
    float mine = 0.f;
    int mybucket = -1;
    if (j < n) {
      mine = K[j].response;
      mybucket = single ? 0 : (int)((unsigned)(int)K[j].x / step_u) * nbv + (int)((unsigned)(int)K[j].y / step_v);
    }
    int better = 0;
    for (int t0 = 0; t0 < n; t0 += UF_THREADS) {
      __syncthreads();
      if (t0 + tid < n) {
        tile[tid] = K[t0 + tid].response;
        btile[tid] = single ? 0 : (int)((unsigned)(int)K[t0 + tid].x / step_u) * nbv + (int)((unsigned)(int)K[t0 + tid].y / step_v);
      }
      __syncthreads();
      const int m = min(UF_THREADS, n - t0);
      if (j < n)
        for (int q = 0; q < m; ++q) {
          const float s = tile[q];
          better += (btile[q] == mybucket && (s > mine || (s == mine && t0 + q < j))) ? 1 : 0;
        }
    }
    if (j < n) ord[j] = (better < cap) ? (int)0x80000000 : 0;
  }
  __threadfence();
  __syncthreads();
  // pass 2: output position of a kept keypoint = kept keypoints with a better score
  for (int j0 = 0; j0 < n; j0 += UF_THREADS) {
    const int j = j0 + tid;
    const float mine = (j < n) ? K[j].response : 0.f;
    const bool kept = (j < n) && (__hip_atomic_load(&ord[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0);
    int pos = 0;
    for (int t0 = 0; t0 < n; t0 += UF_THREADS) {
      __syncthreads();
      if (t0 + tid < n) {
        tile[tid] = K[t0 + tid].response;
        btile[tid] = __hip_atomic_load(&ord[t0 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0 ? 1 : 0;
      }
      __syncthreads();
      const int m = min(UF_THREADS, n - t0);
      if (kept)
        for (int q = 0; q < m; ++q) {
          const float s = tile[q];
          pos += (btile[q] && (s > mine || (s == mine && t0 + q < j))) ? 1 : 0;
        }
    }
    if (kept) T[pos] = K[j];
    const unsigned long long bal = __ballot(kept);
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    if (tid == 0) {
      int t = (j0 == 0) ? 0 : kept_s;
      for (int q = 0; q < UF_THREADS / 64; ++q) t += wsum[q];
      kept_s = t;
    }
    __syncthreads();
  }
  __threadfence();
  __syncthreads();
  const int nkept = kept_s;
  for (int i = tid; i < nkept; i += UF_THREADS) K[i] = T[i];
  if (tid == 0) counters[frame].nkp = nkept;
}

void brisk_launch_bucketing(BriskKeyPoint* kp, BriskFrameCounters* counters, int* order, BriskKeyPoint* tmp, int kp_cap, int rows,
                            int cols, int nbu, int nbv, int max_keypoints, int nframes, hipStream_t s) {
  if (nframes <= 0) return;
  hipLaunchKernelGGL(k_bucketing, dim3(nframes), dim3(UF_THREADS), 0, s, kp, counters, order, tmp, kp_cap, rows, cols, nbu, nbv,
                     max_keypoints);
}
