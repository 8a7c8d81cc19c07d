// brisk_uniformity.hip - keypoint uniformity enforcement as an optional post-filter of the detector.
//
// EnforceKeyPointUniformity (brisk/include/brisk/internal/uniformity-enforcement-inl.h:44-194) with the occupancy mask
// of ScaleSpaceLayer (brisk/include/brisk/internal/scale-space-layer-inl.h:88-97).  In the reference the filter is
// only wired into the Harris ScaleSpaceFeatureDetector (scale-space-layer-inl.h:372-375); the engine offers it behind
// BriskFeatureDetector for BASELINE config 4 ("uniformity-enforced").  The algorithm is a greedy pass over the
// keypoints in descending score order against an occupancy image, i.e. sequential by construction: one workgroup per
// frame walks the sorted list, the 31 x 31 occupancy update of an accepted point is spread over the workgroup.
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
__global__ void __launch_bounds__(UF_THREADS) k_uniformity(BriskKeyPoint* __restrict__ kp, BriskFrameCounters* __restrict__ counters,
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
  if (n == 0) return;
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

void brisk_launch_uniformity(BriskKeyPoint* kp, BriskFrameCounters* counters, int* order, BriskKeyPoint* tmp, uint8_t* occ,
                             long occ_frame, int ow, int kp_cap, float scaling, int max_keypoints, int nframes, hipStream_t s) {
  if (nframes <= 0) return;
  (void)hipMemsetAsync(occ, 0, (size_t)occ_frame * nframes, s);
  hipLaunchKernelGGL(k_uniformity, dim3(nframes), dim3(UF_THREADS), 0, s, kp, counters, order, tmp, occ, occ_frame, ow, kp_cap,
                     scaling, max_keypoints);
}
