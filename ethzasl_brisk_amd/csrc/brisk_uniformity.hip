// brisk_uniformity.hip - keypoint uniformity enforcement as an optional post-filter of the detector.
//
// EnforceKeyPointUniformity (brisk/include/brisk/internal/uniformity-enforcement-inl.h:44-194) with the occupancy mask
// of ScaleSpaceLayer (brisk/include/brisk/internal/scale-space-layer-inl.h:88-97).  In the reference the filter is
// only wired into the Harris ScaleSpaceFeatureDetector (scale-space-layer-inl.h:372-375); the engine offers it behind
// BriskFeatureDetector for BASELINE config 4 ("uniformity-enforced").  The algorithm is a greedy pass over the
// keypoints in descending score order against an occupancy image.  What a point reads from that image is
//   min(255, sum of the mask values that the ACCEPTED earlier points within 15 cells added at its cell)
// (saturating adds of non-negative values commute), so no image is needed: k_uf_rank puts the points into score order,
// k_uf_decide decides all points of a frame at once - a point waits (in LDS) only for the decisions of the better points
// within 15 cells of its own, found through a hash of 16 x 16-cell bins, and sums their contributions directly.
// Frames with more points than the on-chip arrays hold take the
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

// ---- frames with up to UF_LDS_POINTS points: two kernels --------------------------------------------------------
// k_uf_rank    (several workgroups per frame) score rank of every point by counting - (score descending, input index
//              ascending), the walk order of the reference - and, at its rank, the point's occupancy cell and normalised
//              score.  n^2 comparisons, but spread over n / 256 workgroups and two instructions each: 45 us for the 4 745
//              points of a 4K frame, where ONE workgroup ranking and then scanning all earlier points of every point
//              took 3 ms.
// k_uf_decide  (one workgroup per frame) the points of a frame hashed by 16 x 16-cell bins; a point looks at the better
//              points of the 3 x 3 bins around its own (what lies within 15 cells is in there), waits until those within
//              reach are decided and sums what the accepted ones contributed - O(n k) for k neighbours, all points at
//              once.  One lane per point, ranks dealt round-robin: a lane only ever waits for smaller ranks, and the
//              decision is published INSIDE the polling loop (lanes of one wave wait for each other), so the smallest
//              undecided rank always gets decided.
#define UR_THREADS 256
#define UF_TILE 1024
#define UF_HASH 8192
#define UF_NB 6   // neighbours a point keeps in registers (k_uf_decide)
struct UfSorted { int cx, cy; float nsc; };  // (scratch in the first 12 bytes of tmp[rank])

// UR_POINTS points per workgroup, one per lane; the UR_THREADS / 64 waves split the comparison range between them (wave w
// takes tiles w, w + 4, ... of 1024 scores, staged in its own LDS region: no workgroup barrier in the loop), the partial
// ranks are added up in LDS.  A single 4K frame (4 745 points): 75 workgroups, ~2 tiles per wave.
#define UR_POINTS 64
__global__ void __launch_bounds__(UR_THREADS) k_uf_rank(const BriskKeyPoint* __restrict__ kp, const BriskFrameCounters* __restrict__ counters,
                                                        int* __restrict__ order, BriskKeyPoint* __restrict__ tmp, int kp_cap, float scaling) {
  __shared__ __attribute__((aligned(16))) float tile[UR_THREADS / 64][UF_TILE];
  __shared__ int srank[UR_POINTS];
  __shared__ float wmax[UR_THREADS / 64];
  const int frame = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = min(counters[frame].nkp, kp_cap);
  const int j0 = blockIdx.x * UR_POINTS;
  if (n == 0 || n > UF_LDS_POINTS || j0 >= n) return;
  const BriskKeyPoint* K = kp + (long)frame * kp_cap;
  const int j = j0 + lane;
  BriskKeyPoint me;
  me.x = me.y = 0.f; me.response = 0.f;
  if (j < n) me = K[j];
  const float mine = me.response;
  const int own = j0 / UF_TILE * UF_TILE;  // the tile that holds this workgroup's own points
  if (tid < UR_POINTS) srank[tid] = 0;
  __syncthreads();
  int rank = 0;
  float mx = 0.f;
  float* tl = tile[wave];
  for (int t0 = wave * UF_TILE; t0 < n; t0 += (UR_THREADS / 64) * UF_TILE) {
#pragma unroll
    for (int k = 0; k < UF_TILE / 64; ++k) {
      const int q = t0 + k * 64 + lane;
      const float v = q < n ? K[q].response : -3.0e38f;  // (never counted)
      tl[k * 64 + lane] = v;
      mx = fmaxf(mx, v);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // equal scores: the smaller input index first - all of an earlier tile's equals count, none of a later tile's
    if (t0 < own) {
      for (int q = 0; q < UF_TILE; q += 4) {
        const float4 s = *reinterpret_cast<const float4*>(&tl[q]);
        rank += (s.x >= mine ? 1 : 0) + (s.y >= mine ? 1 : 0) + (s.z >= mine ? 1 : 0) + (s.w >= mine ? 1 : 0);
      }
    } else if (t0 > own) {
      for (int q = 0; q < UF_TILE; q += 4) {
        const float4 s = *reinterpret_cast<const float4*>(&tl[q]);
        rank += (s.x > mine ? 1 : 0) + (s.y > mine ? 1 : 0) + (s.z > mine ? 1 : 0) + (s.w > mine ? 1 : 0);
      }
    } else {
      for (int q = 0; q < UF_TILE; ++q) {
        const float sq = tl[q];
        rank += (sq > mine || (sq == mine && t0 + q < j)) ? 1 : 0;
      }
    }
    __builtin_amdgcn_wave_barrier();  // (the tile is overwritten by the wave's next one)
  }
  if (rank) atomicAdd(&srank[lane], rank);
  // the best score of the frame (the workgroup's waves have seen all of them between them)
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if (lane == 0) wmax[wave] = mx;
  __syncthreads();
  if (wave != 0 || j >= n) return;
  float maxScore = wmax[0];
#pragma unroll
  for (int k = 1; k < UR_THREADS / 64; ++k) maxScore = fmaxf(maxScore, wmax[k]);
  rank = srank[lane];
  order[(long)frame * kp_cap + rank] = j;
  UfSorted rec;
  rec.cy = (int)(me.y * scaling + 16);
  rec.cx = (int)(me.x * scaling + 16);
  rec.nsc = sqrtf(sqrtf(mine / maxScore)) * 255.0f;
  *reinterpret_cast<UfSorted*>(tmp + (long)frame * kp_cap + rank) = rec;
}

__device__ __forceinline__ unsigned uf_hash(int bx, int by) {
  return (((unsigned)by * 73856093u) ^ ((unsigned)bx * 19349663u)) & (UF_HASH - 1);
}

__global__ void __launch_bounds__(UF_THREADS) k_uf_decide(BriskKeyPoint* __restrict__ kp, BriskFrameCounters* __restrict__ counters,
                                                          int* __restrict__ order, BriskKeyPoint* __restrict__ tmp, int kp_cap,
                                                          int max_keypoints) {
  extern __shared__ __attribute__((aligned(16))) unsigned char uf_lds[];
  int2* pcell = reinterpret_cast<int2*>(uf_lds);                                   // [UF_LDS_POINTS] (cx, cy) of rank r
  int* hcur = reinterpret_cast<int*>(pcell + UF_LDS_POINTS);                       // [UF_HASH] bucket counts, then fill cursors
  unsigned short* hstart = reinterpret_cast<unsigned short*>(hcur + UF_HASH);      // [UF_HASH + 2] first list entry of a bucket
  unsigned short* blist = hstart + UF_HASH + 2;                                    // [UF_LDS_POINTS] ranks grouped by bucket
  volatile unsigned char* dec = reinterpret_cast<volatile unsigned char*>(blist + UF_LDS_POINTS);  // [UF_LDS_POINTS] 0 pending, 1 accepted, 2 rejected
  __shared__ int wsum[UF_THREADS / 64];
  __shared__ int kept_s, giveup_s;
  const int frame = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = min(counters[frame].nkp, kp_cap);
  if (n == 0 || n > UF_LDS_POINTS) return;  // (the larger frames: k_uniformity_seq)
  BriskKeyPoint* K = kp + (long)frame * kp_cap;
  BriskKeyPoint* T = tmp + (long)frame * kp_cap;
  const int* ord = order + (long)frame * kp_cap;
#ifdef UF_TIMING
  const long long t_start = (long long)wall_clock64();
#define UF_T(i) if (tid == 0) counters[frame].pad[i] = (int)((long long)wall_clock64() - t_start);
#else
#define UF_T(i)
#endif
  for (int b = tid; b < UF_HASH; b += UF_THREADS) hcur[b] = 0;
  if (tid == 0) giveup_s = 0;
  __syncthreads();
  for (int r = tid; r < n; r += UF_THREADS) {
    const UfSorted rec = *reinterpret_cast<const UfSorted*>(T + r);
    pcell[r] = make_int2(rec.cx, rec.cy);
    dec[r] = 0;
    atomicAdd(&hcur[uf_hash(rec.cx >> 4, rec.cy >> 4)], 1);
  }
  __syncthreads();
  {  // exclusive prefix over the buckets: UF_HASH / UF_THREADS consecutive buckets per thread
    constexpr int PER = UF_HASH / UF_THREADS;
    int loc[PER], sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) { loc[q] = hcur[tid * PER + q]; sum += loc[q]; }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off, 64);
      if (lane >= off) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int run = incl - sum;
#pragma unroll
    for (int q = 0; q < UF_THREADS / 64; ++q) run += (q < wave) ? wsum[q] : 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) { hstart[tid * PER + q] = (unsigned short)run; hcur[tid * PER + q] = run; run += loc[q]; }
    if (tid == UF_THREADS - 1) hstart[UF_HASH] = (unsigned short)run;  // (n <= 8192 < 65536)
  }
  __syncthreads();
  for (int r = tid; r < n; r += UF_THREADS) {
    const int2 c = pcell[r];
    blist[atomicAdd(&hcur[uf_hash(c.x >> 4, c.y >> 4)], 1)] = (unsigned short)r;
  }
  __syncthreads();
  // the bucket cursors are done with: their LDS now holds the normalised scores (a neighbour's contribution is read in
  // the decision loop, where a global load per accepted neighbour cost microseconds)
  float* pnsc = reinterpret_cast<float*>(hcur);
  static_assert(UF_HASH >= UF_LDS_POINTS, "the scores alias the bucket cursors");
  for (int r = tid; r < n; r += UF_THREADS) pnsc[r] = reinterpret_cast<const UfSorted*>(T + r)->nsc;
  __syncthreads();
  UF_T(0)
  // decisions: rank r on thread r % UF_THREADS, in increasing order
  for (int r0 = 0; r0 < n; r0 += UF_THREADS) {
    const int r = r0 + tid;
    bool done = r >= n;
    int2 c = make_int2(0, 0);
    float mynsc = 0.f;
    if (!done) { c = pcell[r]; mynsc = pnsc[r]; }
    const int bx = c.x >> 4, by = c.y >> 4;
    // The better points within reach, found ONCE (the bins do not change): rank and what the point adds if accepted, in
    // registers.  The polling loop below then costs a few LDS reads per pass instead of a walk over nine buckets - the
    // passes are what a chain of dependent decisions multiplies (186 -> 30 us for the 4 012 points of a 4K frame).  A point
    // with more than UF_NB such neighbours (tight clusters) walks the buckets on every pass as before.
    int nq[UF_NB], nc[UF_NB];
#pragma unroll
    for (int i = 0; i < UF_NB; ++i) { nq[i] = 0; nc[i] = 0; }
    int cnt = 0;
    if (!done) {
      for (int k = 0; k < 9; ++k) {
        const int nbx = bx + k % 3 - 1, nby = by + k / 3 - 1;
        const unsigned b = uf_hash(nbx, nby);
        const int e1 = hstart[b + 1];
        for (int e = hstart[b]; e < e1; ++e) {
          const int q = blist[e];
          if (q >= r) continue;  // only better points count
          const int2 cq = pcell[q];
          if ((cq.x >> 4) != nbx || (cq.y >> 4) != nby) continue;  // another bin in the same bucket (it has its own turn)
          const int dx = c.x - cq.x, dy = c.y - cq.y;
          if (dx < -15 || dx > 15 || dy < -15 || dy > 15) continue;
          const int contrib = uf_contribution(dx, dy, pnsc[q]);
#pragma unroll
          for (int i = 0; i < UF_NB; ++i)
            if (cnt == i) { nq[i] = q; nc[i] = contrib; }
          ++cnt;
        }
      }
    }
    const bool listed = cnt <= UF_NB;
    int spins = 0;
    while (__any(!done)) {
      if (!done) {
        bool pending = false;
        int sum = 0;
        if (listed) {
#pragma unroll
          for (int i = 0; i < UF_NB; ++i) {
            const unsigned d = (i < cnt) ? (unsigned)dec[nq[i]] : 2u;
            pending = pending || d == 0;
            sum += (d == 1) ? nc[i] : 0;
          }
        } else {
          for (int k = 0; k < 9 && !pending; ++k) {
            const int nbx = bx + k % 3 - 1, nby = by + k / 3 - 1;
            const unsigned b = uf_hash(nbx, nby);
            const int e1 = hstart[b + 1];
            for (int e = hstart[b]; e < e1; ++e) {
              const int q = blist[e];
              if (q >= r) continue;
              const int2 cq = pcell[q];
              if ((cq.x >> 4) != nbx || (cq.y >> 4) != nby) continue;
              const int dx = c.x - cq.x, dy = c.y - cq.y;
              if (dx < -15 || dx > 15 || dy < -15 || dy > 15) continue;
              const unsigned d = dec[q];
              if (d == 0) { pending = true; break; }
              if (d == 1) sum += uf_contribution(dx, dy, pnsc[q]);
            }
          }
        }
        if (!pending) {
          const double s0 = (double)(sum > 255 ? 255 : sum);
          dec[r] = !(mynsc < s0) ? 1 : 2;
          done = true;
        } else if (++spins > (1 << 22)) {  // (never observed) reported as an internal error instead of a hang
          giveup_s = 1;
          dec[r] = 2;
          done = true;
        }
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  UF_T(1)
  if (tid == 0 && giveup_s) atomicOr(&counters[frame].overflow, 8);
  // the accepted points in score order, at most max_keypoints of them.  T[] held the sorted cells and scores until here.
  int run = 0;
  __syncthreads();
  for (int r0 = 0; r0 < n; r0 += UF_THREADS) {
    const int r = r0 + tid;
    const bool a = (r < n) && dec[r] == 1;
    BriskKeyPoint mine;
    if (a) mine = K[ord[r]];
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
    // (T[pos] with pos <= r: the sorted record at pos is no longer needed - every decision has been taken)
    if (a && pos < max_keypoints) T[pos] = mine;
    run += total;
    __syncthreads();
  }
  if (tid == 0) kept_s = min(run, max_keypoints);
  __threadfence();
  __syncthreads();
  UF_T(2)
  const int kept = kept_s;
  for (int i = tid; i < kept; i += UF_THREADS) K[i] = T[i];
  if (tid == 0) counters[frame].nkp = kept;
  UF_T(3)
}

static size_t uf_decide_lds() {
  return (size_t)UF_LDS_POINTS * 8 + (size_t)UF_HASH * 4 + (size_t)(UF_HASH + 2) * 2 + (size_t)UF_LDS_POINTS * 2 + (size_t)UF_LDS_POINTS + 64;
}

void brisk_launch_uniformity(BriskKeyPoint* kp, BriskFrameCounters* counters, int* order, BriskKeyPoint* tmp, uint8_t* occ,
                             long occ_frame, int ow, int kp_cap, float scaling, int max_keypoints, int nframes, hipStream_t s) {
  if (nframes <= 0) return;
  const int cap = kp_cap < UF_LDS_POINTS ? kp_cap : UF_LDS_POINTS;
  hipLaunchKernelGGL(k_uf_rank, dim3((cap + UR_POINTS - 1) / UR_POINTS, nframes), dim3(UR_THREADS), 0, s, kp, counters, order, tmp,
                     kp_cap, scaling);
  (void)hipFuncSetAttribute((const void*)k_uf_decide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)uf_decide_lds());
  hipLaunchKernelGGL(k_uf_decide, dim3(nframes), dim3(UF_THREADS), uf_decide_lds(), s, kp, counters, order, tmp, kp_cap, max_keypoints);
  // frames with more points than the on-chip arrays hold (left untouched above; clears its own image)
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
// it open) - except for one bucket with no more than max_keypoints points, which the reference leaves untouched.  Like the uniformity filter this is offered as a post-filter of BriskFeatureDetector: parity unpinned.
// ------------------------------------------------------------------------------------------------
// Three kernels (round 4; round 3 did both counting passes in ONE workgroup per frame: n^2 / 1024 steps per thread, twice -
// about 2 ms for the 4 600 keypoints of a 4K frame).  k_bk_count<0>: 64 points per workgroup, the comparison range split
// over its four waves (as in k_uf_rank): better points in the same bucket -> kept flag.  k_bk_count<1>: better KEPT points
// -> output position, the keypoint goes to tmp[position].  k_bk_finish: one workgroup per frame copies the kept
// keypoints back and sets the count.
#define BK_POINTS 64
struct BkBuckets { int nbu, nbv, cap, single; unsigned step_u, step_v; };
__device__ __forceinline__ int bk_bucket(const BkBuckets& b, float x, float y) {
  return b.single ? 0 : (int)((unsigned)(int)x / b.step_u) * b.nbv + (int)((unsigned)(int)y / b.step_v);
}
// bucket of every point, once (the divisions by the run-time bucket steps are ~40 instructions each): order[j] = bucket
__global__ void __launch_bounds__(256) k_bk_keys(const BriskKeyPoint* __restrict__ kp, const BriskFrameCounters* __restrict__ counters,
                                                 int* __restrict__ order, int kp_cap, BkBuckets B, int max_keypoints) {
  const int frame = blockIdx.y;
  const int n = min(counters[frame].nkp, kp_cap);
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n || (B.single && n <= max_keypoints)) return;
  const BriskKeyPoint p = kp[(long)frame * kp_cap + j];
  order[(long)frame * kp_cap + j] = bk_bucket(B, p.x, p.y);
}
// MODE 0: key = bucket of a point (order[q], bits 0-30), result = kept flag in bit 31 of order[j]; MODE 1: key = kept flag,
// result = tmp[position]
template <int MODE>
__global__ void __launch_bounds__(UR_THREADS) k_bk_count(const BriskKeyPoint* __restrict__ kp, const BriskFrameCounters* __restrict__ counters,
                                                         int* __restrict__ order, BriskKeyPoint* __restrict__ tmp, int kp_cap, BkBuckets B,
                                                         int max_keypoints) {
  __shared__ __attribute__((aligned(16))) float tsc[UR_THREADS / 64][UF_TILE];
  __shared__ __attribute__((aligned(16))) int tkey[UR_THREADS / 64][UF_TILE];
  __shared__ int scount[BK_POINTS];
  const int frame = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = min(counters[frame].nkp, kp_cap);
  const int j0 = blockIdx.x * BK_POINTS;
  if (j0 >= n || (B.single && n <= max_keypoints)) return;  // (:87-88: one bucket sorts and cuts only when there are too many points)
  const BriskKeyPoint* K = kp + (long)frame * kp_cap;
  int* ord = order + (long)frame * kp_cap;
  const int j = j0 + lane;
  BriskKeyPoint me;
  me.x = me.y = 0.f; me.response = 0.f;
  if (j < n) me = K[j];
  const float mine = me.response;
  // (MODE 1: only kept points need a position; the flags were written by the launch before this one)
  const int mykey = MODE == 0 ? (j < n ? (ord[j] & 0x7FFFFFFF) : -2) : 1;
  const bool active = j < n && (MODE == 0 || ord[j] < 0);
  const int own = j0 / UF_TILE * UF_TILE;
  if (tid < BK_POINTS) scount[tid] = 0;
  __syncthreads();
  int cnt = 0;
  float* ts = tsc[wave];
  int* tk = tkey[wave];
  for (int t0 = wave * UF_TILE; t0 < n; t0 += (UR_THREADS / 64) * UF_TILE) {
#pragma unroll
    for (int k = 0; k < UF_TILE / 64; ++k) {
      const int q = t0 + k * 64 + lane;
      float v = -3.0e38f;
      int key = -1;  // (matches no bucket and no kept flag)
      if (q < n) {
        v = K[q].response;
        // (MODE 0: other workgroups set bit 31 of these words meanwhile - the bucket is in the bits below)
        const int o = __hip_atomic_load(&ord[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        key = MODE == 0 ? (o & 0x7FFFFFFF) : (o < 0 ? 1 : 0);
      }
      ts[k * 64 + lane] = v;
      tk[k * 64 + lane] = key;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (active) {
      // (four scores and four keys per LDS read: one element per iteration is bound by the latency of its two reads)
      if (t0 < own) {
#pragma unroll 2
        for (int q = 0; q < UF_TILE; q += 4) {
          const float4 sv = *reinterpret_cast<const float4*>(&ts[q]);
          const int4 kv = *reinterpret_cast<const int4*>(&tk[q]);
          cnt += ((kv.x == mykey && sv.x >= mine) ? 1 : 0) + ((kv.y == mykey && sv.y >= mine) ? 1 : 0) +
                 ((kv.z == mykey && sv.z >= mine) ? 1 : 0) + ((kv.w == mykey && sv.w >= mine) ? 1 : 0);
        }
      } else if (t0 > own) {
#pragma unroll 2
        for (int q = 0; q < UF_TILE; q += 4) {
          const float4 sv = *reinterpret_cast<const float4*>(&ts[q]);
          const int4 kv = *reinterpret_cast<const int4*>(&tk[q]);
          cnt += ((kv.x == mykey && sv.x > mine) ? 1 : 0) + ((kv.y == mykey && sv.y > mine) ? 1 : 0) +
                 ((kv.z == mykey && sv.z > mine) ? 1 : 0) + ((kv.w == mykey && sv.w > mine) ? 1 : 0);
        }
      } else {
        for (int q = 0; q < UF_TILE; ++q) cnt += (tk[q] == mykey && (ts[q] > mine || (ts[q] == mine && t0 + q < j))) ? 1 : 0;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (cnt) atomicAdd(&scount[lane], cnt);
  __syncthreads();
  if (wave != 0 || !active) return;
  cnt = scount[lane];
  if (MODE == 0) { if (cnt < B.cap) __hip_atomic_fetch_or(&ord[j], (int)0x80000000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  else tmp[(long)frame * kp_cap + cnt] = me;
}

__global__ void __launch_bounds__(UF_THREADS) k_bk_finish(BriskKeyPoint* __restrict__ kp, BriskFrameCounters* __restrict__ counters,
                                                          const int* __restrict__ order, const BriskKeyPoint* __restrict__ tmp, int kp_cap,
                                                          int single, int max_keypoints) {
  __shared__ int wsum[UF_THREADS / 64];
  __shared__ int kept_s;
  const int frame = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = min(counters[frame].nkp, kp_cap);
  if (n == 0 || (single && n <= max_keypoints)) return;
  const int* ord = order + (long)frame * kp_cap;
  int kept = 0;
  for (int j = tid; j < n; j += UF_THREADS) kept += ord[j] < 0 ? 1 : 0;
  for (int off = 32; off > 0; off >>= 1) kept += __shfl_xor(kept, off, 64);
  if (lane == 0) wsum[wave] = kept;
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int q = 0; q < UF_THREADS / 64; ++q) t += wsum[q];
    kept_s = t;
  }
  __syncthreads();
  const int nkept = kept_s;
  BriskKeyPoint* K = kp + (long)frame * kp_cap;
  const BriskKeyPoint* T = tmp + (long)frame * kp_cap;
  for (int i = tid; i < nkept; i += UF_THREADS) K[i] = T[i];
  if (tid == 0) counters[frame].nkp = nkept;
}

void brisk_launch_bucketing(BriskKeyPoint* kp, BriskFrameCounters* counters, int* order, BriskKeyPoint* tmp, int kp_cap, int rows,
                            int cols, int nbu, int nbv, int max_keypoints, int nframes, hipStream_t s) {
  if (nframes <= 0) return;
  BkBuckets B;
  B.nbu = nbu; B.nbv = nbv;
  B.single = (nbu == 1 || nbv == 1) ? 1 : 0;
  B.cap = B.single ? max_keypoints : max_keypoints / (nbu * nbv);
  B.step_u = 1u + (unsigned)(cols - 1) / (unsigned)nbu;
  B.step_v = 1u + (unsigned)(rows - 1) / (unsigned)nbv;
  const dim3 grid((kp_cap + BK_POINTS - 1) / BK_POINTS, nframes);
  hipLaunchKernelGGL(k_bk_keys, dim3((kp_cap + 255) / 256, nframes), dim3(256), 0, s, kp, counters, order, kp_cap, B, max_keypoints);
  hipLaunchKernelGGL(k_bk_count<0>, grid, dim3(UR_THREADS), 0, s, kp, counters, order, tmp, kp_cap, B, max_keypoints);
  hipLaunchKernelGGL(k_bk_count<1>, grid, dim3(UR_THREADS), 0, s, kp, counters, order, tmp, kp_cap, B, max_keypoints);
  hipLaunchKernelGGL(k_bk_finish, dim3(nframes), dim3(UF_THREADS), 0, s, kp, counters, order, tmp, kp_cap, B.single, max_keypoints);
}
