// brisk_describe.hip - descriptor kernels of the MI355X BRISK engine (gfx950): k_desc_prepare (scale index, border
// filter, processing order) and k_describe (pattern sampling, orientation, bits).  The integral image kernel lives in
// brisk_kernels.hip (it shares the pyramid kernel's band sums).
// brisk/src/brisk-descriptor-extractor.cc:612-778 is the reference of everything here.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "brisk_common.h"
#include "brisk_device_describe.h"
#include "brisk_kernels.h"

// ------------------------------------------------------------------------------------------------
#define DP_MAXSORT 2048   // (= DP_SMALL_N: the one-workgroup kernel never sees more)
#ifndef DP_THREADS
#define DP_THREADS 1024
#endif
// Frames with more input keypoints than this take the multi-workgroup path (k_dp_count / k_dp_scan / k_dp_scatter): the
// one-workgroup kernel below ranks its keys by counting (n^2 / 1024 steps per thread) and walks its inputs 1024 at a time.
#define DP_SMALL_N 2048
#define DP_SB DP_THREADS  // buckets of the one-workgroup kernel's processing order (pieces of 64-row bands)
// work area of the multi-workgroup path, ints per frame: [0] keypoints without an angle, [DP_W_HIST ..] bucket histogram /
// cursors (DP_MAXBUCKETS + 1), [DP_W_BLK ..] kept keypoints per block of 1024 inputs (then their exclusive prefix)
#define DP_MAXBUCKETS 4096
#define DP_W_HIST 16
#define DP_W_BLK (DP_W_HIST + DP_MAXBUCKETS + 16)
// k_desc_prepare: per frame, scale index + border filter (brisk-descriptor-extractor.cc:636-662),
// stable compaction into dkp (keypoints) / dscale.  One workgroup per frame.
// ------------------------------------------------------------------------------------------------
// Sort key of the processing order (spatial locality of the keypoints that k_describe has in flight together); the
// low 11 bits carry the keypoint's index so that up to 2048 keys are all different.  mode (experiments, debug bits 4-7):
// 0 = 64-row bands, x inside a band; 1 = Hilbert curve over 64 x 64 pixel tiles; 2 / 3 = bands of 128 / 32 rows;
// 4 = scale class (index >> 4) first, then 64-row bands
__device__ __forceinline__ unsigned dp_order_key(int x, int y, int sc, int j, int mode) {
  const unsigned low = (unsigned)(j & 0x7FF);
  if (mode == 1) {
    unsigned tx = (unsigned)x >> 6, ty = (unsigned)y >> 6, d = 0;
    for (unsigned s = 64; s > 0; s >>= 1) {  // 128 x 128 tiles cover the engine's 8191-pixel limit
      const unsigned rx = (tx & s) ? 1u : 0u, ry = (ty & s) ? 1u : 0u;
      d += s * s * ((3u * rx) ^ ry);
      if (ry == 0) {
        if (rx == 1) { tx = 127u - tx; ty = 127u - ty; }
        const unsigned t = tx; tx = ty; ty = t;
      }
    }
    return (d << 18) | (((unsigned)x & 63u) << 12) | low;
  }
  if (mode == 2) return ((unsigned)(y >> 7) << 24) | (((unsigned)x & 0x1FFFu) << 11) | low;
  if (mode == 3) return ((unsigned)(y >> 5) << 24) | (((unsigned)x & 0x1FFFu) << 11) | low;
  if (mode == 4) return ((unsigned)(sc >> 4) << 30) | (((unsigned)(y >> 6) & 0x7Fu) << 23) | ((((unsigned)x >> 1) & 0xFFFu) << 11) | low;
  return ((unsigned)(y >> 6) << 24) | (((unsigned)x & 0x1FFFu) << 11) | low;
}

__global__ void __launch_bounds__(DP_THREADS) k_desc_prepare(BriskGeom G, BriskPatternDev P, const BriskKeyPoint* kp_in,
                                                       const int* n_in_ptr, long n_in_stride, BriskFrameCounters* counters,
                                                       BriskKeyPoint* dkp, int* dscale, int* dperm, uint4* drec, int kp_cap,
                                                       int* work, long work_stride) {
  __shared__ int wtot[DP_THREADS / 64];
  __shared__ int base, nest;
  __shared__ __attribute__((aligned(16))) unsigned pkey[DP_MAXSORT + 4];
  BRISK_CHAIN_SETPRIO();
  const int frame = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = min(*(const int*)((const char*)n_in_ptr + (long)frame * n_in_stride), kp_cap);
  const BriskKeyPoint* K = kp_in + (long)frame * kp_cap;
  if (n > DP_SMALL_N) {  // the multi-workgroup path takes this frame: its histogram starts from zero
    int* wk = work + (long)frame * work_stride;
    for (int q = tid; q < DP_W_BLK; q += DP_THREADS) wk[q] = 0;
    return;
  }
  // the kept keypoints' records (what k_describe reads) by compacted index, and the border per scale index, on chip: the
  // processing order below writes a record straight from here (it used to go order -> memory -> keypoint -> scale ->
  // record: three dependent round trips beside the integral kernel, which keeps the memory system busy)
  __shared__ uint4 srec[DP_SMALL_N];
  __shared__ int sborder[BRISK_SCALES];
  __shared__ float sthr[BRISK_SCALES];  // the 64 size thresholds of the scale index: one load instead of a chain of them per keypoint
  if (tid == 0) { base = 0; nest = 0; }
  if (tid < BRISK_SCALES) { sborder[tid] = P.size_list[tid]; sthr[tid] = P.size_thresh[tid]; }
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += DP_THREADS) {
    const int i = i0 + tid;
    bool keep = false;
    int sc = 0;
    BriskKeyPoint kp;
    if (i < n) {
      kp = K[i];
      if (!P.scale_invariant) sc = P.basicscale;  // (brisk_scale_index, on the thresholds in LDS)
      else {
#pragma unroll 8
        for (int q = 1; q < BRISK_SCALES; ++q) sc += (kp.size >= sthr[q]) ? 1 : 0;
      }
      const int border = sborder[sc];  // == brisk_inside_border(P, sc, ...)
      keep = !((kp.x < (float)border) || (kp.x >= (float)(G.L[0].w - border)) || (kp.y < (float)border) || (kp.y >= (float)(G.L[0].h - border)));
    }
    // stable compaction: position = kept keypoints before this one (ballots inside the wave, wave totals through LDS)
    const unsigned long long bal = __ballot(keep);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wtot[wave] = __popcll(bal);
    const unsigned long long bal_e = __ballot(keep && kp.angle == -1.0f);  // keypoints whose orientation has to be estimated
    if (lane == 0 && bal_e) atomicAdd(&nest, __popcll(bal_e));
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int k = 0; k < DP_THREADS / 64; ++k) {
      const int t = wtot[k];
      wbase += (k < wave) ? t : 0;
      total += t;
    }
    if (keep) {
      const int j = base + wbase + before;
      dkp[(long)frame * kp_cap + j] = kp;
      if (j < DP_MAXSORT) pkey[j] = dp_order_key((int)kp.x, (int)kp.y, sc, j, (BRISK_DBG_FLAGS(G) >> 4) & 0xF);
      srec[j] = make_uint4(__float_as_uint(kp.x), __float_as_uint(kp.y), __float_as_uint(kp.angle), (unsigned)sc | ((unsigned)j << 8));
    }
    __syncthreads();
    if (tid == 0) base += total;
    __syncthreads();
  }
  if (tid == 0) { counters[frame].ndesc = base; counters[frame].nestimate = nest; counters[frame].desc_ticket = 0; counters[frame].orient_ticket = 0; }
  // Processing order for k_describe: keypoints sorted by 64-row band, then x, so that keypoints sampled at the
  // same time touch the same part of the integral image (the output order stays (layer, y, x)).
  // (m <= DP_SMALL_N here: the keys carry j in their low bits, all different)
  const int m = base;
  if (tid < 4) pkey[m + tid] = 0xFFFFFFFFu;  // the count below reads four keys at a time
  __syncthreads();
  // the keypoints again, in processing order, as one 16-byte record each: k_describe reads them with a single
  // (prefetchable) load instead of the dependent chain order -> keypoint -> scale
  if (((BRISK_DBG_FLAGS(G) >> 4) & 0xF) != 0) {  // the experimental orders (debug bits 4-7): rank by counting all smaller keys
    for (int j = tid; j < m; j += DP_THREADS) {
      const unsigned kj = pkey[j];
      int r = 0;
      for (int q = 0; q < m; q += 4) {
        const uint4 kk = *reinterpret_cast<const uint4*>(&pkey[q]);
        r += (kk.x < kj ? 1 : 0) + (kk.y < kj ? 1 : 0) + (kk.z < kj ? 1 : 0) + (kk.w < kj ? 1 : 0);
      }
      drec[(long)frame * kp_cap + r] = srec[j];
    }
    return;
  }
  // Rank = keys of smaller buckets + smaller keys of the own bucket; a bucket is a piece of a 64-row band (the key's
  // order: band, x), a few keypoints, chained in LDS.  Round 5: counting ALL smaller keys per key (n^2 / 4 LDS reads) was
  // half of the kernel's 44 us for a 1080p frame's 1 200 keypoints - on the critical path of every one-frame compute()
  // and of the window beside the integral kernel in a batch.
  {
    __shared__ int bhead[DP_SB], bstart[DP_SB];
    __shared__ int bnext[DP_SMALL_N];
    const int nbands = ((G.L[0].h - 1) >> 6) + 1;
    int xs = 6;
    while (nbands * (((G.L[0].w - 1) >> xs) + 1) > DP_SB) ++xs;
    const int nbx = ((G.L[0].w - 1) >> xs) + 1;
    auto bucket_of = [&](unsigned key) { return (int)(key >> 24) * nbx + (int)(((key >> 11) & 0x1FFFu) >> xs); };
    for (int b = tid; b < DP_SB; b += DP_THREADS) { bhead[b] = -1; bstart[b] = 0; }
    __syncthreads();
    for (int j = tid; j < m; j += DP_THREADS) {
      const int b = bucket_of(pkey[j]);
      atomicAdd(&bstart[b], 1);
      bnext[j] = atomicExch(&bhead[b], j);
    }
    __syncthreads();
    {  // exclusive prefix over the buckets (one per thread)
      static_assert(DP_SB == DP_THREADS, "one bucket per thread");
      const int v = bstart[tid];
      int incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
      }
      __syncthreads();  // (wtot: the compaction's last use is over)
      if (lane == 63) wtot[wave] = incl;
      __syncthreads();
      int woff = 0;
#pragma unroll
      for (int k = 0; k < DP_THREADS / 64; ++k) woff += (k < wave) ? wtot[k] : 0;
      bstart[tid] = woff + incl - v;
    }
    __syncthreads();
    for (int j = tid; j < m; j += DP_THREADS) {
      const unsigned kj = pkey[j];
      const int b = bucket_of(kj);
      int r = bstart[b];
      for (int q = bhead[b]; q >= 0; q = bnext[q]) r += (pkey[q] < kj) ? 1 : 0;
      drec[(long)frame * kp_cap + r] = srec[j];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same preparation for frames with many keypoints (BASELINE config 5: 100 000 provided keypoints on one frame; 4K
// frames with several thousand), any count, several workgroups per frame, O(n):
//   k_dp_count    block b of a frame takes inputs [1024 b, 1024 b + 1024): scale index + border filter, kept keypoints per
//                 block, histogram of the kept keypoints over square tiles of the frame (at most DP_MAXBUCKETS)
//   k_dp_scan     one workgroup per frame: exclusive prefix over the block counts (-> stable compaction offsets) and over
//                 the tile histogram (-> start of every tile's stretch of the processing order), frame counters
//   k_dp_scatter  keypoint j (input order, filtered: the OUTPUT order of the reference) -> dkp[j]; its record goes to the
//                 next free place of its tile's stretch.  The processing order is tile after tile in raster order - the
//                 spatial locality k_describe's L2 hit rate rests on - and unordered inside a tile (an atomic cursor):
//                 results are written by index j, so they do not depend on it.
// ------------------------------------------------------------------------------------------------
struct DpTiles { int shift, tiles_x, w, h; };
__device__ __forceinline__ int dp_bucket(const DpTiles& T, float x, float y) {
  const int xi = min(max((int)x, 0), T.w - 1), yi = min(max((int)y, 0), T.h - 1);
  return (yi >> T.shift) * T.tiles_x + (xi >> T.shift);
}
// keep flag, scale index and wave-level position of input i
struct DpItem { bool keep; int sc; BriskKeyPoint kp; };
__device__ __forceinline__ DpItem dp_item(const BriskGeom& G, const BriskPatternDev& P, const BriskKeyPoint* K, int i, int n) {
  DpItem it;
  it.keep = false; it.sc = 0;
  if (i < n) {
    it.kp = K[i];
    it.sc = brisk_scale_index(P, it.kp.size);
    it.keep = brisk_inside_border(P, it.sc, it.kp.x, it.kp.y, G.L[0].w, G.L[0].h);
  }
  return it;
}

__global__ void __launch_bounds__(DP_THREADS) k_dp_count(BriskGeom G, BriskPatternDev P, DpTiles T, const BriskKeyPoint* kp_in,
                                                          const int* n_in_ptr, long n_in_stride, int kp_cap, int* work, long work_stride) {
  __shared__ int wtot[DP_THREADS / 64];
  const int frame = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = min(*(const int*)((const char*)n_in_ptr + (long)frame * n_in_stride), kp_cap);
  if (n <= DP_SMALL_N) return;
  int* wk = work + (long)frame * work_stride;
  // (the grid holds a few workgroups per frame, not one per 1024 keypoints of the CAPACITY - 4096 workgroups that exit at
  // once cost 46 us per launch in a 256-frame batch -: a workgroup takes block after block)
  for (int blk = blockIdx.x; blk * DP_THREADS < n; blk += gridDim.x) {
    const DpItem it = dp_item(G, P, kp_in + (long)frame * kp_cap, blk * DP_THREADS + tid, n);
    const unsigned long long bal = __ballot(it.keep);
    if (lane == 0) wtot[wave] = __popcll(bal);
    const unsigned long long bal_e = __ballot(it.keep && it.kp.angle == -1.0f);
    if (lane == 0 && bal_e) atomicAdd(&wk[0], __popcll(bal_e));
    if (it.keep) atomicAdd(&wk[DP_W_HIST + dp_bucket(T, it.kp.x, it.kp.y)], 1);
    __syncthreads();
    if (tid == 0) {
      int t = 0;
#pragma unroll
      for (int k = 0; k < DP_THREADS / 64; ++k) t += wtot[k];
      wk[DP_W_BLK + blk] = t;
    }
    __syncthreads();  // (wtot is rewritten by the next block)
  }
}

// exclusive prefix of v[0 .. n) in place, returns the total (one workgroup of DP_THREADS threads; sh: DP_THREADS / 64 + 1 ints)
__device__ __forceinline__ int dp_scan_inplace(int* v, int n, int* sh) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int carry = 0;
  for (int i0 = 0; i0 < n; i0 += DP_THREADS) {
    const int i = i0 + tid;
    const int x = i < n ? v[i] : 0;
    int incl = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < DP_THREADS / 64; ++k) { const int t = sh[k]; woff += (k < wave) ? t : 0; tot += t; }
    if (i < n) v[i] = carry + woff + incl - x;
    carry += tot;
    __syncthreads();
  }
  return carry;
}

__global__ void __launch_bounds__(DP_THREADS) k_dp_scan(const int* n_in_ptr, long n_in_stride, int kp_cap, int nbuckets,
                                                         BriskFrameCounters* counters, int* work, long work_stride) {
  __shared__ int sh[DP_THREADS / 64 + 1];
  const int frame = blockIdx.x;
  const int n = min(*(const int*)((const char*)n_in_ptr + (long)frame * n_in_stride), kp_cap);
  if (n <= DP_SMALL_N) return;
  int* wk = work + (long)frame * work_stride;
  const int m = dp_scan_inplace(wk + DP_W_BLK, (n + DP_THREADS - 1) / DP_THREADS, sh);
  (void)dp_scan_inplace(wk + DP_W_HIST, nbuckets, sh);
  if (threadIdx.x == 0) {
    counters[frame].ndesc = m; counters[frame].nestimate = wk[0]; counters[frame].desc_ticket = 0; counters[frame].orient_ticket = 0;
  }
}

__global__ void __launch_bounds__(DP_THREADS) k_dp_scatter(BriskGeom G, BriskPatternDev P, DpTiles T, const BriskKeyPoint* kp_in,
                                                            const int* n_in_ptr, long n_in_stride, int kp_cap, int* work,
                                                            long work_stride, BriskKeyPoint* dkp, uint4* drec) {
  __shared__ int wtot[DP_THREADS / 64];
  const int frame = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n = min(*(const int*)((const char*)n_in_ptr + (long)frame * n_in_stride), kp_cap);
  if (n <= DP_SMALL_N) return;
  int* wk = work + (long)frame * work_stride;
  for (int blk = blockIdx.x; blk * DP_THREADS < n; blk += gridDim.x) {
    const DpItem it = dp_item(G, P, kp_in + (long)frame * kp_cap, blk * DP_THREADS + tid, n);
    const unsigned long long bal = __ballot(it.keep);
    if (lane == 0) wtot[wave] = __popcll(bal);
    __syncthreads();
    if (it.keep) {
      int j = wk[DP_W_BLK + blk] + __popcll(bal & ((1ull << lane) - 1ull));
#pragma unroll
      for (int k = 0; k < DP_THREADS / 64; ++k) j += (k < wave) ? wtot[k] : 0;
      dkp[(long)frame * kp_cap + j] = it.kp;
      const int pos = atomicAdd(&wk[DP_W_HIST + dp_bucket(T, it.kp.x, it.kp.y)], 1);
      drec[(long)frame * kp_cap + pos] = make_uint4(__float_as_uint(it.kp.x), __float_as_uint(it.kp.y), __float_as_uint(it.kp.angle),
                                                    (unsigned)it.sc | ((unsigned)j << 8));
    }
    __syncthreads();  // (wtot is rewritten by the next block)
  }
}

// ------------------------------------------------------------------------------------------------
// k_describe: persistent waves, one RUN of consecutive keypoints (in processing order) per ticket.
//
// What bounds this kernel is the vector L1 / L2 path of the gathers (tools/microbench_gather.hip, profiles/r03_*): a
// describe-shaped pass costs the same at 1 and at 8 waves per SIMD, but 2.3 x less when the keypoints that are in
// flight on an XCD at the same time are neighbours (their lines are then L2 hits).  So:
//  * keypoints are dealt in their spatial processing order through one ticket counter per frame, and every wave first
//    serves the frames of its own XCD (frame % 8 == XCC id; speed only: afterwards it serves whatever is left, so the
//    result does not depend on where the hardware places a workgroup);
//  * the samples of a run are laid out flat, lane = (keypoint, pattern point): RUN x npoints samples in
//    ceil(RUN x npoints / 64) rounds of one sample per lane - a 66-point pattern no longer pays a second, 97 % idle
//    round per keypoint - and the orientations of a run are computed together (one fp64 atan2 per run, not per
//    keypoint);
//  * the number of resident waves is a launch parameter (persistent grid, dynamic LDS as the occupancy limiter).
// Per round the lane's parameters for the NEXT round (table entry of (scale, point), rotated unit offset) are already
// in flight.  Long pairs: integer wave reductions (order independent); bits: 64-wide ballots.
// ------------------------------------------------------------------------------------------------
#ifndef DS_WAVES
#define DS_WAVES 2
#endif
#define DS_MAXRUN 8
#define DS_MAXQ 256        // frames per work queue (more frames per launch: more queues)
#ifndef DS_BLOCKS_PER_CU
#define DS_BLOCKS_PER_CU 3
#endif
#define DS_LP_LDS 1024
#define DS_REG_LONG 14   // long pairs per lane held in registers (REGTAB): 896 pairs
#define DS_REG_SHORT 8   // short pairs per lane: 512 bits
#define DS_GETREG_XCC_ID (20 | (3 << 11))  // s_getreg_b32 hwreg(HW_REG_XCC_ID, 0, 4)

typedef uint32_t __attribute__((ext_vector_type(2))) ds_u32x2;

// sum over the 64 lanes, returned in every lane (DPP inclusive scan, lane 63 holds the total)
__device__ __forceinline__ int ds_wave_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return __builtin_amdgcn_readlane(v, 63);
}

struct DsLane {  // a lane's sample of one round
  float kx, ky;
  int4 tab;      // {mult, sigma (float bits), scaling | shift << 24, magic} of (scale, point): BriskPatternDev::tab
  int ti;        // scale * npoints + point
  double2 uv;    // unit offset of the point at the keypoint's rotation
  int slot;      // keypoint-in-run * npoints + point
};
typedef BriskBoxPrep DsPrep;  // address / weight stage of a sample (brisk_device_describe.h)
// The two displaced corner pixels of the reference quirk (brisk-descriptor-extractor.cc:453) come out
// of the integral image as well - pixel (u, v) = I[v+1][u+1] - I[v+1][u] - I[v][u+1] + I[v][u], exact in wrap-around
// arithmetic: the bottom row pair's first row is loaded three columns wide and row y_bottom - 1 adds one 8-byte gather per
// side - instead of two byte gathers from the frame, which then is no part of the kernel's working set (2.1 MB per 1080p
// frame, a fifth of the compulsory misses).  Only a box that ends in the image's last column still reads the frame (the
// reference's linear address is the first pixel of the next row): a rare, separate pair of loads.
typedef uint32_t __attribute__((ext_vector_type(3))) ds_u32x3;
struct DsRaw {
  ds_u32x2 p00, p02, p10, p12, p30, p32, ql, qr;
  ds_u32x3 p20, p22;
  unsigned br, bl;
};
// SmoothedIntensity split into address / load / combine stages: the arithmetic is brisk_box_prep / brisk_box_acc /
// brisk_box_divide of brisk_device_describe.h (host + device: tests/emul runs it on the CPU)
__device__ __forceinline__ DsPrep ds_prep(float xf, float yf, float sigma_half, int tab_z, int tab_w) {
  return brisk_box_prep(xf, yf, sigma_half, tab_z, tab_w);
}
// The 4 x 4 integral samples as eight gathers through a buffer descriptor of the frame's integral image: four 32-bit
// offsets per lane (the corners), the second row of each pair through the scalar offset; plus what the two displaced
// bottom corners of the reference quirk need (above).
// I24: the integral image in 3-byte elements (values modulo 2^24): the same ten gathers at byte offsets 3 x - the
// hardware takes 8- and 12-byte gathers at any byte alignment at the rate of aligned ones (tools/microbench_unaligned.hip:
// 255 vs 259 G gathers/s out of L2) -, 8 bytes hold the 6 of two adjacent columns, 12 bytes the 9 of three.
template <bool I24>
__device__ __forceinline__ void ds_load(DsRaw& r, const DsPrep& p, __amdgpu_buffer_rsrc_t rs_img, int stride, int cols,
                                        __amdgpu_buffer_rsrc_t rs_int, int istride) {
  constexpr int ES = I24 ? 3 : 4;
  const int rowb = istride * ES;
  const int o_t = p.y_top * rowb, o_b = p.y_bottom * rowb;
  const int o_tl = o_t + p.x_left * ES, o_tr = o_t + p.x_right * ES, o_bl = o_b + p.x_left * ES, o_br = o_b + p.x_right * ES;
  r.p00 = __builtin_amdgcn_raw_buffer_load_b64(rs_int, o_tl, 0, 0);
  r.p02 = __builtin_amdgcn_raw_buffer_load_b64(rs_int, o_tr, 0, 0);
  r.p10 = __builtin_amdgcn_raw_buffer_load_b64(rs_int, o_tl, rowb, 0);
  r.p12 = __builtin_amdgcn_raw_buffer_load_b64(rs_int, o_tr, rowb, 0);
  r.ql = __builtin_amdgcn_raw_buffer_load_b64(rs_int, o_bl - rowb + ES, 0, 0);   // row y_bottom - 1, columns x_left + 1, + 2
  r.qr = __builtin_amdgcn_raw_buffer_load_b64(rs_int, o_br - rowb + ES, 0, 0);   // row y_bottom - 1, columns x_right + 1, + 2
  r.p20 = __builtin_amdgcn_raw_buffer_load_b96(rs_int, o_bl, 0, 0);
  r.p22 = __builtin_amdgcn_raw_buffer_load_b96(rs_int, o_br, 0, 0);
  r.p30 = __builtin_amdgcn_raw_buffer_load_b64(rs_int, o_bl, rowb, 0);
  r.p32 = __builtin_amdgcn_raw_buffer_load_b64(rs_int, o_br, rowb, 0);
  r.br = 0; r.bl = 0;
  const bool wrap = p.quirk && (p.x_right + 1 >= cols || p.x_left + 1 >= cols);
  if (__builtin_expect(wrap, 0)) {
    const int qy = max(p.y_bottom - 1, 0), xr = p.x_right + 1, xl = p.x_left + 1, o_q = qy * stride, wr = stride - cols;
    r.br = __builtin_amdgcn_raw_buffer_load_b8(rs_img, o_q + xr + (xr >= cols ? wr : 0), 0, 0);
    r.bl = __builtin_amdgcn_raw_buffer_load_b8(rs_img, o_q + xl + (xl >= cols ? wr : 0), 0, 0);
    r.br |= 0x100u; r.bl |= 0x100u;  // (marks "from the frame")
  }
}
// two / three adjacent 24-bit columns out of the 8 / 12 bytes that start at the first one.  Only the low 24 bits of the
// results mean anything: every use is a four-corner difference that brisk_box_acc masks to 24 bits, so the byte above is
// left as it comes (one v_alignbit per column instead of shifts and masks).
__device__ __forceinline__ void ds_unpack2(ds_u32x2 v, uint32_t& a, uint32_t& b) {
  a = v.x;
  b = __builtin_amdgcn_alignbit(v.y, v.x, 24);
}
template <bool I24>
__device__ __forceinline__ int ds_combine(const DsPrep& p, const DsRaw& r) {
  uint32_t i00, i01, i02, i03, i10, i11, i12, i13, i20, i21, i2x, i22, i23, i2y, i30, i31, i32, i33, ql0, ql1, qr0, qr1;
  if (I24) {
    ds_unpack2(r.p00, i00, i01); ds_unpack2(r.p02, i02, i03); ds_unpack2(r.p10, i10, i11); ds_unpack2(r.p12, i12, i13);
    ds_unpack2(ds_u32x2{r.p20.x, r.p20.y}, i20, i21); i2x = __builtin_amdgcn_alignbit(r.p20.z, r.p20.y, 16);
    ds_unpack2(ds_u32x2{r.p22.x, r.p22.y}, i22, i23); i2y = __builtin_amdgcn_alignbit(r.p22.z, r.p22.y, 16);
    ds_unpack2(r.p30, i30, i31); ds_unpack2(r.p32, i32, i33); ds_unpack2(r.ql, ql0, ql1); ds_unpack2(r.qr, qr0, qr1);
  } else {
    i00 = r.p00.x; i01 = r.p00.y; i02 = r.p02.x; i03 = r.p02.y; i10 = r.p10.x; i11 = r.p10.y; i12 = r.p12.x; i13 = r.p12.y;
    i20 = r.p20.x; i21 = r.p20.y; i2x = r.p20.z; i22 = r.p22.x; i23 = r.p22.y; i2y = r.p22.z;
    i30 = r.p30.x; i31 = r.p30.y; i32 = r.p32.x; i33 = r.p32.y; ql0 = r.ql.x; ql1 = r.ql.y; qr0 = r.qr.x; qr1 = r.qr.y;
  }
  constexpr uint32_t mask = I24 ? 0xFFFFFFu : 0xFFFFFFFFu;
  unsigned qbr = (i2y - i23 - qr1 + qr0) & mask;  // pixel (x_right + 1, y_bottom - 1)
  unsigned qbl = (i2x - i21 - ql1 + ql0) & mask;  // pixel (x_left + 1, y_bottom - 1)
  if (r.br & 0x100u) { qbr = r.br & 0xFFu; qbl = r.bl & 0xFFu; }
  const uint32_t acc = brisk_box_acc(p, i00, i01, i02, i03, i10, i11, i12, i13, i20, i21, i22, i23, i30, i31, i32, i33, qbr, qbl, mask);
  if (__any(p.shift < 0)) return brisk_box_divide(p, acc);  // (a pattern with degenerate boxes: the plain division is compiled in but skipped)
  return brisk_div_by_magic((int)acc, p.magic, p.shift);
}
// ---- round 5: the same ten gathers, dealt so that the two lanes of a lane pair read the SAME row ------------------------------
// What the vector L1 charges a gather for is one tag look-up per lane whose line no neighbouring lane shares
// (profiles/r05_microbench_il2.txt: ten gathers that all hit cost 0.63 ms with a line per lane, 0.29 with a line per lane pair).
// A lane is still a pattern point, but of the lane pair (2 k, 2 k + 1) = samples (a, b) five gathers serve a - the even lane
// reads a's left column pairs, the odd lane a's right ones: the same rows, one line for narrow boxes - and five serve b the same
// way.  Every lane reduces its side of both samples to six numbers (brisk_box_side), hands the partner's six over by DPP and
// finishes its own sample (brisk_box_acc_pair).  Microbenchmark: 1.416 -> 1.280 ms on BASELINE config 2's keypoints.
__device__ __forceinline__ int ds_dpp_even(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xA0, 0xf, 0xf, false); }     // quad_perm(0,0,2,2)
__device__ __forceinline__ int ds_dpp_odd(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xF5, 0xf, 0xf, false); }      // quad_perm(1,1,3,3)
__device__ __forceinline__ int ds_dpp_partner(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false); }  // quad_perm(1,0,3,2)
struct DsRawPair {
  ds_u32x2 at0, at1, aq, ab1, bt0, bt1, bq, bb1;  // a*: the gathers that serve the even lane's sample, b*: the odd lane's
  ds_u32x3 ab0, bb0;
  unsigned br, bl;                                // the lane's OWN sample: displaced corner pixels from the frame (last column)
  bool a_quirk, b_quirk;
};
template <bool I24>
__device__ __forceinline__ void ds_load_pair(DsRawPair& r, const DsPrep& p, bool odd, bool pair_valid, bool valid, __amdgpu_buffer_rsrc_t rs_img,
                                             int stride, int cols, __amdgpu_buffer_rsrc_t rs_int, int istride) {
  constexpr int ES = I24 ? 3 : 4;
  const int rowb = istride * ES;
  const int o_t = p.y_top * rowb, o_b = p.y_bottom * rowb;
  const int o_tl = o_t + p.x_left * ES, o_tr = o_t + p.x_right * ES, o_bl = o_b + p.x_left * ES, o_br = o_b + p.x_right * ES;
  // (DPP reads are statements of their own: inside a conditional expression they would run with half of the lanes switched
  // off and read nothing from them)
  const int e_tr = ds_dpp_even(o_tr), o_tl_ = ds_dpp_odd(o_tl), e_br = ds_dpp_even(o_br), o_bl_ = ds_dpp_odd(o_bl);
  const int qk = p.quirk ? 1 : 0;
  const int e_qk = ds_dpp_even(qk), o_qk = ds_dpp_odd(qk);
  r.a_quirk = e_qk != 0; r.b_quirk = o_qk != 0;
  const int a_top = odd ? e_tr : o_tl, b_top = odd ? o_tr : o_tl_;
  const int a_bot = odd ? e_br : o_bl, b_bot = odd ? o_br : o_bl_;
  if (pair_valid) {
    r.at0 = __builtin_amdgcn_raw_buffer_load_b64(rs_int, a_top, 0, 0);
    r.bt0 = __builtin_amdgcn_raw_buffer_load_b64(rs_int, b_top, 0, 0);
    r.at1 = __builtin_amdgcn_raw_buffer_load_b64(rs_int, a_top, rowb, 0);
    r.bt1 = __builtin_amdgcn_raw_buffer_load_b64(rs_int, b_top, rowb, 0);
    r.aq = __builtin_amdgcn_raw_buffer_load_b64(rs_int, a_bot - rowb + ES, 0, 0);  // row y_bottom - 1, columns c + 1, c + 2
    r.bq = __builtin_amdgcn_raw_buffer_load_b64(rs_int, b_bot - rowb + ES, 0, 0);
    r.ab0 = __builtin_amdgcn_raw_buffer_load_b96(rs_int, a_bot, 0, 0);
    r.bb0 = __builtin_amdgcn_raw_buffer_load_b96(rs_int, b_bot, 0, 0);
    r.ab1 = __builtin_amdgcn_raw_buffer_load_b64(rs_int, a_bot, rowb, 0);
    r.bb1 = __builtin_amdgcn_raw_buffer_load_b64(rs_int, b_bot, rowb, 0);
  }
  r.br = 0; r.bl = 0;
  const bool wrap = valid && p.quirk && (p.x_right + 1 >= cols || p.x_left + 1 >= cols);
  if (__builtin_expect(wrap, 0)) {  // (as ds_load: a box that ends in the image's last column reads the frame)
    const int qy = max(p.y_bottom - 1, 0), xr = p.x_right + 1, xl = p.x_left + 1, o_q = qy * stride, wr = stride - cols;
    r.br = __builtin_amdgcn_raw_buffer_load_b8(rs_img, o_q + xr + (xr >= cols ? wr : 0), 0, 0);
    r.bl = __builtin_amdgcn_raw_buffer_load_b8(rs_img, o_q + xl + (xl >= cols ? wr : 0), 0, 0);
    r.br |= 0x100u; r.bl |= 0x100u;
  }
}
template <bool I24>
__device__ __forceinline__ BriskBoxSide ds_side(ds_u32x2 t0, ds_u32x2 t1, ds_u32x2 q, ds_u32x3 b0, ds_u32x2 b1, bool quirk, bool right) {
  uint32_t t0x, t0y, t1x, t1y, qx, qy, b0x, b0y, b0z, b1x, b1y;
  if (I24) {
    ds_unpack2(t0, t0x, t0y); ds_unpack2(t1, t1x, t1y); ds_unpack2(q, qx, qy); ds_unpack2(ds_u32x2{b0.x, b0.y}, b0x, b0y); ds_unpack2(b1, b1x, b1y);
    b0z = __builtin_amdgcn_alignbit(b0.z, b0.y, 16);
  } else {
    t0x = t0.x; t0y = t0.y; t1x = t1.x; t1y = t1.y; qx = q.x; qy = q.y; b0x = b0.x; b0y = b0.y; b0z = b0.z; b1x = b1.x; b1y = b1.y;
  }
  return brisk_box_side(t0x, t0y, t1x, t1y, qx, qy, b0x, b0y, b0z, b1x, b1y, quirk, right, I24 ? 0xFFFFFFu : 0xFFFFFFFFu);
}
template <bool I24>
__device__ __forceinline__ int ds_combine_pair(const DsPrep& p, const DsRawPair& r, bool odd) {
  constexpr uint32_t mask = I24 ? 0xFFFFFFu : 0xFFFFFFFFu;
  const BriskBoxSide sa = ds_side<I24>(r.at0, r.at1, r.aq, r.ab0, r.ab1, r.a_quirk, odd);
  const BriskBoxSide sb = ds_side<I24>(r.bt0, r.bt1, r.bq, r.bb0, r.bb1, r.b_quirk, odd);
  // own: my side of my sample; par: the other side of my sample, which the partner lane computed in ITS other set
  BriskBoxSide own, par;
#define DS_X(f) own.f = odd ? sb.f : sa.f; par.f = (uint32_t)ds_dpp_partner((int)(odd ? sa.f : sb.f));
  DS_X(ct) DS_X(cb) DS_X(st) DS_X(dt) DS_X(db) DS_X(dm)
#undef DS_X
  if (r.br & 0x100u) {  // displaced corner pixels from the frame: bl belongs to the left side, br to the right one
    const unsigned pbl = r.bl & 0xFFu, pbr = r.br & 0xFFu;
    own.cb = odd ? pbr : pbl;
    par.cb = odd ? pbl : pbr;
  }
  const uint32_t acc = brisk_box_acc_pair(p, own, par, odd, mask);
  if (__any(p.shift < 0)) return brisk_box_divide(p, acc);
  return brisk_div_by_magic((int)acc, p.magic, p.shift);
}
// one sample: address stage + gathers (issue), and its combine stage
// GENERIC: the pattern has points on the bilinear branch of SmoothedIntensity (sigma < 0.5, :391-408; only with a small
// patternScale): every sample through the generic function of brisk_device_describe.h
struct DsFrame {  // wave-uniform: the frame a ticket belongs to
  const uint8_t* img;
  const uint32_t* integ;
  __amdgpu_buffer_rsrc_t rs_img, rs_int;
};
struct DsSide {  // what a run's first pass requests for the runs behind it
  int t;       // ticket (lane 0; every lane with static dealing)
  uint4 rec;   // the next run's records (lane = keypoint of the run)
};
struct DsTicket {  // wave-uniform
  int frame, k0, cnt, total;
  bool ok;
};

// One persistent launch.  What the vector memory path delivers depends on how many of the gathered lines are L2 hits
// (profiles/r03_microbench_describe_shape*: 53 samples/ns chip-wide at 99 % hits, 23 at 52 %, HBM misses cost twice
// what Infinity Cache misses cost): a keypoint's two passes run back to back in one wave (they sample the same patch: a
// two-launch split doubles the compulsory misses - measured, and in tools/cache_sim.py), the keypoints an XCD has in
// flight are one contiguous stretch of the spatial processing order, tickets are taken two runs ahead and the next
// run's records are requested before the current run's gathers.
template <int RUN, bool GENERIC, bool REGTAB, bool I24>
__global__ void __launch_bounds__(DS_WAVES * 64) k_describe(BriskGeom G, BriskPatternDev P, const uint8_t* __restrict__ pyr,
                                                            const uint32_t* __restrict__ integral, int istride,
                                                            long iframe_elems, BriskFrameCounters* counters,
                                                            BriskKeyPoint* dkp, uint4* __restrict__ drec,
                                                            uint8_t* desc, int kp_cap, int desc_pitch, int nframes, int run_fixed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ds_lds[];
  // LDS: long pairs {i, j, wdx, wdy} | short pairs i | j << 16 | per wave: records and rotations of a run, values, the
  // work queue's cumulative run / keypoint counts
  const int np = P.npoints;
  // REGTAB (at most 896 long pairs with 16-bit weights, at most 512 short pairs - the built-in patterns): every lane
  // keeps ITS pairs in registers for the kernel's lifetime (pair p belongs to lane p % 64), so a keypoint's long-pair sum
  // is 28 independent LDS reads + arithmetic instead of 14 dependent table -> values round trips.  Otherwise the tables
  // live in LDS (long pairs beyond DS_LP_LDS: in global memory).
  const bool lp_in_lds = !REGTAB && P.nlong <= DS_LP_LDS;
  int4* lp_s = reinterpret_cast<int4*>(ds_lds);
  unsigned* sp_s = reinterpret_cast<unsigned*>(lp_s + (lp_in_lds ? P.nlong : 0));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int table_bytes = REGTAB ? 0 : ((lp_in_lds ? P.nlong * 16 : 0) + ((P.nshort * 4 + 15) & ~15));
  const int vals_bytes = (RUN * np * 4 + 15) & ~15;
  const int per_wave = 2 * RUN * 16 + ((RUN * 4 + 15) & ~15) + vals_bytes + 2 * DS_MAXQ * 4;
  unsigned char* wbase = ds_lds + table_bytes + wave * per_wave;
  uint4* krec2 = reinterpret_cast<uint4*>(wbase);          // [2][RUN]: the records of this run and of the next one
  int* kth = reinterpret_cast<int*>(wbase + 2 * RUN * 16);  // [RUN]
  int* vals = reinterpret_cast<int*>(wbase + 2 * RUN * 16 + ((RUN * 4 + 15) & ~15));
  int* cum = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(vals) + vals_bytes);  // [DS_MAXQ] runs up to and including entry j
  int* cnt_q = cum + DS_MAXQ;                                                              // [DS_MAXQ] keypoints of entry j
  unsigned lp_ij[DS_REG_LONG], lp_w[DS_REG_LONG], sp_ij[DS_REG_SHORT];  // i | j << 8, wdx (low half) | wdy << 16; i | j << 16
  if (REGTAB) {
#pragma unroll
    for (int t = 0; t < DS_REG_LONG; ++t) {
      const int p = t * 64 + lane;
      lp_ij[t] = 0; lp_w[t] = 0;  // beyond the table: weight 0 contributes nothing
      if (p < P.nlong) {
        const int4 q = reinterpret_cast<const int4*>(P.long_pairs)[p];
        lp_ij[t] = (unsigned)q.x | ((unsigned)q.y << 8);
        lp_w[t] = ((unsigned)q.z & 0xFFFFu) | ((unsigned)q.w << 16);
      }
    }
#pragma unroll
    for (int t = 0; t < DS_REG_SHORT; ++t) {
      const int p = t * 64 + lane;
      sp_ij[t] = 0xFFFFFFFFu;  // beyond the table: bit 0
      if (p < P.nshort) sp_ij[t] = (unsigned)P.short_pairs[2 * p] | ((unsigned)P.short_pairs[2 * p + 1] << 16);
    }
  } else {
    if (lp_in_lds)
      for (int p = threadIdx.x; p < P.nlong; p += DS_WAVES * 64) lp_s[p] = reinterpret_cast<const int4*>(P.long_pairs)[p];
    for (int p = threadIdx.x; p < P.nshort; p += DS_WAVES * 64) sp_s[p] = (unsigned)P.short_pairs[2 * p] | ((unsigned)P.short_pairs[2 * p + 1] << 16);
    __syncthreads();  // the only workgroup barrier: from here on the waves are independent
  }

  const int xcc = (int)(__builtin_amdgcn_s_getreg(DS_GETREG_XCC_ID) & 7);
#ifdef DS_TIMING  // experiments: s_memtime ticks per phase of a run, summed per wave into counters[0].dphase[] (tools/describe_phases.py)
  unsigned dt_acc[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long dt_last = 0;
#define DS_T0() do { dt_last = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#define DS_T(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dt_acc[i] += (unsigned)(t_ - dt_last); dt_last = t_; } while (0)
#else
#define DS_T0() do { } while (0)
#define DS_T(i) do { } while (0)
#endif
  const int stride = G.L[0].stride, cols = G.L[0].w;
  const int img_bytes = (G.L[0].h - 1) * stride + cols;
  const unsigned inv20 = (1u << 20) / (unsigned)np + 1u;  // s / np == (s * inv20) >> 20 for s < 2048
  const int nwords = P.strings / 8;                        // 64-bit words of a descriptor
  const int4* tab4 = reinterpret_cast<const int4*>(P.tab);
  const double2* uv2 = reinterpret_cast<const double2*>(P.uv);

  auto wave_sync = [&]() {  // LDS written by some lanes, read by others of the same wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // parameters of sample s of the run (clamped to the run's last sample: surplus lanes repeat it and are masked at the
  // loads)
  auto fetch = [&](const uint4* krec, int total, int s, bool rotated) {
    DsLane L;
    const int sc = min(s, total - 1);
    const int kq = (int)(((unsigned)sc * inv20) >> 20);
    const int pt = sc - kq * np;
    const uint4 rec = krec[kq];
    const int theta = rotated ? kth[kq] : 0;
    L.kx = __uint_as_float(rec.x); L.ky = __uint_as_float(rec.y);
    L.ti = (int)(rec.w & 0xFF) * np + pt;
    L.tab = tab4[L.ti];
    L.uv = uv2[theta * np + pt];
    L.slot = sc;
    return L;
  };
  // one sampling pass over the run: one sample per lane and round, lane = (keypoint, pattern point); the parameters of
  // the next round are requested before the current round's gathers.  (Measured and dropped: all rounds of a pass in
  // flight together - parameters of every round, then every round's gathers, then the combines: 234 VGPRs, no faster;
  // dedicated sampler waves fed by helper waves through LDS: profiles/r03_describe_roles_experiment.txt.)
  // (first: the first round's parameters if they were requested at the end of the previous run; by value - a reference
  // or a captured variable keeps the record on the stack)
  // side: the run's first pass also takes the ticket of the run after the next one and requests the next run's records,
  // directly in front of its first burst of gathers (take_fn, nsrc, ncnt; returned).  The vector memory counter is in
  // order: anywhere earlier, the first wait behind them - for the first round's parameters - would wait for the atomic's
  // microseconds too; here they travel with the gathers, whose wait is a full one anyway.
  auto pass = [&](const DsFrame& F, const uint4* krec, int total, bool rotated, bool use_first, DsLane first, bool side,
                  auto&& take_fn, const uint4* nsrc, int ncnt) {
    DsSide out;
    out.t = 0; out.rec = make_uint4(0, 0, 0, 0);
    DsLane Lc = first;
    if (!use_first) Lc = fetch(krec, total, lane, rotated);
    for (int s0 = 0; s0 < total; s0 += 64) {
      const bool valid = s0 + lane < total;
      DsLane Ln = Lc;
      if (s0 + 64 < total) Ln = fetch(krec, total, s0 + 64 + lane, rotated);
      const double mm = (double)__int_as_float(Lc.tab.x);
      const float sigma = __int_as_float(Lc.tab.y);
      const float xf = (float)(mm * Lc.uv.x) + Lc.kx, yf = (float)(mm * Lc.uv.y) + Lc.ky;
      int value;
      if (GENERIC) {
        BriskSamplePoint sp;
        sp.x = (float)(mm * Lc.uv.x); sp.y = (float)(mm * Lc.uv.y); sp.sigma = sigma;
        sp.scaling = P.scaling[2 * Lc.ti]; sp.scaling2 = P.scaling[2 * Lc.ti + 1];
        value = valid ? brisk_smoothed_intensity(F.img, stride, cols, F.integ, istride, Lc.kx, Lc.ky, sp) : 0;
      } else {
        const DsPrep pr = ds_prep(xf, yf, sigma, Lc.tab.z, Lc.tab.w);
        if (side && s0 == 0) {
          out.t = take_fn();
          if (lane < ncnt) out.rec = nsrc[lane];
        }
#ifdef DS_NO_PAIRS  // (A / B builds: one sample's ten gathers per lane everywhere)
        constexpr bool pairs = false;
#else
        // (the 3-byte image only: with 4-byte elements - dense frames - the two column pairs of a row share a line less often and
        // the exchange costs more than it saves: 64 frames at threshold 30 4.36 -> 4.34 k frames/s, 100 000 keypoints 0.263 -> 0.269 ms)
        constexpr bool pairs = I24;
#endif
        if (!pairs) {
          DsRaw raw;
          __builtin_amdgcn_s_setprio(1);  // a wave that has its gathers to issue goes first (1 % of the kernel)
          if (valid) ds_load<I24>(raw, pr, F.rs_img, stride, cols, F.rs_int, istride);
          __builtin_amdgcn_s_setprio(0);
          value = ds_combine<I24>(pr, raw);
        } else {
          DsRawPair raw;
          const bool odd = lane & 1;
          __builtin_amdgcn_s_setprio(1);
          ds_load_pair<I24>(raw, pr, odd, s0 + (lane & ~1) < total, valid, F.rs_img, stride, cols, F.rs_int, istride);
          __builtin_amdgcn_s_setprio(0);
          value = ds_combine_pair<I24>(pr, raw, odd);
        }
      }
      if (valid) vals[Lc.slot] = value;
      Lc = Ln;
    }
    if (GENERIC && side) {  // (the generic sampler has no burst: here)
      out.t = take_fn();
      if (lane < ncnt) out.rec = nsrc[lane];
    }
    wave_sync();
    return out;
  };
  auto frame_of = [&](int frame) {
    DsFrame F;
    F.img = brisk_layer_img(G, pyr, frame, 0);
    F.integ = integral + (long)frame * iframe_elems;
    // timing experiments (debug bits 29 / 30): a descriptor without records drops every gather through it
    F.rs_img = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(F.img), 0, (BRISK_DBG_FLAGS(G) & (1 << 30)) ? 0 : img_bytes, 0x00020000);
    F.rs_int = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(F.integ), 0, (BRISK_DBG_FLAGS(G) & (1 << 29)) ? 0 : (int)(iframe_elems * 4), 0x00020000);
    return F;
  };

  // Work queue: the frames are dealt to `ngroups` groups (frame % ngroups; 8 = one group per XCD), each group is ONE
  // queue of runs - its frames from the last to the first (the integral images are written in frame order just before
  // this kernel: the last ones are what the Infinity Cache still holds), each frame's keypoints in processing order -
  // behind one ticket counter (in the counters of the group's first frame).  A wave serves the group of its own XCD
  // first (speed only), then the others (so that the result never depends on where the hardware places a workgroup).
  const int ngroups = nframes < 8 ? 1 : (nframes <= 8 * DS_MAXQ ? 8 : (nframes + DS_MAXQ - 1) / DS_MAXQ);
  for (int gi = 0; gi < ngroups; ++gi) {
    const int g = (xcc + gi) % ngroups;
    if (g >= nframes) continue;
    const int m = (nframes - 1 - g) / ngroups + 1;  // frames of the group; queue entry j is frame g + (m - 1 - j) * ngroups
    __builtin_amdgcn_wave_barrier();
    int carry = 0;
    for (int j0 = 0; j0 < m; j0 += 64) {
      const int j = j0 + lane;
      const int nkp = j < m ? counters[g + (m - 1 - j) * ngroups].ndesc : 0;
      // keypoints per ticket of this frame (RUN is the kernel's maximum): sparse frames small runs - the stretch of
      // keypoints an XCD has in flight must stay local (3 at BASELINE config 2's ~480 keypoints per megapixel) -, dense
      // frames larger ones (fuller sampling rounds, fewer tickets: 30 k keypoints per frame 6.6 -> 5.8 ms per 64 frames),
      // and never fewer tickets than there are waves to take them (one frame per call: 1 keypoint per ticket)
      int runf = run_fixed;
      if (runf == 0) {
        const long density = (long)nkp * (1 << 20) / ((long)cols * G.L[0].h);  // keypoints per megapixel
        runf = density < 1500 ? 3 : density < 6000 ? 4 : 8;  // (3, not 2, since round 5: 198 of 256 sampling lanes instead of 132 of 192, 2.60 -> 2.56 ms per 512 frames)
        const int waves_serving = (int)(gridDim.x * DS_WAVES) / ngroups;
        runf = min(runf, max(1, nkp / max(waves_serving, 1)));
      }
      runf = min(max(runf, 1), RUN);
      int r = (nkp + runf - 1) / runf;
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(r, off, 64);
        if (lane >= off) r += t;
      }
      if (j < m) { cum[j] = carry + r; cnt_q[j] = nkp | (runf << 24); }
      carry += __shfl(r, 63, 64);
    }
    wave_sync();
    const int total_runs = __builtin_amdgcn_readfirstlane(carry);
    int* ticket = &counters[g].desc_ticket;
    int qj = 0;  // queue entry of the last decoded ticket: a wave's tickets only grow
    auto decode = [&](int tk) {
      DsTicket T;
      T.ok = tk < total_runs;
      T.frame = 0; T.k0 = 0; T.cnt = 0; T.total = 0;
      if (T.ok) {
        while (tk >= cum[qj]) ++qj;
        qj = __builtin_amdgcn_readfirstlane(qj);
        T.frame = g + (m - 1 - qj) * ngroups;
        const int run = __builtin_amdgcn_readfirstlane(tk - (qj ? cum[qj - 1] : 0));
        const int cq = __builtin_amdgcn_readfirstlane(cnt_q[qj]);
        const int runf = cq >> 24;
        T.k0 = run * runf;
        T.cnt = min(runf, (cq & 0xFFFFFF) - T.k0);
        T.total = T.cnt * np;
      }
      return T;
    };
    // Fewer than 8 frames (one queue for the whole chip; typically one frame per call, where latency matters): the runs
    // are dealt statically, wave w takes runs w, w + waves, ... - no atomics.  Otherwise lane 0 takes tickets from the
    // group's counter.  An agent-scope atomic takes microseconds, so a ticket is taken two runs ahead and - round 5 - READ
    // where the wave has just waited for its last gathers anyway: the vector memory counter is in order, whatever was
    // issued before those gathers has arrived with them.  Until then the compiler had the wave wait for every ticket the
    // moment it was taken (its atomic optimizer turns a one-lane atomic on a uniform address into mbcnt / readfirstlane
    // code that needs the result at once; the pointer is made opaque below so that it keeps its hands off) and for the next
    // run's records - requested two instructions earlier - at the top of every run (the LDS write of THIS run's records
    // sat behind them: a full wait).  Now the top of a run waits for nothing: the records of the next run go to the second
    // LDS buffer and the first sampling round of its orientation pass is requested at the END of this run, under the bit
    // tests and the descriptor stores.
    const bool static_deal = ngroups == 1;
    int static_next = (int)(blockIdx.x * DS_WAVES + wave);
    typedef __attribute__((address_space(1))) int* ds_gptr;  // (global address space: a flat atomic would count as an LDS access as well)
    ds_gptr ticket_opaque = (ds_gptr)ticket;
    asm volatile("" : "+v"(ticket_opaque));  // (not provably uniform any more: no atomic optimizer)
    auto take = [&]() {
      int t = total_runs;
      if (static_deal) {
        t = min(static_next, total_runs);
        static_next += (int)(gridDim.x * DS_WAVES);
      } else if (lane == 0) {
        t = __hip_atomic_fetch_add(ticket_opaque, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return t;
    };
    // groups of other XCDs - normally finished by their own waves - get a plain look first
    if (gi > 0 && __builtin_amdgcn_readfirstlane(__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= total_runs) continue;
    const int t1 = take();
    const int t2 = take();
    DsTicket cur = decode(__builtin_amdgcn_readfirstlane(t1));
    DsTicket nxt = decode(__builtin_amdgcn_readfirstlane(t2));
    int kb = 0;  // record buffer of the current run
    uint4 myrec = make_uint4(0, 0, 0, 0);
    if (lane < cur.cnt) myrec = (drec + (long)cur.frame * kp_cap)[cur.k0 + lane];
    if (lane < cur.cnt) krec2[lane] = myrec;
    bool have_first = false;  // (wave-uniform) `first` holds the first round of this run's orientation pass
    DsLane first;
    first.kx = 0.f; first.ky = 0.f; first.tab = make_int4(0, 0, 0, 0); first.ti = 0; first.uv = make_double2(0., 0.); first.slot = 0;
    DS_T0();
    while (cur.ok) {
      uint4* krec = krec2 + kb * RUN;
      const uint4* nsrc = drec + (long)nxt.frame * kp_cap + nxt.k0;
      const DsFrame F = frame_of(cur.frame);
      const bool estimate = P.rotation_invariant && lane < cur.cnt && __uint_as_float(myrec.z) == -1.0f;
      int theta = 0;
      DsSide side;
      side.t = 0; side.rec = make_uint4(0, 0, 0, 0);
      const bool two_passes = __any(estimate);
      DS_T(0);
      if (two_passes) {
        // orientation (:714-739): unrotated pattern, long pairs
        wave_sync();
        side = pass(F, krec, cur.total, false, have_first, first, true, take, nsrc, nxt.cnt);
        DS_T(1);
        int md0 = 0, md1 = 0;
        for (int kq = 0; kq < cur.cnt; ++kq) {
          const int* v = vals + kq * np;
          int d0 = 0, d1 = 0;
          if (REGTAB) {
            int va[DS_REG_LONG], vb[DS_REG_LONG];
#pragma unroll
            for (int t = 0; t < DS_REG_LONG; ++t) { va[t] = v[lp_ij[t] & 0xFF]; vb[t] = v[(lp_ij[t] >> 8) & 0xFF]; }
#pragma unroll
            for (int t = 0; t < DS_REG_LONG; ++t) {
              const int delta_t = va[t] - vb[t];
              d0 += delta_t * (int)(short)(lp_w[t] & 0xFFFFu) / 1024;
              d1 += delta_t * ((int)lp_w[t] >> 16) / 1024;
            }
          } else if (lp_in_lds) {
            for (int p = lane; p < P.nlong; p += 64) {
              const int4 q = lp_s[p];
              const int delta_t = v[q.x] - v[q.y];
              d0 += delta_t * q.z / 1024;
              d1 += delta_t * q.w / 1024;
            }
          } else {
            for (int p = lane; p < P.nlong; p += 64) {
              int a, b;
              brisk_long_pair(v, P.long_pairs + 4 * p, &a, &b);
              d0 += a;
              d1 += b;
            }
          }
          // wave sums (integers: order independent) by DPP row shifts / broadcasts - six dependent LDS permutes per sum
          // (__shfl_xor) cost ~0.25 us per keypoint of a wave's time
          d0 = ds_wave_sum(d0);
          d1 = ds_wave_sum(d1);
          if (lane == kq) { md0 = d0; md1 = d1; }
        }
        if (estimate) {  // one fp64 atan2 for the whole run
          const float ang = brisk_angle_from_direction(md0, md1);
          dkp[(long)cur.frame * kp_cap + (int)(myrec.w >> 8)].angle = ang;
          theta = brisk_theta_from_angle(ang, true);
        }
        __builtin_amdgcn_wave_barrier();  // every lane is done with vals[] of the orientation pass
      }
      if (P.rotation_invariant && !estimate) theta = brisk_theta_from_angle(__uint_as_float(myrec.z), false);
      if (lane < cur.cnt) kth[lane] = theta;
      wave_sync();
      DS_T(2);
      {
        const DsSide s2 = pass(F, krec, cur.total, true, false, first, !two_passes, take, nsrc, nxt.cnt);
        if (!two_passes) side = s2;
      }
      DS_T(3);
      const int t3 = side.t;
      const uint4 nrec = side.rec;
      // The wave has just waited for the rotated pass's last gathers: the ticket taken at the top of the run and the next
      // run's records have arrived with them.  Decode the ticket and put the records into the other buffer.
      const DsTicket nn = decode(__builtin_amdgcn_readfirstlane(t3));
      uint4* krec_n = krec2 + (kb ^ 1) * RUN;
      if (lane < nxt.cnt) krec_n[lane] = nrec;
      const bool est_n = P.rotation_invariant && lane < nxt.cnt && __uint_as_float(nrec.z) == -1.0f;
      have_first = nxt.ok && __any(est_n);
      // bit p = values[i] > values[j], LSB first in little-endian u32 words (:538-564)
      for (int kq = 0; kq < cur.cnt; ++kq) {
        const int* v = vals + kq * np;
        const int k = (int)((unsigned)__builtin_amdgcn_readfirstlane((int)krec[kq].w) >> 8);  // (index < 2^23: brisk_hip_set_capacity)
        unsigned long long mine = 0;
        if (REGTAB) {
          int va[DS_REG_SHORT], vb[DS_REG_SHORT];
#pragma unroll
          for (int w = 0; w < DS_REG_SHORT; ++w) { va[w] = v[sp_ij[w] & 0xFF]; vb[w] = v[(sp_ij[w] >> 16) & 0xFF]; }
#pragma unroll
          for (int w = 0; w < DS_REG_SHORT; ++w) {
            const unsigned long long mk = __ballot(sp_ij[w] != 0xFFFFFFFFu && va[w] > vb[w]);
            if (lane == w) mine = mk;
          }
        } else {
          for (int w = 0; w < nwords; ++w) {
            const int p = w * 64 + lane;
            bool bit = false;
            if (p < P.nshort) { const unsigned q = sp_s[p]; bit = v[q & 0xFFFF] > v[q >> 16]; }
            const unsigned long long mk = __ballot(bit);
            if (lane == w) mine = mk;
          }
        }
        if (lane < nwords)
          *reinterpret_cast<unsigned long long*>(desc + ((long)cur.frame * kp_cap + k) * desc_pitch + lane * 8) = mine;
      }
      // (behind the loop with the descriptor stores: in front of it the compiler drains the memory counter at the loop's
      // entry; what hides this request is the top of the next run)
      if (have_first) {
        wave_sync();
        first = fetch(krec_n, nxt.total, lane, false);
      }
      __builtin_amdgcn_wave_barrier();  // vals[], kth[] and this run's records are reused
      cur = nxt;
      nxt = nn;
      myrec = nrec;
      kb ^= 1;
      DS_T(4);
#ifdef DS_TIMING
      dt_acc[5] += 1;
#endif
    }
  }
#ifdef DS_TIMING
  if (lane == 0)
    for (int k = 0; k < 6; ++k) atomicAdd(&counters[0].dphase[k], (int)(dt_acc[k] >> (k < 5 ? 4 : 0)));  // (16-tick units: 512 frames overflow an int otherwise)
#endif
}

long brisk_dp_work_ints(int kp_cap) { return DP_W_BLK + (kp_cap + DP_THREADS - 1) / DP_THREADS + 16; }

static size_t describe_lds_bytes(const BriskPatternDev& P, int run, bool regtab) {
  const size_t table = regtab ? 0 : (P.nlong <= DS_LP_LDS ? (size_t)P.nlong * 16 : 0) + (((size_t)P.nshort * 4 + 15) & ~(size_t)15);
  const size_t per_wave = 2 * (size_t)run * 16 + (((size_t)run * 4 + 15) & ~(size_t)15) + (((size_t)run * P.npoints * 4 + 15) & ~(size_t)15) + 2 * DS_MAXQ * 4;
  return table + DS_WAVES * per_wave;
}

typedef void (*ds_kernel_t)(BriskGeom, BriskPatternDev, const uint8_t*, const uint32_t*, int, long, BriskFrameCounters*, BriskKeyPoint*, uint4*,
                            uint8_t*, int, int, int, int);

void brisk_launch_describe(const BriskGeom& G, const BriskPatternDev& P, const BriskDetectBuffers& B,
                           const BriskDescribeBuffers& Dd, int nframes, const BriskKeyPoint* kp_in, const int* n_in,
                           long n_in_stride, hipStream_t s, BriskProfiler* prof, const BriskOverlap* ov, int n_in_max) {
  brisk_prof_mark(prof, BRISK_STG_INTEGRAL, s);
  if (!ov && !((BRISK_DBG_FLAGS(G) & (1 << 19)) && (BRISK_DBG_FLAGS(G) & (1 << 27))))  // (bits 19 + 27: brisk_capi.hip, timing experiments)
    brisk_launch_integral(G, B.pyr, B.bandsum, Dd.integral, Dd.istride, Dd.iframe_elems, B.band_h, nframes, s, Dd.ibits, B.counters);
  brisk_prof_mark(prof, BRISK_STG_DESC_PREPARE, s);
  hipLaunchKernelGGL(k_desc_prepare, dim3(nframes), dim3(DP_THREADS), 0, s, G, P, kp_in, n_in, n_in_stride, B.counters, Dd.dkp,
                     Dd.dscale, Dd.dperm, Dd.drec, B.kp_cap, Dd.dp_work, Dd.dp_work_stride);
  // frames with more keypoints than the one-workgroup kernel takes (the others exit at once; a host call knows its
  // count - n_in_max - and does not launch them for nothing: three launches are 15 us of a single frame's latency)
  if (B.kp_cap > DP_SMALL_N && (n_in_max < 0 || n_in_max > DP_SMALL_N)) {
    DpTiles T;
    T.w = G.L[0].w; T.h = G.L[0].h; T.shift = 6;
    while ((((T.w - 1) >> T.shift) + 1) * (((T.h - 1) >> T.shift) + 1) > DP_MAXBUCKETS) ++T.shift;
    T.tiles_x = ((T.w - 1) >> T.shift) + 1;
    const int nbuckets = T.tiles_x * (((T.h - 1) >> T.shift) + 1);
    // workgroups per frame: what the capacity needs at most, but no more than fill the chip twice over (they loop)
    const int dp_blocks = min((B.kp_cap + DP_THREADS - 1) / DP_THREADS, max(1, 512 / nframes));
    const dim3 grid(dp_blocks, nframes);
    hipLaunchKernelGGL(k_dp_count, grid, dim3(DP_THREADS), 0, s, G, P, T, kp_in, n_in, n_in_stride, B.kp_cap, Dd.dp_work, Dd.dp_work_stride);
    hipLaunchKernelGGL(k_dp_scan, dim3(nframes), dim3(DP_THREADS), 0, s, n_in, n_in_stride, B.kp_cap, nbuckets, B.counters, Dd.dp_work,
                       Dd.dp_work_stride);
    hipLaunchKernelGGL(k_dp_scatter, grid, dim3(DP_THREADS), 0, s, G, P, T, kp_in, n_in, n_in_stride, B.kp_cap, Dd.dp_work,
                       Dd.dp_work_stride, Dd.dkp, Dd.drec);
  }
  // with `ov` the integral image is already running beside the detector's tail (brisk_launch_detect) and
  // k_desc_prepare (one workgroup per frame, does not read it): join only in front of the sampling kernel
  if (ov) (void)hipStreamWaitEvent(s, ov->join, 0);
  brisk_prof_mark(prof, BRISK_STG_DESCRIBE, s);
  {
    // Persistent grid: `wpc` waves per CU (the dynamic LDS request keeps the hardware from placing more workgroups on a
    // CU than that), runs of `run` keypoints per ticket.  Test / tuning knobs: debug bits 8-11 = run, 12-15 = workgroups
    // per CU.  Few frames: single keypoints per ticket (latency), many: runs of 4 (fewer, fuller sampling rounds).
    const int ncu = brisk_device_cus();
    // keypoints per ticket: chosen per frame inside the kernel (from the frame's keypoint density) up to DS_MAXRUN;
    // debug bits 8-11 fix it (tuning experiments)
    int run_fixed = (BRISK_DBG_FLAGS(G) >> 8) & 0xF;
    if (P.has_bilinear) run_fixed = 1;
    if (run_fixed > DS_MAXRUN) run_fixed = DS_MAXRUN;
    int bpc = (BRISK_DBG_FLAGS(G) >> 12) & 0xF;
    if (!bpc) bpc = DS_BLOCKS_PER_CU;
    const bool regtab = P.reg_tables && !P.has_bilinear;
    const int run_max = P.has_bilinear ? 1 : DS_MAXRUN;
    size_t lds = describe_lds_bytes(P, run_max, regtab);
    const size_t lds_limit = 160 * 1024 / (size_t)(bpc + 1) + 512;  // bpc + 1 workgroups of this size do not fit a CU
    if (lds < lds_limit && !(BRISK_DBG_FLAGS(G) & (1 << 28))) lds = lds_limit;  // bit 28: no padding (other kernels share the CUs)
    const bool i24 = Dd.ibits == 24;  // (decided by the caller from BriskPatternDev::int24_ok: never with the bilinear branch)
    ds_kernel_t fn = regtab ? (i24 ? k_describe<DS_MAXRUN, false, true, true> : k_describe<DS_MAXRUN, false, true, false>)
                            : (i24 ? k_describe<DS_MAXRUN, false, false, true> : k_describe<DS_MAXRUN, false, false, false>);
    if (P.has_bilinear) fn = k_describe<1, true, false, false>;
    (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fn, dim3(ncu * bpc), dim3(DS_WAVES * 64), lds, s, G, P, B.pyr, Dd.integral, Dd.istride, Dd.iframe_elems,
                       B.counters, Dd.dkp, Dd.drec, Dd.desc, B.kp_cap, Dd.desc_pitch, nframes, run_fixed);
  }
  brisk_prof_mark(prof, BRISK_STG_DESCRIBE + 1, s);
}
