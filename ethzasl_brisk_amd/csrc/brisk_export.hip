// brisk_export.hip - the batch path's exit to HOST memory (brisk_hip_batch_download_all, brisk_capi.hip).
//
// The reference hands a call's results to the caller's std::vector<cv::KeyPoint> / descriptor cv::Mat
// (brisk/src/brisk-feature-detector.cc:77-85, brisk-descriptor-extractor.cc:601-604).  A batch call leaves them in HBM in
// per-frame slots of `kp_cap` rows; these three kernels turn the slots of ALL frames into exact, prefix-summed rows in
// host memory without a host round trip for the sizes:
//   k_export_offsets  one workgroup: per-frame counts / flags and the exclusive prefix sums of the rows (frames that do
//                     not fit the caller's row capacity are cut, together with every frame behind them)
//   k_export_rows     keypoints and descriptor rows of every frame -> a packed device slab at those offsets (on the
//                     batch's stream: the next batch may overwrite the slots as soon as this is done)
//   k_export_egress   slab -> host memory, the exact bytes only, written by the device over the link (on the context's
//                     egress stream, beside the next batch's kernels and H2D copies)
#include <hip/hip_runtime.h>

#include "brisk_common.h"
#include "brisk_kernels.h"

#define EX_THREADS 1024

__global__ void __launch_bounds__(EX_THREADS) k_export_offsets(const BriskFrameCounters* __restrict__ counters, int nframes, int which,
                                                               long long rows_cap, int cut_flag, int* __restrict__ counts,
                                                               int* __restrict__ flags, long long* __restrict__ offsets) {
  __shared__ long long part[EX_THREADS / 64];
  __shared__ long long red[EX_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (nframes + EX_THREADS - 1) / EX_THREADS;
  const int f0 = min(tid * per, nframes), f1 = min(f0 + per, nframes);
  auto want = [&](int f) -> long long {
    const BriskFrameCounters& c = counters[f];
    return c.overflow ? 0 : (long long)(which ? c.ndesc : c.nkp);
  };
  long long sum = 0;
  for (int f = f0; f < f1; ++f) sum += want(f);
  // exclusive scan of the threads' sums: inside the wave, then over the 16 wave totals
  long long incl = sum;
  for (int off = 1; off < 64; off <<= 1) {
    const long long v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) part[wave] = incl;
  __syncthreads();
  long long base = incl - sum, total = 0;
  for (int q = 0; q < EX_THREADS / 64; ++q) {
    const long long v = part[q];
    base += q < wave ? v : 0;
    total += v;
  }
  // the first frame that does not fit: rows are stored up to its prefix (every frame behind it is cut as well: prefixes grow)
  long long stop = total;
  {
    long long p = base;
    for (int f = f0; f < f1; ++f) {
      const long long w = want(f);
      if (w > 0 && p + w > rows_cap) stop = min(stop, p);
      p += w;
    }
  }
  for (int off = 32; off > 0; off >>= 1) stop = min(stop, (long long)__shfl_xor(stop, off, 64));
  if (lane == 0) red[wave] = stop;
  __syncthreads();
  for (int q = 0; q < EX_THREADS / 64; ++q) stop = min(stop, red[q]);
  long long p = base;
  for (int f = f0; f < f1; ++f) {
    const BriskFrameCounters& c = counters[f];
    const long long w = want(f);
    const bool cut = w > 0 && p + w > rows_cap;
    counts[f] = which ? c.ndesc : c.nkp;
    flags[f] = c.overflow | (cut ? cut_flag : 0);
    offsets[f] = min(p, stop);
    p += w;
  }
  if (tid == 0) offsets[nframes] = stop;
}

// rows of frame blockIdx.y -> slab rows [offsets[f], offsets[f + 1]); descriptor rows `ddw` dwords wide (the caller's
// row stride), the bytes behind the descriptor zero
__global__ void __launch_bounds__(256) k_export_rows(const BriskKeyPoint* __restrict__ kps, const uint8_t* __restrict__ desc, int kp_cap,
                                                     int dev_pitch, int sdw, int ddw, const long long* __restrict__ offsets,
                                                     uint32_t* __restrict__ s_kps, uint32_t* __restrict__ s_desc) {
  const int f = blockIdx.y;
  const long long o = offsets[f];
  const int n = (int)(offsets[f + 1] - o);
  if (n <= 0) return;
  const int kw = (int)(sizeof(BriskKeyPoint) / 4);
  const uint32_t* src_k = reinterpret_cast<const uint32_t*>(kps + (long)f * kp_cap);
  uint32_t* dst_k = s_kps + o * kw;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * kw; i += gridDim.x * blockDim.x) dst_k[i] = src_k[i];
  if (!desc) return;
  uint32_t* dst_d = s_desc + o * ddw;
  const uint8_t* src_d = desc + (long)f * kp_cap * dev_pitch;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * ddw; i += gridDim.x * blockDim.x) {
    const int row = i / ddw, c = i - row * ddw;
    dst_d[i] = c < sdw ? *reinterpret_cast<const uint32_t*>(src_d + (long)row * dev_pitch + 4 * c) : 0u;
  }
}

// dwords [0, n) of src (16-byte aligned) -> dst (host memory, 4-byte aligned): 16-byte stores where dst allows them
__device__ __forceinline__ void ex_copy_words(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, long long n, long gt, long gn) {
  if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    const long long nv = n >> 2;
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    for (long long i = gt; i < nv; i += gn) d4[i] = s4[i];
    for (long long i = (nv << 2) + gt; i < n; i += gn) dst[i] = src[i];
  } else {
    for (long long i = gt; i < n; i += gn) dst[i] = src[i];
  }
}

__global__ void __launch_bounds__(256) k_export_egress(const int* __restrict__ s_counts, const int* __restrict__ s_flags,
                                                       const long long* __restrict__ s_offsets, const uint32_t* __restrict__ s_kps,
                                                       const uint32_t* __restrict__ s_desc, int nframes, int ddw, int* h_counts,
                                                       int* h_flags, long long* h_offsets, uint32_t* h_kps, uint32_t* h_desc) {
  const long gt = (long)blockIdx.x * blockDim.x + threadIdx.x, gn = (long)gridDim.x * blockDim.x;
  const long long rows = s_offsets[nframes];
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < nframes; i += blockDim.x) { h_counts[i] = s_counts[i]; h_flags[i] = s_flags[i]; }
    for (int i = threadIdx.x; i <= nframes; i += blockDim.x) h_offsets[i] = s_offsets[i];
  }
  ex_copy_words(h_kps, s_kps, rows * (long long)(sizeof(BriskKeyPoint) / 4), gt, gn);
  if (h_desc) ex_copy_words(h_desc, s_desc, rows * ddw, gt, gn);
}

void brisk_launch_export_pack(const BriskFrameCounters* counters, const BriskKeyPoint* kps, const uint8_t* desc, int kp_cap, int dev_pitch,
                              int strings, int nframes, int which, long long rows_cap, int desc_stride, int cut_flag,
                              const BriskExportSlab& S, hipStream_t s) {
  hipLaunchKernelGGL(k_export_offsets, dim3(1), dim3(EX_THREADS), 0, s, counters, nframes, which, rows_cap, cut_flag, S.counts, S.flags,
                     S.offsets);
  // workgroups per frame: a large batch fills the chip with one or two, a small one spreads its frames
  const int bx = nframes >= 256 ? 2 : nframes >= 32 ? 8 : 32;
  hipLaunchKernelGGL(k_export_rows, dim3(bx, nframes), dim3(256), 0, s, kps, desc, kp_cap, dev_pitch, strings / 4, desc_stride / 4, S.offsets,
                     S.kps, S.desc);
}

void brisk_launch_export_egress(const BriskExportSlab& S, int nframes, int desc_stride, int* h_counts, int* h_flags, long long* h_offsets,
                                void* h_kps, void* h_desc, hipStream_t s) {
  // the link bounds this kernel, not the chip: 48 workgroups keep ~200 KB of stores in flight and leave the CUs to the next
  // batch's kernels running beside it
  hipLaunchKernelGGL(k_export_egress, dim3(48), dim3(256), 0, s, S.counts, S.flags, S.offsets, S.kps, S.desc, nframes, desc_stride / 4,
                     h_counts, h_flags, h_offsets, static_cast<uint32_t*>(h_kps), static_cast<uint32_t*>(h_desc));
}
