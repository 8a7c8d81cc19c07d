// brisk_kernels.hip - hand-written HIP kernels (gfx950 / CDNA4) of the BRISK detect+describe engine.
//
// Kernel inventory (one launch handles a whole batch of frames; blockIdx.y or a decoded index = frame):
//   k_smap_clear        zeroes what the previous batch left in the score-state map (sparse)
//   k_pyramid_fused     layer-0 copy, band sums, half and two-third sampling chains from 96x96 blocks (HBM-bound)
//   k_pyramid_even      layer-0 copy + half sampling chain from 64x64 blocks (descriptor-only call: layer 0 alone)
//   k_pyramid_level     one level at a time (layers beyond the fused depth)
//   k_detect            threshold map + adaptive OAST 9_16 test, LDS-tiled                (VALU-bound)
//   k_score_blocks      lane-parallel scores around every candidate (own 5x5, 4x4 above / below)
//   k_classify_refine   IsMax2D steps 1-2 + 3-D refinement, one lane per candidate (+ _direct safety net)
//   k_tie_resolve       order-faithful replay of the lazy score cache for ties            (1 WG per frame and layer, latency-bound)
//   k_finalize          (layer, y, x) ordering + keypoint output
//   k_integral_final    exclusive 2-D prefix sum (u32) from the band column sums          (HBM-bound, side stream)
//   k_desc_prepare      scale index + border filter + stable compaction + processing order
//   k_describe          pattern sampling, orientation, 384/512 bit tests                  (dominant kernel; gather / L2-bound)
// (the Hamming matcher kernels live in brisk_match.hip)
// No MFMA: the path is byte/integer stencil + gather work.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "brisk_common.h"
#include "brisk_device_detect.h"
#include "brisk_kernels.h"

// tuning experiments only (BRISK_HIP_TUNING builds): integer knob from the environment; the release library has no
// environment knobs - every one of them is its default
#ifdef BRISK_HIP_TUNING
static int env_knob(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}
#else
#define env_knob(name, dflt) (dflt)
#endif


// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ BriskLayerView make_view(const BriskGeom& G, uint8_t* pyr, uint16_t* smap, int frame,
                                                    int l) {
  BriskLayerView v;
  const long base = (long)frame * G.pyr_elems + G.L[l].off;
  v.img = brisk_layer_img(G, pyr, frame, l);
  v.smap = smap + base;
  v.w = G.L[l].w;
  v.h = G.L[l].h;
  v.stride = G.L[l].stride;
  brisk_block_clear(&v.blk);
  brisk_block_clear(&v.blk58);
  v.miss = 0;
  return v;
}

// 16-bit field update inside the u32 word that holds it
__device__ __forceinline__ void smap_or(uint16_t* smap, long idx, unsigned bits) {
  unsigned* w = reinterpret_cast<unsigned*>(smap + (idx & ~1L));
  atomicOr(w, bits << ((idx & 1) ? 16 : 0));
}
__device__ __forceinline__ void smap_xor(uint16_t* smap, long idx, unsigned bits) {
  unsigned* w = reinterpret_cast<unsigned*>(smap + (idx & ~1L));
  atomicXor(w, bits << ((idx & 1) ? 16 : 0));
}
// L1-bypassing read (other waves of this launch update smap with atomics)
__device__ __forceinline__ unsigned smap_load_fresh(const uint16_t* smap, long idx) {
  const unsigned* w = reinterpret_cast<const unsigned*>(smap + (idx & ~1L));
  const unsigned v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (idx & 1) ? (v >> 16) : (v & 0xFFFFu);
}

// XCD-affine frame mapping for kernels whose blocks gather sparsely inside one frame: workgroups are dealt round-robin
// over the 8 XCDs, so with a 1-D grid all blocks with equal blockIdx.x % 8 share an XCD (and its L2).  Every frame gets
// `bpf` blocks on ONE residue: the lines its neighbouring candidates / keypoints share are then fetched into one L2
// instead of up to eight.  (With fewer than 8 frames the affinity would idle XCDs: plain mapping.)
__device__ __forceinline__ bool xcd_frame_block(int nframes, int bpf, int* frame, int* block_in_frame) {
  if (nframes >= 8) {
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    *frame = (jj / bpf) * 8 + xcd;
    *block_in_frame = jj % bpf;
  } else {
    *frame = blockIdx.x / bpf;
    *block_in_frame = blockIdx.x % bpf;
  }
  return *frame < nframes;
}
static inline int xcd_grid(int nframes, int bpf) { return nframes >= 8 ? (nframes + 7) / 8 * 8 * bpf : nframes * bpf; }

// ------------------------------------------------------------------------------------------------
// k_pyramid_level: builds destination layer `dl` from source layer `sl` (mode 0 half, 1 two-third).
// One thread produces 4 horizontally adjacent output pixels (one dword store).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pyramid_level(BriskGeom G, uint8_t* __restrict__ pyr, int sl, int dl,
                                                        int mode) {
  const int frame = blockIdx.y;
  const uint8_t* src = pyr + (long)frame * G.pyr_elems + G.L[sl].off;
  uint8_t* dst = pyr + (long)frame * G.pyr_elems + G.L[dl].off;
  const int sw = G.L[sl].w, sstride = G.L[sl].stride;
  const int dw = G.L[dl].w, dh = G.L[dl].h, dstride = G.L[dl].stride;
  const int words = dstride / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)words * dh; i += (long)gridDim.x * blockDim.x) {
    const int y = (int)(i / words), x = (int)(i % words) * 4;
    unsigned v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (x + k < dw) {
        const unsigned px = mode ? brisk_twothird_px(src, sstride, sw, x + k, y) : brisk_half_px(src, sstride, sw, x + k, y);
        v |= px << (8 * k);
      }
    }
    *reinterpret_cast<unsigned*>(dst + (long)y * dstride + x) = v;
  }
}

// ------------------------------------------------------------------------------------------------
// Fused pyramid kernels.  Halfsample8 output (c, r) depends only on the source block (2c..2c+1, 2r..2r+1) and
// Twothirdsample8 output pairs only on a 3x3 source block, so whole chains of levels can be produced from one
// aligned source block without any halo:
//   k_pyramid_even : 64x64 block of the input frame -> layer 0 copy (aligned pyramid layout), L2 32x32, L4 16x16,
//                    L6 8x8; the intermediate levels stay in LDS.
//   k_pyramid_fused: 96x96 block of the input frame -> layer 0 copy, L2 / L4 / L6 and L1 64x64 (two-third), L3, L5, L7.
// Layers beyond the fused depth (more than 4 octaves) are produced by k_pyramid_level.
// The exact column classes of the reference depend on absolute output columns and source widths, which
// brisk_half_px / brisk_twothird_px receive unchanged (LDS tiles are addressed with absolute coordinates).
// ------------------------------------------------------------------------------------------------
struct TileRef {  // view of an LDS tile that covers image pixels [x0, x0+tw) x [y0, y0+th)
  const uint8_t* p;
  int x0, y0, tw;
};
// pointer such that ptr[r * stride + c] addresses absolute pixel (c, r) of the tiled layer
__device__ __forceinline__ const uint8_t* tile_origin(const TileRef& t) { return t.p - (long)t.y0 * t.tw - t.x0; }

// one level of a fused chain: a (tw x tw) tile at absolute origin (ox, oy) of the destination layer from the LDS tile
// `so` (addressed with absolute source coordinates, row pitch stw) of a source layer of width sw.  Four horizontally
// adjacent outputs per thread: one dword store to the pyramid (and to the next LDS tile) instead of four byte stores;
// tile origins are multiples of 8, layer offsets / strides multiples of 64.
// rounded-up average of four byte pairs at once: (a + b + 1) >> 1 per byte = (a | b) - ((a ^ b) >> 1)
__device__ __forceinline__ unsigned swar_avg(unsigned a, unsigned b) { return (a | b) - (((a ^ b) >> 1) & 0x7F7F7F7Fu); }

template <bool TWOTHIRD>
__device__ __forceinline__ void pyramid_tile_level(const uint8_t* so, int stw, int sw, uint8_t* __restrict__ dst, int dstride,
                                                   int dw, int dh, int ox, int oy, int tw, uint8_t* dl) {
  const int qw = tw >> 2;
  // output columns below `fast_cols` belong to the reference's SIMD class of nested rounded averages (Halfsample8:
  // the 32-column double blocks, image-down-sampling.cc:308-330; Twothirdsample8: the 15 -> 10 column blocks,
  // :714-751): four outputs at a time on packed bytes.  The other column classes take the per-pixel functions.
  const int fast_cols = TWOTHIRD ? (sw / 15) * 10 : 16 * ((sw / 16) / 2);
  for (int i = threadIdx.x; i < tw * qw; i += 256) {
    const int r = i / qw, c = (i % qw) * 4;
    const int gx = ox + c, gy = oy + r;
    unsigned v = 0;
    if (gy < dh) {
      if (gx + 3 < fast_cols) {
        if (TWOTHIRD) {
          // source columns 3 (gx / 2) .. + 5 (2-byte aligned) of the outer row (A or C) and the middle row B
          // (the six bytes start at a multiple of 6: two ALIGNED dwords hold them, shifted by 0 or 2 bytes - reads at a
          // 2-byte alignment stall the LDS pipeline: SQ_LDS_UNALIGNED_STALL was 70 % of this kernel's LDS time)
          const int sx = 3 * (gx >> 1), sh = sx & 2;
          const unsigned* wa = reinterpret_cast<const unsigned*>(so + (long)(3 * (gy >> 1) + ((gy & 1) ? 2 : 0)) * stw + (sx & ~3));
          const unsigned* wb = reinterpret_cast<const unsigned*>(so + (long)(3 * (gy >> 1) + 1) * stw + (sx & ~3));
          const unsigned a0 = wa[0], a1 = wa[1], b0 = wb[0], b1 = wb[1];
          const unsigned a_lo = __builtin_amdgcn_alignbyte(a1, a0, sh), a_hi = (a1 >> (8 * sh)) & 0xFFFFu;
          const unsigned b_lo = __builtin_amdgcn_alignbyte(b1, b0, sh), b_hi = (b1 >> (8 * sh)) & 0xFFFFu;
          const unsigned u_lo = swar_avg(swar_avg(a_lo, b_lo), a_lo), u_hi = swar_avg(swar_avg(a_hi, b_hi), a_hi);  // u0..u3, u4 u5
          const unsigned X = __builtin_amdgcn_perm(u_hi, u_lo, 0x05030200u);  // u0 u2 u3 u5: the outer columns
          const unsigned M = __builtin_amdgcn_perm(u_hi, u_lo, 0x04040101u);  // u1 u1 u4 u4: the middle columns
          v = swar_avg(swar_avg(X, M), X);
        } else {
          const uint2 T = *reinterpret_cast<const uint2*>(so + (long)(2 * gy) * stw + 2 * gx);
          const uint2 B = *reinterpret_cast<const uint2*>(so + (long)(2 * gy + 1) * stw + 2 * gx);
          const unsigned V0 = swar_avg(T.x, B.x), V1 = swar_avg(T.y, B.y);         // vertical: v0..v3, v4..v7
          const unsigned E = __builtin_amdgcn_perm(V1, V0, 0x06040200u);           // v0 v2 v4 v6
          const unsigned O = __builtin_amdgcn_perm(V1, V0, 0x07050301u);           // v1 v3 v5 v7
          v = swar_avg(E, O);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (gx + q < dw)
            v |= (unsigned)(TWOTHIRD ? brisk_twothird_px(so, stw, sw, gx + q, gy) : brisk_half_px(so, stw, sw, gx + q, gy)) << (8 * q);
      }
      if (gx < dstride) *reinterpret_cast<unsigned*>(dst + (long)gy * dstride + gx) = v;
    }
    if (dl) *reinterpret_cast<unsigned*>(&dl[r * tw + c]) = v;
  }
}

__global__ void __launch_bounds__(256) k_pyramid_even(BriskGeom G, const uint8_t* __restrict__ frames, long frame_pitch,
                                                       int row_pitch, uint8_t* __restrict__ pyr, int nlevels, int tiles_x,
                                                       int tiles_y, uint32_t* __restrict__ bandsum, int istride) {
  __shared__ __attribute__((aligned(16))) uint8_t t0[64 * 64], t2[32 * 32], t4[16 * 16];
  __shared__ unsigned psum[4][64];
  const int frame = blockIdx.y;
  // XCD-aware tile order (see k_detect): x-adjacent 64-byte tile rows share 128-byte lines, keep them in one L2
  const int nt = tiles_x * tiles_y;
  const int res = blockIdx.x & 7;
  const int tsw = res * (nt >> 3) + min(res, nt & 7) + (blockIdx.x >> 3);
  const int bx = (tsw % tiles_x) * 64, by = (tsw / tiles_x) * 64;
  const int w = G.L[0].w, h = G.L[0].h;
  const uint8_t* src = frames + (long)frame * frame_pitch;
  uint8_t* P = pyr + (long)frame * G.pyr_elems;
  const bool aligned = ((row_pitch | (uintptr_t)src) & 3) == 0;
  // stage the 64x64 source block (zero outside the image) and write the layer-0 copy.  Interior blocks of 16-byte
  // aligned frames: one 16-byte load and store per thread (a row of the block = 4 lanes).  Otherwise four dword loads
  // per thread, issued back to back (unconditional, on a safe address when the dword is not fully inside the row).
  const bool wide = (((unsigned)row_pitch | (uintptr_t)src) & 15) == 0 && by + 64 <= h && bx + 64 <= w;
  if (wide) {
    const int r = threadIdx.x >> 2, c16 = (threadIdx.x & 3) * 16;
    const uint4 v = *reinterpret_cast<const uint4*>(src + (long)(by + r) * row_pitch + bx + c16);
    *reinterpret_cast<uint4*>(P + G.L[0].off + (long)(by + r) * G.L[0].stride + bx + c16) = v;
    *reinterpret_cast<uint4*>(&t0[r * 64 + c16]) = v;
  } else {
    const long safe_off = -(long)((uintptr_t)src & 3);
    unsigned stg[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = threadIdx.x + k * 256;
      const int r = i >> 4, c4 = (i & 15) * 4;
      const int gy = by + r, gx = bx + c4;
      const bool full = aligned && gy < h && gx + 3 < w;
      stg[k] = *reinterpret_cast<const unsigned*>(src + (full ? (long)gy * row_pitch + gx : safe_off));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = threadIdx.x + k * 256;
      const int r = i >> 4, c4 = (i & 15) * 4;
      const int gy = by + r, gx = bx + c4;
      const bool full = aligned && gy < h && gx + 3 < w;
      unsigned v = full ? stg[k] : 0u;
      if (gy < h) {
        if (!full)
          for (int q = 0; q < 4; ++q)
            if (gx + q < w) v |= (unsigned)src[(long)gy * row_pitch + gx + q] << (8 * q);
        if (gx < G.L[0].stride) *reinterpret_cast<unsigned*>(P + G.L[0].off + (long)gy * G.L[0].stride + gx) = v;
      }
      *reinterpret_cast<unsigned*>(&t0[r * 64 + c4]) = v;
    }
  }
  __syncthreads();
  // column sums of this 64-row band (the integral image kernel's carry rows): integral column = pixel column + 1
  {
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    unsigned sum = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += t0[(q * 16 + r) * 64 + c];
    psum[q][c] = sum;
    __syncthreads();
    if (threadIdx.x < 64) {
      const unsigned tot = psum[0][c] + psum[1][c] + psum[2][c] + psum[3][c];
      const int nb = tiles_y, band = by >> 6;
      uint32_t* row = bandsum + ((long)frame * nb + band) * istride;
      if (bx + c + 1 < istride) row[bx + c + 1] = tot;
      if (bx == 0 && c == 0) row[0] = 0;
    }
  }
  // successive halvings: level k has tile (64 >> k)^2 at origin (bx >> k, by >> k)
  TileRef srct = {t0, bx, by, 64};
  uint8_t* dst_lds[3] = {t2, t4, nullptr};
  int sw = w;
#pragma unroll
  for (int k = 1; k <= 3; ++k) {
    if (k > nlevels) break;
    const int l = 2 * k;
    const int tw = 64 >> k;
    const int ox = bx >> k, oy = by >> k;
    const int dw = G.L[l].w, dh = G.L[l].h, dstride = G.L[l].stride;
    uint8_t* dl = dst_lds[k - 1];
    pyramid_tile_level<false>(tile_origin(srct), srct.tw, sw, P + G.L[l].off, dstride, dw, dh, ox, oy, tw, dl);
    __syncthreads();
    srct.p = dl; srct.x0 = ox; srct.y0 = oy; srct.tw = tw;
    sw = dw;
  }
}

// ------------------------------------------------------------------------------------------------
// k_pyramid_fused: both chains from ONE read of the frame.  96x96 block of the input frame -> layer-0 copy (aligned
// pyramid layout), the column sums of the block's 96-row band (carry rows of the integral image kernel), L2 48x48,
// L4 24x24, L6 12x12 (half sampling) and L1 64x64 (two-third), L3 32x32, L5 16x16, L7 8x8.  Separate kernels
// for the two chains read layer 0 a second time for the odd chain (2.07 MB of the 8.3 MB per 1080p frame); they remain for the
// descriptor-only call (k_pyramid_even: layer 0 + 64-row band sums).
// ------------------------------------------------------------------------------------------------
#define PF_BAND 96
__global__ void __launch_bounds__(256) k_pyramid_fused(BriskGeom G, const uint8_t* __restrict__ frames, long frame_pitch,
                                                        int row_pitch, uint8_t* __restrict__ pyr, int even_levels,
                                                        int odd_levels, int tiles_x, int tiles_y,
                                                        uint32_t* __restrict__ bandsum, int istride) {
  __shared__ __attribute__((aligned(16))) uint8_t t0[96 * 96], t1[64 * 64], t2[48 * 48], t3[32 * 32], t4[24 * 24], t5[16 * 16];
  __shared__ uint2 psum[8][24];
  const int frame = blockIdx.y;
  // XCD-aware tile order (see k_detect): x-adjacent tile rows share 128-byte lines, keep them in one L2
  const int nt = tiles_x * tiles_y;
  const int res = blockIdx.x & 7;
  const int tsw = res * (nt >> 3) + min(res, nt & 7) + (blockIdx.x >> 3);
  const int bx = (tsw % tiles_x) * 96, by = (tsw / tiles_x) * 96;
  const int w = G.L[0].w, h = G.L[0].h, s0 = G.L[0].stride;
  const uint8_t* src = frames + (long)frame * frame_pitch;
  uint8_t* P = pyr + (long)frame * G.pyr_elems;
  uint8_t* L0 = P + G.L[0].off;
  // stage the 96x96 source block (zero outside the image) and write the layer-0 copy.  Interior blocks of 16-byte
  // aligned frames: 16-byte loads and stores (a block row = 6 lanes), issued back to back.  Otherwise nine dword
  // loads per thread (unconditional, on a safe address when the dword is not fully inside the row).
  const bool wide = (((unsigned)row_pitch | (uintptr_t)src) & 15) == 0 && by + 96 <= h && bx + 96 <= w;
  if (wide) {
    uint4 stg[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = min((int)threadIdx.x + k * 256, 96 * 6 - 1);
      stg[k] = *reinterpret_cast<const uint4*>(src + (long)(by + i / 6) * row_pitch + bx + (i % 6) * 16);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = threadIdx.x + k * 256;
      if (i < 96 * 6) {
        if (!G.l0_ext) *reinterpret_cast<uint4*>(L0 + (long)(by + i / 6) * s0 + bx + (i % 6) * 16) = stg[k];
        *reinterpret_cast<uint4*>(&t0[i * 16]) = stg[k];
      }
    }
  } else {
    const bool aligned = ((row_pitch | (uintptr_t)src) & 3) == 0;
    const long safe_off = -(long)((uintptr_t)src & 3);
    unsigned stg[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int i = threadIdx.x + k * 256;
      const int r = i / 24, c4 = (i % 24) * 4;
      const int gy = by + r, gx = bx + c4;
      const bool full = aligned && gy < h && gx + 3 < w;
      stg[k] = *reinterpret_cast<const unsigned*>(src + (full ? (long)gy * row_pitch + gx : safe_off));
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int i = threadIdx.x + k * 256;
      const int r = i / 24, c4 = (i % 24) * 4;
      const int gy = by + r, gx = bx + c4;
      const bool full = aligned && gy < h && gx + 3 < w;
      unsigned v = full ? stg[k] : 0u;
      if (gy < h) {
        if (!full)
          for (int q = 0; q < 4; ++q)
            if (gx + q < w) v |= (unsigned)src[(long)gy * row_pitch + gx + q] << (8 * q);
        if (gx < s0 && !G.l0_ext) *reinterpret_cast<unsigned*>(L0 + (long)gy * s0 + gx) = v;
      }
      *reinterpret_cast<unsigned*>(&t0[i * 4]) = v;
    }
  }
  __syncthreads();
  // column sums of this 96-row band (the integral image kernel's carry rows): integral column = pixel column + 1
  // (four columns per dword read, even / odd bytes summed in 16-bit halves: 96 rows x 255 < 65536)
  if (threadIdx.x < 192) {
    const int cq = threadIdx.x % 24, rg = threadIdx.x / 24;  // dword column, group of 12 rows
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int r = 0; r < 12; ++r) {
      const unsigned v = *reinterpret_cast<const unsigned*>(&t0[(rg * 12 + r) * 96 + cq * 4]);
      lo += v & 0x00FF00FFu;
      hi += (v >> 8) & 0x00FF00FFu;
    }
    psum[rg][cq] = make_uint2(lo, hi);
  }
  // first level of both chains (both read the frame block): L2 48x48 and L1 64x64
  {
    const TileRef st = {t0, bx, by, 96};
    if (even_levels >= 1)
      pyramid_tile_level<false>(tile_origin(st), 96, w, P + G.L[2].off, G.L[2].stride, G.L[2].w, G.L[2].h, bx >> 1, by >> 1, 48, t2);
    if (odd_levels >= 0)
      pyramid_tile_level<true>(tile_origin(st), 96, w, P + G.L[1].off, G.L[1].stride, G.L[1].w, G.L[1].h, bx / 3 * 2, by / 3 * 2, 64, t1);
  }
  __syncthreads();
  if (threadIdx.x < 96) {
    const int c = threadIdx.x, cq = c >> 2, k = c & 3;
    unsigned tot = 0;
#pragma unroll
    for (int rg = 0; rg < 8; ++rg) {
      const uint2 p = psum[rg][cq];
      const unsigned w = (k & 1) ? p.y : p.x;     // columns 1, 3 in the odd-byte sums
      tot += (k & 2) ? (w >> 16) : (w & 0xFFFFu);
    }
    uint32_t* row = bandsum + ((long)frame * tiles_y + by / PF_BAND) * istride;
    if (bx + c + 1 < istride) row[bx + c + 1] = tot;
    if (bx == 0 && c == 0) row[0] = 0;
  }
  // second level: L4 24x24 from L2, L3 32x32 from L1
  {
    const TileRef s2 = {t2, bx >> 1, by >> 1, 48}, s1 = {t1, bx / 3 * 2, by / 3 * 2, 64};
    if (even_levels >= 2)
      pyramid_tile_level<false>(tile_origin(s2), 48, G.L[2].w, P + G.L[4].off, G.L[4].stride, G.L[4].w, G.L[4].h, bx >> 2, by >> 2, 24, t4);
    if (odd_levels >= 1)
      pyramid_tile_level<false>(tile_origin(s1), 64, G.L[1].w, P + G.L[3].off, G.L[3].stride, G.L[3].w, G.L[3].h, bx / 3, by / 3, 32, t3);
  }
  __syncthreads();
  // third level: L6 12x12 from L4, L5 16x16 from L3
  {
    const TileRef s4 = {t4, bx >> 2, by >> 2, 24}, s3 = {t3, bx / 3, by / 3, 32};
    if (even_levels >= 3)
      pyramid_tile_level<false>(tile_origin(s4), 24, G.L[4].w, P + G.L[6].off, G.L[6].stride, G.L[6].w, G.L[6].h, bx >> 3, by >> 3, 12, nullptr);
    if (odd_levels >= 2)
      pyramid_tile_level<false>(tile_origin(s3), 32, G.L[3].w, P + G.L[5].off, G.L[5].stride, G.L[5].w, G.L[5].h, bx / 6, by / 6, 16, t5);
  }
  __syncthreads();
  if (odd_levels >= 3) {  // L7 8x8 from L5
    const TileRef s5 = {t5, bx / 6, by / 6, 16};
    pyramid_tile_level<false>(tile_origin(s5), 16, G.L[5].w, P + G.L[7].off, G.L[7].stride, G.L[7].w, G.L[7].h, bx / 12, by / 12, 8, nullptr);
  }
}

// ------------------------------------------------------------------------------------------------
// k_detect: per-pixel threshold map (37-px disc contrast, brisk-layer.cc:278-598) + contrast-adaptive OAST 9_16
// segment test (oast9-16.cc:79-100).
//
// Tile = 64x64 output pixels per 256-thread workgroup; the (64+8)x(64+6) u8 halo tile is staged in LDS with
// coalesced dword loads issued back to back (row pitch 80 bytes: the four row groups of a wave start 80 dwords
// apart, so the window reads of a wave are bank-conflict free).
// Phase A (all pixels, dense): the pre-gate brisk_pregate_pair - a necessary condition for a detection that needs
// only the centre and the four compass pixels of the ring - on two pixels per instruction (packed 16-bit lanes;
// byte pairs are pulled out of the window dwords with v_perm_b32).  Every thread owns 4 columns x 4 rows; the
// pixels that pass (about 1 %) are compacted into an LDS queue: per-thread bit mask, wave prefix sum of the
// counts (DPP), one LDS atomic per wave.
// Phase B (survivors only, one lane each): exact 37-pixel disc contrast -> adaptive threshold -> closed-form
// segment test on the ring (a subset of the disc pixels already in registers).  A detection writes its contrast
// score into the score-state map (all zero otherwise, see k_smap_clear) and appends a candidate to the frame's
// list (order is restored later from the (layer,y,x) key).
// grid.x enumerates the tiles of all layers, grid.y = frame.
// ------------------------------------------------------------------------------------------------
#define DT_W BRISK_DETECT_TILE_W
#define DT_H BRISK_DETECT_TILE_H
#define DT_R BRISK_DETECT_ROWS_PER_THREAD
#define DT_PITCH 80                                   // LDS row pitch (bytes)
#define DT_ROWDW ((DT_W + 8) / 4)                     // data dwords per tile row
#define DT_LH (DT_H + 6)
#define DT_WR (DT_R + 6)                              // window rows per thread
#define DT_NLD ((DT_LH * DT_ROWDW + 255) / 256)       // staging dwords per thread
static_assert(DT_H == 16 * DT_R && DT_W == 64, "256 threads = 16 column groups x 16 row groups");

// inclusive prefix sum over the 64 lanes of a wave (DPP row shifts + row broadcasts, no LDS)
__device__ __forceinline__ int wave_inclusive_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return v;
}

__global__ void __launch_bounds__(256) k_detect(BriskGeom G, BriskTileTable T, const uint8_t* __restrict__ pyr,
                                                 uint16_t* __restrict__ smap, BriskCand* __restrict__ cand,
                                                 BriskFrameCounters* __restrict__ counters, int cand_cap) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[DT_LH * DT_PITCH];
  __shared__ uint16_t queue[DT_H * DT_W];                              // tile-local pixel index py * 64 + px
  __shared__ int qcount;
  const int frame = blockIdx.y;
#ifdef DT_TIMING  // experiments: s_memtime ticks per phase and wave, summed into counters[frame].tdet[] (tools/detect_phases.py)
  unsigned long long dt_last = __builtin_amdgcn_s_memtime();
  unsigned dt_acc[7] = {0, 0, 0, 0, 0, 0, 0};
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define DT_T(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dt_acc[i] += (unsigned)(t_ - dt_last); dt_last = t_; } while (0)
#else
#define DT_T(i) do { } while (0)
#endif
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so blocks with equal blockIdx.x % 8
  // share an XCD (and its L2) within a frame.  Give each residue a contiguous eighth of the frame's tile list so
  // that the halo rows/columns shared by neighbouring tiles are fetched into one L2 instead of eight.
  // WHICH eighth an XCD takes rotates with the frame: the eighths differ in cost (layer 0's full tiles first, the small
  // layers' partial tiles last; texture), workgroups are dealt to the XCDs in strict rotation, and with the same eighth
  // on the same XCD frame after frame the XCD with the cheapest one idles while the dispatcher waits for the others -
  // 64 4K frames (3960 tiles: a multiple of 8, so nothing rotated by itself): 1.15 -> 0.78 ms.  gridDim.x is the tile count
  // rounded up to a multiple of 8, so blockIdx.x & 7 IS the XCD for every frame; the surplus workgroups exit.
  const int nt = T.total_tiles;
  const int res = ((blockIdx.x & 7) + frame) & 7;
  const int slot = blockIdx.x >> 3;
  if (slot >= (nt >> 3) + (res < (nt & 7) ? 1 : 0)) return;
  const int tid_sw = res * (nt >> 3) + min(res, nt & 7) + slot;
  int l = 0;
  while (l + 1 < G.nlayers && tid_sw >= T.first_tile[l + 1]) ++l;
  const int t = tid_sw - T.first_tile[l];
  const int tx = t % T.tiles_x[l], ty = t / T.tiles_x[l];
  const int w = G.L[l].w, h = G.L[l].h, stride = G.L[l].stride;
  const long base = (long)frame * G.pyr_elems + G.L[l].off;  // (score-state map)
  const uint8_t* img = brisk_layer_img(G, pyr, frame, l);
  const int x0 = tx * DT_W, y0 = ty * DT_H;
  const int thr = G.threshold;

  // stage halo tile: rows y0-3 .. y0+DT_H+2, columns x0-4 .. x0+DT_W+3 (dword granularity).  All loads of a thread
  // are issued before the first LDS write (one memory round trip per workgroup instead of one per load) on clamped
  // addresses.  What lands in the tile for positions outside the image is irrelevant: no valid centre (3 px inside
  // the border) reads it, and phase B rejects the invalid centres phase A may let through.
  {
    // thread = (dword column, row group): DT_ROWDW columns x DT_RG row groups (256 / 18 = 14: threads 252 .. 255 idle), rows
    // rg, rg + 14, ... - the column clamp and the LDS address are computed once, the row clamp is three operations per
    // dword (the staging was 100 VALU instructions per thread, a quarter of the kernel's, with a running (row, column)
    // pair per dword)
    constexpr int DT_RG = 256 / DT_ROWDW;
    static_assert(DT_RG * DT_NLD >= DT_LH, "the row groups cover the halo tile");
    unsigned stg[DT_NLD];
    const int col = threadIdx.x % DT_ROWDW, rg = threadIdx.x / DT_ROWDW;
    const uint8_t* colp = img + min(max(x0 - 4 + col * 4, 0), stride - 4);
#pragma unroll
    for (int k = 0; k < DT_NLD; ++k) {
      const int rr = min(rg + DT_RG * k, DT_LH - 1);  // (surplus rows reload the last one)
      const int cy = min(max(y0 - 3 + rr, 0), h - 1);
      stg[k] = *reinterpret_cast<const unsigned*>(colp + (long)cy * stride);
    }
    uint8_t* tp = &tile[rg * DT_PITCH + col * 4];
#pragma unroll
    for (int k = 0; k < DT_NLD; ++k)
      if (rg < DT_RG && rg + DT_RG * k < DT_LH) *reinterpret_cast<unsigned*>(tp + DT_RG * k * DT_PITCH) = stg[k];
  }
  if (threadIdx.x == 0) qcount = 0;
  DT_T(0);  // tile decode + staging (address arithmetic, global loads, LDS writes)
  __syncthreads();
  DT_T(1);  // waiting for the workgroup's other waves at the barrier

  // ---- phase A: 4 columns x DT_R rows per thread.  Window rows 0..DT_R+5 = image rows gy-3 .. gy+DT_R+2, three
  // dwords each = image columns gx-4 .. gx+7.  Pairs of horizontally adjacent pixels as packed 16-bit lanes:
  //   C*  the pixels themselves (they are centre for output row r-3, N for r, S for r-6),
  //   W*/E*  the pixels 3 to the left / right of them (centre rows only).
  const int lx = (threadIdx.x & 15) * 4, ly = (threadIdx.x >> 4) * DT_R;
  const int lane = threadIdx.x & 63;
  {
    const BriskPregate pg = brisk_pregate_make(thr, G.lower_threshold);
    unsigned R[DT_WR][3];
#pragma unroll
    for (int r = 0; r < DT_WR; ++r) {
      const unsigned* p = reinterpret_cast<const unsigned*>(&tile[(ly + r) * DT_PITCH + lx]);
      R[r][0] = p[0]; R[r][1] = p[1]; R[r][2] = p[2];
    }
    unsigned CA[DT_WR], CB[DT_WR];
#pragma unroll
    for (int r = 0; r < DT_WR; ++r) {
      CA[r] = __builtin_amdgcn_perm(0u, R[r][1], 0x0c010c00u);   // (col 0, col 1)
      CB[r] = __builtin_amdgcn_perm(0u, R[r][1], 0x0c030c02u);   // (col 2, col 3)
    }
    DT_T(2);  // window: 30 LDS dword reads + the centre / N / S byte pairs
    // pass flags of the thread's 16 pixels: every pair's two sign bits (15 and 31) are shifted in from the top, so
    // pair k = 2 * rr + (j >> 1) ends at bits 16 - 2 DT_R + k (column j even) and 32 - 2 DT_R + k (column j odd)
    unsigned m = 0;
#pragma unroll
    for (int rr = 0; rr < DT_R; ++rr) {
      const int ci = rr + 3;
      const unsigned WA = __builtin_amdgcn_perm(0u, R[ci][0], 0x0c020c01u);        // (col -3, col -2)
      const unsigned WB = __builtin_amdgcn_perm(R[ci][1], R[ci][0], 0x0c040c03u);  // (col -1, col 0)
      const unsigned EA = __builtin_amdgcn_perm(R[ci][2], R[ci][1], 0x0c040c03u);  // (col 3, col 4)
      const unsigned EB = __builtin_amdgcn_perm(0u, R[ci][2], 0x0c020c01u);        // (col 5, col 6)
      const unsigned gA = brisk_pregate_pair(CA[ci], CA[rr], CA[rr + 6], WA, EA, pg);
      const unsigned gB = brisk_pregate_pair(CB[ci], CB[rr], CB[rr + 6], WB, EB, pg);
      m = (m >> 1) | gA;
      m = (m >> 1) | gB;
    }
    DT_T(3);  // W / E byte pairs + the packed pre-gate on 16 pixels
    // compaction: wave prefix sum of the per-thread counts, one LDS atomic per wave
    const int cnt = __popc(m);
    const int incl = wave_inclusive_scan(cnt);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (total) {
      int qb = 0;
      if (lane == 0) qb = atomicAdd(&qcount, total);
      qb = __builtin_amdgcn_readfirstlane(qb);
      int pos = qb + incl - cnt;
      while (m) {
        const int bit = __ffs(m) - 1;
        m &= m - 1;
        const int k = (bit & 15) - (16 - 2 * DT_R), half = bit >> 4;
        queue[pos++] = (uint16_t)((ly + (k >> 1)) * DT_W + lx + ((k & 1) << 1) + half);
      }
    }
  }
  DT_T(4);  // compaction of the survivors
  __syncthreads();
  DT_T(5);  // barrier in front of phase B

  // ---- phase B: exact contrast + closed-form segment test on the survivors (one lane each)
  const int nq = qcount;
  const int cmp = (thr * G.lower_threshold) / 100;
  const float kthr = brisk_b2_factor(thr);
  for (int i = threadIdx.x; i < nq; i += 256) {
    const int idx = queue[i];
    const int py = idx / DT_W, px = idx % DT_W;
    const int gx = x0 + px, gyy = y0 + py;
    if (gx < 3 || gyy < 3 || gx > w - 4 || gyy > h - 4) continue;
    const uint8_t* p = &tile[(py + 3) * DT_PITCH + px + 4];
    // the 37 disc pixels, row by row (half widths 1, 2, 3, 3, 3, 2, 1)
    int v[7][7];
#pragma unroll
    for (int dy = -3; dy <= 3; ++dy) {
      const int hw = (dy == -3 || dy == 3) ? 1 : (dy == -2 || dy == 2) ? 2 : 3;
#pragma unroll
      for (int dx = -3; dx <= 3; ++dx) v[dy + 3][dx + 3] = (dx >= -hw && dx <= hw) ? (int)p[dy * DT_PITCH + dx] : -1;
    }
    int mx = 0, mn = 255;
#pragma unroll
    for (int dy = 0; dy < 7; ++dy)
#pragma unroll
      for (int dx = 0; dx < 7; ++dx)
        if (v[dy][dx] >= 0) { mx = max(mx, v[dy][dx]); mn = min(mn, v[dy][dx]); }
    const int tt = mx - mn;
    if (tt < cmp) continue;
    const int tc = min(max(tt, G.lower_threshold), BRISK_UPPER_THRESHOLD);
    const int b2 = brisk_b2_fast(tc, kthr);  // == (tc * thr) / 100 without quarter-rate integer multiplies
    const int c = v[3][3];
    if (mx - c <= b2 && c - mn <= b2) continue;  // no ring pixel can differ by more than b2
    // ring order of brisk_oast9_16_M (agast/include/agast/oast9-16.h:99-116), pixels i and i + 8 in one packed lane pair
    const uint32_t cc = (uint32_t)c * 0x10001u;
    uint32_t P[8];
    P[0] = brisk_pk_ring_pair(v[3][0], v[3][6], cc); P[1] = brisk_pk_ring_pair(v[2][0], v[4][6], cc);
    P[2] = brisk_pk_ring_pair(v[1][1], v[5][5], cc); P[3] = brisk_pk_ring_pair(v[0][2], v[6][4], cc);
    P[4] = brisk_pk_ring_pair(v[0][3], v[6][3], cc); P[5] = brisk_pk_ring_pair(v[0][4], v[6][2], cc);
    P[6] = brisk_pk_ring_pair(v[1][5], v[5][1], cc); P[7] = brisk_pk_ring_pair(v[2][6], v[4][0], cc);
    if (brisk_oast9_16_M_from_pk(P) > b2) {
      const int D = tt;
      // the score-state map is all zero between batches (k_smap_clear): only detections are written
      smap[base + (long)gyy * stride + gx] = (uint16_t)D;
      if (D <= 2) counters[frame].low_score = 1;  // (thresholds below 20 only) the lazy cache would not treat this as cached: ordered path for the frame
      const int ci = atomicAdd(&counters[frame].ncand, 1);
      if (ci < cand_cap) {
        BriskCand cnd;
        cnd.x = (uint16_t)gx; cnd.y = (uint16_t)gyy; cnd.layer = (uint8_t)l; cnd.D = (uint8_t)D;
        cnd.status = 0; cnd.flags = 0; cnd.fp_x0 = 0; cnd.fp_y0 = 0; cnd.fp_mask = 0; cnd.pad = 0;
        cnd.kx = cnd.ky = cnd.ksize = cnd.kresp = 0.f;
        cnd.key = ((unsigned)l << 26) | ((unsigned)gyy << 13) | (unsigned)gx;
        cand[(long)frame * cand_cap + ci] = cnd;
      } else {
        atomicOr(&counters[frame].overflow, 1);
      }
    }
  }
  DT_T(6);  // phase B: exact contrast + segment test on the survivors
#ifdef DT_TIMING
  if ((threadIdx.x & 63) == 0 && ((blockIdx.x >> 3) & 15) == 3) {  // (one tile in 16 reports: 2 M atomics per launch took ten times the kernel)
    for (int k = 0; k < 7; ++k) atomicAdd(&counters[frame].tdet[k], (int)dt_acc[k]);
    atomicAdd(&counters[frame].tdet[7], 1);
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// k_smap_clear: restores the "all zero" state of the score-state map after a batch by visiting what the batch
// wrote: every candidate's own entry (D + state bits) and the 4x4 touch footprint it may have set on the layer
// above.  Runs at the start of the NEXT batch on the same buffers (so the maps of a finished batch stay readable).
// A frame whose candidate list overflowed (entries without a record) is cleared completely.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_smap_clear(BriskGeom G, uint16_t* __restrict__ smap, const BriskCand* __restrict__ cand,
                                                    const BriskFrameCounters* __restrict__ counters, int cand_cap) {
  const int frame = blockIdx.y;
  uint16_t* fs = smap + (long)frame * G.pyr_elems;
  const int ncand = counters[frame].ncand;
  if (ncand > cand_cap || (counters[frame].overflow & 1) || counters[frame].full_clear) {
    uint4* p = reinterpret_cast<uint4*>(fs);  // pyr_elems is a multiple of 256 elements
    const long n16 = (long)G.pyr_elems * 2 / 16;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) p[i] = make_uint4(0, 0, 0, 0);
    return;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ncand; i += gridDim.x * blockDim.x) {
    const BriskCand c = cand[(long)frame * cand_cap + i];
    const int l = c.layer;
    fs[G.L[l].off + (long)c.y * G.L[l].stride + c.x] = 0;
    if (c.fp_mask && l + 1 < G.nlayers) {
      uint16_t* a = fs + G.L[l + 1].off;
      const int st = G.L[l + 1].stride;
      for (int b = 0; b < 16; ++b)
        if (c.fp_mask & (1u << b)) a[(long)(c.fp_y0 + (b >> 2)) * st + c.fp_x0 + (b & 3)] = 0;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_score_blocks: lane-parallel evaluation of every candidate's score blocks - own 5x5 (bytes 0-24), 4x4 on the
// layer above (bytes 32-47), 4x4 on the layer below or, on layer 0, the AGAST 5_8 3x3 (bytes 48-63) - one pixel per lane (16 ring loads + closed-form
// segment test), 64 bytes per candidate.  A wave takes SB_PER_WAVE candidates, unrolled so that their loads
// overlap (the kernel is latency-bound otherwise).
// ------------------------------------------------------------------------------------------------
#define SB_WAVES 4
#ifndef SB_PER_WAVE
#define SB_PER_WAVE 3
#endif
// ring offsets (dx, dy): 9_16 in the order of brisk_oast9_16_M, 5_8 in the order of brisk_agast5_8_M
__device__ __forceinline__ constexpr int sb_dx16(int j) { constexpr int t[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3}; return t[j]; }
__device__ __forceinline__ constexpr int sb_dy16(int j) { constexpr int t[16] = {0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1}; return t[j]; }
__device__ __forceinline__ constexpr int sb_dx8(int j) { constexpr int t[8] = {-1, -1, 0, 1, 1, 1, 0, -1}; return t[j]; }
__device__ __forceinline__ constexpr int sb_dy8(int j) { constexpr int t[8] = {0, -1, -1, -1, 0, 1, 1, 1}; return t[j]; }

// image patches of one candidate in LDS: [0] own layer 11x11 around (x, y), [1] layer above 10x10 around the 4x4
// block, [2] layer below 10x10; rows of 16 bytes starting at a 4-byte aligned column (SB_PROWS x 4 dwords each)
#define SB_PROWS 11
#define SB_PSLOTS (3 * SB_PROWS * 4)  // (patch, row, dword) load slots per candidate

__global__ void __launch_bounds__(SB_WAVES * 64) k_score_blocks(BriskGeom G, const uint8_t* __restrict__ pyr,
                                                                const uint16_t* __restrict__ smap,
                                                                const BriskCand* __restrict__ cand,
                                                                const BriskFrameCounters* __restrict__ counters,
                                                                uint8_t* __restrict__ blocks, int cand_cap, int nframes, int bpf) {
  int frame, bx;
  if (!xcd_frame_block(nframes, bpf, &frame, &bx)) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (counters[frame].low_score) return;  // (the frame runs the ordered path)
  const int n = min(counters[frame].ncand, cand_cap);
  const uint8_t* fimg = pyr + (long)frame * G.pyr_elems;
  const uint8_t* img0 = G.l0_ext ? G.l0_ext + (long)frame * G.l0_pitch : nullptr;  // layer 0 read in place
  const uint16_t* fsm = smap + (long)frame * G.pyr_elems;
  // lane role: 0 own 5x5 (lanes 0-24), 1 4x4 above (32-47), 2 4x4 below or the 5_8 3x3 on layer 0 (48-63), 3 unused
  const int role = (lane < 25) ? 0 : (lane >= 32 && lane < 48) ? 1 : (lane >= 48) ? 2 : 3;
  __shared__ int4 lgeo[BRISK_MAX_LAYERS];
  __shared__ __attribute__((aligned(16))) uint8_t patch[SB_WAVES][SB_PER_WAVE][3][SB_PROWS][16];
  if (threadIdx.x < BRISK_MAX_LAYERS) lgeo[threadIdx.x] = make_int4(G.L[threadIdx.x].w, G.L[threadIdx.x].h, G.L[threadIdx.x].stride, G.L[threadIdx.x].off);
  __syncthreads();
#ifdef SB_TIMING
  int sb_acc[4] = {0, 0, 0, 0};
  long long sb_last = (long long)wall_clock64();
#define SB_T(i) { const long long now_ = (long long)wall_clock64(); sb_acc[i] += (int)(now_ - sb_last); sb_last = now_; }
#else
#define SB_T(i)
#endif
  for (int base = (bx * SB_WAVES + wave) * SB_PER_WAVE; base < n; base += bpf * SB_WAVES * SB_PER_WAVE) {
    // -- round trip 1: the candidate headers (x, y, layer) of the wave's SB_PER_WAVE candidates
    uint2 hdr[SB_PER_WAVE];
#pragma unroll
    for (int k = 0; k < SB_PER_WAVE; ++k)
      hdr[k] = *reinterpret_cast<const uint2*>(&cand[(long)frame * cand_cap + min(base + k, n - 1)]);
    // -- round trip 2: the three image patches of every candidate as coalesced dword loads (132 dwords per
    // candidate instead of 57 x 17 byte gathers), plus the smap entry of each lane's pixel.  Loads are
    // unconditional on clamped addresses: clamping only ever moves data that no in-image ring reads.
    unsigned pv[SB_PER_WAVE][3];
    unsigned smv[SB_PER_WAVE];
    int cofs[SB_PER_WAVE];   // byte offset of the lane's pixel inside its patch
    bool ok[SB_PER_WAVE], is8[SB_PER_WAVE];
#pragma unroll
    for (int k = 0; k < SB_PER_WAVE; ++k) {
      // (the header is the same in every lane: as scalars, the anchors and patch origins below are scalar arithmetic)
      const int hx = __builtin_amdgcn_readfirstlane((int)hdr[k].x);
      const int x = hx & 0xFFFF, y = (int)((unsigned)hx >> 16);
      const int l = __builtin_amdgcn_readfirstlane((int)(hdr[k].y & 0xFF));
      const bool has_above = !G.single_layer && (l + 1 < G.nlayers);
      const bool has_below = !G.single_layer && (l > 0);
      const bool has_58 = !G.single_layer && (l == 0);
      int ax = 0, ay = 0, bx = 0, by = 0;
      brisk_block_anchor(true, (l & 1) != 0, x, y, &ax, &ay);
      brisk_block_anchor(false, (l & 1) != 0, x, y, &bx, &by);
      // patch origins (top-left pixel each patch must contain) on the three layers
      const int ox0 = x - 5, oy0 = y - 5, ox1 = ax - 3, oy1 = ay - 3, ox2 = bx - 3, oy2 = by - 3;
      // (a) patch loads: load round t fetches patch t (own / above / below), lane = row * 4 + dword (44 of the 64 lanes).
      // The patch's layer, origin and base address are the same in every lane - scalars -; a lane only clamps its row and
      // column (with the slots dealt 64 at a time across the patches, which patch a lane loads differed between lanes: a
      // layer-geometry read from LDS, a pointer select and 64-bit address arithmetic per slot, a third of the kernel's
      // vector instructions - tools/score_block_phases.py, ISA between the clock reads)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int r = min(lane >> 2, SB_PROWS - 1), d = lane & 3;
        const int pl = (t == 0) ? l : (t == 1) ? (has_above ? l + 1 : l) : (has_below ? l - 1 : l);
        const int pw = G.L[pl].stride, ph = G.L[pl].h;
        const int pox = (t == 0) ? ox0 : (t == 1) ? ox1 : ox2;
        const int poy = (t == 0) ? oy0 : (t == 1) ? oy1 : oy2;
        const int gx = min(max((pox & ~3) + 4 * d, 0), pw - 4);
        const int gy = min(max(poy + r, 0), ph - 1);
        const uint8_t* limg = (pl == 0 && img0) ? img0 : fimg + G.L[pl].off;
        pv[k][t] = *reinterpret_cast<const unsigned*>(limg + (gy * pw + gx));
      }
      // (b) the lane's pixel
      int ll = l, px = 0, py = 0, which = 0;
      bool valid = false;
      is8[k] = false;
      if (role == 0) {
        px = x - 2 + lane % 5; py = y - 2 + lane / 5; valid = true;
      } else if (role == 1) {
        ll = has_above ? l + 1 : l; px = ax + (lane & 3); py = ay + ((lane - 32) >> 2); valid = has_above; which = 1;
      } else if (role == 2) {
        const int p = lane - 48;
        if (has_below) {
          ll = l - 1; px = bx + (p & 3); py = by + (p >> 2); valid = true; which = 2;
        } else if (has_58 && p < 9) {
          px = x - 1 + p % 3; py = y - 1 + p / 3; valid = true; is8[k] = true;
        }
      }
      const int4 gg = lgeo[ll];  // geometry of the lane's layer (LDS copy: no vector-memory round trip)
      const int w = gg.x, h = gg.y, st = gg.z, off = gg.w;
      const int bd = is8[k] ? 2 : 3;
      ok[k] = valid && px >= bd && py >= bd && px < w - bd && py < h - bd;
      smv[k] = fsm[(long)off + (ok[k] ? (long)py * st + px : 0)];
      const int pox = (which == 0) ? ox0 : (which == 1) ? ox1 : ox2;
      const int poy = (which == 0) ? oy0 : (which == 1) ? oy1 : oy2;
      cofs[k] = which * (SB_PROWS * 16) + (py - poy) * 16 + (px - (pox & ~3));
    }
    SB_T(0)
    // -- patches to LDS (the wave's own region: wave-level ordering only)
#pragma unroll
    for (int k = 0; k < SB_PER_WAVE; ++k)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        if (lane < SB_PROWS * 4) *reinterpret_cast<unsigned*>(&patch[wave][k][t][0][0] + lane * 4) = pv[k][t];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    SB_T(1)
    // -- evaluation: ring bytes from LDS, closed-form segment tests on registers
#pragma unroll
    for (int k = 0; k < SB_PER_WAVE; ++k) {
      const uint8_t* pc = &patch[wave][k][0][0][0] + (ok[k] ? cofs[k] : 5 * 16 + 8);
      const int c = pc[0];
      // every lane reads the 9_16 ring at fixed offsets (the lanes of the 5_8 block too: their ring lies inside the own-layer
      // patch, the result is not used) - a per-lane choice between the two rings cost a select and an add per read
      // ring differences (d[i], d[i + 8]) as signed 16-bit halves: the 16 arcs on packed lanes (brisk_oast9_16_M_from_pk)
      uint32_t P[8];
      const uint32_t cc = (uint32_t)c * 0x10001u;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t a = pc[sb_dx16(i) + sb_dy16(i) * 16], b = pc[sb_dx16(i + 8) + sb_dy16(i + 8) * 16];
        P[i] = brisk_pk_ring_pair(a, b, cc);
      }
      int v = brisk_Kp_from_M(brisk_oast9_16_M_from_pk(P));
      const int D = BRISK_SM_D(smv[k]);
      if (D > 2) v = D;
      if (__any(is8[k])) {  // (wave-uniform per candidate: only layer-0 candidates carry the 5_8 block)
        int d8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) d8[j] = (int)pc[sb_dx8(j) + sb_dy8(j) * 16] - c;
        const int v8 = brisk_Kp_from_M(brisk_agast5_8_M_from_d(d8));
        if (is8[k]) v = v8;
      }
      if (!ok[k]) v = 0;
      if (base + k < n) blocks[((long)frame * cand_cap + base + k) * 64 + lane] = (uint8_t)v;
    }
    __builtin_amdgcn_wave_barrier();  // the patches are overwritten by the wave's next candidates
    SB_T(2)
#ifdef SB_TIMING
    sb_acc[3] += 1;
#endif
  }
#ifdef SB_TIMING
  if (lane == 0) for (int i = 0; i < 4; ++i) atomicAdd(const_cast<int*>(&counters[frame].sphase[i]), sb_acc[i]);
#endif
}

// ------------------------------------------------------------------------------------------------
// k_classify_refine: IsMax2D steps 1-2 + 3-D refinement, one lane per candidate.  The lane pulls the
// candidate's 64 score bytes (k_score_blocks) into registers and runs the scalar logic on them: no memory
// access and no score evaluation code in this kernel.  A block miss (never observed) hands the candidate to
// k_classify_refine_direct.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_classify_refine(BriskGeom G, uint8_t* pyr, uint16_t* smap, BriskCand* cand,
                                                         BriskFrameCounters* counters, const uint8_t* blocks, int* tie_idx,
                                                         int cand_cap, int tie_cap, int nframes, int bpf) {
  int frame, bx;
  if (!xcd_frame_block(nframes, bpf, &frame, &bx)) return;
  if (counters[frame].low_score) return;  // (the frame runs the ordered path)
  const int n = min(counters[frame].ncand, cand_cap);
  for (int mine = bx * blockDim.x + threadIdx.x; mine < n; mine += bpf * blockDim.x) {
#ifdef CR_TIMING
    const long long cr_t0 = (long long)wall_clock64();
#endif
    BriskCand* c = &cand[(long)frame * cand_cap + mine];
    const int x = c->x, y = c->y, l = c->layer, D = c->D;
    const bool has_above = !G.single_layer && (l + 1 < G.nlayers);
    const bool has_below = !G.single_layer && (l > 0);
    const bool has_58 = !G.single_layer && (l == 0);
    BriskLayerView Lo = make_view(G, pyr, smap, frame, l);
    BriskLayerView La = make_view(G, pyr, smap, frame, has_above ? l + 1 : l);
    BriskLayerView Lb = make_view(G, pyr, smap, frame, has_below ? l - 1 : l);
    const uint4* blk = reinterpret_cast<const uint4*>(blocks + ((long)frame * cand_cap + mine) * 64);
    const uint4 o = blk[0], o2 = blk[1], a = blk[2], b = blk[3];
    {  // 3x3 centre of the own 5x5 block: bytes 6,7,8, 11,12,13, 16,17,18
      const unsigned w1 = o.y, w2 = o.z, w3 = o.w, w4 = o2.x;
      Lo.blk.w0 = ((w1 >> 16) & 0xFFFFu) | ((w2 & 0xFFu) << 16) | (w2 & 0xFF000000u);                 // b6 b7 b8 b11
      Lo.blk.w1 = (w3 & 0xFFFFu) | ((w4 & 0xFFFFu) << 16);                                              // b12 b13 b16 b17
      Lo.blk.w2 = (w4 >> 16) & 0xFFu;                                                                   // b18
      Lo.blk.w3 = 0;
      Lo.blk.x0 = x - 1; Lo.blk.y0 = y - 1; Lo.blk.cw = 3; Lo.blk.ch = 3;
    }
    if (has_58) { Lo.blk58.w0 = b.x; Lo.blk58.w1 = b.y; Lo.blk58.w2 = b.z; Lo.blk58.w3 = b.w; Lo.blk58.x0 = x - 1; Lo.blk58.y0 = y - 1; Lo.blk58.cw = 3; Lo.blk58.ch = 3; }
    if (has_above) {
      int ax, ay;
      brisk_block_anchor(true, (l & 1) != 0, x, y, &ax, &ay);
      La.blk.w0 = a.x; La.blk.w1 = a.y; La.blk.w2 = a.z; La.blk.w3 = a.w; La.blk.x0 = ax; La.blk.y0 = ay; La.blk.cw = 4; La.blk.ch = 4;
    }
    if (has_below) {
      int bx, by;
      brisk_block_anchor(false, (l & 1) != 0, x, y, &bx, &by);
      Lb.blk.w0 = b.x; Lb.blk.w1 = b.y; Lb.blk.w2 = b.z; Lb.blk.w3 = b.w; Lb.blk.x0 = bx; Lb.blk.y0 = by; Lb.blk.cw = 4; Lb.blk.ch = 4;
    }
    int nprobed = 0;
    unsigned flags = 0;
    BriskKeyPoint kp;
    kp.x = kp.y = kp.size = kp.response = 0.f;
    BriskTouch touch;
    touch.on = false; touch.mask = 0; touch.x0 = 0; touch.y0 = 0;
#ifdef CR_TIMING
    touch.tlast = cr_t0;
    for (int i = 0; i < 6; ++i) touch.tacc[i] = 0;
#endif
    bool e5 = false;
    const unsigned status = brisk_classify<false>(Lo, x, y, D, &nprobed);
    BRISK_CR_T(&touch, 0)
    if (status != BRISK_ST_REJ && brisk_refine<false>(G, Lb, Lo, La, l, x, y, &kp, &e5, &touch)) flags |= 1;
#ifdef CR_TIMING
    { const long long now_ = (long long)wall_clock64(); atomicAdd(&counters[frame].cphase[7], (int)(now_ - touch.tlast)); touch.tlast = now_; }  // (lanes that ended early: waiting for the others)
#endif
    if ((Lo.miss | La.miss | Lb.miss) || (BRISK_DBG_FLAGS(G) & 1)) {  // leave the candidate to k_classify_refine_direct
      c->status = 0xFF;
      atomicAdd(&counters[frame].nredo, 1);
      continue;
    }
    // the tie list's slot first: the only memory operation of the record whose result is needed - requested in front of the
    // stores and touches below, which it would otherwise queue behind (memory operations return in order; the write-out
    // was the longest phase of the kernel, tools/classify_phases.py)
    int tie_j = 0;
    if (status == BRISK_ST_TIE) tie_j = atomicAdd(&counters[frame].ntie[l], 1);
    unsigned bits = ((unsigned)nprobed << 8) | (status << 12);
    if (e5) { flags |= 2; bits |= BRISK_SM_E5; }
    if (flags & 1) { c->kx = kp.x; c->ky = kp.y; c->ksize = kp.size; c->kresp = kp.response; }
    c->fp_x0 = (int16_t)touch.x0; c->fp_y0 = (int16_t)touch.y0; c->fp_mask = (uint16_t)touch.mask;
    c->status = (uint8_t)status;
    c->flags = (uint8_t)flags;
    if (status == BRISK_ST_PASS && touch.mask) {  // event e3: score-touch the layer above
      for (int bb = 0; bb < 16; ++bb)
        if (touch.mask & (1u << bb)) smap_or(La.smap, (long)(touch.y0 + (bb >> 2)) * La.stride + touch.x0 + (bb & 3), BRISK_SM_TOUCH);
    }
    smap_or(Lo.smap, (long)y * Lo.stride + x, bits);
    if (status == BRISK_ST_TIE) {
      if (tie_j < tie_cap) tie_idx[((long)frame * BRISK_MAX_LAYERS + l) * tie_cap + tie_j] = mine;
      else atomicOr(&counters[frame].overflow, 2);
    }
#ifdef CR_TIMING
    BRISK_CR_T(&touch, 5)
    for (int i = 0; i < 6; ++i) atomicAdd(&counters[frame].cphase[i], touch.tacc[i]);
    atomicAdd(&counters[frame].cphase[6], 1);
#endif
  }
}

// ------------------------------------------------------------------------------------------------
// k_classify_refine_direct: safety net of k_classify_refine.  Handles the candidates whose score blocks
// did not cover an access (status 0xFF; none has ever been observed) with direct evaluation, one thread per
// candidate.  Exits immediately when the frame has no such candidate.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_classify_refine_direct(BriskGeom G, uint8_t* pyr, uint16_t* smap, BriskCand* cand,
                                                                BriskFrameCounters* counters, int* tie_idx, int cand_cap,
                                                                int tie_cap) {
  const int frame = blockIdx.y;
  if (counters[frame].nredo == 0 || counters[frame].low_score) return;
  const int n = min(counters[frame].ncand, cand_cap);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    BriskCand* c = &cand[(long)frame * cand_cap + i];
    if (c->status != 0xFF) continue;
    const int l = c->layer, x = c->x, y = c->y, D = c->D;
    const bool has_above = !G.single_layer && (l + 1 < G.nlayers);
    const bool has_below = !G.single_layer && (l > 0);
    const BriskLayerView Lo = make_view(G, pyr, smap, frame, l);
    const BriskLayerView La = make_view(G, pyr, smap, frame, has_above ? l + 1 : l);
    const BriskLayerView Lb = make_view(G, pyr, smap, frame, has_below ? l - 1 : l);
    int nprobed = 0;
    unsigned flags = 0;
    BriskKeyPoint kp;
    kp.x = kp.y = kp.size = kp.response = 0.f;
    BriskTouch touch;
    touch.on = false; touch.mask = 0; touch.x0 = 0; touch.y0 = 0;
    bool e5 = false;
    const unsigned status = brisk_classify<true>(Lo, x, y, D, &nprobed);
    if (status != BRISK_ST_REJ && brisk_refine<true>(G, Lb, Lo, La, l, x, y, &kp, &e5, &touch)) flags |= 1;
    unsigned bits = ((unsigned)nprobed << 8) | (status << 12);
    if (e5) { flags |= 2; bits |= BRISK_SM_E5; }
    if (flags & 1) { c->kx = kp.x; c->ky = kp.y; c->ksize = kp.size; c->kresp = kp.response; }
    c->fp_x0 = (int16_t)touch.x0; c->fp_y0 = (int16_t)touch.y0; c->fp_mask = (uint16_t)touch.mask;
    c->status = (uint8_t)status;
    c->flags = (uint8_t)flags;
    if (status == BRISK_ST_PASS && touch.mask) {
      for (int b = 0; b < 16; ++b)
        if (touch.mask & (1u << b)) smap_or(La.smap, (long)(touch.y0 + (b >> 2)) * La.stride + touch.x0 + (b & 3), BRISK_SM_TOUCH);
    }
    if (status == BRISK_ST_TIE) {
      const int j = atomicAdd(&counters[frame].ntie[l], 1);
      if (j < tie_cap) tie_idx[((long)frame * BRISK_MAX_LAYERS + l) * tie_cap + j] = i;
      else atomicOr(&counters[frame].overflow, 2);
    }
    smap_or(Lo.smap, (long)y * Lo.stride + x, bits);
  }
}

// ------------------------------------------------------------------------------------------------
// Ordered path (AGAST thresholds 1..19): the reference's GetKeypoints (brisk-scale-space.cc:92-287) walked in its own
// order on its own lazy score cache (low byte of the score-state map).  Below threshold 20 a detection can store a
// score <= 2, which the cache treats as "not cached" (brisk-layer.cc:118-132), and the history-free / history-
// dependent split of the fast path no longer holds; this path makes no such assumption, it is the sequential
// algorithm itself: slow (one lane per frame), bit-exact for every threshold, and used only where the fast path
// does not apply.
//   k_order_candidates     all candidates of a frame into (layer, y, x) order: counting sort over (layer, row)
//                          buckets in global scratch, rows sorted by x
//   k_ordered_keypoints    one lane per frame: literal IsMax2D + refinement per candidate, keypoints in order
// ------------------------------------------------------------------------------------------------
#define OC_THREADS 256
__global__ void __launch_bounds__(OC_THREADS) k_order_candidates(BriskGeom G, const BriskCand* cand, BriskFrameCounters* counters,
                                                                 unsigned* order_scratch, int* row_scratch, long row_scratch_stride,
                                                                 int cand_cap, int only_flagged) {
  const int frame = blockIdx.x, tid = threadIdx.x;
  if (only_flagged && !counters[frame].low_score) return;  // (thresholds below 20: only the frames that need the ordered path)
  const int n = min(counters[frame].ncand, cand_cap);
  const BriskCand* C = cand + (long)frame * cand_cap;
  unsigned* order = order_scratch + (long)frame * cand_cap * 2;  // [0, n): candidate indices in key order
  int* rows = row_scratch + (long)frame * row_scratch_stride;     // [0, R]: start of every (layer, row) bucket; [R+1, 2R+1): fill
  __shared__ int rbase[BRISK_MAX_LAYERS + 1];
  if (tid == 0) {
    int acc = 0;
    for (int l = 0; l < G.nlayers; ++l) { rbase[l] = acc; acc += G.L[l].h; }
    rbase[G.nlayers] = acc;
  }
  __syncthreads();
  const int R = rbase[G.nlayers];
  if (2 * R + 2 > row_scratch_stride) {  // (cannot happen with the capacities the host checks)
    if (tid == 0) atomicOr(&counters[frame].overflow, 8);
    return;
  }
  for (int r = tid; r < 2 * R + 2; r += OC_THREADS) rows[r] = 0;
  __threadfence_block();
  __syncthreads();
  for (int i = tid; i < n; i += OC_THREADS) atomicAdd(&rows[rbase[C[i].layer] + C[i].y + 1], 1);
  __threadfence_block();
  __syncthreads();
  if (tid == 0) {  // inclusive prefix over the row counts: rows[r] = first slot of bucket r (serial: a few thousand rows)
    int acc = 0;
    for (int r = 1; r <= R; ++r) { acc += rows[r]; rows[r] = acc; }
  }
  __threadfence_block();
  __syncthreads();
  int* fill = rows + R + 1;
  for (int i = tid; i < n; i += OC_THREADS) {
    const int b = rbase[C[i].layer] + C[i].y;
    order[rows[b] + atomicAdd(&fill[b], 1)] = (unsigned)i;
  }
  __threadfence_block();
  __syncthreads();
  for (int b = tid; b < R; b += OC_THREADS) {  // inside a row: insertion sort by x (rows hold a handful of candidates)
    const int lo = rows[b], hi = rows[b + 1];
    for (int i = lo + 1; i < hi; ++i) {
      const unsigned v = order[i];
      const int xv = C[v].x;
      int j = i - 1;
      while (j >= lo && C[order[j]].x > xv) { order[j + 1] = order[j]; --j; }
      order[j + 1] = v;
    }
  }
}

__global__ void __launch_bounds__(64) k_ordered_keypoints(BriskGeom G, uint8_t* pyr, uint16_t* smap, const BriskCand* cand,
                                                           BriskFrameCounters* counters, const unsigned* order_scratch,
                                                           BriskKeyPoint* kp_out, int cand_cap, int kp_cap, const uint8_t* mask,
                                                           long mask_pitch_frame, int mask_row_pitch, int no_scale_nms, int only_flagged) {
  if (threadIdx.x != 0) return;
  const int frame = blockIdx.x;
  if (only_flagged && !counters[frame].low_score) return;
  const int n = min(counters[frame].ncand, cand_cap);
  counters[frame].full_clear = 1;  // the cache is written wherever a score was asked for: the next batch clears the whole map
  if (counters[frame].overflow & 1) return;
  BriskOrderedOut out;
  out.kp = kp_out + (long)frame * kp_cap;
  out.cap = kp_cap;
  out.n = 0;
  out.mask = mask ? mask + (long)frame * mask_pitch_frame : nullptr;
  out.mask_row_pitch = mask_row_pitch;
  const bool undefined = brisk_ordered_walk(G, pyr + (long)frame * G.pyr_elems, smap + (long)frame * G.pyr_elems,
                                            cand + (long)frame * cand_cap, order_scratch + (long)frame * cand_cap * 2, n,
                                            no_scale_nms != 0, &out);
  if (undefined) atomicOr(&counters[frame].overflow, 16);
  counters[frame].nkp = min(out.n, kp_cap);
  if (out.n > kp_cap) atomicOr(&counters[frame].overflow, 4);
}

// ComputeScale (brisk-feature-detector.cc:87-92): provided keypoints, one frame.
// k_compute_scale is the reference's walk on one lane (brisk_compute_scale_walk: ~19 us per point and layer - 3 000 points
// on a 1080p frame took 0.45 s).  Round 5: the walk's phases are order-free among themselves (brisk_cs_* in
// brisk_device_detect.h), so they run one lane per (layer, provided point):
//   k_cs_admit   a workgroup per layer: which points the layer admits (stable compaction -> adm[layer][j]), their four
//                threshold-0 touches
//   k_cs_scores  GetAgastPoints on the provided lists (the score at the reference's linear offset; flags the inputs on which
//                the reference reads beyond the image)
//   k_cs_refine  the per-point refinement of GetKeypoints -> tmp[layer][j] = keypoint, valid
//   k_cs_emit    the valid ones in (layer, provided) order
// Benign race, stated: k_cs_admit and k_cs_refine let many lanes read-modify-write the same 16-bit cells of the score-state map
// (brisk_S_literal) without atomics.  Every such access is a threshold-1 access of the literal cache, and what a lane stores in
// a cell is a pure function of the IMAGE around the cell (the AGAST score of that pixel), never of the order of accesses: all
// racing writers store the same value, and a reader sees either "not cached" (and computes that same value itself) or the
// value.  Provided lists whose points share cells and neighbourhoods across layers - duplicates, a dense grid - are compared
// with the sequential algorithm (the oracle) in tests/test_gpu_round6.py.
// A layer that admits no provided point DETECTS instead (brisk-layer.cc:99-105) and its raster-ordered list feeds the same
// loop: that case - and a point count beyond the scratch buffers - stays on the one-lane walk (k_compute_scale runs when
// counters[0].pad[0] is set; the touches k_cs_admit already made are the ones the walk makes first, and are idempotent).
#define CS_THREADS 256
__global__ void __launch_bounds__(CS_THREADS) k_cs_admit(BriskGeom G, uint8_t* pyr, uint16_t* smap, const BriskKeyPoint* in, int n_in,
                                                          BriskFrameCounters* counters, int* adm) {
  __shared__ int wtot[CS_THREADS / 64];
  __shared__ int base;
  const int l = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int k0 = 0; k0 < n_in; k0 += CS_THREADS) {
    const int k = k0 + tid;
    float kx = 0.f, ky = 0.f;
    const bool ok = k < n_in && brisk_provided_on_layer(G, l, in[k], &kx, &ky);
    const unsigned long long bal = __ballot(ok);
    if (lane == 0) wtot[wave] = __popcll(bal);
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int q = 0; q < CS_THREADS / 64; ++q) { wbase += (q < wave) ? wtot[q] : 0; total += wtot[q]; }
    if (ok) {
      adm[(long)l * n_in + base + wbase + __popcll(bal & ((1ull << lane) - 1ull))] = k;
      brisk_cs_touch(G, pyr, smap, l, kx, ky);
    }
    __syncthreads();
    if (tid == 0) base += total;
    __syncthreads();
  }
  if (tid == 0) {
    counters[0].ntie[l] = base;  // (the tie lists are not used by this call: admitted points of layer l)
    counters[0].full_clear = 1;   // the cache is written wherever a score was asked for: the next batch clears the whole map
    if (base == 0) atomicOr(&counters[0].pad[0], 1);  // the layer detects instead: the one-lane walk takes the call
  }
}

__global__ void __launch_bounds__(CS_THREADS) k_cs_scores(BriskGeom G, uint8_t* pyr, uint16_t* smap, const BriskKeyPoint* in, int n_in,
                                                           BriskFrameCounters* counters, const int* adm) {
  if (counters[0].pad[0]) return;
  const int l = blockIdx.y, j = blockIdx.x * CS_THREADS + threadIdx.x;
  if (j >= counters[0].ntie[l]) return;
  float kx, ky;
  (void)brisk_provided_on_layer(G, l, in[adm[(long)l * n_in + j]], &kx, &ky);
  if (brisk_cs_score(G, pyr, smap, l, kx, ky)) atomicOr(&counters[0].overflow, 16);
}

__global__ void __launch_bounds__(64) k_cs_refine(BriskGeom G, uint8_t* pyr, uint16_t* smap, const BriskKeyPoint* in, int n_in, int suppress,
                                                   BriskFrameCounters* counters, const int* adm, unsigned* tmp) {
  if (counters[0].pad[0] || (counters[0].overflow & 16)) return;
  const int l = blockIdx.y, j = blockIdx.x * 64 + threadIdx.x;
  const int cnt = counters[0].ntie[l];
  const bool flat = !suppress && !G.single_layer;  // :131-170: layer l at the coordinates of layer 0's j-th entry
  if (flat && cnt > counters[0].ntie[0]) {          // agastPoints.at(0)[n] throws std::out_of_range
    if (j == 0) atomicOr(&counters[0].overflow, 16);
    return;
  }
  if (j >= cnt) return;
  BriskKeyPoint kp;
  bool valid = true;
  float kx, ky;
  if (flat) {
    const BriskKeyPoint src0 = in[adm[j]];
    (void)brisk_provided_on_layer(G, 0, src0, &kx, &ky);
    brisk_cs_flat(G, pyr, smap, l, src0, kx, ky, &kp);
  } else {
    const BriskKeyPoint src = in[adm[(long)l * n_in + j]];
    (void)brisk_provided_on_layer(G, l, src, &kx, &ky);
    valid = brisk_cs_refine(G, pyr, smap, l, src, kx, ky, &kp);
  }
  unsigned* t = tmp + ((long)l * n_in + j) * 8;
  t[7] = valid ? 1u : 0u;
  if (valid) {
    t[0] = __float_as_uint(kp.x); t[1] = __float_as_uint(kp.y); t[2] = __float_as_uint(kp.size); t[3] = __float_as_uint(kp.angle);
    t[4] = __float_as_uint(kp.response); t[5] = (unsigned)kp.octave; t[6] = (unsigned)kp.class_id;
  }
}

__global__ void __launch_bounds__(1024) k_cs_emit(BriskGeom G, BriskFrameCounters* counters, const unsigned* tmp, int n_in,
                                                   BriskKeyPoint* kp_out, int kp_cap) {
  if (counters[0].pad[0] || (counters[0].overflow & 16)) return;
  __shared__ int wtot[16];
  __shared__ int base;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int l = 0; l < G.nlayers; ++l) {
    const int cnt = counters[0].ntie[l];
    for (int j0 = 0; j0 < cnt; j0 += 1024) {
      const int j = j0 + tid;
      const unsigned* t = tmp + ((long)l * n_in + j) * 8;
      const bool ok = j < cnt && t[7] != 0;
      const unsigned long long bal = __ballot(ok);
      if (lane == 0) wtot[wave] = __popcll(bal);
      __syncthreads();
      int wbase = 0, total = 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) { wbase += (q < wave) ? wtot[q] : 0; total += wtot[q]; }
      const int pos = base + wbase + __popcll(bal & ((1ull << lane) - 1ull));
      if (ok && pos < kp_cap) {
        BriskKeyPoint kp;
        kp.x = __uint_as_float(t[0]); kp.y = __uint_as_float(t[1]); kp.size = __uint_as_float(t[2]); kp.angle = __uint_as_float(t[3]);
        kp.response = __uint_as_float(t[4]); kp.octave = (int)t[5]; kp.class_id = (int)t[6];
        kp_out[pos] = kp;
      }
      __syncthreads();
      if (tid == 0) base += total;
      __syncthreads();
    }
  }
  if (tid == 0) {
    counters[0].nkp = min(base, kp_cap);
    if (base > kp_cap) atomicOr(&counters[0].overflow, 4);
  }
}

// the one-lane walk: everything when `always`, else only the calls k_cs_admit handed over (a layer without admitted points)
__global__ void __launch_bounds__(64) k_compute_scale(BriskGeom G, uint8_t* pyr, uint16_t* smap, const BriskKeyPoint* in, int n_in,
                                                       int suppress, BriskFrameCounters* counters, unsigned* det_scratch,
                                                       int det_cap, BriskKeyPoint* kp_out, int kp_cap, int always) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (!always && !counters[0].pad[0]) return;
  counters[0].full_clear = 1;
  BriskOrderedOut out;
  out.kp = kp_out; out.cap = kp_cap; out.n = 0; out.mask = nullptr; out.mask_row_pitch = 0;
  bool cap_exceeded = false;
  const bool undefined = brisk_compute_scale_walk(G, pyr, smap, in, n_in, suppress != 0, det_scratch, det_cap, &out, &cap_exceeded);
  if (undefined) atomicOr(&counters[0].overflow, 16);
  if (cap_exceeded) atomicOr(&counters[0].overflow, 1);
  counters[0].nkp = min(out.n, kp_cap);
  if (out.n > kp_cap) atomicOr(&counters[0].overflow, 4);
}

// ------------------------------------------------------------------------------------------------
// k_tie_resolve: one workgroup per (frame, layer).  A layer's tie candidates are rank-sorted into raster order and
// dealt round-robin to the waves; a wave spins until every raster-earlier tie candidate within Chebyshev distance 4 of
// its candidate is decided (the earliest undecided candidate never waits, so the scheme cannot deadlock), then replays
// the lazy score cache with one lane per pixel (8 probe values + the 5x5 raw block).
//
// Layers of one frame run concurrently as a pipeline: layer l + 1 needs the e3 touches (4x4 footprints on its map) of
// the ties of layer l that pass.  The workgroup of layer l publishes its progress ("every tie above row Y is decided
// and its touches are performed") in counters[frame].tie_prog[l]; a wave of layer l + 1 waits for the rows of layer l
// whose footprints can reach its tie's 5x5 block.  Workgroups take their (frame, layer) from a ticket counter, so a
// workgroup only ever waits for one that started before it (layer l draws its ticket before layer l + 1 of the same
// frame): no assumption on the dispatch order of blockIdx, no deadlock.  All cross-workgroup data (map entries,
// progress) move through agent-scope atomics; all workgroups of a frame share blockIdx.x % 8, i.e. an XCD and its L2.
// Layers with more ties than the on-chip arrays hold are rank-sorted through global scratch and processed in chunks of
// TR_CHUNK consecutive ranks with the same on-chip scheme.
// ------------------------------------------------------------------------------------------------
#ifndef TR_WAVES
#define TR_WAVES 16
#endif
#define TR_THREADS (TR_WAVES * 64)
#define TR_WIN 9
#ifndef TR_CHUNK
#define TR_CHUNK 3072
#endif
#define TR_PROG_DONE 0x7FFFFFFF
#define TR_DONE_BIT 4
#define TR_BM_WORDS 512  // 16384 cells of 8x8 pixels (or 16x16, ... for larger layers)
#ifdef TR_TIMING  // experiments: per-phase time of the decision loop (10 ns units) summed into counters[frame].tphase[]
#define TR_T(i) { const long long now_ = (long long)wall_clock64(); tacc[i] += (int)(now_ - tlast); tlast = now_; }
#else
#define TR_T(i)
#endif
#ifdef TR_TIMELINE
#define TR_STAMP(c_, l_, k_) if (c_) counters[frame].tl[(l_) * 8 + (k_)] = (int)wall_clock64();
#else
#define TR_STAMP(c_, l_, k_)
#endif

// Raster order of a layer's ties beyond the on-chip capacity, in O(n) (ranking by counting smaller keys took n^2 / 1024
// steps per thread: 24 s for the 970 k ties of an all-tie 3645 x 2524 frame, 0.4 s now): ties counted per image row (LDS
// histogram, two 16-bit counters per word), rows prefix-summed, ties scattered into their row's bucket in global scratch,
// every bucket ordered by x (a wave per row, rank by counting - 64 keys per load, handed round by lane broadcasts; a row
// holds at most its width) and written back over the layer's list, which is then in raster order.  pool: 12288 words of
// LDS (4096 words of row counters + 8192 row offsets).  Not inlined: a cold path, and the tie kernel must stay at 104
// VGPRs at most (k_tie_resolve).
// (tie_sort_large: in brisk_tie_kernel.inc, one copy per kernel - a function with two callers is a real call whose
// register needs add to both kernels: 94 -> 108 VGPRs)

// k_tie_resolve (batches of 32 frames and more) must stay at 96 VGPRs at most: beside its 16 waves a CU must still have
// room for one 512-thread workgroup of the integral kernel - 2 waves x 56 allocated VGPRs per SIMD in the 24-bit form (51
// used) -, or the two stop overlapping (seen twice: 0.74 -> 0.84 ms for the window in round 3; 72.4 -> 70.2 k frames/s in
// round 4 when a variant of this kernel reached 99).  tie_sort_large is called, not inlined, at a point where little is
// live: 94 VGPRs.
// k_tie_resolve_small (fewer than 32 frames: nothing competes for the CUs) runs the static step of the cache replay with
// the lane's precomputed act / not-self masks (brisk_state_masks; TR_KERNEL_MASKS): 13 % fewer instructions per tie, 98
// VGPRs; one 640 x 480 frame: detect 294 -> 281 us, 1080p 415 -> 386 us.
#define TR_KERNEL_NAME k_tie_resolve
#define TR_SORT_NAME tie_sort_large
#define TR_KERNEL_MASKS 0
#define TR_KERNEL_PAIR 0
#include "brisk_tie_kernel.inc"
#undef TR_KERNEL_NAME
#undef TR_KERNEL_MASKS
#undef TR_SORT_NAME
#define TR_KERNEL_NAME k_tie_resolve_small
#define TR_SORT_NAME tie_sort_large_small
#define TR_KERNEL_MASKS 1
#include "brisk_tie_kernel.inc"
#undef TR_KERNEL_NAME
#undef TR_SORT_NAME
#undef TR_KERNEL_PAIR
// k_tie_resolve_pair (round 5): the small form with TWO ties per wave, one per half wave (brisk_tie_kernel.inc)
#define TR_KERNEL_NAME k_tie_resolve_pair
#define TR_SORT_NAME tie_sort_large_pair
#define TR_KERNEL_PAIR 1
#include "brisk_tie_kernel.inc"
#undef TR_KERNEL_NAME
#undef TR_KERNEL_PAIR
#undef TR_KERNEL_MASKS

// ------------------------------------------------------------------------------------------------
// k_finalize: keypoints of a frame in (layer, y, x) order.  One workgroup per frame; a key's rank is found through
// row buckets on chip (up to FN_SMALL keypoints; denser frames: k_finalize_large).
// ------------------------------------------------------------------------------------------------
#ifndef FN_THREADS
#define FN_THREADS 512
#endif
#define FN_BUCKETS 2048
#define FN_SMALL 3072   // up to this many keypoints the keys stay on chip (k_finalize); more: k_finalize_large
#define FN_SB 1024      // row buckets of k_finalize's on-chip ordering (a multiple of FN_THREADS)
__global__ void __launch_bounds__(FN_THREADS) k_finalize(BriskGeom G, const BriskCand* cand, BriskFrameCounters* counters,
                                                          unsigned* keys_scratch, BriskKeyPoint* kp_out, int cand_cap,
                                                          int kp_cap, const uint8_t* mask, long mask_pitch_frame,
                                                          int mask_row_pitch) {
  __shared__ int nvalid;
  __shared__ unsigned skey[FN_SMALL];
  __shared__ unsigned sidx[FN_SMALL];
  BRISK_CHAIN_SETPRIO();
  const int frame = blockIdx.x, tid = threadIdx.x;
  if (counters[frame].low_score) return;  // (the frame runs the ordered path, which writes its keypoints itself)
  const int n = min(counters[frame].ncand, cand_cap);
  const BriskCand* C = cand + (long)frame * cand_cap;
  unsigned* keys = keys_scratch + (long)frame * cand_cap * 2;  // [key][cand index] (k_finalize_large reads them)
  if (tid == 0) nvalid = 0;
  __syncthreads();
  // four candidates per thread and round: their status words and keys are requested together (one memory round trip per
  // round instead of two per candidate - the kernel runs beside the integral kernel, which keeps the memory system busy)
  constexpr int FN_U = 4;
  for (int i0 = tid; i0 < n; i0 += FN_U * FN_THREADS) {
    unsigned sw[FN_U], ck[FN_U];
#pragma unroll
    for (int u = 0; u < FN_U; ++u) {
      const int i = min(i0 + u * FN_THREADS, n - 1);
      sw[u] = *reinterpret_cast<const unsigned*>(reinterpret_cast<const uint8_t*>(&C[i]) + 4);  // layer, D, status, flags
      ck[u] = C[i].key;
    }
#pragma unroll
    for (int u = 0; u < FN_U; ++u) {
      const int i = i0 + u * FN_THREADS;
      bool valid = i < n && ((sw[u] >> 16) & 0xFFu) == BRISK_ST_PASS && ((sw[u] >> 24) & 1u);
      if (valid && mask) {  // RemoveInvalidKeyPoints (brisk-feature-detector.cc:49-66)
        const BriskCand& c = C[i];
        const uint8_t* m = mask + (long)frame * mask_pitch_frame;
        valid = m[(long)(int)(c.ky + 0.5f) * mask_row_pitch + (int)(c.kx + 0.5f)] != 0;
      }
      if (valid) {
        const int j = atomicAdd(&nvalid, 1);
        keys[2 * j] = ck[u];
        keys[2 * j + 1] = (unsigned)i;
        if (j < FN_SMALL) { skey[j] = ck[u]; sidx[j] = (unsigned)i; }
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  const int nv = nvalid;
  if (nv > FN_SMALL) {  // dense frame: the O(n) ordering of k_finalize_large takes over
    if (tid == 0) counters[frame].nvalid_large = nv;
    return;
  }
  // Rank of a key = keys of smaller buckets + smaller keys of its own bucket.  A bucket is FN_ROWS_PER_BUCKET consecutive
  // rows of the pyramid (layers stacked: the key is (layer, y, x)) - a handful of keypoints -, its members hang on a linked
  // list in LDS.  Round 5: the kernel used to count ALL smaller keys per key, n^2 / 4 LDS reads - 25 of its 40 us for the
  // 1 200 keypoints of a 1080p frame, on the critical path of every one-frame call and of the window beside the integral
  // kernel in a batch.
  {
    __shared__ int rbase[BRISK_MAX_LAYERS + 1];
    __shared__ int bhead[FN_SB];       // last member of the bucket (-1: empty)
    __shared__ int bstart[FN_SB];      // keys in smaller buckets
    __shared__ int bnext[FN_SMALL];    // next member of the same bucket
    __shared__ int wsum[FN_THREADS / 64];
    if (tid == 0) {
      int acc = 0;
      for (int l = 0; l < G.nlayers; ++l) { rbase[l] = acc; acc += G.L[l].h; }
      rbase[G.nlayers] = acc;
    }
    for (int b = tid; b < FN_SB; b += FN_THREADS) { bhead[b] = -1; bstart[b] = 0; }
    __syncthreads();
    int shift = 0;
    while ((rbase[G.nlayers] >> shift) >= FN_SB) ++shift;
    auto bucket_of = [&](unsigned key) { return (rbase[key >> 26] + (int)((key >> 13) & 0x1FFF)) >> shift; };
    for (int j = tid; j < nv; j += FN_THREADS) {
      const int b = bucket_of(skey[j]);
      atomicAdd(&bstart[b], 1);
      bnext[j] = atomicExch(&bhead[b], j);
    }
    __syncthreads();
    {  // exclusive prefix over the buckets: FN_SB / FN_THREADS consecutive buckets per thread
      constexpr int PER = FN_SB / FN_THREADS;
      int loc[PER], sum = 0;
#pragma unroll
      for (int q = 0; q < PER; ++q) { loc[q] = bstart[tid * PER + q]; sum += loc[q]; }
      const int incl = wave_inclusive_scan(sum);
      const int lane = tid & 63, wave = tid >> 6;
      if (lane == 63) wsum[wave] = incl;
      __syncthreads();
      int woff = 0;
#pragma unroll
      for (int q = 0; q < FN_THREADS / 64; ++q) woff += (q < wave) ? wsum[q] : 0;
      int run = woff + incl - sum;
#pragma unroll
      for (int q = 0; q < PER; ++q) { bstart[tid * PER + q] = run; run += loc[q]; }
    }
    __syncthreads();
    for (int j = tid; j < nv; j += FN_THREADS) {
      const unsigned k = skey[j];
      const BriskCand& c = C[sidx[j]];
      const float4 r = *reinterpret_cast<const float4*>(&c.kx);  // (requested before the list walk)
      const int lyr = c.layer;
      const int b = bucket_of(k);
      int rank = bstart[b];
      for (int q = bhead[b]; q >= 0; q = bnext[q]) rank += (skey[q] < k) ? 1 : 0;
      if (rank < kp_cap) {
        BriskKeyPoint kp;
        kp.x = r.x; kp.y = r.y; kp.size = r.z; kp.angle = -1.0f; kp.response = r.w;
        kp.octave = G.single_layer ? 0 : lyr; kp.class_id = -1;
        kp_out[(long)frame * kp_cap + rank] = kp;
      }
    }
  }
  if (tid == 0) {
    counters[frame].nkp = min(nv, kp_cap);
    if (nv > kp_cap) atomicOr(&counters[frame].overflow, 4);
  }
}

// Ordering of dense frames (more than FN_SMALL keypoints) in O(n): the key is (layer, y, x), so a counting sort over
// (layer, row) buckets puts every keypoint into its row's range and the position inside the range is the number of
// smaller x of the same row (rows hold a handful of keypoints).  Exits at once for the other frames.
// scratch2: 2 * (number of valid candidates) words per frame (the tie lists are free by now).
__global__ void __launch_bounds__(FN_THREADS) k_finalize_large(BriskGeom G, const BriskCand* cand, BriskFrameCounters* counters,
                                                                const unsigned* keys_scratch, unsigned* scratch2,
                                                                long scratch2_stride, BriskKeyPoint* kp_out, int cand_cap,
                                                                int kp_cap) {
  const int frame = blockIdx.x, tid = threadIdx.x;
  const int nv = counters[frame].nvalid_large;
  if (nv <= 0) return;
  __shared__ int bstart[FN_BUCKETS + 1];
  __shared__ int bfill[FN_BUCKETS];
  __shared__ int rbase[BRISK_MAX_LAYERS + 1];
  __shared__ int wsum[FN_THREADS / 64];
  const BriskCand* C = cand + (long)frame * cand_cap;
  const unsigned* keys = keys_scratch + (long)frame * cand_cap * 2;  // [key][cand index] from k_finalize
  unsigned* tmp = scratch2 + (long)frame * scratch2_stride;          // the same pairs grouped by bucket
  if (tid == 0) {
    int acc = 0;
    for (int l = 0; l < G.nlayers; ++l) { rbase[l] = acc; acc += G.L[l].h; }
    rbase[G.nlayers] = acc;
  }
  for (int b = tid; b < FN_BUCKETS; b += FN_THREADS) { bstart[b] = 0; bfill[b] = 0; }
  __syncthreads();
  int shift = 0;
  while ((rbase[G.nlayers] >> shift) >= FN_BUCKETS) ++shift;  // several rows per bucket if the pyramid has more rows
  for (int j = tid; j < nv; j += FN_THREADS) {
    const unsigned key = keys[2 * j];
    atomicAdd(&bstart[(rbase[key >> 26] + (int)((key >> 13) & 0x1FFF)) >> shift], 1);
  }
  __syncthreads();
  {  // exclusive prefix over the buckets: FN_BUCKETS / FN_THREADS consecutive buckets per thread
    constexpr int PER = FN_BUCKETS / FN_THREADS;
    int loc[PER], sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) { loc[q] = bstart[tid * PER + q]; sum += loc[q]; }
    int incl = sum;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off, 64);
      if (lane >= off) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
#pragma unroll
    for (int q = 0; q < FN_THREADS / 64; ++q) woff += (q < wave) ? wsum[q] : 0;
    int run = woff + incl - sum;
#pragma unroll
    for (int q = 0; q < PER; ++q) { bstart[tid * PER + q] = run; run += loc[q]; }
    if (tid == FN_THREADS - 1) bstart[FN_BUCKETS] = run;
  }
  __syncthreads();
  for (int j = tid; j < nv; j += FN_THREADS) {  // group by bucket (unordered inside a bucket)
    const unsigned key = keys[2 * j];
    const int b = (rbase[key >> 26] + (int)((key >> 13) & 0x1FFF)) >> shift;
    const int pos = bstart[b] + atomicAdd(&bfill[b], 1);
    tmp[2 * pos] = key;
    tmp[2 * pos + 1] = keys[2 * j + 1];
  }
  __threadfence();
  __syncthreads();
  for (int pos = tid; pos < nv; pos += FN_THREADS) {
    const unsigned key = tmp[2 * pos];
    const int b = (rbase[key >> 26] + (int)((key >> 13) & 0x1FFF)) >> shift;
    int rank = bstart[b];
    for (int q = bstart[b]; q < bstart[b + 1]; ++q) rank += (tmp[2 * q] < key) ? 1 : 0;
    if (rank < kp_cap) {
      const BriskCand& c = C[tmp[2 * pos + 1]];
      BriskKeyPoint kp;
      kp.x = c.kx; kp.y = c.ky; kp.size = c.ksize; kp.angle = -1.0f; kp.response = c.kresp;
      kp.octave = G.single_layer ? 0 : c.layer; kp.class_id = -1;
      kp_out[(long)frame * kp_cap + rank] = kp;
    }
  }
  if (tid == 0) {
    counters[frame].nkp = min(nv, kp_cap);
    if (nv > kp_cap) atomicOr(&counters[frame].overflow, 4);
  }
}

// ------------------------------------------------------------------------------------------------
// Integral image (brisk/include/brisk/internal/integral-image.h:56-161): exclusive 2-D prefix sums, u32,
// (h+1) x (w+1) with row stride istride.  The image is read once more and the integral written once:
//   (k_pyramid_fused / k_pyramid_even)  column sums of every 96- / 64-row band, produced while the frame block is in LDS anyway
//   k_integral_final     per band: carry row = prefix over the bands above, then row after row the
//                        workgroup scans the row (wave shuffles + one barrier) and adds it to the running
//                        column accumulators it keeps in registers; 16-byte aligned stores.
// Thread t of a 512-thread workgroup owns integral columns 4t..4t+3 (= pixels 4t-1..4t+2) of a 2048-column chunk.
// ------------------------------------------------------------------------------------------------
#define II_THREADS 512
#define II_BAND 64
#define II_CHUNK (II_THREADS * 4)

// exclusive offset of `total` among the workgroup's threads (thread order) + the workgroup total
__device__ __forceinline__ unsigned ii_wg_scan(unsigned total, unsigned (*wave_tot)[II_THREADS / 64], int buf, unsigned* wg_total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned incl = (unsigned)wave_inclusive_scan((int)total);  // DPP row shifts / broadcasts: no LDS traffic
  if (lane == 63) wave_tot[buf][wave] = incl;
  __syncthreads();
  unsigned woff = 0, tot = 0;
#pragma unroll
  for (int q = 0; q < II_THREADS / 64; ++q) {
    const unsigned v = wave_tot[buf][q];
    woff += (q < wave) ? v : 0;
    tot += v;
  }
  *wg_total = tot;
  return woff + incl - total;
}

#ifndef II_ROWS
#define II_ROWS 3  // pixel rows per workgroup barrier (1: 0.79 ms beside the tie chain, 2: 0.71, 3: 0.69, 4: 0.84)
#endif
#define II_MAXCHUNKS 4  // 4 x 2048 columns >= the 8191-pixel width limit of the engine
// raw dwords that hold pixel columns c0-1 .. c0+2 of one image row (unconditional loads on clamped addresses)
__device__ __forceinline__ uint2 ii_fetch(const uint8_t* row, int stride, int c0 /* first integral column */) {
  const int a = min(max(c0 - 4, 0), stride - 4), b = min(c0, stride - 4);
  return make_uint2(*reinterpret_cast<const unsigned*>(row + a), *reinterpret_cast<const unsigned*>(row + b));
}
__device__ __forceinline__ void ii_unpack(uint2 raw, int stride, int w, int c0, unsigned* px) {
  const unsigned prev = (c0 >= 4 && c0 - 4 < stride) ? raw.x : 0u, cur = (c0 < stride) ? raw.y : 0u;
  px[0] = (c0 - 1 >= 0 && c0 - 1 < w) ? (prev >> 24) : 0;
  px[1] = (c0 < w) ? (cur & 0xFF) : 0;
  px[2] = (c0 + 1 < w) ? ((cur >> 8) & 0xFF) : 0;
  px[3] = (c0 + 2 < w) ? ((cur >> 16) & 0xFF) : 0;
}

// NCH = number of 2048-column chunks (template: the running sums and the prefetched row live in registers, and the
// kernel has to stay at <= 64 VGPRs so that four 512-thread workgroups share a CU)
// B24: the integral image modulo 2^24 in 3-byte elements (row pitch istride * 3 bytes) - a thread's four columns are 12
// bytes, one dwordx3 store; what k_describe needs of it are box sums below 2^24 (BriskPatternDev::int24_ok)
typedef uint32_t __attribute__((ext_vector_type(3))) ii_u32x3;
__device__ __forceinline__ void ii_store4(bool b24, uint32_t* out, long row, int istride, int c0, unsigned a0, unsigned a1, unsigned a2, unsigned a3) {
  if (b24) {
    ii_u32x3 v;
    v.x = (a0 & 0xFFFFFFu) | (a1 << 24);
    v.y = ((a1 >> 8) & 0xFFFFu) | (a2 << 16);
    v.z = ((a2 >> 16) & 0xFFu) | (a3 << 8);
    *reinterpret_cast<ii_u32x3*>(reinterpret_cast<uint8_t*>(out) + (row * istride + c0) * 3) = v;
  } else {
    *reinterpret_cast<uint4*>(out + row * istride + c0) = make_uint4(a0, a1, a2, a3);
  }
}
// SUB (calls of a few frames): a band is cut into `nsub` pieces of `sub_h` rows, one workgroup each - a band is a chain of
// band_h / II_ROWS barrier steps, and a single 1080p frame has 12 or 17 of them for 256 CUs: 40 us of a one-frame call.
// A piece's carry row = the bands above + the column sums of its band's rows above it, which it adds up itself.
template <int NCH, bool B24, bool SUB>
__global__ void __launch_bounds__(II_THREADS) k_integral_final(BriskGeom G, const uint8_t* __restrict__ pyr,
                                                               const uint32_t* __restrict__ bandsum,
                                                               uint32_t* __restrict__ integral, int istride, long iframe_elems,
                                                               int nbands, int band_h, BriskFrameCounters* counters, int nsub, int sub_h) {
  __shared__ unsigned wave_tot[2][II_THREADS / 64];
  const int frame = blockIdx.y, band = SUB ? (int)blockIdx.x / nsub : (int)blockIdx.x;
  const int sub = SUB ? (int)blockIdx.x - band * nsub : 0;
  // B24: 3-byte elements, for every frame of the call (k_describe is instantiated for one form per launch); the flag in
  // the counters is what the debug download reads
  constexpr bool b24 = B24;
  if (counters && blockIdx.x == 0 && threadIdx.x == 0) counters[frame].i24 = b24 ? 1 : 0;
  const int w = G.L[0].w, h = G.L[0].h, stride = G.L[0].stride;
  const uint8_t* img = brisk_layer_img(G, pyr, frame, 0);
  uint32_t* out = integral + (long)frame * iframe_elems;
  const int yb = band * band_h;                                       // first row of the band
  const int y0 = SUB ? yb + sub * sub_h : yb;                          // rows of this workgroup
  const int y1 = SUB ? min(min(h, yb + band_h), y0 + sub_h) : min(h, y0 + band_h);
  if (SUB && y0 >= y1) return;
  int buf = 0;
  unsigned acc[NCH][4];  // running integral values of this thread's columns (row above the current one)
  // first pixel row of the band: in flight while the carry row is computed
  uint2 raw[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) raw[ch] = ii_fetch(img + (long)min(y0, h - 1) * stride, stride, ch * II_CHUNK + threadIdx.x * 4);
  // carry row of the band: column sums of the bands above, prefix over the columns (left to right over the chunks)
  {
    unsigned carry = 0;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c0 = ch * II_CHUNK + threadIdx.x * 4;
      uint4 C = make_uint4(0, 0, 0, 0);
      if (c0 <= w)
        for (int b = 0; b < band; ++b) {
          const uint4 v = *reinterpret_cast<const uint4*>(bandsum + ((long)frame * nbands + b) * istride + c0);
          C.x += v.x; C.y += v.y; C.z += v.z; C.w += v.w;
        }
      if (SUB && c0 <= w) {  // the band's rows above this piece (independent loads, six rows in flight)
        for (int r0 = yb; r0 < y0; r0 += 6) {
          uint2 rr[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) rr[k] = ii_fetch(img + (long)min(r0 + k, y0 - 1) * stride, stride, c0);
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            unsigned px[4];
            ii_unpack(rr[k], stride, w, c0, px);
            if (r0 + k < y0) { C.x += px[0]; C.y += px[1]; C.z += px[2]; C.w += px[3]; }
          }
        }
      }
      if (c0 + 1 > w) C.y = 0;  // integral columns beyond w are padding (their band sums are not written)
      if (c0 + 2 > w) C.z = 0;
      if (c0 + 3 > w) C.w = 0;
      const unsigned s0 = C.x, s1 = s0 + C.y, s2 = s1 + C.z, s3 = s2 + C.w;
      unsigned tot;
      const unsigned o = ii_wg_scan(s3, wave_tot, buf, &tot) + carry;
      buf ^= 1;
      carry += tot;
      acc[ch][0] = o + s0; acc[ch][1] = o + s1; acc[ch][2] = o + s2; acc[ch][3] = o + s3;
      if (band == 0 && sub == 0 && c0 <= w) ii_store4(b24, out, 0, istride, c0, 0, 0, 0, 0);  // integral row 0
    }
  }
  // II_ROWS rows per step: their row scans share ONE workgroup barrier (the loop is a chain of barriers); the pixels of
  // the next II_ROWS rows are requested before the scans (one memory round trip per step would be exposed otherwise).
  __shared__ unsigned wave_totr[2][II_THREADS / 64][II_ROWS];
  uint2 rawr[II_ROWS][NCH];
#pragma unroll
  for (int k = 0; k < II_ROWS; ++k)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
      rawr[k][ch] = (k == 0) ? raw[ch] : ii_fetch(img + (long)min(y0 + k, h - 1) * stride, stride, ch * II_CHUNK + threadIdx.x * 4);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int y = y0; y < y1; y += II_ROWS) {
    uint2 nxt[II_ROWS][NCH];
#pragma unroll
    for (int k = 0; k < II_ROWS; ++k) {
      const int yn = min(y + II_ROWS + k, y1 - 1);
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) nxt[k][ch] = ii_fetch(img + (long)yn * stride, stride, ch * II_CHUNK + threadIdx.x * 4);
    }
    unsigned carry[II_ROWS];
#pragma unroll
    for (int k = 0; k < II_ROWS; ++k) carry[k] = 0;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c0 = ch * II_CHUNK + threadIdx.x * 4;
      unsigned s[II_ROWS][4], incl[II_ROWS];
#pragma unroll
      for (int k = 0; k < II_ROWS; ++k) {
        unsigned px[4] = {0, 0, 0, 0};
        if (c0 <= w && y + k < y1) ii_unpack(rawr[k][ch], stride, w, c0, px);
        s[k][0] = px[0]; s[k][1] = s[k][0] + px[1]; s[k][2] = s[k][1] + px[2]; s[k][3] = s[k][2] + px[3];
        incl[k] = (unsigned)wave_inclusive_scan((int)s[k][3]);
        if (lane == 63) wave_totr[buf][wave][k] = incl[k];
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < II_ROWS; ++k) {
        unsigned wo = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < II_THREADS / 64; ++q) {
          const unsigned v = wave_totr[buf][q][k];
          wo += (q < wave) ? v : 0;
          tot += v;
        }
        const unsigned o = wo + incl[k] - s[k][3] + carry[k];
        carry[k] += tot;
        acc[ch][0] += o + s[k][0]; acc[ch][1] += o + s[k][1]; acc[ch][2] += o + s[k][2]; acc[ch][3] += o + s[k][3];
        if (c0 <= w && y + k < y1) ii_store4(b24, out, (long)(y + k + 1), istride, c0, acc[ch][0], acc[ch][1], acc[ch][2], acc[ch][3]);
      }
      buf ^= 1;
    }
#pragma unroll
    for (int k = 0; k < II_ROWS; ++k)
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) rawr[k][ch] = nxt[k][ch];
  }
}

void brisk_launch_integral(const BriskGeom& G, const uint8_t* pyr, const uint32_t* bandsum, uint32_t* integral, int istride,
                            long iframe_elems, int band_h, int nframes, hipStream_t s, int ibits, BriskFrameCounters* counters) {
  const int nbands = (G.L[0].h + band_h - 1) / band_h;
  const int nchunks = (G.L[0].w + 1 + II_CHUNK - 1) / II_CHUNK;
  // calls of up to 4 frames: bands in pieces of 5 or 6 barrier steps (a single frame's 12 / 17 bands leave the chip idle and
  // are chains of 32 / 22 steps)
  static const int sub_knob = env_knob("BRISK_II_SUB", -1);
  const int sub_h = ((sub_knob > 0 ? sub_knob : (band_h >= 96 ? 18 : 15)) + II_ROWS - 1) / II_ROWS * II_ROWS;
  const bool sub = sub_knob != 0 && nframes <= 4 && sub_h < band_h;
  const int nsub = sub ? (band_h + sub_h - 1) / sub_h : 1;
  const dim3 grid(nbands * nsub, nframes), block(II_THREADS);
  static const int pad_lds = env_knob("BRISK_II_LDS", 0);  // tuning experiments: dynamic LDS bytes = fewer workgroups per CU
  const bool b24 = ibits == 24 && counters;  // (the per-frame flag lives in the counters)
#define II_LAUNCH1(NCH, LDS, B24_, SUB_)                                                                                                   \
  hipLaunchKernelGGL((k_integral_final<NCH, B24_, SUB_>), grid, block, LDS, s, G, pyr, bandsum, integral, istride, iframe_elems, nbands,   \
                     band_h, counters, nsub, sub_h);
#define II_LAUNCH(NCH, LDS)                                                                                                                  \
  if (b24) { if (sub) { II_LAUNCH1(NCH, LDS, true, true) } else { II_LAUNCH1(NCH, LDS, true, false) } }                                      \
  else { if (sub) { II_LAUNCH1(NCH, LDS, false, true) } else { II_LAUNCH1(NCH, LDS, false, false) } }
  if (nchunks <= 1) { II_LAUNCH(1, pad_lds) }
  else if (nchunks == 2) { II_LAUNCH(2, 0) }
  else { II_LAUNCH(II_MAXCHUNKS, 0) }
#undef II_LAUNCH
#undef II_LAUNCH1
}

// AGAST candidates of a batch, summed into host-visible (pinned, mapped) memory: what the NEXT batch of the context goes
// by when it chooses the element size of its integral image (brisk_capi.hip: integral_format) - read there without any
// synchronisation, so possibly a batch or two old; it steers speed only, never a result.
__global__ void __launch_bounds__(256) k_batch_density(const BriskFrameCounters* __restrict__ counters, int nframes, int cand_cap,
                                                       long long* __restrict__ host_word) {
  __shared__ long long part[4];
  long long sum = 0;
  for (int f = threadIdx.x; f < nframes; f += 256) sum += min(counters[f].ncand, cand_cap);
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    const long long total = part[0] + part[1] + part[2] + part[3];
    // candidates in the low 40 bits, frames above: one 8-byte store
    __hip_atomic_store(host_word, (total & 0xFFFFFFFFFFll) | ((long long)nframes << 40), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
void brisk_launch_batch_density(const BriskFrameCounters* counters, int nframes, int cand_cap, long long* host_word, hipStream_t s) {
  hipLaunchKernelGGL(k_batch_density, dim3(1), dim3(256), 0, s, counters, nframes, cand_cap, host_word);
}

// ------------------------------------------------------------------------------------------------
// k_publish_single: the results of a ONE-frame host-buffer call (frame slot 0) written straight into pinned host memory
// by the device - counter record, the keypoints and descriptor rows the counters announce - and a sequence word behind
// them that the host polls.  Replaces three blocking copies (count, keypoints, descriptor rows: 20 us each) by one small
// kernel; the host sees the word a microsecond after the last store instead of waking up from a stream wait.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_publish_single(const BriskFrameCounters* __restrict__ counters, const BriskKeyPoint* __restrict__ kps,
                                                        const uint8_t* __restrict__ desc, int which, int max_kp, int dev_pitch,
                                                        uint8_t* host, unsigned o_cnt, unsigned o_kp, unsigned o_desc, int* done, unsigned seq) {
  const int tid = threadIdx.x, gt = blockIdx.x * blockDim.x + tid, gn = gridDim.x * blockDim.x;
  const int cnt = (counters[0].overflow & 7) ? 0 : (which ? counters[0].ndesc : counters[0].nkp);
  const bool fits = cnt <= max_kp;
  const int n = fits ? cnt : 0;
  if (blockIdx.x == 0)
    for (int i = tid; i < (int)(sizeof(BriskFrameCounters) / 4); i += blockDim.x)
      reinterpret_cast<int*>(host + o_cnt)[i] = reinterpret_cast<const int*>(counters)[i];
  {
    const int* src = reinterpret_cast<const int*>(kps);
    int* dst = reinterpret_cast<int*>(host + o_kp);
    const int nd = n * (int)(sizeof(BriskKeyPoint) / 4);
    for (int i = gt; i < nd; i += gn) dst[i] = src[i];
  }
  if (desc) {
    const uint2* src = reinterpret_cast<const uint2*>(desc);
    uint2* dst = reinterpret_cast<uint2*>(host + o_desc);
    const long nd = (long)n * dev_pitch / 8;
    for (long i = gt; i < nd; i += gn) dst[i] = src[i];
  }
  __threadfence_system();
  __syncthreads();
  if (tid == 0) {
    const int old = __hip_atomic_fetch_add(done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (int)gridDim.x - 1) {  // every workgroup's stores are out
      __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      __hip_atomic_store(reinterpret_cast<unsigned*>(host), seq | (fits ? 0u : 0x80000000u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
void brisk_launch_publish_single(const BriskFrameCounters* counters, const BriskKeyPoint* kps, const uint8_t* desc, int which, int max_kp,
                                 int dev_pitch, int expect, uint8_t* host, unsigned o_cnt, unsigned o_kp, unsigned o_desc, int* done,
                                 unsigned seq, hipStream_t s) {
  const long bytes = (long)expect * ((long)sizeof(BriskKeyPoint) + (desc ? dev_pitch : 0));
  int grid = (int)((bytes + 16383) / 16384);
  grid = grid < 1 ? 1 : (grid > 32 ? 32 : grid);
  hipLaunchKernelGGL(k_publish_single, dim3(grid), dim3(256), 0, s, counters, kps, desc, which, max_kp, dev_pitch, host, o_cnt, o_kp, o_desc,
                     done, seq);
}

// ------------------------------------------------------------------------------------------------
// launch wrappers (called from the C ABI implementation)
// ------------------------------------------------------------------------------------------------
// compute units of the current device (cached per device ordinal): what the "one workgroup per CU" launch shapes go by
int brisk_device_cus() {
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int v = __atomic_load_n(&cus[dev], __ATOMIC_RELAXED);
  if (!v) {
    hipDeviceProp_t prop;
    v = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    __atomic_store_n(&cus[dev], v, __ATOMIC_RELAXED);
  }
  return v;
}

static inline int grid_for(long items, int per_block, int cap) {
  long b = (items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

const char* brisk_stage_name(int i) {
  static const char* n[BRISK_PROF_STAGES] = {"k_pyramid", "k_detect", "k_classify_refine", "k_tie_resolve", "k_finalize",
                                             "k_postfilter", "k_integral_final", "k_desc_prepare", "k_describe"};
  return (i >= 0 && i < BRISK_PROF_STAGES) ? n[i] : "?";
}

void brisk_prof_begin_call(BriskProfiler* P) {
  if (!P || !P->on) return;
  if (!P->created) {
    for (int c = 0; c < BRISK_PROF_MAX_CALLS; ++c)
      for (int k = 0; k <= BRISK_PROF_STAGES; ++k) (void)hipEventCreate(&P->ev[c][k]);
    for (int c = 0; c < BRISK_PROF_MAX_CALLS; ++c)
      for (int k = 0; k < 2; ++k) (void)hipEventCreate(&P->side_ev[c][k]);
    P->created = true;
  }
  const int c = P->calls % BRISK_PROF_MAX_CALLS;
  for (int k = 0; k <= BRISK_PROF_STAGES; ++k) P->used[c][k] = false;
  P->side_used[c] = false;
}

void brisk_prof_mark_side(BriskProfiler* P, int which, hipStream_t side) {
  if (!P || !P->on) return;
  const int c = P->calls % BRISK_PROF_MAX_CALLS;
  (void)hipEventRecord(P->side_ev[c][which], side);
  if (which == 1) P->side_used[c] = true;
}

void brisk_prof_destroy(BriskProfiler* P) {
  if (!P || !P->created) return;
  for (int c = 0; c < BRISK_PROF_MAX_CALLS; ++c) {
    for (int k = 0; k <= BRISK_PROF_STAGES; ++k) (void)hipEventDestroy(P->ev[c][k]);
    for (int k = 0; k < 2; ++k) (void)hipEventDestroy(P->side_ev[c][k]);
  }
  P->created = false;
}

void brisk_prof_mark(BriskProfiler* P, int slot, hipStream_t s) {
  if (!P || !P->on) return;
  const int c = P->calls % BRISK_PROF_MAX_CALLS;
  (void)hipEventRecord(P->ev[c][slot], s);
  P->used[c][slot] = true;
}

// ---- the box's own streaming ceiling (reported next to the roofline numbers): float4 copy / read-only pass
__global__ void __launch_bounds__(256) k_stream_copy(const float4* __restrict__ in, float4* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void __launch_bounds__(256) k_stream_read(const float4* __restrict__ in, float* __restrict__ out, long n) {
  float s = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float4 v = in[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 1.2345f) out[0] = s;
}
void brisk_launch_stream_probe(const void* a, void* b, size_t bytes, int mode, hipStream_t s) {
  const long n = (long)(bytes / 16);
  if (mode == 0) hipLaunchKernelGGL(k_stream_copy, dim3(4096), dim3(256), 0, s, (const float4*)a, (float4*)b, n);
  else hipLaunchKernelGGL(k_stream_read, dim3(4096), dim3(256), 0, s, (const float4*)a, (float*)b, n);
}

void brisk_launch_smap_clear(const BriskGeom& Gprev, const BriskDetectBuffers& B, int nframes, hipStream_t s) {
  if (nframes > 0) hipLaunchKernelGGL(k_smap_clear, dim3(16, nframes), dim3(256), 0, s, Gprev, B.smap, B.cand, B.counters, B.cand_cap);
}

static void launch_pyramid(const BriskGeom& G, const BriskDetectBuffers& B, int nframes, const uint8_t* frames, long frame_pitch,
                           int row_pitch, hipStream_t s) {
  {
    // layer-0 copy, band sums and both chains (L2, L4, L6 / L1, L3, L5, L7) from 96x96 blocks of the frame
    const int even_levels = G.nlayers >= 7 ? 3 : G.nlayers >= 5 ? 2 : G.nlayers >= 3 ? 1 : 0;
    const int odd_levels = G.nlayers >= 8 ? 3 : G.nlayers >= 6 ? 2 : G.nlayers >= 4 ? 1 : G.nlayers >= 2 ? 0 : -1;
    const int ftx = (G.L[0].stride + 95) / 96, fty = (G.L[0].h + PF_BAND - 1) / PF_BAND;
    hipLaunchKernelGGL(k_pyramid_fused, dim3(ftx * fty, nframes), dim3(256), 0, s, G, frames, frame_pitch, row_pitch, B.pyr,
                       even_levels, odd_levels, ftx, fty, B.bandsum, B.istride);
    for (int l = 8; l < G.nlayers; ++l) {  // more than 4 octaves: remaining levels one by one
      const long items = (long)(G.L[l].stride / 4) * G.L[l].h;
      hipLaunchKernelGGL(k_pyramid_level, dim3(grid_for(items, 256, 2048), nframes), dim3(256), 0, s, G, B.pyr, l - 2, l, 0);
    }
  }
}

void brisk_launch_detect(const BriskGeom& G, const BriskTileTable& T, const BriskDetectBuffers& B, int nframes,
                         const uint8_t* frames, long frame_pitch, int row_pitch, const uint8_t* mask,
                         long mask_frame_pitch, int mask_row_pitch, hipStream_t s, BriskProfiler* prof,
                         const BriskOverlap* ov) {
  (void)hipMemsetAsync(B.counters, 0, sizeof(BriskFrameCounters) * (size_t)nframes, s);
  brisk_prof_mark(prof, BRISK_STG_PYRAMID, s);
  // timing probes only (tools/overlap_probe3.py; the batch's results are meaningless): 1 = the call ends behind k_detect,
  // 2 = and skips the pyramid kernel (k_detect alone, on whatever pyramid the buffers hold)
  const int probe_stop = env_knob("BRISK_DETECT_PROBE", 0);  // (read per call: the probe switches it on after its set-up batches)
  if (probe_stop != 2) launch_pyramid(G, B, nframes, frames, frame_pitch, row_pitch, s);
  // The integral image only needs layer 0 and the band sums (pyramid kernel) and is HBM-bound; tie resolution
  // (one workgroup per frame, a chain of dependent decisions) and the final ordering are latency-bound and leave
  // most of the chip idle.  The integral kernel runs beside them on a second, low-priority stream (forked in front
  // of the tie kernel) and is joined before the descriptor kernels.
  static const int fork_at = env_knob("BRISK_INTEGRAL_FORK", 0);  // experiments: 1 = beside k_detect, 2 = beside the score blocks
  auto fork_integral = [&]() {
    (void)hipEventRecord(ov->fork, s);
    (void)hipStreamWaitEvent(ov->side, ov->fork, 0);
    brisk_prof_mark_side(prof, 0, ov->side);
    brisk_launch_integral(G, B.pyr, B.bandsum, ov->Dd->integral, ov->Dd->istride, ov->Dd->iframe_elems, B.band_h, nframes, ov->side,
                          ov->Dd->ibits, B.counters);
    brisk_prof_mark_side(prof, 1, ov->side);
    (void)hipEventRecord(ov->join, ov->side);
  };
  if (ov && fork_at == 1) fork_integral();
  brisk_prof_mark(prof, BRISK_STG_DETECT, s);
  hipLaunchKernelGGL(k_detect, dim3((T.total_tiles + 7) / 8 * 8, nframes), dim3(256), 0, s, G, T, B.pyr, B.smap, B.cand, B.counters,
                     B.cand_cap);
  if (probe_stop) return;
  if (ov && fork_at == 2) fork_integral();
  brisk_prof_mark(prof, BRISK_STG_CLASSIFY, s);
  // Ordered path (the sequential algorithm on its literal cache): always for the multi-layer no-scale-NMS branch and for
  // ComputeScale's pyramid; below threshold 20 only for the frames in which k_detect stored a score <= 2
  // (BriskFrameCounters::low_score) - the other frames of such a batch go through the fast-path kernels below, which skip
  // the flagged ones.
  const bool ordered_all = G.no_scale_nms || G.lower_threshold != BRISK_LOWER_THRESHOLD;
  const bool ordered_some = !ordered_all && G.threshold < BRISK_FAST_PATH_MIN_THRESHOLD;
  auto launch_ordered = [&](int only_flagged) {
    const long row_stride = (long)BRISK_MAX_LAYERS * B.tie_cap;
    hipLaunchKernelGGL(k_order_candidates, dim3(nframes), dim3(OC_THREADS), 0, s, G, B.cand, B.counters, B.keys, B.tie_idx,
                       row_stride, B.cand_cap, only_flagged);
    hipLaunchKernelGGL(k_ordered_keypoints, dim3(nframes), dim3(64), 0, s, G, B.pyr, B.smap, B.cand, B.counters, B.keys, B.kp_out,
                       B.cand_cap, B.kp_cap, mask, mask_frame_pitch, mask_row_pitch, G.no_scale_nms, only_flagged);
  };
  if (ordered_all) {
    brisk_prof_mark(prof, BRISK_STG_TIES, s);
    if (ov && fork_at != 1 && fork_at != 2) fork_integral();
    launch_ordered(0);
    brisk_prof_mark(prof, BRISK_STG_FINALIZE, s);
    brisk_prof_mark(prof, BRISK_STG_POSTFILTER, s);
    return;
  }
  // blocks per frame: few for large batches (every block then walks several rounds of its frame's candidates on one
  // XCD), many for small ones (a single frame must spread over the chip)
  static const int sb_knob = env_knob("BRISK_SB_BLOCKS", 0), cr_knob = env_knob("BRISK_CR_BLOCKS", 0);
  // (round 5, tools/sweep_cr_sb.sh: 64 frames per call with 256 score blocks per frame instead of 32 - 64 dense frames at threshold
  // 30 4.23 -> 4.33 k frames/s, threshold 50 14.65 -> 14.9 k, 64 4K frames 16.04 -> 16.23 k; 128 frames with 64: 68.07 -> 68.55 k;
  // more classification blocks per frame lose everywhere: 48 / 96 / 192 at 64 dense frames 4.22 / 4.12 / 3.95 k)
  const int sb_blocks = sb_knob ? sb_knob : (nframes > 64 ? (nframes >= 256 ? 32 : 64) : 256), cr_blocks = cr_knob ? cr_knob : (nframes >= 64 ? 24 : 64);
  {
    const int sb_bpf = grid_for(B.cand_cap, SB_WAVES * SB_PER_WAVE, sb_blocks), cr_bpf = grid_for(B.cand_cap, 64, cr_blocks);
    hipLaunchKernelGGL(k_score_blocks, dim3(xcd_grid(nframes, sb_bpf)), dim3(SB_WAVES * 64), 0, s, G, B.pyr, B.smap, B.cand,
                       B.counters, B.blocks, B.cand_cap, nframes, sb_bpf);
    hipLaunchKernelGGL(k_classify_refine, dim3(xcd_grid(nframes, cr_bpf)), dim3(64), 0, s, G, B.pyr, B.smap, B.cand, B.counters,
                       B.blocks, B.tie_idx, B.cand_cap, B.tie_cap, nframes, cr_bpf);
  }
  hipLaunchKernelGGL(k_classify_refine_direct, dim3(8, nframes), dim3(64), 0, s, G, B.pyr, B.smap, B.cand, B.counters,
                     B.tie_idx, B.cand_cap, B.tie_cap);
  brisk_prof_mark(prof, BRISK_STG_TIES, s);
  if (ov && fork_at == 0) fork_integral();
  {
    // Small batches (up to 64 frames x 8 layers): one workgroup per (frame, layer), the layers of a frame run as a
    // pipeline (a frame is ready in the time of its largest layer instead of the sum; dense frames gain most).  Larger
    // batches: groups of consecutive layers so that about one workgroup per CU is at work, i.e. one workgroup per frame
    // from 256 frames on - the chip is full anyway and a workgroup that waits for the layer below only takes a slot
    // (128 frames: 0.69 ms with one layer per workgroup, 0.35 with four, 0.45 with eight).  With 8 or more frames
    // every XCD residue of blockIdx gets whole frames.
    static const int lpw_knob = env_knob("BRISK_TR_LPW", 0);
    // (pyramids of more than 8 layers - 4K frames with 6 octaves: 23 ... 1026 ties per layer - in groups that leave a
    // quarter of the CUs to the integral kernel beside them: 64 frames x 12 layers 1.03 -> 0.89 ms)
    const int per_wg = G.nlayers > 8 ? 192 : 256;
    // (the fewest layers per workgroup whose ticket count still fits: 192 frames x 8 layers by six gave 384 workgroups on
    // 256 CUs - 0.67 ms; one workgroup per frame: 0.40)
    int lpw_auto = G.nlayers;
    for (int c = 1; c <= G.nlayers; ++c)
      if (nframes * ((G.nlayers + c - 1) / c) <= per_wg) { lpw_auto = c; break; }
    const int lpw = lpw_knob ? min(lpw_knob, G.nlayers) : (nframes * G.nlayers <= 512 ? 1 : lpw_auto);
    int tr_grid = (nframes >= 8 ? (nframes + 7) / 8 * 8 : nframes) * ((G.nlayers + lpw - 1) / lpw);
    static const int persist_knob = env_knob("BRISK_TR_PERSIST", -1), pgrid_env = env_knob("BRISK_TR_PGRID", 0);
    const int pgrid_knob = pgrid_env > 0 ? pgrid_env : brisk_device_cus();  // one workgroup per CU of THIS device (256 on an MI355X)
    const int persist = persist_knob >= 0 ? persist_knob : (tr_grid > pgrid_knob ? 1 : 0);
    if (persist && tr_grid > pgrid_knob) tr_grid = pgrid_knob;
    // waves per workgroup: 16 (15 deciding).  Where a workgroup owns a whole frame of a large batch (one workgroup per CU,
    // the integral kernel beside it): 12 - a second 512-thread workgroup of the integral kernel then fits on the CU (3 x 96
    // + 4 x 56 VGPRs per SIMD), the tie kernel itself gets slower and the batch 0.9 % faster (72.95 -> 73.6 k frames/s; 14:
    // 72.7, 10: 72.5, 8: 70.9).  Not for the ticketed forms: 64 4K frames 15.2 -> 14.7 k frames/s with 12 (there the tie
    // kernel is what the window waits for).
    static const int waves_knob = env_knob("BRISK_TR_WAVES", 0);
    const int tr_waves = waves_knob ? min(max(waves_knob, 2), TR_WAVES) : ((!persist && lpw >= G.nlayers && nframes >= 192) ? 12 : TR_WAVES);
    // two ties per wave (k_tie_resolve_pair) up to 64 frames per call: 64 dense frames 3.83 -> 4.15 k frames/s (threshold 30),
    // 64 4K frames 15.45 -> 15.78 k, one 4K frame's tie stage 0.386 -> 0.32 ms; from 128 frames on its 107 VGPRs cost more
    // beside the integral kernel than the pairs win (128: 68.1 -> 67.2 k frames/s, 512: 76.0 -> 74.6 k).
    static const int pair_knob = env_knob("BRISK_TR_PAIR", 1);  // A / B runs: 0 = one tie per wave everywhere, 2 = pairs in all batches
    static const int pair_min_knob = env_knob("BRISK_TR_PAIR_MIN", 0);
    static const int pair_mode_knob = env_knob("BRISK_TR_PAIR_MODE", 1);
    const int pair_min = (pair_min_knob ? pair_min_knob : 4 * (tr_waves - 1)) | (pair_mode_knob << 16);
    // small calls: every layer's ties dealt to `bands` workgroups by image row (brisk_tie_kernel.inc), as many as leave one
    // workgroup per CU (8, 4 or 2): one 4K frame's tie stage 0.267 -> 0.179 ms with eight (four: 0.209, two: 0.249), one 1080p frame
    // 121 -> 99 us (four: 104), VGA 92 -> 82 us; 3 / 4 / 8 frames per call + 7 / 6 / 2.5 % with four (six bands lose: 115 us at 1080p)
    static const int bands_knob = env_knob("BRISK_TR_BANDS", 0);
    int bands = 1;
    if (lpw == 1 && !persist) {
      bands = bands_knob ? min(max(bands_knob, 1), BRISK_TIE_MAX_BANDS) : (tr_grid * 8 <= pgrid_knob ? 8 : tr_grid * 4 <= pgrid_knob ? 4 : tr_grid * 2 <= pgrid_knob ? 2 : 1);
      // (a performance choice, not a residency requirement: tickets are drawn at run time and a ticket only waits for tickets
      // drawn before it, so forward progress holds with any number of resident workgroups - more bands than CUs only queue)
      if (tr_grid * bands > pgrid_knob) bands = 1;
    }
    if ((nframes <= 64 && pair_knob) || pair_knob == 2)
      hipLaunchKernelGGL(k_tie_resolve_pair, dim3(tr_grid * bands), dim3(tr_waves * 64), 0, s, G, B.pyr, B.smap, B.cand, B.counters, B.tie_idx,
                         B.blocks, B.keys, B.cand_cap, B.tie_cap, nframes, lpw, persist, pair_min, bands);
    else if (nframes < 32)
      hipLaunchKernelGGL(k_tie_resolve_small, dim3(tr_grid), dim3(tr_waves * 64), 0, s, G, B.pyr, B.smap, B.cand, B.counters, B.tie_idx,
                         B.blocks, B.keys, B.cand_cap, B.tie_cap, nframes, lpw, persist, 0, 1);
    else
      hipLaunchKernelGGL(k_tie_resolve, dim3(tr_grid), dim3(tr_waves * 64), 0, s, G, B.pyr, B.smap, B.cand, B.counters, B.tie_idx,
                         B.blocks, B.keys, B.cand_cap, B.tie_cap, nframes, lpw, persist, 0, 1);
  }
  brisk_prof_mark(prof, BRISK_STG_FINALIZE, s);
  hipLaunchKernelGGL(k_finalize, dim3(nframes), dim3(FN_THREADS), 0, s, G, B.cand, B.counters, B.keys, B.kp_out, B.cand_cap,
                     B.kp_cap, mask, mask_frame_pitch, mask_row_pitch);
  hipLaunchKernelGGL(k_finalize_large, dim3(nframes), dim3(FN_THREADS), 0, s, G, B.cand, B.counters, B.keys,
                     reinterpret_cast<unsigned*>(B.tie_idx), (long)BRISK_MAX_LAYERS * B.tie_cap, B.kp_out, B.cand_cap, B.kp_cap);
  if (ordered_some) launch_ordered(1);  // (after the fast-path kernels, which left the flagged frames' lists and scratch alone)
  brisk_prof_mark(prof, BRISK_STG_POSTFILTER, s);
}

void brisk_launch_compute_scale(const BriskGeom& G, const BriskDetectBuffers& B, const uint8_t* frame, int row_pitch,
                                const BriskKeyPoint* d_in, int n_in, int suppress, hipStream_t s) {
  (void)hipMemsetAsync(B.counters, 0, sizeof(BriskFrameCounters), s);
  launch_pyramid(G, B, 1, frame, 0, row_pitch, s);
  // one lane per (layer, point) where the scratch buffers hold the lists - adm: nlayers x n_in ints in B.keys (2 x cand_cap
  // words), tmp: 32 bytes per entry in B.blocks (64 x cand_cap bytes) -, the one-lane walk otherwise and for the calls in
  // which a layer admits no point (decided on the device: counters[0].pad[0])
  static const int seq_knob = env_knob("BRISK_CS_SEQUENTIAL", 0);  // tests / A-B runs: 1 = always the one-lane walk
  const bool parallel = !seq_knob && (long)G.nlayers * n_in <= 2L * B.cand_cap;
  if (parallel) {
    int* adm = reinterpret_cast<int*>(B.keys);
    unsigned* tmp = reinterpret_cast<unsigned*>(B.blocks);
    hipLaunchKernelGGL(k_cs_admit, dim3(G.nlayers), dim3(CS_THREADS), 0, s, G, B.pyr, B.smap, d_in, n_in, B.counters, adm);
    hipLaunchKernelGGL(k_cs_scores, dim3((n_in + CS_THREADS - 1) / CS_THREADS, G.nlayers), dim3(CS_THREADS), 0, s, G, B.pyr, B.smap, d_in,
                       n_in, B.counters, adm);
    hipLaunchKernelGGL(k_cs_refine, dim3((n_in + 63) / 64, G.nlayers), dim3(64), 0, s, G, B.pyr, B.smap, d_in, n_in, suppress, B.counters,
                       adm, tmp);
    hipLaunchKernelGGL(k_cs_emit, dim3(1), dim3(1024), 0, s, G, B.counters, tmp, n_in, B.kp_out, B.kp_cap);
  }
  // (the walk's detection scratch is B.keys as well: it only runs where the lists above are not used any more)
  hipLaunchKernelGGL(k_compute_scale, dim3(1), dim3(64), 0, s, G, B.pyr, B.smap, d_in, n_in, suppress, B.counters, B.keys,
                     2 * B.cand_cap, B.kp_out, B.kp_cap, parallel ? 0 : 1);
}

void brisk_launch_layer0_only(const BriskGeom& G, const BriskDetectBuffers& B, int nframes, const uint8_t* frames,
                              long frame_pitch, int row_pitch, hipStream_t s) {
  const int etx = (G.L[0].stride + 63) / 64, ety = (G.L[0].h + 63) / 64;
  hipLaunchKernelGGL(k_pyramid_even, dim3(etx * ety, nframes), dim3(256), 0, s, G, frames, frame_pitch, row_pitch, B.pyr, 0,
                     etx, ety, B.bandsum, B.istride);
}

