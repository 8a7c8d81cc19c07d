// microbench_lds_patch.hip - LDS patch staging for the descriptor's sampling pattern, measured before k_describe is
// touched (round-4 review item 1; brisk/src/brisk-descriptor-extractor.cc:370-530, 618-662).
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/mblds tools/microbench_lds_patch.hip \
//         ethzasl_brisk_amd/csrc/brisk_pattern.cpp
//   python3 tools/gen_lds_patch_input.py /tmp/lds_in.bin && /tmp/mblds /tmp/lds_in.bin
//
// Workload: the keypoints of BASELINE config 2 (4 distinct 1080p frames, the oracle's positions, scale indices and
// rotations) on 256 frame slots, dealt like k_describe deals them (frame f -> queue f % 8 = XCD, spatial processing
// order).  Every variant computes the SAME 132 SmoothedIntensity values per keypoint with the engine's own arithmetic
// (brisk_box_prep / brisk_box_acc / brisk_div_by_magic) - the checksums must agree - plus `filler` dependent integer
// operations per keypoint for the long pairs / bits the real kernel runs beside the sampling.
//   gather_u32 / gather_i24   the shipped formulation: runs of 2 keypoints, 3 rounds per pass, ten buffer gathers per
//                             sample from the global integral image (32-bit / 3-byte elements)
//   lds_*                     one keypoint per wave: its patch of the integral image staged in LDS with coalesced row
//                             loads, then 4 rounds (64 + 2 samples per pass) of ten ds_read2 per sample
//     lds_u16_from_i24 / _u32   low 16 bits of the integral rows (every region sum of a box with interior <= 256 px is
//                               below 2^16: scale indices <= 28, patch side <= 101 px), 2 bytes per element
//     lds_u32_from_u32          the rows as they are (review variant ii)
//     lds_u16_from_pix / lds_u32_from_pix   u8 pixels -> local integral by DPP row scans (review variant i)
// Classes by patch side 2 * sizeList_[scale] + 1: <= 67, <= 101, <= 151, <= 201, all.
// Output: one JSON row per (variant, class): samples/ns chip-wide, us per keypoint and CU, checksum.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../ethzasl_brisk_amd/csrc/brisk_common.h"
#include "../ethzasl_brisk_amd/csrc/brisk_device_describe.h"
#include "../ethzasl_brisk_amd/csrc/brisk_pattern.h"

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);  \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

#define GETREG_XCC_ID (20 | (3 << 11))
typedef uint32_t __attribute__((ext_vector_type(2))) u32x2;
typedef uint32_t __attribute__((ext_vector_type(3))) u32x3;
typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
typedef uint32_t __attribute__((ext_vector_type(2), aligned(4))) u32x2a4;

struct MbArgs {
  const uint32_t* integ32;  // [NF] (h + 1) x (w + 1) u32
  const uint8_t* integ24;   // [NF] the same in 3-byte elements
  const uint8_t* pix;       // [NF] h x w u8
  const uint8_t* integ_il2; // [NF] 3-byte elements, rows interleaved in pairs: (y, x) at ((y >> 1) * 2 iw + 2 x + (y & 1)) * 3
  long il2_bytes;
  long f32_elems, f24_bytes, pix_bytes;
  int iw, ih, w, h;
  const int4* tab4;
  const double2* uv2;
  const int* size_list;
  int np;
  const uint4* kps;     // {x bits, y bits, scale, theta}
  const uint4* tasks;   // [8][max_tasks] {frame slot, kp0, kp1 or -1, 0}
  const int* ntasks;    // [8]
  int max_tasks;
  int* tickets;         // [8], 32 ints apart
  unsigned long long* checksum;
  int filler;
  unsigned long long* phase;  // [8] summed s_memtime ticks per phase of k_lds2 (lane 0 of every wave)
};

__device__ __forceinline__ int wave_sum_i(int v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

struct Raw {
  u32x2 p00, p02, p10, p12, p30, p32, ql, qr;
  u32x3 p20, p22;
};
// what-if forms (timing only, wrong results): 1 = without the two gathers of the displaced corners (8 gathers, 18 dwords),
// 2 = every pair gather three dwords wide (10 gathers, 30 dwords, the same lines), 3 = every gather ONE dword (10 gathers, 10 dwords)
template <bool I24, int WHATIF = 0>
__device__ __forceinline__ void gather_load(Raw& r, const BriskBoxPrep& p, __amdgpu_buffer_rsrc_t rs, int istride) {
  constexpr int ES = I24 ? 3 : 4;
  const int rowb = istride * ES;
  const int o_t = p.y_top * rowb, o_b = p.y_bottom * rowb;
  const int o_tl = o_t + p.x_left * ES, o_tr = o_t + p.x_right * ES, o_bl = o_b + p.x_left * ES, o_br = o_b + p.x_right * ES;
  if (WHATIF == 2) {
    auto w = [&](int o, int so) { const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, o, so, 0); return u32x2{v.x + v.z, v.y}; };
    r.p00 = w(o_tl, 0); r.p02 = w(o_tr, 0); r.p10 = w(o_tl, rowb); r.p12 = w(o_tr, rowb);
    r.ql = w(o_bl - rowb + ES, 0); r.qr = w(o_br - rowb + ES, 0); r.p30 = w(o_bl, rowb); r.p32 = w(o_br, rowb);
    r.p20 = __builtin_amdgcn_raw_buffer_load_b96(rs, o_bl, 0, 0);
    r.p22 = __builtin_amdgcn_raw_buffer_load_b96(rs, o_br, 0, 0);
    return;
  }
  if (WHATIF == 4) {  // no gathers at all: the address stage's values stand in for the data
    const uint32_t a = (uint32_t)o_tl, b = (uint32_t)o_br;
    r.p00 = u32x2{a, b}; r.p02 = u32x2{b, a}; r.p10 = u32x2{a + 1, b}; r.p12 = u32x2{b + 1, a}; r.ql = u32x2{a, b + 2}; r.qr = u32x2{b, a + 2};
    r.p30 = u32x2{a + 3, b}; r.p32 = u32x2{b + 3, a}; r.p20 = u32x3{a, b, a}; r.p22 = u32x3{b, a, b};
    return;
  }
  if (WHATIF >= 5) {  // the ten gathers at their real widths from ONE line per lane (always L1 hits after the first round);
    // 6 / 7 / 8: two / four / sixteen adjacent lanes share their line
    const int ln = (int)(threadIdx.x & 63);
    const int o = WHATIF == 5 ? ln * 128 : WHATIF == 6 ? (ln >> 1) * 128 + (ln & 1) * 3 : WHATIF == 7 ? (ln >> 2) * 128 + (ln & 3) * 3 : (ln >> 4) * 128 + (ln & 15);
    r.p00 = __builtin_amdgcn_raw_buffer_load_b64(rs, o, 0, 0); r.p02 = __builtin_amdgcn_raw_buffer_load_b64(rs, o + 9, 0, 0);
    r.p10 = __builtin_amdgcn_raw_buffer_load_b64(rs, o + 18, 0, 0); r.p12 = __builtin_amdgcn_raw_buffer_load_b64(rs, o + 27, 0, 0);
    r.ql = __builtin_amdgcn_raw_buffer_load_b64(rs, o + 36, 0, 0); r.qr = __builtin_amdgcn_raw_buffer_load_b64(rs, o + 45, 0, 0);
    r.p20 = __builtin_amdgcn_raw_buffer_load_b96(rs, o + 54, 0, 0); r.p22 = __builtin_amdgcn_raw_buffer_load_b96(rs, o + 69, 0, 0);
    r.p30 = __builtin_amdgcn_raw_buffer_load_b64(rs, o + 84, 0, 0); r.p32 = __builtin_amdgcn_raw_buffer_load_b64(rs, o + 96, 0, 0);
    return;
  }
  if (WHATIF == 3) {
    auto w = [&](int o, int so) { const uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(rs, o, so, 0); return u32x2{v, v >> 3}; };
    r.p00 = w(o_tl, 0); r.p02 = w(o_tr, 0); r.p10 = w(o_tl, rowb); r.p12 = w(o_tr, rowb);
    r.ql = w(o_bl - rowb + ES, 0); r.qr = w(o_br - rowb + ES, 0); r.p30 = w(o_bl, rowb); r.p32 = w(o_br, rowb);
    const u32x2 a = w(o_bl, 0), b = w(o_br, 0);
    r.p20 = u32x3{a.x, a.y, a.x ^ 1}; r.p22 = u32x3{b.x, b.y, b.x ^ 1};
    return;
  }
  r.p00 = __builtin_amdgcn_raw_buffer_load_b64(rs, o_tl, 0, 0);
  r.p02 = __builtin_amdgcn_raw_buffer_load_b64(rs, o_tr, 0, 0);
  r.p10 = __builtin_amdgcn_raw_buffer_load_b64(rs, o_tl, rowb, 0);
  r.p12 = __builtin_amdgcn_raw_buffer_load_b64(rs, o_tr, rowb, 0);
  if (WHATIF == 1) {
    r.ql = r.p00; r.qr = r.p02;
  } else {
    r.ql = __builtin_amdgcn_raw_buffer_load_b64(rs, o_bl - rowb + ES, 0, 0);
    r.qr = __builtin_amdgcn_raw_buffer_load_b64(rs, o_br - rowb + ES, 0, 0);
  }
  r.p20 = __builtin_amdgcn_raw_buffer_load_b96(rs, o_bl, 0, 0);
  r.p22 = __builtin_amdgcn_raw_buffer_load_b96(rs, o_br, 0, 0);
  r.p30 = __builtin_amdgcn_raw_buffer_load_b64(rs, o_bl, rowb, 0);
  r.p32 = __builtin_amdgcn_raw_buffer_load_b64(rs, o_br, rowb, 0);
}
__device__ __forceinline__ void unpack2(u32x2 v, uint32_t& a, uint32_t& b) {
  a = v.x;
  b = __builtin_amdgcn_alignbit(v.y, v.x, 24);
}
template <bool I24>
__device__ __forceinline__ int gather_combine(const BriskBoxPrep& p, const Raw& r) {
  uint32_t i00, i01, i02, i03, i10, i11, i12, i13, i20, i21, i2x, i22, i23, i2y, i30, i31, i32, i33, ql0, ql1, qr0, qr1;
  if (I24) {
    unpack2(r.p00, i00, i01); unpack2(r.p02, i02, i03); unpack2(r.p10, i10, i11); unpack2(r.p12, i12, i13);
    unpack2(u32x2{r.p20.x, r.p20.y}, i20, i21); i2x = __builtin_amdgcn_alignbit(r.p20.z, r.p20.y, 16);
    unpack2(u32x2{r.p22.x, r.p22.y}, i22, i23); i2y = __builtin_amdgcn_alignbit(r.p22.z, r.p22.y, 16);
    unpack2(r.p30, i30, i31); unpack2(r.p32, i32, i33); unpack2(r.ql, ql0, ql1); unpack2(r.qr, qr0, qr1);
  } else {
    i00 = r.p00.x; i01 = r.p00.y; i02 = r.p02.x; i03 = r.p02.y; i10 = r.p10.x; i11 = r.p10.y; i12 = r.p12.x; i13 = r.p12.y;
    i20 = r.p20.x; i21 = r.p20.y; i2x = r.p20.z; i22 = r.p22.x; i23 = r.p22.y; i2y = r.p22.z;
    i30 = r.p30.x; i31 = r.p30.y; i32 = r.p32.x; i33 = r.p32.y; ql0 = r.ql.x; ql1 = r.ql.y; qr0 = r.qr.x; qr1 = r.qr.y;
  }
  constexpr uint32_t mask = I24 ? 0xFFFFFFu : 0xFFFFFFFFu;
  const unsigned qbr = (i2y - i23 - qr1 + qr0) & mask;
  const unsigned qbl = (i2x - i21 - ql1 + ql0) & mask;
  const uint32_t acc = brisk_box_acc(p, i00, i01, i02, i03, i10, i11, i12, i13, i20, i21, i22, i23, i30, i31, i32, i33, qbr, qbl, mask);
  return brisk_div_by_magic((int)acc, p.magic, p.shift);
}

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- the shipped formulation -----------------------------------------------------------------------------------
template <bool I24, int WHATIF = 0>
__global__ void __launch_bounds__(128) k_gather(MbArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int* vals = reinterpret_cast<int*>(lds + wave * 1024);
  const int xcc = (int)(__builtin_amdgcn_s_getreg(GETREG_XCC_ID) & 7);
  const int np = A.np;
  long long sum = 0;
  for (int gi = 0; gi < 8; ++gi) {
    const int g = (xcc + gi) & 7;
    const int nt = A.ntasks[g];
    for (;;) {
      int t = 0;
      if (lane == 0) t = atomicAdd(&A.tickets[g * 32], 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= nt) break;
      const uint4 task = A.tasks[(long)g * A.max_tasks + t];
      const int cnt = (int)task.z >= 0 ? 2 : 1, total = cnt * np;
      const uint4 rec0 = A.kps[task.y], rec1 = A.kps[cnt == 2 ? task.z : task.y];
      const __amdgpu_buffer_rsrc_t rs =
          I24 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(A.integ24 + (long)task.x * A.f24_bytes), 0, (int)A.f24_bytes, 0x00020000)
              : __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(A.integ32 + (long)task.x * A.f32_elems), 0, (int)(A.f32_elems * 4), 0x00020000);
      int ksum = 0;
      for (int pass = 0; pass < 2; ++pass) {
        for (int s0 = 0; s0 < total; s0 += 64) {
          const int s = s0 + lane;
          const bool valid = s < total;
          const int sc = min(s, total - 1);
          const int kq = sc >= np ? 1 : 0, pt = sc - kq * np;
          const uint4 rec = kq ? rec1 : rec0;
          const int theta = pass ? (int)rec.w : 0;
          const int4 tab = A.tab4[(int)rec.z * np + pt];
          const double2 uv = A.uv2[theta * np + pt];
          const double mm = (double)__int_as_float(tab.x);
          const float xf = (float)(mm * uv.x) + __uint_as_float(rec.x), yf = (float)(mm * uv.y) + __uint_as_float(rec.y);
          const BriskBoxPrep pr = brisk_box_prep(xf, yf, __int_as_float(tab.y), tab.z, tab.w);
          Raw raw;
          if (valid) gather_load<I24, WHATIF>(raw, pr, rs, A.iw);
          const int value = gather_combine<I24>(pr, raw);
          if (valid) { vals[sc] = value; ksum += value; }
        }
        wave_sync();
      }
      int f = vals[lane] + ksum;
      { int f1 = f ^ 5, f2 = f + 7, f3 = f * 3; for (int k = 0; k < A.filler * cnt / 12; ++k) { f += (f >> 3) ^ k; f1 += (f1 >> 3) ^ k; f2 += (f2 >> 3) ^ k; f3 += (f3 >> 3) ^ k; } f ^= f1 ^ f2 ^ f3; }
      if (f == 0x12345678) ksum += 1;
      sum += ksum;
      __builtin_amdgcn_wave_barrier();
    }
  }
  const int tot_lo = wave_sum_i((int)(sum & 0xFFFFFF)), tot_hi = wave_sum_i((int)(sum >> 24));
  if (lane == 0) atomicAdd(A.checksum, (unsigned long long)tot_lo + ((unsigned long long)tot_hi << 24));
}



// ---- round 5: two lanes per sample, one per box side --------------------------------------------------------------------
// What the vector L1 charges a gather for is one tag look-up per lane whose line no neighbouring lane shares (the what-if
// rows above): here lane 2 k reads the left column pairs of sample k's five rows and lane 2 k + 1 the right ones - the same row,
// the same line when the box is narrow.  Each side reduces its eleven values to six (corner pixel top / bottom, column strip, three
// inner-column differences), the right lane's six cross to the left lane by DPP, which finishes the sample.
__device__ __forceinline__ int dpp_partner(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false); }  // quad_perm(1,0,3,2)
__global__ void __launch_bounds__(128) k_gather_lr(MbArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int* vals = reinterpret_cast<int*>(lds + wave * 1024);
  const int xcc = (int)(__builtin_amdgcn_s_getreg(GETREG_XCC_ID) & 7);
  const int np = A.np;
  const bool right = lane & 1;
  const int rowb = A.iw * 3;
  long long sum = 0;
  for (int gi = 0; gi < 8; ++gi) {
    const int g = (xcc + gi) & 7;
    const int nt = A.ntasks[g];
    for (;;) {
      int t = 0;
      if (lane == 0) t = atomicAdd(&A.tickets[g * 32], 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= nt) break;
      const uint4 task = A.tasks[(long)g * A.max_tasks + t];
      const int cnt = (int)task.z >= 0 ? 2 : 1, total = cnt * np;
      const uint4 rec0 = A.kps[task.y], rec1 = A.kps[cnt == 2 ? task.z : task.y];
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(A.integ24 + (long)task.x * A.f24_bytes), 0, (int)A.f24_bytes, 0x00020000);
      int ksum = 0;
      for (int pass = 0; pass < 2; ++pass) {
        for (int s0 = 0; s0 < total; s0 += 32) {
          const int s = s0 + (lane >> 1);
          const bool valid = s < total;
          const int sc = min(s, total - 1);
          const int kq = sc >= np ? 1 : 0, pt = sc - kq * np;
          const uint4 rec = kq ? rec1 : rec0;
          const int theta = pass ? (int)rec.w : 0;
          const int4 tab = A.tab4[(int)rec.z * np + pt];
          const double2 uv = A.uv2[theta * np + pt];
          const double mm = (double)__int_as_float(tab.x);
          const float xf = (float)(mm * uv.x) + __uint_as_float(rec.x), yf = (float)(mm * uv.y) + __uint_as_float(rec.y);
          const BriskBoxPrep pr = brisk_box_prep(xf, yf, __int_as_float(tab.y), tab.z, tab.w);
          const int xo = (right ? pr.x_right : pr.x_left) * 3;
          const int o_t = pr.y_top * rowb + xo, o_b = pr.y_bottom * rowb + xo;
          u32x2 t0 = {0, 0}, t1 = {0, 0}, q = {0, 0}, b1 = {0, 0};
          u32x3 b0 = {0, 0, 0};
          if (valid) {
            t0 = __builtin_amdgcn_raw_buffer_load_b64(rs, o_t, 0, 0);
            t1 = __builtin_amdgcn_raw_buffer_load_b64(rs, o_t, rowb, 0);
            q = __builtin_amdgcn_raw_buffer_load_b64(rs, o_b - rowb + 3, 0, 0);
            b0 = __builtin_amdgcn_raw_buffer_load_b96(rs, o_b, 0, 0);
            b1 = __builtin_amdgcn_raw_buffer_load_b64(rs, o_b, rowb, 0);
          }
          constexpr uint32_t m = 0xFFFFFFu;
          uint32_t t0x, t0y, t1x, t1y, qx, qy, b0x, b0y, b1x, b1y;
          unpack2(t0, t0x, t0y); unpack2(t1, t1x, t1y); unpack2(q, qx, qy); unpack2(u32x2{b0.x, b0.y}, b0x, b0y); unpack2(b1, b1x, b1y);
          const uint32_t b0z = __builtin_amdgcn_alignbit(b0.z, b0.y, 16);
          // the side's six numbers
          const uint32_t corner_t = (t1y - t0y - t1x + t0x) & m;
          const uint32_t corner_b = (pr.quirk ? (b0z - b0y - qy + qx) : (b1y - b0y - b1x + b0x)) & m;
          const uint32_t strip = (b0y - b0x - t1y + t1x) & m;
          const uint32_t t0i = right ? t0x : t0y, t1i = right ? t1x : t1y, b0i = right ? b0x : b0y, b1i = right ? b1x : b1y;  // the inner column
          const uint32_t d_top = t1i - t0i, d_bot = b1i - b0i, d_mid = b0i - t1i;
          // the right lane's numbers cross to the left lane
          const uint32_t Rct = (uint32_t)dpp_partner((int)corner_t), Rcb = (uint32_t)dpp_partner((int)corner_b), Rst = (uint32_t)dpp_partner((int)strip);
          const uint32_t Rdt = (uint32_t)dpp_partner((int)d_top), Rdb = (uint32_t)dpp_partner((int)d_bot), Rdm = (uint32_t)dpp_partner((int)d_mid);
          const uint32_t top = (Rdt - d_top) & m, bottom = (Rdb - d_bot) & m, middle = (Rdm - d_mid) & m;
          const uint32_t acc = pr.A * corner_t + pr.B * Rct + pr.C * Rcb + pr.D * corner_b + pr.r_y_1_i * top + pr.r_y1_i * bottom +
                               pr.r_x_1_i * strip + pr.r_x1_i * Rst + (unsigned)pr.scaling * middle;
          const int value = brisk_div_by_magic((int)acc, pr.magic, pr.shift);
          if (valid && !right) { vals[sc] = value; ksum += value; }
        }
        wave_sync();
      }
      int f = vals[lane] + ksum;
      { int f1 = f ^ 5, f2 = f + 7, f3 = f * 3; for (int k = 0; k < A.filler * cnt / 12; ++k) { f += (f >> 3) ^ k; f1 += (f1 >> 3) ^ k; f2 += (f2 >> 3) ^ k; f3 += (f3 >> 3) ^ k; } f ^= f1 ^ f2 ^ f3; }
      if (f == 0x12345678) ksum += 1;
      sum += ksum;
      __builtin_amdgcn_wave_barrier();
    }
  }
  const int tot_lo = wave_sum_i((int)(sum & 0xFFFFFF)), tot_hi = wave_sum_i((int)(sum >> 24));
  if (lane == 0) atomicAdd(A.checksum, (unsigned long long)tot_lo + ((unsigned long long)tot_hi << 24));
}


// ---- round 5: one lane per sample, but every gather instruction reads the SAME row on the two lanes of a pair -----------------
// Lane pair (2 k, 2 k + 1) = samples (a, b).  Five gathers serve a: the even lane reads a's left column pairs, the odd lane a's
// right ones; five serve b the same way.  Every lane reduces its side of both samples to six numbers, hands the partner's six over
// (one DPP each) and finishes its own sample: the ALU work of the shipped form plus the exchange, the tag look-ups of gather_lr.
__device__ __forceinline__ int dpp_even(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xA0, 0xf, 0xf, false); }  // quad_perm(0,0,2,2)
__device__ __forceinline__ int dpp_odd(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xF5, 0xf, 0xf, false); }   // quad_perm(1,1,3,3)
struct SideSix { uint32_t ct, cb, st, dt, db, dm; };
__device__ __forceinline__ SideSix side_six(u32x2 t0, u32x2 t1, u32x2 q, u32x3 b0, u32x2 b1, bool quirk, bool right) {
  constexpr uint32_t m = 0xFFFFFFu;
  uint32_t t0x, t0y, t1x, t1y, qx, qy, b0x, b0y, b1x, b1y;
  unpack2(t0, t0x, t0y); unpack2(t1, t1x, t1y); unpack2(q, qx, qy); unpack2(u32x2{b0.x, b0.y}, b0x, b0y); unpack2(b1, b1x, b1y);
  const uint32_t b0z = __builtin_amdgcn_alignbit(b0.z, b0.y, 16);
  SideSix r;
  r.ct = (t1y - t0y - t1x + t0x) & m;
  r.cb = (quirk ? (b0z - b0y - qy + qx) : (b1y - b0y - b1x + b0x)) & m;
  r.st = (b0y - b0x - t1y + t1x) & m;
  const uint32_t t0i = right ? t0x : t0y, t1i = right ? t1x : t1y, b0i = right ? b0x : b0y, b1i = right ? b1x : b1y;
  r.dt = t1i - t0i; r.db = b1i - b0i; r.dm = b0i - t1i;
  return r;
}
__global__ void __launch_bounds__(128) k_gather_sw(MbArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int* vals = reinterpret_cast<int*>(lds + wave * 1024);
  const int xcc = (int)(__builtin_amdgcn_s_getreg(GETREG_XCC_ID) & 7);
  const int np = A.np;
  const bool odd = lane & 1;
  const int rowb = A.iw * 3;
  long long sum = 0;
  for (int gi = 0; gi < 8; ++gi) {
    const int g = (xcc + gi) & 7;
    const int nt = A.ntasks[g];
    for (;;) {
      int t = 0;
      if (lane == 0) t = atomicAdd(&A.tickets[g * 32], 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= nt) break;
      const uint4 task = A.tasks[(long)g * A.max_tasks + t];
      const int cnt = (int)task.z >= 0 ? 2 : 1, total = cnt * np;
      const uint4 rec0 = A.kps[task.y], rec1 = A.kps[cnt == 2 ? task.z : task.y];
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(A.integ24 + (long)task.x * A.f24_bytes), 0, (int)A.f24_bytes, 0x00020000);
      int ksum = 0;
      for (int pass = 0; pass < 2; ++pass) {
        for (int s0 = 0; s0 < total; s0 += 64) {
          const int s = s0 + lane;
          const bool valid = s < total;
          const bool pair_valid = s0 + (lane & ~1) < total;
          const int sc = min(s, total - 1);
          const int kq = sc >= np ? 1 : 0, pt = sc - kq * np;
          const uint4 rec = kq ? rec1 : rec0;
          const int theta = pass ? (int)rec.w : 0;
          const int4 tab = A.tab4[(int)rec.z * np + pt];
          const double2 uv = A.uv2[theta * np + pt];
          const double mm = (double)__int_as_float(tab.x);
          const float xf = (float)(mm * uv.x) + __uint_as_float(rec.x), yf = (float)(mm * uv.y) + __uint_as_float(rec.y);
          const BriskBoxPrep pr = brisk_box_prep(xf, yf, __int_as_float(tab.y), tab.z, tab.w);
          const int o_t = pr.y_top * rowb, o_b = pr.y_bottom * rowb;
          const int o_tl = o_t + pr.x_left * 3, o_tr = o_t + pr.x_right * 3, o_bl = o_b + pr.x_left * 3, o_br = o_b + pr.x_right * 3;
          // set A serves the even lane's sample, set B the odd lane's; a lane reads its side (even: left, odd: right) of both
          // (the DPP reads are statements of their own: inside a conditional expression they would run with half of the lanes
          // switched off and read nothing from them)
          const int e_tr = dpp_even(o_tr), o_tl_ = dpp_odd(o_tl), e_br = dpp_even(o_br), o_bl_ = dpp_odd(o_bl);
          const int a_top = odd ? e_tr : o_tl, b_top = odd ? o_tr : o_tl_;
          const int a_bot = odd ? e_br : o_bl, b_bot = odd ? o_br : o_bl_;
          const int qk = pr.quirk ? 1 : 0;
          const bool a_quirk = dpp_even(qk) != 0, b_quirk = dpp_odd(qk) != 0;
          u32x2 at0 = {0, 0}, at1 = {0, 0}, aq = {0, 0}, ab1 = {0, 0}, bt0 = {0, 0}, bt1 = {0, 0}, bq = {0, 0}, bb1 = {0, 0};
          u32x3 ab0 = {0, 0, 0}, bb0 = {0, 0, 0};
          if (pair_valid) {
            at0 = __builtin_amdgcn_raw_buffer_load_b64(rs, a_top, 0, 0);
            bt0 = __builtin_amdgcn_raw_buffer_load_b64(rs, b_top, 0, 0);
            at1 = __builtin_amdgcn_raw_buffer_load_b64(rs, a_top, rowb, 0);
            bt1 = __builtin_amdgcn_raw_buffer_load_b64(rs, b_top, rowb, 0);
            aq = __builtin_amdgcn_raw_buffer_load_b64(rs, a_bot - rowb + 3, 0, 0);
            bq = __builtin_amdgcn_raw_buffer_load_b64(rs, b_bot - rowb + 3, 0, 0);
            ab0 = __builtin_amdgcn_raw_buffer_load_b96(rs, a_bot, 0, 0);
            bb0 = __builtin_amdgcn_raw_buffer_load_b96(rs, b_bot, 0, 0);
            ab1 = __builtin_amdgcn_raw_buffer_load_b64(rs, a_bot, rowb, 0);
            bb1 = __builtin_amdgcn_raw_buffer_load_b64(rs, b_bot, rowb, 0);
          }
          const SideSix sa = side_six(at0, at1, aq, ab0, ab1, a_quirk, odd), sb = side_six(bt0, bt1, bq, bb0, bb1, b_quirk, odd);
          // mine: my side of my sample; theirs: the partner's side of my sample (the partner computed it in the other set)
#define SW_X(f) const uint32_t own_##f = odd ? sb.f : sa.f, par_##f = (uint32_t)dpp_partner((int)(odd ? sa.f : sb.f));
          SW_X(ct) SW_X(cb) SW_X(st) SW_X(dt) SW_X(db) SW_X(dm)
#undef SW_X
          constexpr uint32_t m = 0xFFFFFFu;
          // right minus left: the even lane's own side is the left one
          const uint32_t tt = par_dt - own_dt, tb = par_db - own_db, tm = par_dm - own_dm;
          const uint32_t top = (odd ? 0u - tt : tt) & m, bottom = (odd ? 0u - tb : tb) & m, middle = (odd ? 0u - tm : tm) & m;
          const unsigned w_own_t = odd ? pr.B : pr.A, w_par_t = odd ? pr.A : pr.B, w_own_b = odd ? pr.C : pr.D, w_par_b = odd ? pr.D : pr.C;
          const unsigned w_own_s = odd ? pr.r_x1_i : pr.r_x_1_i, w_par_s = odd ? pr.r_x_1_i : pr.r_x1_i;
          const uint32_t acc = w_own_t * own_ct + w_par_t * par_ct + w_own_b * own_cb + w_par_b * par_cb + pr.r_y_1_i * top + pr.r_y1_i * bottom +
                               w_own_s * own_st + w_par_s * par_st + (unsigned)pr.scaling * middle;
          const int value = brisk_div_by_magic((int)acc, pr.magic, pr.shift);
          if (valid) { vals[sc] = value; ksum += value; }
        }
        wave_sync();
      }
      int f = vals[lane] + ksum;
      { int f1 = f ^ 5, f2 = f + 7, f3 = f * 3; for (int k = 0; k < A.filler * cnt / 12; ++k) { f += (f >> 3) ^ k; f1 += (f1 >> 3) ^ k; f2 += (f2 >> 3) ^ k; f3 += (f3 >> 3) ^ k; } f ^= f1 ^ f2 ^ f3; }
      if (f == 0x12345678) ksum += 1;
      sum += ksum;
      __builtin_amdgcn_wave_barrier();
    }
  }
  const int tot_lo = wave_sum_i((int)(sum & 0xFFFFFF)), tot_hi = wave_sum_i((int)(sum >> 24));
  if (lane == 0) atomicAdd(A.checksum, (unsigned long long)tot_lo + ((unsigned long long)tot_hi << 24));
}

// ---- round 6: gather_sw with a TWO-DEEP round pipeline ---------------------------------------------------------------
// The shipped loop runs a round's phases one after the other (addresses -> ten gathers -> wait -> combine -> LDS); at 1.5 waves per
// SIMD nothing covers a wave's combine stage while its gathers are out, and the kernel sits at 38 % of its VALU and of its L1
// look-up rate at the same time.  Here round r + 1's addresses and gathers are issued BEFORE round r's combine: one wave's VALU work
// overlaps its own vector memory traffic, and the keypoints in flight per XCD stay what they are (what killed "more waves").
// CROSS = false: the pipeline drains at the end of a pass (the rotated pass needs the orientation of the unrotated one);
// CROSS = true: it runs on into the task's second pass (an upper bound for a form that overlaps run k's rotated pass with run k + 1's
// unrotated one).
struct SwRound {
  unsigned A, B, C, D, r_x_1_i, r_y_1_i, r_x1_i, r_y1_i;
  int scaling, magic, shift, sc;
  bool valid, a_quirk, b_quirk;
  u32x2 at0, at1, aq, ab1, bt0, bt1, bq, bb1;
  u32x3 ab0, bb0;
};
template <bool CROSS>
__global__ void __launch_bounds__(128) k_gather_sw2(MbArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int* vals = reinterpret_cast<int*>(lds + wave * 1024);
  const int xcc = (int)(__builtin_amdgcn_s_getreg(GETREG_XCC_ID) & 7);
  const int np = A.np;
  const bool odd = lane & 1;
  const int rowb = A.iw * 3;
  long long sum = 0;
  for (int gi = 0; gi < 8; ++gi) {
    const int g = (xcc + gi) & 7;
    const int nt = A.ntasks[g];
    for (;;) {
      int t = 0;
      if (lane == 0) t = atomicAdd(&A.tickets[g * 32], 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= nt) break;
      const uint4 task = A.tasks[(long)g * A.max_tasks + t];
      const int cnt = (int)task.z >= 0 ? 2 : 1, total = cnt * np;
      const uint4 rec0 = A.kps[task.y], rec1 = A.kps[cnt == 2 ? task.z : task.y];
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(A.integ24 + (long)task.x * A.f24_bytes), 0, (int)A.f24_bytes, 0x00020000);
      int ksum = 0;
      const int nr = (total + 63) >> 6;  // rounds per pass (wave-uniform): 2 or 3
      auto issue = [&](int pass, int r) {
        SwRound R;
        const int s0 = r * 64, s = s0 + lane;
        R.valid = s < total;
        const bool pair_valid = s0 + (lane & ~1) < total;
        const int sc = min(s, total - 1);
        R.sc = sc;
        const int kq = sc >= np ? 1 : 0, pt = sc - kq * np;
        const uint4 rec = kq ? rec1 : rec0;
        const int theta = pass ? (int)rec.w : 0;
        const int4 tab = A.tab4[(int)rec.z * np + pt];
        const double2 uv = A.uv2[theta * np + pt];
        const double mm = (double)__int_as_float(tab.x);
        const float xf = (float)(mm * uv.x) + __uint_as_float(rec.x), yf = (float)(mm * uv.y) + __uint_as_float(rec.y);
        const BriskBoxPrep pr = brisk_box_prep(xf, yf, __int_as_float(tab.y), tab.z, tab.w);
        R.A = pr.A; R.B = pr.B; R.C = pr.C; R.D = pr.D; R.r_x_1_i = pr.r_x_1_i; R.r_y_1_i = pr.r_y_1_i; R.r_x1_i = pr.r_x1_i; R.r_y1_i = pr.r_y1_i;
        R.scaling = pr.scaling; R.magic = pr.magic; R.shift = pr.shift;
        const int o_t = pr.y_top * rowb, o_b = pr.y_bottom * rowb;
        const int o_tl = o_t + pr.x_left * 3, o_tr = o_t + pr.x_right * 3, o_bl = o_b + pr.x_left * 3, o_br = o_b + pr.x_right * 3;
        const int e_tr = dpp_even(o_tr), o_tl_ = dpp_odd(o_tl), e_br = dpp_even(o_br), o_bl_ = dpp_odd(o_bl);
        // (lanes beyond the run: an offset beyond the buffer - the load returns 0 without an access, and no branch splits the burst)
        const int oob = pair_valid ? 0 : 0x40000000;
        const int a_top = (odd ? e_tr : o_tl) | oob, b_top = (odd ? o_tr : o_tl_) | oob;
        const int a_bot = (odd ? e_br : o_bl) | oob, b_bot = (odd ? o_br : o_bl_) | oob;
        const int qk = pr.quirk ? 1 : 0;
        R.a_quirk = dpp_even(qk) != 0; R.b_quirk = dpp_odd(qk) != 0;
        R.at0 = __builtin_amdgcn_raw_buffer_load_b64(rs, a_top, 0, 0);
        R.bt0 = __builtin_amdgcn_raw_buffer_load_b64(rs, b_top, 0, 0);
        R.at1 = __builtin_amdgcn_raw_buffer_load_b64(rs, a_top, rowb, 0);
        R.bt1 = __builtin_amdgcn_raw_buffer_load_b64(rs, b_top, rowb, 0);
        R.aq = __builtin_amdgcn_raw_buffer_load_b64(rs, a_bot - rowb + 3, 0, 0);
        R.bq = __builtin_amdgcn_raw_buffer_load_b64(rs, b_bot - rowb + 3, 0, 0);
        R.ab0 = __builtin_amdgcn_raw_buffer_load_b96(rs, a_bot, 0, 0);
        R.bb0 = __builtin_amdgcn_raw_buffer_load_b96(rs, b_bot, 0, 0);
        R.ab1 = __builtin_amdgcn_raw_buffer_load_b64(rs, a_bot, rowb, 0);
        R.bb1 = __builtin_amdgcn_raw_buffer_load_b64(rs, b_bot, rowb, 0);
        return R;
      };
      auto finish = [&](const SwRound& R) {
        const SideSix sa = side_six(R.at0, R.at1, R.aq, R.ab0, R.ab1, R.a_quirk, odd), sb = side_six(R.bt0, R.bt1, R.bq, R.bb0, R.bb1, R.b_quirk, odd);
#define SW_X(f) const uint32_t own_##f = odd ? sb.f : sa.f, par_##f = (uint32_t)dpp_partner((int)(odd ? sa.f : sb.f));
        SW_X(ct) SW_X(cb) SW_X(st) SW_X(dt) SW_X(db) SW_X(dm)
#undef SW_X
        constexpr uint32_t m = 0xFFFFFFu;
        const uint32_t tt = par_dt - own_dt, tb = par_db - own_db, tm = par_dm - own_dm;
        const uint32_t top = (odd ? 0u - tt : tt) & m, bottom = (odd ? 0u - tb : tb) & m, middle = (odd ? 0u - tm : tm) & m;
        const unsigned w_own_t = odd ? R.B : R.A, w_par_t = odd ? R.A : R.B, w_own_b = odd ? R.C : R.D, w_par_b = odd ? R.D : R.C;
        const unsigned w_own_s = odd ? R.r_x1_i : R.r_x_1_i, w_par_s = odd ? R.r_x_1_i : R.r_x1_i;
        const uint32_t acc = w_own_t * own_ct + w_par_t * par_ct + w_own_b * own_cb + w_par_b * par_cb + R.r_y_1_i * top + R.r_y1_i * bottom +
                             w_own_s * own_st + w_par_s * par_st + (unsigned)R.scaling * middle;
        const int value = brisk_div_by_magic((int)acc, R.magic, R.shift);
        if (R.valid) { vals[R.sc] = value; ksum += value; }
      };
      if (!CROSS) {
        for (int pass = 0; pass < 2; ++pass) {
          const SwRound R0 = issue(pass, 0);
          const SwRound R1 = issue(pass, 1);
          finish(R0);
          if (nr == 3) {
            const SwRound R2 = issue(pass, 2);
            finish(R1);
            finish(R2);
          } else {
            finish(R1);
          }
          wave_sync();
        }
      } else {
        // the task's 2 nr rounds as one sequence, two in flight
        SwRound Ra = issue(0, 0);
        const int nseq = 2 * nr;
        for (int q = 1; q < nseq; q += 2) {
          const SwRound Rb = issue(q >= nr ? 1 : 0, q >= nr ? q - nr : q);
          finish(Ra);
          if (q + 1 < nseq) Ra = issue(q + 1 >= nr ? 1 : 0, q + 1 >= nr ? q + 1 - nr : q + 1);
          finish(Rb);
        }
        if (nseq & 1) finish(Ra);  // (never: nseq is even)
        wave_sync();
      }
      int f = vals[lane] + ksum;
      { int f1 = f ^ 5, f2 = f + 7, f3 = f * 3; for (int k = 0; k < A.filler * cnt / 12; ++k) { f += (f >> 3) ^ k; f1 += (f1 >> 3) ^ k; f2 += (f2 >> 3) ^ k; f3 += (f3 >> 3) ^ k; } f ^= f1 ^ f2 ^ f3; }
      if (f == 0x12345678) ksum += 1;
      sum += ksum;
      __builtin_amdgcn_wave_barrier();
    }
  }
  const int tot_lo = wave_sum_i((int)(sum & 0xFFFFFF)), tot_hi = wave_sum_i((int)(sum >> 24));
  if (lane == 0) atomicAdd(A.checksum, (unsigned long long)tot_lo + ((unsigned long long)tot_hi << 24));
}

// ---- round 5, second half: the 3-byte integral image with its rows interleaved in pairs --------------------------------
// Element (y, x) at byte ((y >> 1) * 2 iw + 2 x + (y & 1)) * 3: a 128-byte line holds 21 columns of TWO rows.  The 2 x 2 top
// block of a box side is one 12-byte gather when y_top is even, two when it is odd; the bottom rows y_bottom - 1 .. + 1 are
// always two gathers (16 + 12 bytes): 8 gather instructions per sample instead of 10 (one of them disabled - out-of-range
// offset, no access - in half of the lanes), 7 lines instead of 10 where nothing is shared.
struct RawIl2 {
  u32x3 a[2], b[2], d[2];
  u32x4 c[2];
};
__device__ __forceinline__ void il2_side(RawIl2& r, int s, __amdgpu_buffer_rsrc_t rs, int rowb2, int x, int yt, int yb) {
  const int xo = x * 6;
  const int kt = yt >> 1, kb = (yb - 1) >> 1;
  const bool todd = yt & 1, bodd = (yb - 1) & 1;
  const int ot = kt * rowb2 + xo, ob = kb * rowb2 + xo;
  r.a[s] = __builtin_amdgcn_raw_buffer_load_b96(rs, ot + (todd ? 3 : 0), 0, 0);
  r.b[s] = __builtin_amdgcn_raw_buffer_load_b96(rs, todd ? ot + rowb2 : 0x7FFF0000, 0, 0);
  r.c[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, bodd ? ob + rowb2 : ob + 3, 0, 0);
  r.d[s] = __builtin_amdgcn_raw_buffer_load_b96(rs, bodd ? ob + 9 : ob + rowb2, 0, 0);
}
__device__ __forceinline__ void il2_unpack(const RawIl2& r, int s, bool todd, bool bodd, uint32_t& v00, uint32_t& v01, uint32_t& v10,
                                           uint32_t& v11, uint32_t& i20, uint32_t& i21, uint32_t& i22, uint32_t& i30, uint32_t& i31,
                                           uint32_t& q0, uint32_t& q1) {
  const u32x3 a = r.a[s], b = r.b[s], d = r.d[s];
  const u32x4 c = r.c[s];
  v00 = a.x;
  v01 = __builtin_amdgcn_alignbit(a.z, a.y, 16);
  const uint32_t a3 = __builtin_amdgcn_alignbit(a.y, a.x, 24), a9 = a.z >> 8;
  const uint32_t b6 = __builtin_amdgcn_alignbit(b.z, b.y, 16);
  v10 = todd ? b.x : a3;
  v11 = todd ? b6 : a9;
  i20 = c.x;
  i21 = __builtin_amdgcn_alignbit(c.z, c.y, 16);
  i22 = c.w;
  const uint32_t c3 = __builtin_amdgcn_alignbit(c.y, c.x, 24), c9 = __builtin_amdgcn_alignbit(c.w, c.z, 8);
  const uint32_t d0 = d.x, d6 = __builtin_amdgcn_alignbit(d.z, d.y, 16);
  q0 = bodd ? d0 : c3;
  q1 = bodd ? d6 : c9;
  i30 = bodd ? c3 : d0;
  i31 = bodd ? c9 : d6;
}
__global__ void __launch_bounds__(128) k_gather_il2(MbArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int* vals = reinterpret_cast<int*>(lds + wave * 1024);
  const int xcc = (int)(__builtin_amdgcn_s_getreg(GETREG_XCC_ID) & 7);
  const int np = A.np;
  const int rowb2 = A.iw * 6;
  long long sum = 0;
  for (int gi = 0; gi < 8; ++gi) {
    const int g = (xcc + gi) & 7;
    const int nt = A.ntasks[g];
    for (;;) {
      int t = 0;
      if (lane == 0) t = atomicAdd(&A.tickets[g * 32], 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= nt) break;
      const uint4 task = A.tasks[(long)g * A.max_tasks + t];
      const int cnt = (int)task.z >= 0 ? 2 : 1, total = cnt * np;
      const uint4 rec0 = A.kps[task.y], rec1 = A.kps[cnt == 2 ? task.z : task.y];
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(A.integ_il2 + (long)task.x * A.il2_bytes), 0, (int)A.il2_bytes, 0x00020000);
      int ksum = 0;
      for (int pass = 0; pass < 2; ++pass) {
        for (int s0 = 0; s0 < total; s0 += 64) {
          const int s = s0 + lane;
          const bool valid = s < total;
          const int sc = min(s, total - 1);
          const int kq = sc >= np ? 1 : 0, pt = sc - kq * np;
          const uint4 rec = kq ? rec1 : rec0;
          const int theta = pass ? (int)rec.w : 0;
          const int4 tab = A.tab4[(int)rec.z * np + pt];
          const double2 uv = A.uv2[theta * np + pt];
          const double mm = (double)__int_as_float(tab.x);
          const float xf = (float)(mm * uv.x) + __uint_as_float(rec.x), yf = (float)(mm * uv.y) + __uint_as_float(rec.y);
          const BriskBoxPrep pr = brisk_box_prep(xf, yf, __int_as_float(tab.y), tab.z, tab.w);
          RawIl2 raw;
          if (valid) {
            il2_side(raw, 0, rs, rowb2, pr.x_left, pr.y_top, pr.y_bottom);
            il2_side(raw, 1, rs, rowb2, pr.x_right, pr.y_top, pr.y_bottom);
          }
          const bool todd = pr.y_top & 1, bodd = (pr.y_bottom - 1) & 1;
          uint32_t i00, i01, i02, i03, i10, i11, i12, i13, i20, i21, i2x, i22, i23, i2y, i30, i31, i32, i33, ql0, ql1, qr0, qr1;
          il2_unpack(raw, 0, todd, bodd, i00, i01, i10, i11, i20, i21, i2x, i30, i31, ql0, ql1);
          il2_unpack(raw, 1, todd, bodd, i02, i03, i12, i13, i22, i23, i2y, i32, i33, qr0, qr1);
          constexpr uint32_t mask = 0xFFFFFFu;
          const unsigned qbr = (i2y - i23 - qr1 + qr0) & mask;
          const unsigned qbl = (i2x - i21 - ql1 + ql0) & mask;
          const uint32_t acc = brisk_box_acc(pr, i00, i01, i02, i03, i10, i11, i12, i13, i20, i21, i22, i23, i30, i31, i32, i33, qbr, qbl, mask);
          const int value = brisk_div_by_magic((int)acc, pr.magic, pr.shift);
          if (valid) { vals[sc] = value; ksum += value; }
        }
        wave_sync();
      }
      int f = vals[lane] + ksum;
      { int f1 = f ^ 5, f2 = f + 7, f3 = f * 3; for (int k = 0; k < A.filler * cnt / 12; ++k) { f += (f >> 3) ^ k; f1 += (f1 >> 3) ^ k; f2 += (f2 >> 3) ^ k; f3 += (f3 >> 3) ^ k; } f ^= f1 ^ f2 ^ f3; }
      if (f == 0x12345678) ksum += 1;
      sum += ksum;
      __builtin_amdgcn_wave_barrier();
    }
  }
  const int tot_lo = wave_sum_i((int)(sum & 0xFFFFFF)), tot_hi = wave_sum_i((int)(sum >> 24));
  if (lane == 0) atomicAdd(A.checksum, (unsigned long long)tot_lo + ((unsigned long long)tot_hi << 24));
}

// ---- LDS patch variants ------------------------------------------------------------------------------------------
// patch geometry of a keypoint with border b (sizeList_[scale]): integral columns x0 .. x0 + pw - 1, rows y0 .. y0 + ph - 1
// with x0 = int(kx) - b, y0 = int(ky) - b; every sample needs columns >= x0 + 1 and <= x0 + 2 b + 2 (the three-wide
// bottom reads included), rows >= y0 + 1 and <= y0 + 2 b + 1; pw is padded to a multiple of 4 beyond 2 b + 6 (the aligned
// dword pairs of the 16-bit form reach 3 elements past an even column)
__host__ __device__ inline int patch_pw(int b) { return (2 * b + 8 + 3) & ~3; }
__host__ __device__ inline int patch_ph(int b) { return 2 * b + 3; }

// SRC 0 = u32 integral, 1 = 3-byte integral, 2 = pixels
template <int SRC, bool U16>
__device__ __forceinline__ void stage_patch(const MbArgs& A, int frame, unsigned char* patch, int x0, int y0, int pw, int ph, int lane) {
  const int pitchB = pw * (U16 ? 2 : 4);
  if (SRC == 2) {
    // pixels -> local integral (origin at the patch corner: every four-corner difference equals the global one).
    // lane l owns pixel columns 2 l - 1, 2 l (relative) = integral columns 2 l, 2 l + 1 of the NEXT row
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(A.pix + (long)frame * A.pix_bytes), 0, (int)A.pix_bytes, 0x00020000);
    const bool act = 2 * lane < pw;
    uint32_t c0 = 0, c1 = 0;
    if (act) {
      if (U16) *reinterpret_cast<uint32_t*>(patch + lane * 4) = 0;
      else *reinterpret_cast<u32x2*>(patch + lane * 8) = u32x2{0, 0};
    }
    const int colb = x0 + 2 * lane - 1;
    for (int r = 0; r + 1 < ph; r += 2) {
      // two pixel rows per scan: row sums stay below 2^15, so two of them share a register
      const int o0 = (y0 + r) * A.w + colb;
      uint32_t a0 = 0, b0 = 0, a1 = 0, b1 = 0;
      if (act) {
        if (lane) { a0 = __builtin_amdgcn_raw_buffer_load_b8(rs, o0, 0, 0); a1 = __builtin_amdgcn_raw_buffer_load_b8(rs, o0 + A.w, 0, 0); }
        b0 = __builtin_amdgcn_raw_buffer_load_b8(rs, o0 + 1, 0, 0);
        b1 = __builtin_amdgcn_raw_buffer_load_b8(rs, o0 + 1 + A.w, 0, 0);
      }
      int v = (int)((a0 + b0) | ((a1 + b1) << 16));
      v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
      v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
      v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
      v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
      v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
      v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
      const uint32_t S0 = (uint32_t)v & 0xFFFFu, S1 = (uint32_t)v >> 16;
      c0 += S0 - b0; c1 += S0;
      if (act) {
        if (U16) *reinterpret_cast<uint32_t*>(patch + (r + 1) * pitchB + lane * 4) = (c0 & 0xFFFFu) | (c1 << 16);
        else *reinterpret_cast<u32x2*>(patch + (r + 1) * pitchB + lane * 8) = u32x2{c0, c1};
      }
      c0 += S1 - b1; c1 += S1;
      if (act && r + 2 < ph) {
        if (U16) *reinterpret_cast<uint32_t*>(patch + (r + 2) * pitchB + lane * 4) = (c0 & 0xFFFFu) | (c1 << 16);
        else *reinterpret_cast<u32x2*>(patch + (r + 2) * pitchB + lane * 8) = u32x2{c0, c1};
      }
    }
    return;
  }
  // integral rows copied: a lane moves 4 elements, 64 / (pw / 4) rows per instruction, 8 instructions in flight
  const int nl = pw >> 2;
  const int rpi = 64 / nl;  // (pw <= 256)
  const int lr = lane / nl, lc = lane - lr * nl;
  const bool lact = lr < rpi;
  const __amdgpu_buffer_rsrc_t rs =
      SRC == 1 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(A.integ24 + (long)frame * A.f24_bytes), 0, (int)A.f24_bytes, 0x00020000)
               : __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(A.integ32 + (long)frame * A.f32_elems), 0, (int)(A.f32_elems * 4), 0x00020000);
  constexpr int ES = SRC == 1 ? 3 : 4;
  constexpr int NB = 8;
  const int gbase = ((y0 + lr) * A.iw + x0 + 4 * lc) * ES;
  const int gstep = rpi * A.iw * ES;
  const int lbase = lr * pitchB + lc * (U16 ? 8 : 16);
  const int lstep = rpi * pitchB;
  for (int r0 = 0, blk = 0; r0 < ph; r0 += rpi * NB, blk += NB) {
    u32x4 d[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int r = r0 + k * rpi + lr;
      const int go = (lact && r < ph) ? gbase + (blk + k) * gstep : 0x7FFF0000;  // (no branch around a load, see k_lds3)
      if (SRC == 1) { const u32x3 t = __builtin_amdgcn_raw_buffer_load_b96(rs, go, 0, 0); d[k] = u32x4{t.x, t.y, t.z, 0}; }
      else d[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, go, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int r = r0 + k * rpi + lr;
      if (lact && r < ph) {
        unsigned char* dst = patch + lbase + (blk + k) * lstep;
        if (U16) {
          u32x2 o;
          if (SRC == 1) { o.x = __builtin_amdgcn_perm(d[k].y, d[k].x, 0x04030100u); o.y = __builtin_amdgcn_perm(d[k].z, d[k].y, 0x06050302u); }
          else { o.x = __builtin_amdgcn_perm(d[k].y, d[k].x, 0x05040100u); o.y = __builtin_amdgcn_perm(d[k].w, d[k].z, 0x05040100u); }
          *reinterpret_cast<u32x2*>(dst) = o;
        } else {
          u32x4 o = d[k];
          if (SRC == 1) {
            o.x = d[k].x; o.y = __builtin_amdgcn_alignbit(d[k].y, d[k].x, 24); o.z = __builtin_amdgcn_alignbit(d[k].z, d[k].y, 16); o.w = d[k].z >> 8;
          }
          *reinterpret_cast<u32x4*>(dst) = o;
        }
      }
    }
  }
}

template <bool U16, uint32_t MASK>
__device__ __forceinline__ int lds_sample(const unsigned char* patch, int pitchB, int x0, int y0, const BriskBoxPrep& p) {
  const int cxl = p.x_left - x0, cxr = p.x_right - x0, ryt = p.y_top - y0, ryb = p.y_bottom - y0;
  uint32_t i00, i01, i02, i03, i10, i11, i12, i13, i20, i21, i2x, i22, i23, i2y, i30, i31, i32, i33, ql0, ql1, qr0, qr1;
  if (U16) {
    const int shl = (cxl & 1) * 16, shr = (cxr & 1) * 16;
    const unsigned char* aL = patch + (cxl >> 1) * 4 + ryt * pitchB;
    const unsigned char* aR = patch + (cxr >> 1) * 4 + ryt * pitchB;
    const int db = (ryb - ryt) * pitchB;
    const u32x2 L0 = *reinterpret_cast<const u32x2a4*>(aL), R0 = *reinterpret_cast<const u32x2a4*>(aR);
    const u32x2 L1 = *reinterpret_cast<const u32x2a4*>(aL + pitchB), R1 = *reinterpret_cast<const u32x2a4*>(aR + pitchB);
    const u32x2 Lq = *reinterpret_cast<const u32x2a4*>(aL + db - pitchB), Rq = *reinterpret_cast<const u32x2a4*>(aR + db - pitchB);
    const u32x2 L2 = *reinterpret_cast<const u32x2a4*>(aL + db), R2 = *reinterpret_cast<const u32x2a4*>(aR + db);
    const u32x2 L3 = *reinterpret_cast<const u32x2a4*>(aL + db + pitchB), R3 = *reinterpret_cast<const u32x2a4*>(aR + db + pitchB);
    // elements x, x + 1 of a row = the 32 bits at bit offset (x & 1) * 16 of its two dwords; only the low 16 bits of
    // each operand matter (brisk_box_acc masks every difference)
    i00 = __builtin_amdgcn_alignbit(L0.y, L0.x, shl); i01 = i00 >> 16;
    i02 = __builtin_amdgcn_alignbit(R0.y, R0.x, shr); i03 = i02 >> 16;
    i10 = __builtin_amdgcn_alignbit(L1.y, L1.x, shl); i11 = i10 >> 16;
    i12 = __builtin_amdgcn_alignbit(R1.y, R1.x, shr); i13 = i12 >> 16;
    i20 = __builtin_amdgcn_alignbit(L2.y, L2.x, shl); i21 = i20 >> 16; i2x = L2.y >> shl;
    i22 = __builtin_amdgcn_alignbit(R2.y, R2.x, shr); i23 = i22 >> 16; i2y = R2.y >> shr;
    i30 = __builtin_amdgcn_alignbit(L3.y, L3.x, shl); i31 = i30 >> 16;
    i32 = __builtin_amdgcn_alignbit(R3.y, R3.x, shr); i33 = i32 >> 16;
    ql0 = __builtin_amdgcn_alignbit(Lq.y, Lq.x, shl) >> 16; ql1 = Lq.y >> shl;
    qr0 = __builtin_amdgcn_alignbit(Rq.y, Rq.x, shr) >> 16; qr1 = Rq.y >> shr;
  } else {
    const unsigned char* aL = patch + cxl * 4 + ryt * pitchB;
    const unsigned char* aR = patch + cxr * 4 + ryt * pitchB;
    const int db = (ryb - ryt) * pitchB;
    const u32x2 L0 = *reinterpret_cast<const u32x2a4*>(aL), R0 = *reinterpret_cast<const u32x2a4*>(aR);
    const u32x2 L1 = *reinterpret_cast<const u32x2a4*>(aL + pitchB), R1 = *reinterpret_cast<const u32x2a4*>(aR + pitchB);
    const u32x2 Lq = *reinterpret_cast<const u32x2a4*>(aL + db - pitchB + 4), Rq = *reinterpret_cast<const u32x2a4*>(aR + db - pitchB + 4);
    const u32x2 L2 = *reinterpret_cast<const u32x2a4*>(aL + db), R2 = *reinterpret_cast<const u32x2a4*>(aR + db);
    const uint32_t L2z = *reinterpret_cast<const uint32_t*>(aL + db + 8), R2z = *reinterpret_cast<const uint32_t*>(aR + db + 8);
    const u32x2 L3 = *reinterpret_cast<const u32x2a4*>(aL + db + pitchB), R3 = *reinterpret_cast<const u32x2a4*>(aR + db + pitchB);
    i00 = L0.x; i01 = L0.y; i02 = R0.x; i03 = R0.y; i10 = L1.x; i11 = L1.y; i12 = R1.x; i13 = R1.y;
    i20 = L2.x; i21 = L2.y; i2x = L2z; i22 = R2.x; i23 = R2.y; i2y = R2z;
    i30 = L3.x; i31 = L3.y; i32 = R3.x; i33 = R3.y; ql0 = Lq.x; ql1 = Lq.y; qr0 = Rq.x; qr1 = Rq.y;
  }
  const unsigned qbr = (i2y - i23 - qr1 + qr0) & MASK;
  const unsigned qbl = (i2x - i21 - ql1 + ql0) & MASK;
  const uint32_t acc = brisk_box_acc(p, i00, i01, i02, i03, i10, i11, i12, i13, i20, i21, i22, i23, i30, i31, i32, i33, qbr, qbl, MASK);
  return brisk_div_by_magic((int)acc, p.magic, p.shift);
}

template <int SRC, bool U16>
__global__ void __launch_bounds__(64) k_lds(MbArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  int* vals = reinterpret_cast<int*>(lds);  // 66 (<= 128) values
  unsigned char* patch = lds + 512;
  const int xcc = (int)(__builtin_amdgcn_s_getreg(GETREG_XCC_ID) & 7);
  const int np = A.np;
  constexpr uint32_t MASK = U16 ? 0xFFFFu : (SRC == 1 ? 0xFFFFFFu : 0xFFFFFFFFu);
  long long sum = 0;
  for (int gi = 0; gi < 8; ++gi) {
    const int g = (xcc + gi) & 7;
    const int nt = A.ntasks[g];
    for (;;) {
      int t = 0;
      if (lane == 0) t = atomicAdd(&A.tickets[g * 32], 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= nt) break;
      const uint4 task = A.tasks[(long)g * A.max_tasks + t];
      const uint4 rec = A.kps[task.y];
      const float kx = __uint_as_float(rec.x), ky = __uint_as_float(rec.y);
      const int b = A.size_list[rec.z];
      const int x0 = (int)kx - b, y0 = (int)ky - b, pw = patch_pw(b), ph = patch_ph(b);
      const int pitchB = pw * (U16 ? 2 : 4);
      stage_patch<SRC, U16>(A, (int)task.x, patch, x0, y0, pw, ph, lane);
      wave_sync();
      int ksum = 0;
      for (int pass = 0; pass < 2; ++pass) {
        for (int s0 = 0; s0 < np; s0 += 64) {
          const int s = s0 + lane;
          const bool valid = s < np;
          const int pt = min(s, np - 1);
          const int theta = pass ? (int)rec.w : 0;
          const int4 tab = A.tab4[(int)rec.z * np + pt];
          const double2 uv = A.uv2[theta * np + pt];
          const double mm = (double)__int_as_float(tab.x);
          const float xf = (float)(mm * uv.x) + kx, yf = (float)(mm * uv.y) + ky;
          const BriskBoxPrep pr = brisk_box_prep(xf, yf, __int_as_float(tab.y), tab.z, tab.w);
          if (valid) {
            const int value = lds_sample<U16, MASK>(patch, pitchB, x0, y0, pr);
            vals[pt] = value;
            ksum += value;
          }
        }
        wave_sync();
      }
      int f = vals[lane] + ksum;
      { int f1 = f ^ 5, f2 = f + 7, f3 = f * 3; for (int k = 0; k < A.filler / 12; ++k) { f += (f >> 3) ^ k; f1 += (f1 >> 3) ^ k; f2 += (f2 >> 3) ^ k; f3 += (f3 >> 3) ^ k; } f ^= f1 ^ f2 ^ f3; }
      if (f == 0x12345678) ksum += 1;
      sum += ksum;
      __builtin_amdgcn_wave_barrier();
    }
  }
  const int tot_lo = wave_sum_i((int)(sum & 0xFFFFFF)), tot_hi = wave_sum_i((int)(sum >> 24));
  if (lane == 0) atomicAdd(A.checksum, (unsigned long long)tot_lo + ((unsigned long long)tot_hi << 24));
}

// ---- lds2: the 16-bit patch variant with its latencies taken out of the wave's dependency chain: tickets and keypoint
// records one keypoint ahead, the table entries of both rounds and the unrotated offsets requested in front of the
// staging loads, NB staging loads in flight; LEFT = false leaves the two 2-sample rounds out (what a formulation that
// hands points 64 / 65 to a separate, fully occupied gather round would run: the checksum then differs)
template <int SRC, int NB, bool LEFT>
__global__ void __launch_bounds__(64) k_lds2(MbArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  int* vals = reinterpret_cast<int*>(lds);
  unsigned char* patch = lds + 512;
  const int xcc = (int)(__builtin_amdgcn_s_getreg(GETREG_XCC_ID) & 7);
  const int np = A.np;
  constexpr int ES = SRC == 1 ? 3 : 4;
  long long sum = 0;
  unsigned long long ph_t[5] = {0, 0, 0, 0, 0};
  auto now = [&]() {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return t;
  };
  for (int gi = 0; gi < 8; ++gi) {
    const int g = (xcc + gi) & 7;
    const int nt = A.ntasks[g];
    auto take = [&]() {
      int t = nt;
      if (lane == 0) t = atomicAdd(&A.tickets[g * 32], 1);
      return t;
    };
    int t1 = take(), t2 = take();
    int tc = __builtin_amdgcn_readfirstlane(t1);
    uint4 task = make_uint4(0, 0, 0, 0), rec = make_uint4(0, 0, 0, 0);
    if (tc < nt) { task = A.tasks[(long)g * A.max_tasks + tc]; rec = A.kps[task.y]; }
    while (tc < nt) {
      const unsigned long long T0 = now();
      const int tn = __builtin_amdgcn_readfirstlane(t2);
      t2 = take();
      uint4 ntask = make_uint4(0, 0, 0, 0), nrec = make_uint4(0, 0, 0, 0);
      if (tn < nt) { ntask = A.tasks[(long)g * A.max_tasks + tn]; nrec = A.kps[ntask.y]; }
      const float kx = __uint_as_float(rec.x), ky = __uint_as_float(rec.y);
      const int sc = (int)rec.z, theta = (int)rec.w;
      const int b = A.size_list[sc];
      const int x0 = (int)kx - b, y0 = (int)ky - b, pw = patch_pw(b), ph = patch_ph(b);
      const int pitchB = pw * 2;
      const int ptB = min(64 + lane, np - 1);
      const int4 tabA = A.tab4[sc * np + lane], tabB = A.tab4[sc * np + ptB];
      const double2 uvA0 = A.uv2[lane], uvB0 = A.uv2[ptB];
      const unsigned long long T1 = now();
      // staging
      {
        const int nl = pw >> 2, rpi = 64 / nl;
        const int lr = lane / nl, lc = lane - lr * nl;
        const bool lact = lr < rpi;
        const __amdgpu_buffer_rsrc_t rs =
            SRC == 1 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(A.integ24 + (long)task.x * A.f24_bytes), 0, (int)A.f24_bytes, 0x00020000)
                     : __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(A.integ32 + (long)task.x * A.f32_elems), 0, (int)(A.f32_elems * 4), 0x00020000);
        const int gbase = ((y0 + lr) * A.iw + x0 + 4 * lc) * ES, gstep = rpi * A.iw * ES;
        const int lbase = lr * pitchB + lc * 8, lstep = rpi * pitchB;
        for (int r0 = 0, blk = 0; r0 < ph; r0 += rpi * NB, blk += NB) {
          u32x4 d[NB];
#pragma unroll
          for (int k = 0; k < NB; ++k) {
            const int r = r0 + k * rpi + lr;
            const int go = (lact && r < ph) ? gbase + (blk + k) * gstep : 0x7FFF0000;  // (no branch around a load, see k_lds3)
            if (SRC == 1) { const u32x3 t = __builtin_amdgcn_raw_buffer_load_b96(rs, go, 0, 0); d[k] = u32x4{t.x, t.y, t.z, 0}; }
            else d[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, go, 0, 0);
          }
#pragma unroll
          for (int k = 0; k < NB; ++k) {
            const int r = r0 + k * rpi + lr;
            if (lact && r < ph) {
              u32x2 o;
              if (SRC == 1) { o.x = __builtin_amdgcn_perm(d[k].y, d[k].x, 0x04030100u); o.y = __builtin_amdgcn_perm(d[k].z, d[k].y, 0x06050302u); }
              else { o.x = __builtin_amdgcn_perm(d[k].y, d[k].x, 0x05040100u); o.y = __builtin_amdgcn_perm(d[k].w, d[k].z, 0x05040100u); }
              *reinterpret_cast<u32x2*>(patch + lbase + (blk + k) * lstep) = o;
            }
          }
        }
      }
      wave_sync();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long T2 = now();
      int ksum = 0;
      auto sample = [&](const int4& tab, const double2& uv, int pt, bool valid) {
        const double mm = (double)__int_as_float(tab.x);
        const float xf = (float)(mm * uv.x) + kx, yf = (float)(mm * uv.y) + ky;
        const BriskBoxPrep pr = brisk_box_prep(xf, yf, __int_as_float(tab.y), tab.z, tab.w);
        if (valid) {
          const int value = lds_sample<true, 0xFFFFu>(patch, pitchB, x0, y0, pr);
          vals[pt] = value;
          ksum += value;
        }
      };
      sample(tabA, uvA0, lane, true);
      if (LEFT) sample(tabB, uvB0, ptB, 64 + lane < np);
      wave_sync();
      const unsigned long long T3 = now();
      const double2 uvA1 = A.uv2[theta * np + lane], uvB1 = A.uv2[theta * np + ptB];  // (theta: known after the first pass)
      int f = vals[lane] + ksum;
      { int f1 = f ^ 5, f2 = f + 7, f3 = f * 3; for (int k = 0; k < A.filler / 24; ++k) { f += (f >> 3) ^ k; f1 += (f1 >> 3) ^ k; f2 += (f2 >> 3) ^ k; f3 += (f3 >> 3) ^ k; } f ^= f1 ^ f2 ^ f3; }
      const unsigned long long T4 = now();
      sample(tabA, uvA1, lane, true);
      if (LEFT) sample(tabB, uvB1, ptB, 64 + lane < np);
      wave_sync();
      const unsigned long long T5 = now();
      f += vals[lane];
      { int f1 = f ^ 5, f2 = f + 7, f3 = f * 3; for (int k = 0; k < A.filler / 24; ++k) { f += (f >> 3) ^ k; f1 += (f1 >> 3) ^ k; f2 += (f2 >> 3) ^ k; f3 += (f3 >> 3) ^ k; } f ^= f1 ^ f2 ^ f3; }
      if (f == 0x12345678) ksum += 1;
      sum += ksum;
      __builtin_amdgcn_wave_barrier();
      tc = tn; task = ntask; rec = nrec;
      const unsigned long long T6 = now();
      ph_t[0] += T1 - T0; ph_t[1] += T2 - T1; ph_t[2] += T3 - T2; ph_t[3] += T5 - T4; ph_t[4] += (T4 - T3) + (T6 - T5);
    }
  }
  const int tot_lo = wave_sum_i((int)(sum & 0xFFFFFF)), tot_hi = wave_sum_i((int)(sum >> 24));
  if (lane == 0) {
    atomicAdd(A.checksum, (unsigned long long)tot_lo + ((unsigned long long)tot_hi << 24));
    for (int k = 0; k < 5; ++k) atomicAdd(&A.phase[k], ph_t[k]);
  }
}

// ---- lds3: lds2 with the NEXT keypoint's patch rows prefetched into registers (up to 52 x 12 bytes per lane: a 7-wave
// CU has the registers) while the current keypoint is sampled from LDS: issue order per keypoint = rotated offsets, next
// table entries, next patch rows (the vector memory counter is in order: what pass 1 waits for is in front of the rows);
// the unrotated offsets live in registers for the kernel's lifetime.  filler = independent chains (ILP as in the pair code).
template <bool LEFT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_lds3(MbArgs A, const uint4* __restrict__ r_tasks, const uint4* __restrict__ r_kps, const int* __restrict__ r_size,
                                                                                          const int* __restrict__ r_ntasks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  int* vals = reinterpret_cast<int*>(lds);
  int4* ltab = reinterpret_cast<int4*>(lds + 320);       // [2] table entries of points 64, 65 (this keypoint), [2] of the next
  double2* luv = reinterpret_cast<double2*>(lds + 384);  // [2] unrotated, [2] rotated offsets of points 64, 65
  unsigned char* patch = lds + 512;
  const int xcc = (int)(__builtin_amdgcn_s_getreg(GETREG_XCC_ID) & 7);
  const int np = A.np;
  constexpr int NBT = 52;
  const int ptB = min(64 + lane, np - 1);
  const double2 uvA0 = A.uv2[lane];
  if (lane < 2) luv[lane] = A.uv2[ptB];
  long long sum = 0;
  struct Geo { int x0, y0, pitchB, ph, rpi, lr, lbase, lstep; bool lact; };
  u32x3 d[NBT];
  auto issue = [&](const uint4& task, const uint4& rec, Geo& G) {
    const int b = r_size[(int)rec.z];
    const int pw = patch_pw(b);
    G.x0 = (int)__uint_as_float(rec.x) - b; G.y0 = (int)__uint_as_float(rec.y) - b; G.pitchB = pw * 2; G.ph = patch_ph(b);
    const int nl = pw >> 2;
    G.rpi = 64 / nl;
    G.lr = lane / nl;
    const int lc = lane - G.lr * nl;
    G.lact = G.lr < G.rpi;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(A.integ24 + (long)task.x * A.f24_bytes), 0, (int)A.f24_bytes, 0x00020000);
    const int gbase = ((G.y0 + G.lr) * A.iw + G.x0 + 4 * lc) * 3, gstep = __builtin_amdgcn_readfirstlane(G.rpi * A.iw * 3);
    G.lbase = G.lr * G.pitchB + lc * 8; G.lstep = G.rpi * G.pitchB;
#pragma unroll
    for (int k = 0; k < NBT; ++k) {
      // (no branch around a load: the compiler's wait-count insertion would wait for every load at the join;
      // lanes beyond the patch read past the descriptor's end: zero, no memory access)
      d[k] = __builtin_amdgcn_raw_buffer_load_b96(rs, (G.lact && k * G.rpi + G.lr < G.ph) ? gbase : 0x7FFF0000, k * gstep, 0);
    }
  };
  auto commit = [&](const Geo& G) {
#pragma unroll
    for (int k = 0; k < NBT; ++k) {
      u32x2 o;
      o.x = __builtin_amdgcn_perm(d[k].y, d[k].x, 0x04030100u);
      o.y = __builtin_amdgcn_perm(d[k].z, d[k].y, 0x06050302u);
      const bool act = G.lact && k * G.rpi + G.lr < G.ph;
      const int off = act ? 512 + G.lbase + k * G.lstep : 448;  // (448: a dump slot)
      *reinterpret_cast<u32x2*>(lds + off) = o;
    }
  };
  for (int gi = 0; gi < 8; ++gi) {
    const int g = (xcc + gi) & 7;
    const int nt = r_ntasks[g];
    auto take = [&]() {
      int t = nt;
      if (lane == 0) t = atomicAdd(&A.tickets[g * 32], 1);
      return t;
    };
    int t1 = take(), t2 = take();
    int tc = __builtin_amdgcn_readfirstlane(t1);
    uint4 task = make_uint4(0, 0, 0, 0), rec = make_uint4(0, 0, 0, 0);
    int4 tabA = make_int4(0, 0, 0, 0);
    int tsel = 0;
    Geo G = {};
    if (tc < nt) {
      task = r_tasks[(long)g * A.max_tasks + tc]; rec = r_kps[task.y];
      tabA = A.tab4[(int)rec.z * np + lane];
      if (lane < 2) ltab[lane] = A.tab4[(int)rec.z * np + ptB];
      issue(task, rec, G);
    }
    while (tc < nt) {
      const int tn = __builtin_amdgcn_readfirstlane(t2);
      t2 = take();
      uint4 ntask = make_uint4(0, 0, 0, 0), nrec = make_uint4(0, 0, 0, 0);
      if (tn < nt) { ntask = r_tasks[(long)g * A.max_tasks + tn]; nrec = r_kps[ntask.y]; }
      const float kx = __uint_as_float(rec.x), ky = __uint_as_float(rec.y);
      const int theta = (int)rec.w;
      commit(G);
      wave_sync();
      const int x0 = G.x0, y0 = G.y0, pitchB = G.pitchB;
      int ksum = 0;
      auto sample = [&](const int4& tab, const double2& uv, int pt, bool valid) {
        const double mm = (double)__int_as_float(tab.x);
        const float xf = (float)(mm * uv.x) + kx, yf = (float)(mm * uv.y) + ky;
        const BriskBoxPrep pr = brisk_box_prep(xf, yf, __int_as_float(tab.y), tab.z, tab.w);
        if (valid) {
          const int value = lds_sample<true, 0xFFFFu>(patch, pitchB, x0, y0, pr);
          vals[pt] = value;
          ksum += value;
        }
      };
      sample(tabA, uvA0, lane, true);
      if (LEFT) sample(ltab[tsel + (lane & 1)], luv[lane & 1], ptB, 64 + lane < np);
      wave_sync();
      int f0 = vals[lane] + ksum, f1 = f0 ^ 5, f2 = f0 + 7, f3 = f0 * 3;
      for (int k = 0; k < A.filler / 24; ++k) { f0 += (f0 >> 3) ^ k; f1 += (f1 >> 3) ^ k; f2 += (f2 >> 3) ^ k; f3 += (f3 >> 3) ^ k; }
      const double2 uvA1 = A.uv2[theta * np + lane];  // (theta: known after the first pass)
      if (lane < 2) luv[2 + lane] = A.uv2[theta * np + ptB];
      int4 ntabA = tabA;
      Geo NG = G;
      const int4 ctabA = tabA;
      // pass 1 of this keypoint reads the patch in LDS; the next keypoint's rows travel meanwhile
      // (the registers d[] are free: committed above)
      if (tn < nt) {
        ntabA = A.tab4[(int)nrec.z * np + lane];
        if (lane < 2) ltab[(tsel ^ 2) + lane] = A.tab4[(int)nrec.z * np + ptB];
        issue(ntask, nrec, NG);
      }
      sample(ctabA, uvA1, lane, true);
      wave_sync();
      if (LEFT) sample(ltab[tsel + (lane & 1)], luv[2 + (lane & 1)], ptB, 64 + lane < np);
      wave_sync();
      f0 += vals[lane];
      for (int k = 0; k < A.filler / 24; ++k) { f0 += (f0 >> 3) ^ k; f1 += (f1 >> 3) ^ k; f2 += (f2 >> 3) ^ k; f3 += (f3 >> 3) ^ k; }
      if ((f0 ^ f1 ^ f2 ^ f3) == 0x12345678) ksum += 1;
      sum += ksum;
      __builtin_amdgcn_wave_barrier();
      tc = tn; task = ntask; rec = nrec; tabA = ntabA; tsel ^= 2; G = NG;
    }
  }
  const int tot_lo = wave_sum_i((int)(sum & 0xFFFFFF)), tot_hi = wave_sum_i((int)(sum >> 24));
  if (lane == 0) atomicAdd(A.checksum, (unsigned long long)tot_lo + ((unsigned long long)tot_hi << 24));
}

// ---- host --------------------------------------------------------------------------------------------------------
struct KpRec { float x, y; int s, t; };

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s <input.bin> [filler=300] [only=<variant substring>]\n", argv[0]); return 2; }
  const int filler = argc > 2 ? atoi(argv[2]) : 300;
  const char* only = argc > 3 ? argv[3] : "";
  FILE* fi = fopen(argv[1], "rb");
  if (!fi) { perror(argv[1]); return 1; }
  int hdr[3];
  if (fread(hdr, 4, 3, fi) != 3) return 1;
  const int nd = hdr[0], w = hdr[1], h = hdr[2];
  std::vector<std::vector<KpRec>> kp(nd);
  for (int f = 0; f < nd; ++f) {
    int n;
    if (fread(&n, 4, 1, fi) != 1) return 1;
    kp[f].resize(n);
    if (fread(kp[f].data(), sizeof(KpRec), n, fi) != (size_t)n) return 1;
  }
  std::vector<std::vector<uint8_t>> img(nd, std::vector<uint8_t>((size_t)w * h));
  for (int f = 0; f < nd; ++f)
    if (fread(img[f].data(), 1, (size_t)w * h, fi) != (size_t)w * h) return 1;
  fclose(fi);

  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const int NF = 256;
  const int iw = w + 1, ih = h + 1;
  const long f32_elems = (long)iw * ih, f24_bytes = ((f32_elems * 3 + 15) & ~15L) + 16, pix_bytes = (long)w * h;
  uint32_t* d_i32; uint8_t* d_i24; uint8_t* d_pix; uint8_t* d_il2;
  const long il2_bytes = (((long)((ih + 1) / 2) * iw * 6 + 15) & ~15L) + 32;
  CHECK(hipMalloc(&d_il2, NF * il2_bytes));
  CHECK(hipMalloc(&d_i32, NF * f32_elems * 4));
  CHECK(hipMalloc(&d_i24, NF * f24_bytes));
  CHECK(hipMalloc(&d_pix, NF * pix_bytes + 64));
  {
    std::vector<uint32_t> I(f32_elems);
    std::vector<uint8_t> I3(f24_bytes);
    std::vector<uint8_t> I2(il2_bytes);
    for (int f = 0; f < nd; ++f) {
      std::fill(I.begin(), I.end(), 0u);
      for (int y = 0; y < h; ++y) {
        uint32_t run = 0;
        for (int x = 0; x < w; ++x) {
          run += img[f][(size_t)y * w + x];
          I[(size_t)(y + 1) * iw + x + 1] = I[(size_t)y * iw + x + 1] + run;
        }
      }
      for (long i = 0; i < f32_elems; ++i) { I3[3 * i] = I[i] & 0xFF; I3[3 * i + 1] = (I[i] >> 8) & 0xFF; I3[3 * i + 2] = (I[i] >> 16) & 0xFF; }
      for (int y = 0; y < ih; ++y)
        for (int x = 0; x < iw; ++x) {
          const uint32_t v = I[(size_t)y * iw + x];
          uint8_t* q = &I2[((size_t)(y >> 1) * iw * 2 + 2 * x + (y & 1)) * 3];
          q[0] = v & 0xFF; q[1] = (v >> 8) & 0xFF; q[2] = (v >> 16) & 0xFF;
        }
      for (int s = f; s < NF; s += nd) {
        CHECK(hipMemcpy(d_il2 + s * il2_bytes, I2.data(), il2_bytes, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_i32 + s * f32_elems, I.data(), f32_elems * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_i24 + s * f24_bytes, I3.data(), f24_bytes, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_pix + s * pix_bytes, img[f].data(), pix_bytes, hipMemcpyHostToDevice));
      }
    }
  }
  BriskPatternHost H;
  std::string err;
  if (!brisk_pattern_build_default(2, 1.0f, &H, &err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
  const int np = H.npoints;
  std::vector<int> tab(64 * np * 4);
  for (int i = 0; i < 64 * np; ++i) {
    memcpy(&tab[4 * i], &H.mult[i], 4);
    memcpy(&tab[4 * i + 1], &H.sigma[i], 4);
    brisk_pack_tab(H.scaling[2 * i], H.scaling[2 * i + 1], &tab[4 * i + 2], &tab[4 * i + 3]);
  }
  int4* d_tab; double2* d_uv; int* d_size;
  CHECK(hipMalloc(&d_tab, tab.size() * 4));
  CHECK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&d_uv, H.uv.size() * 8));
  CHECK(hipMemcpy(d_uv, H.uv.data(), H.uv.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&d_size, 64 * 4));
  CHECK(hipMemcpy(d_size, H.size_list.data(), 64 * 4, hipMemcpyHostToDevice));
  // keypoint records, all distinct frames concatenated
  std::vector<uint4> recs;
  std::vector<int> base(nd);
  for (int f = 0; f < nd; ++f) {
    base[f] = (int)recs.size();
    for (const KpRec& k : kp[f]) {
      uint4 r;
      memcpy(&r.x, &k.x, 4); memcpy(&r.y, &k.y, 4); r.z = (unsigned)k.s; r.w = (unsigned)k.t;
      recs.push_back(r);
    }
  }
  uint4* d_kps;
  CHECK(hipMalloc(&d_kps, recs.size() * 16));
  CHECK(hipMemcpy(d_kps, recs.data(), recs.size() * 16, hipMemcpyHostToDevice));
  const int max_tasks = NF / 8 * 2048;
  uint4* d_tasks; int* d_ntasks; int* d_tickets; unsigned long long* d_sum;
  CHECK(hipMalloc(&d_tasks, (size_t)8 * max_tasks * 16));
  CHECK(hipMalloc(&d_ntasks, 8 * 4));
  CHECK(hipMalloc(&d_tickets, 8 * 32 * 4));
  CHECK(hipMalloc(&d_sum, 8));
  unsigned long long* d_phase;
  CHECK(hipMalloc(&d_phase, 64));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));

  MbArgs A;
  A.integ_il2 = d_il2; A.il2_bytes = il2_bytes;
  A.integ32 = d_i32; A.integ24 = d_i24; A.pix = d_pix; A.f32_elems = f32_elems; A.f24_bytes = f24_bytes; A.pix_bytes = pix_bytes;
  A.iw = iw; A.ih = ih; A.w = w; A.h = h; A.tab4 = d_tab; A.uv2 = d_uv; A.size_list = d_size; A.np = np; A.kps = d_kps;
  A.tasks = d_tasks; A.ntasks = d_ntasks; A.max_tasks = max_tasks; A.tickets = d_tickets; A.checksum = d_sum; A.filler = filler; A.phase = d_phase;

  struct Cls { const char* name; int lo, hi; };  // patch side range (lo, hi]
  const Cls classes[] = {{"side<=67", 0, 67}, {"67<side<=101", 67, 101}, {"side<=101", 0, 101}, {"101<side<=151", 101, 151},
                         {"151<side<=201", 151, 201}, {"side>201", 201, 100000}, {"all", 0, 100000}};
  struct Var { const char* name; int kind; int src; bool u16; int max_side; };
  const Var vars[] = {{"gather_sw2_drain", 12, 1, false, 100000}, {"gather_sw2_cross", 13, 1, false, 100000}, {"gather_sw", 11, 1, false, 100000}, {"gather_lr", 10, 1, false, 100000}, {"gather_il2", 9, 1, false, 100000}, {"gather_i24_whatif_8gathers", 0, 11, false, 100000},
                      {"gather_i24_whatif_30dwords", 0, 12, false, 100000}, {"gather_i24_whatif_10dwords", 0, 13, false, 100000}, {"gather_i24_whatif_nogathers", 0, 14, false, 100000},
                      {"gather_i24_whatif_l1hits", 0, 15, false, 100000}, {"gather_i24_whatif_l1hits_pairs", 0, 16, false, 100000},
                      {"gather_i24_whatif_l1hits_quads", 0, 17, false, 100000}, {"gather_i24_whatif_l1hits_16", 0, 18, false, 100000}, {"gather_i24", 0, 1, false, 100000},    {"gather_u32", 0, 0, false, 100000},   {"lds_u16_from_i24", 1, 1, true, 101},
                      {"lds_u16_from_u32", 1, 0, true, 101}, {"lds_u32_from_u32", 1, 0, false, 201}, {"lds_u32_from_i24", 1, 1, false, 201},
                      {"lds_u16_from_pix", 1, 2, true, 101}, {"lds_u32_from_pix", 1, 2, false, 101},
                      {"lds2_i24_nb13", 2, 1, true, 101}, {"lds2_i24_nb26", 3, 1, true, 101}, {"lds2_u32_nb13", 4, 0, true, 101},
                      {"lds2_i24_nb13_noleft", 5, 1, true, 101}, {"lds2_i24_nb26_noleft", 6, 1, true, 101},
                      {"lds3_i24", 7, 1, true, 101}, {"lds3_i24_noleft", 8, 1, true, 101}};
  printf("{\"device\": \"%s\", \"cus\": %d, \"frames\": %d, \"filler\": %d, \"rows\": [\n", prop.gcnArchName, ncu, NF, filler);
  bool first = true;
  for (const Cls& C : classes) {
    for (const Var& V : vars) {
      if (only[0] && !strstr(V.name, only)) continue;
      if (C.hi > V.max_side) continue;
      // tasks of this class: queue g = frames g, g + 8, ... (last first, as the engine), keypoints in processing order;
      // the gather variants take them in runs of two
      std::vector<std::vector<uint4>> tq(8);
      long nkp = 0;
      int bmax = 0;
      for (int g = 0; g < 8; ++g)
        for (int s = NF - 8 + g; s >= 0; s -= 8) {
          const int f = s % nd;
          std::vector<int> sel;
          for (int i = 0; i < (int)kp[f].size(); ++i) {
            const int b = H.size_list[kp[f][i].s], side = 2 * b + 1;
            if (side <= C.lo || side > C.hi) continue;
            if ((int)kp[f][i].x + b + 2 > w) continue;  // (the displaced-corner wrap of the last column: not part of this benchmark)
            sel.push_back(i);
            bmax = std::max(bmax, b);
          }
          nkp += (long)sel.size();
          if (V.kind == 0 || (V.kind >= 9 && V.kind <= 13)) {
            for (size_t i = 0; i < sel.size(); i += 2)
              tq[g].push_back(make_uint4((unsigned)s, (unsigned)(base[f] + sel[i]), i + 1 < sel.size() ? (unsigned)(base[f] + sel[i + 1]) : 0xFFFFFFFFu, 0));
          } else {
            for (int i : sel) tq[g].push_back(make_uint4((unsigned)s, (unsigned)(base[f] + i), 0xFFFFFFFFu, 0));
          }
        }
      int nt[8];
      for (int g = 0; g < 8; ++g) {
        nt[g] = (int)tq[g].size();
        if (nt[g] > max_tasks) { fprintf(stderr, "task overflow\n"); return 1; }
        if (nt[g]) CHECK(hipMemcpy(d_tasks + (size_t)g * max_tasks, tq[g].data(), (size_t)nt[g] * 16, hipMemcpyHostToDevice));
      }
      CHECK(hipMemcpy(d_ntasks, nt, 32, hipMemcpyHostToDevice));
      if (!nkp) continue;
      std::vector<int> wpcs;
      size_t lds = 0;
      if (V.kind == 0 || (V.kind >= 9 && V.kind <= 13)) {
        wpcs = {3};
      } else {
        lds = 512 + (size_t)patch_pw(bmax) * patch_ph(bmax) * (V.u16 ? 2 : 4);
        if (lds > 160 * 1024) continue;
        const int fit = (int)(160 * 1024 / lds);
        wpcs.push_back(std::min(fit, 16));
        if (fit > 8) wpcs.push_back(8);
      }
      for (int wpc : wpcs) {
        double best_ms = 1e30;
        unsigned long long sum = 0;
        for (int rep = 0; rep < 3; ++rep) {
          CHECK(hipMemsetAsync(d_tickets, 0, 8 * 32 * 4, 0));
          CHECK(hipMemsetAsync(d_sum, 0, 8, 0));
          CHECK(hipMemsetAsync(d_phase, 0, 64, 0));
          CHECK(hipEventRecord(e0, 0));
          if (V.kind == 0 || (V.kind >= 9 && V.kind <= 13)) {
            const size_t l = 160 * 1024 / 4 + 512;
            auto fn = V.kind == 12 ? k_gather_sw2<false> : V.kind == 13 ? k_gather_sw2<true> : V.kind == 11 ? k_gather_sw : V.kind == 10 ? k_gather_lr : V.kind == 9 ? k_gather_il2 : V.src == 11 ? k_gather<true, 1> : V.src == 12 ? k_gather<true, 2> : V.src == 13 ? k_gather<true, 3> : V.src == 14 ? k_gather<true, 4> : V.src == 15 ? k_gather<true, 5> : V.src == 16 ? k_gather<true, 6> : V.src == 17 ? k_gather<true, 7> : V.src == 18 ? k_gather<true, 8> :
                      V.src == 1 ? k_gather<true> : k_gather<false>;
            CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l));
            hipLaunchKernelGGL(fn, dim3(ncu * wpc), dim3(128), l, 0, A);
          } else {
            void (*fn)(MbArgs) = nullptr;
            if (V.src == 1 && V.u16) fn = k_lds<1, true>;
            if (V.src == 0 && V.u16) fn = k_lds<0, true>;
            if (V.src == 0 && !V.u16) fn = k_lds<0, false>;
            if (V.src == 1 && !V.u16) fn = k_lds<1, false>;
            if (V.src == 2 && V.u16) fn = k_lds<2, true>;
            if (V.src == 2 && !V.u16) fn = k_lds<2, false>;
            if (V.kind == 2) fn = k_lds2<1, 13, true>;
            if (V.kind == 3) fn = k_lds2<1, 26, true>;
            if (V.kind == 4) fn = k_lds2<0, 13, true>;
            if (V.kind == 5) fn = k_lds2<1, 13, false>;
            if (V.kind == 6) fn = k_lds2<1, 26, false>;
            if (V.kind >= 7) {
              auto f3 = V.kind == 7 ? k_lds3<true> : k_lds3<false>;
              CHECK(hipFuncSetAttribute((const void*)f3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
              hipLaunchKernelGGL(f3, dim3(ncu * wpc), dim3(64), lds, 0, A, A.tasks, A.kps, A.size_list, A.ntasks);
            } else {
              CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
              hipLaunchKernelGGL(fn, dim3(ncu * wpc), dim3(64), lds, 0, A);
            }
          }
          CHECK(hipEventRecord(e1, 0));
          CHECK(hipEventSynchronize(e1));
          CHECK(hipGetLastError());
          float ms;
          CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (rep) best_ms = std::min(best_ms, (double)ms);
          CHECK(hipMemcpy(&sum, d_sum, 8, hipMemcpyDeviceToHost));
        }
        const double samples = (double)nkp * 2 * np;
        unsigned long long php[8];
        CHECK(hipMemcpy(php, d_phase, 64, hipMemcpyDeviceToHost));
        printf("%s{\"class\": \"%s\", \"variant\": \"%s\", \"keypoints\": %ld, \"max_border\": %d, \"lds_bytes\": %zu, \"waves_per_cu\": %d, "
               "\"ms\": %.4f, \"samples_per_ns_chip\": %.2f, \"us_per_keypoint_cu\": %.3f, \"checksum\": %llu, "
               "\"wave_us_per_keypoint\": {\"ticket_params\": %.2f, \"staging\": %.2f, \"pass0\": %.2f, \"pass1\": %.2f, \"other\": %.2f}}",
               first ? "" : ",\n", C.name, V.name, nkp, bmax, lds, (V.kind == 0 || (V.kind >= 9 && V.kind <= 13)) ? wpc * 2 : wpc, best_ms, samples / (best_ms * 1e6),
               best_ms * 1e3 * ncu / (double)nkp, sum, php[0] * 0.01 / nkp, php[1] * 0.01 / nkp, php[2] * 0.01 / nkp, php[3] * 0.01 / nkp,
               php[4] * 0.01 / nkp);
        first = false;
        fflush(stdout);
      }
    }
  }
  printf("\n]}\n");
  return 0;
}
