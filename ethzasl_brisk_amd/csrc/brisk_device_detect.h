// brisk_device_detect.h - per-item device logic of the scale-space detector.
//
// Everything here is `__host__ __device__` so that the HIP kernels (brisk_kernels.hip) and the
// test-only CPU emulation harness (tests/emul) run the SAME code.  No function in this file is
// a CPU fallback of the product: the shipped library only instantiates them inside kernels.
//
// Float semantics follow the reference expressions literally (un-suffixed literals are double);
// compile with -ffp-contract=off.  Reference paths are relative to /root/reference.
#pragma once
#include "brisk_common.h"

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
BRISK_HD int brisk_min(int a, int b) { return a < b ? a : b; }
BRISK_HD int brisk_max(int a, int b) { return a > b ? a : b; }
BRISK_HD int brisk_min3(int a, int b, int c) { return brisk_min(brisk_min(a, b), c); }
BRISK_HD int brisk_max3(int a, int b, int c) { return brisk_max(brisk_max(a, b), c); }

// Block of up to 16 pre-evaluated score values, held in four 32-bit words (registers, not memory):
// the kernels evaluate a candidate's score patches with one lane per pixel and then run the scalar
// classification / refinement logic on these blocks.  A read outside the block raises *miss and the
// caller redoes the candidate with direct evaluation, so a block can never change a result.
struct BriskScoreBlock {
  uint32_t w0, w1, w2, w3;
  int x0, y0, cw, ch;  // block origin and extent (cw * ch <= 16); cw == 0: no block
};

struct BriskLayerView {
  const uint8_t* img;   // layer image
  uint16_t* smap;       // score-state map (same stride)
  int w, h, stride;
  BriskScoreBlock blk;    // brisk_V() values
  BriskScoreBlock blk58;  // brisk_V58<DIRECT>() values (3x3)
  mutable int miss;       // set to 1 when a block is enabled but does not cover an access
};

BRISK_HD void brisk_block_clear(BriskScoreBlock* b) { b->w0 = b->w1 = b->w2 = b->w3 = 0; b->x0 = b->y0 = b->cw = b->ch = 0; }
BRISK_HD void brisk_block_from_bytes(BriskScoreBlock* b, const uint8_t* v, int n, int x0, int y0, int cw, int ch) {
  uint32_t w[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) w[i >> 2] |= (uint32_t)v[i] << (8 * (i & 3));
  b->w0 = w[0]; b->w1 = w[1]; b->w2 = w[2]; b->w3 = w[3];
  b->x0 = x0; b->y0 = y0; b->cw = cw; b->ch = ch;
}
BRISK_HD int brisk_block_get(const BriskScoreBlock& b, int idx) {
  const uint32_t w = (idx < 8) ? ((idx < 4) ? b.w0 : b.w1) : ((idx < 12) ? b.w2 : b.w3);
  return (int)((w >> ((idx & 3) * 8)) & 0xFFu);
}

// Large helpers that are called from several places are kept out of line in device code (register
// pressure / code size of the refinement kernel); on the host they are ordinary static functions.
#if defined(__HIPCC__)
#ifdef BRISK_OUTLINE_HELPERS
#define BRISK_HD_OUTLINE __host__ __device__ static __attribute__((noinline))
#else
#define BRISK_HD_OUTLINE __host__ __device__ static inline
#endif
#else
#define BRISK_HD_OUTLINE static
#endif


// ---------------------------------------------------------------------------------------------
// Down-sampling (brisk/src/image-down-sampling.cc)
// ---------------------------------------------------------------------------------------------
BRISK_HD int brisk_avg(int a, int b) { return (a + b + 1) >> 1; }

// Halfsample8 (:142-392): output pixel (c, r) of a source of width sw.  Column classes by SIMD
// block position: [0,16*end) double rounded average, then 8 truncating columns if the number of
// 16-blocks is odd, then (sw%16)/2 columns with the (a+b+c+d+2)/4 rule.
BRISK_HD uint8_t brisk_half_px(const uint8_t* src, int sstride, int sw, int c, int r) {
  const uint8_t* p1 = src + (long)(2 * r) * sstride + 2 * c;
  const uint8_t* p2 = p1 + sstride;
  const int hsize = sw / 16;
  const int n1 = 16 * (hsize / 2);
  const int n2 = n1 + 8 * (hsize % 2);
  const int a = p1[0], b = p1[1], cc = p2[0], d = p2[1];
  if (c < n2) {
    const int v0 = brisk_avg(a, cc), v1 = brisk_avg(b, d);
    return (uint8_t)(c < n1 ? brisk_avg(v0, v1) : (v0 + v1) / 2);
  }
  return (uint8_t)((a + b + cc + d + 2) / 4);
}

// Twothirdsample8 (:550-787): output pixel (c, r).  SIMD blocks of 15 source / 10 output columns
// use nested rounding averages; the ((sw/3)*3)%15 tail columns use the /9 formula.
BRISK_HD uint8_t brisk_twothird_px(const uint8_t* src, int sstride, int sw, int c, int r) {
  const int t = c >> 1, cx = c & 1;   // source triple, left/right output of the pair
  const int g = r >> 1, ry = r & 1;   // source row triple, upper/lower output
  const uint8_t* pa = src + (long)(3 * g + (ry ? 2 : 0)) * sstride + 3 * t;  // outer row (A or C)
  const uint8_t* pb = src + (long)(3 * g + 1) * sstride + 3 * t;             // middle row B
  const int simd_cols = (sw / 15) * 10;
  const int o = cx ? 2 : 0;  // outer column (p0 or p2)
  if (c < simd_cols) {
    const int uo = brisk_avg(brisk_avg(pa[o], pb[o]), pa[o]);
    const int um = brisk_avg(brisk_avg(pa[1], pb[1]), pa[1]);
    return (uint8_t)brisk_avg(brisk_avg(uo, um), uo);
  }
  const unsigned A1 = pa[o], A2 = pa[1], B1 = pb[o], B2 = pb[1];
  return (uint8_t)(((4 * A1 + 2 * (A2 + B1 + 1) + B2 + 1) / 9) & 0xFF);
}

// ---------------------------------------------------------------------------------------------
// AGAST scores in closed form: the generated decision trees
// (agast/src/oast9-16.cc:100-1843, oast9-16-nms.cc:64-1962, agast5-8-nms.cc:59-336) are the
// segment test, so "corner at b" <=> M > b with
//   M = max( max_arcs min_arc(p_i - c), max_arcs min_arc(c - p_i) ).
// ---------------------------------------------------------------------------------------------

// M for OAST 9_16 from the 16 ring differences d[i] = ring_i - centre.
BRISK_HD int brisk_oast9_16_M_from_d(const int* d) {
  int lo3[16], hi3[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    lo3[i] = brisk_min3(d[i], d[(i + 1) & 15], d[(i + 2) & 15]);
    hi3[i] = brisk_max3(d[i], d[(i + 1) & 15], d[(i + 2) & 15]);
  }
  int best_bright = -256, best_dark = 256;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    best_bright = brisk_max(best_bright, brisk_min3(lo3[i], lo3[(i + 3) & 15], lo3[(i + 6) & 15]));
    best_dark = brisk_min(best_dark, brisk_max3(hi3[i], hi3[(i + 3) & 15], hi3[(i + 6) & 15]));
  }
  return brisk_max(best_bright, -best_dark);
}

// M for OAST 9_16 at p (ring order agast/include/agast/oast9-16.h:99-116).
BRISK_HD int brisk_oast9_16_M(const uint8_t* p, int s) {
  const int c = p[0];
  int d[16];
  d[0] = p[-3] - c;          d[1] = p[-3 - s] - c;      d[2] = p[-2 - 2 * s] - c;  d[3] = p[-1 - 3 * s] - c;
  d[4] = p[-3 * s] - c;      d[5] = p[1 - 3 * s] - c;   d[6] = p[2 - 2 * s] - c;   d[7] = p[3 - s] - c;
  d[8] = p[3] - c;           d[9] = p[3 + s] - c;       d[10] = p[2 + 2 * s] - c;  d[11] = p[1 + 3 * s] - c;
  d[12] = p[3 * s] - c;      d[13] = p[-1 + 3 * s] - c; d[14] = p[-2 + 2 * s] - c; d[15] = p[-3 + s] - c;
  return brisk_oast9_16_M_from_d(d);
}

// M for AGAST 5_8 from the 8 ring differences.
BRISK_HD int brisk_agast5_8_M_from_d(const int* d) {
  int best_bright = -256, best_dark = 256;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int lo = brisk_min(brisk_min3(d[i], d[(i + 1) & 7], d[(i + 2) & 7]), brisk_min(d[(i + 3) & 7], d[(i + 4) & 7]));
    const int hi = brisk_max(brisk_max3(d[i], d[(i + 1) & 7], d[(i + 2) & 7]), brisk_max(d[(i + 3) & 7], d[(i + 4) & 7]));
    best_bright = brisk_max(best_bright, lo);
    best_dark = brisk_min(best_dark, hi);
  }
  return brisk_max(best_bright, -best_dark);
}

// M for AGAST 5_8 (ring order agast/include/agast/agast5-8.h:66-75).
BRISK_HD int brisk_agast5_8_M(const uint8_t* p, int s) {
  const int c = p[0];
  int d[8];
  d[0] = p[-1] - c; d[1] = p[-1 - s] - c; d[2] = p[-s] - c; d[3] = p[1 - s] - c;
  d[4] = p[1] - c;  d[5] = p[1 + s] - c;  d[6] = p[s] - c;  d[7] = p[-1 + s] - c;
  return brisk_agast5_8_M_from_d(d);
}

// K' = clamp(M - 1, 0, 254): cornerScore(b) == max(b, K') for b >= 0
// (oast9-16-nms.cc:39-42,1964-1975: bisection on [b, 255), 255 never tested).
BRISK_HD int brisk_Kp_from_M(int M) { return brisk_min(brisk_max(M - 1, 0), 254); }

// Disc contrast = thrmap (brisk/src/brisk-layer.cc:278-598), interior pixels only.
BRISK_HD void brisk_disc_minmax(const uint8_t* p, int s, int* mn_, int* mx_) {
  int mn = 255, mx = 0;
#pragma unroll
  for (int dy = -3; dy <= 3; ++dy) {
    const int r = (dy == -3 || dy == 3) ? 1 : (dy == -2 || dy == 2) ? 2 : 3;
    for (int dx = -r; dx <= r; ++dx) {
      const int v = p[dy * s + dx];
      mn = brisk_min(mn, v);
      mx = brisk_max(mx, v);
    }
  }
  *mn_ = mn;
  *mx_ = mx;
}

// (tc * thr) / 100 without integer multiplies (quarter rate on CDNA): with k = float(thr) * 0.01f the fused
// multiply-add tc * k + 0.005 truncates to the exact quotient for every tc in [0, 255], thr in [1, 255]
// (exhaustively checked by tests/test_emul_parity.py::test_b2_fast_exact).
BRISK_HD float brisk_b2_factor(int thr) { return (float)thr * 0.01f; }
BRISK_HD int brisk_b2_fast(int tc, float k) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (int)__fmaf_rn((float)tc, k, 0.005f);
#else
  return (int)fmaf((float)tc, k, 0.005f);
#endif
}

// ---------------------------------------------------------------------------------------------
// Pre-gate of k_detect: a cheap NECESSARY condition for a detection, evaluated for two horizontally
// adjacent pixels at once in packed 16-bit lanes (v_pk_* on the device, plain C on the host).
//   * a 9-of-16 arc contains two adjacent compass points, i.e. one of {N, S} and one of {W, E} (ring radius 3);
//   * the disc contrast t (37 px, brisk-layer.cc:278-598) is at least the range t5 of {N, S, W, E}, all of
//     which lie in the disc, and the adaptive threshold b2 = clamp(t, 10, 230) * thr / 100 (oast9-16.cc:79-100)
//     is monotone in t, so b2 >= b2' := (clamp(t5, 10, 230) * K) >> s with K / 2^s <= thr / 100.
// A detection therefore implies  min(max(dN, dS), max(dW, dE)) > b2'  (bright arc) or
// max(min(dN, dS), min(dW, dE)) < -b2'  (dark arc), d* = compass pixel - centre.  About 1.3 % of the pixels of
// the benchmark frames pass; only those get the exact 37-pixel contrast and the segment test.
// ---------------------------------------------------------------------------------------------
struct BriskPregate {
  uint32_t K;      // multiplier in both 16-bit lanes
  uint32_t shift;  // shift in both 16-bit lanes
  uint32_t lower;  // lowerThreshold_ in both 16-bit lanes
};
BRISK_HD BriskPregate brisk_pregate_make(int thr, int lower_threshold = BRISK_LOWER_THRESHOLD) {
  int s = 8;
  while (s > 0 && 230 * ((thr << s) / 100) > 65535) --s;
  const uint32_t K = (uint32_t)((thr << s) / 100);
  BriskPregate g;
  g.K = K | (K << 16);
  g.shift = (uint32_t)s | ((uint32_t)s << 16);
  g.lower = (uint32_t)lower_threshold * 0x10001u;
  return g;
}

#if defined(__HIP_DEVICE_COMPILE__)
typedef short brisk_v2s __attribute__((ext_vector_type(2)));
typedef unsigned short brisk_v2u __attribute__((ext_vector_type(2)));
#define BRISK_PK_S(x) __builtin_bit_cast(brisk_v2s, (uint32_t)(x))
#define BRISK_PK_U(x) __builtin_bit_cast(brisk_v2u, (uint32_t)(x))
__device__ __forceinline__ uint32_t brisk_pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (brisk_v2s)(BRISK_PK_S(a) - BRISK_PK_S(b))); }
__device__ __forceinline__ uint32_t brisk_pk_max(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(BRISK_PK_S(a), BRISK_PK_S(b))); }
__device__ __forceinline__ uint32_t brisk_pk_min(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(BRISK_PK_S(a), BRISK_PK_S(b))); }
__device__ __forceinline__ uint32_t brisk_pk_mul(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (brisk_v2u)(BRISK_PK_U(a) * BRISK_PK_U(b))); }
__device__ __forceinline__ uint32_t brisk_pk_shr(uint32_t a, uint32_t s) { return __builtin_bit_cast(uint32_t, (brisk_v2u)(BRISK_PK_U(a) >> BRISK_PK_U(s))); }
#else
#define BRISK_PK_LANES(expr)                                                              \
  const int a0 = (int16_t)(a & 0xFFFFu), a1 = (int16_t)(a >> 16);                         \
  const int b0 = (int16_t)(b & 0xFFFFu), b1 = (int16_t)(b >> 16);                         \
  (void)a0; (void)a1; (void)b0; (void)b1;                                                 \
  return expr;
static inline uint32_t brisk_pk_pack(int lo, int hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); }
static inline uint32_t brisk_pk_sub(uint32_t a, uint32_t b) { BRISK_PK_LANES(brisk_pk_pack(a0 - b0, a1 - b1)) }
static inline uint32_t brisk_pk_max(uint32_t a, uint32_t b) { BRISK_PK_LANES(brisk_pk_pack(a0 > b0 ? a0 : b0, a1 > b1 ? a1 : b1)) }
static inline uint32_t brisk_pk_min(uint32_t a, uint32_t b) { BRISK_PK_LANES(brisk_pk_pack(a0 < b0 ? a0 : b0, a1 < b1 ? a1 : b1)) }
static inline uint32_t brisk_pk_mul(uint32_t a, uint32_t b) { return brisk_pk_pack((int)(((a & 0xFFFFu) * (b & 0xFFFFu)) & 0xFFFFu), (int)(((a >> 16) * (b >> 16)) & 0xFFFFu)); }
static inline uint32_t brisk_pk_shr(uint32_t a, uint32_t s) { return brisk_pk_pack((int)((a & 0xFFFFu) >> (s & 0xFFFFu)), (int)((a >> 16) >> (s >> 16))); }
#endif

// a against b with its 16-bit halves exchanged: (min(a.lo, b.hi), min(a.hi, b.lo)) - on the device the exchange is the
// instruction's operand-half selector (VOP3P op_sel), i.e. free
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t brisk_pk_min_sw(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_min_i16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t brisk_pk_max_sw(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_max_i16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
#else
static inline uint32_t brisk_pk_min_sw(uint32_t a, uint32_t b) { return brisk_pk_min(a, (b >> 16) | (b << 16)); }
static inline uint32_t brisk_pk_max_sw(uint32_t a, uint32_t b) { return brisk_pk_max(a, (b >> 16) | (b << 16)); }
#endif

// brisk_oast9_16_M_from_d on packed lanes: P[i] = (d[i], d[i + 8]) as signed 16-bit halves, i = 0 ... 7.  The minimum over
// the nine consecutive ring differences starting at i (and, in the other half, at i + 8) by doubling - windows of 2, 4, 8,
// then the ninth element - where an index beyond 7 is the same register with its halves exchanged: 32 packed operations per
// polarity for all 16 arcs instead of 48 three-input ones, and the maximum over the arcs on packed lanes too.
// (ring pixels i and i + 8 minus the centre in both halves: one packed lane pair of brisk_oast9_16_M_from_pk)
BRISK_HD uint32_t brisk_pk_ring_pair(uint32_t px_i, uint32_t px_i8, uint32_t centre_both) { return brisk_pk_sub(px_i | (px_i8 << 16), centre_both); }
BRISK_HD int brisk_oast9_16_M_from_pk(const uint32_t* P) {
  uint32_t lo2[8], hi2[8], lo4[8], hi4[8], lo8[8], hi8[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    lo2[i] = (i < 7) ? brisk_pk_min(P[i], P[i + 1]) : brisk_pk_min_sw(P[7], P[0]);
    hi2[i] = (i < 7) ? brisk_pk_max(P[i], P[i + 1]) : brisk_pk_max_sw(P[7], P[0]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    lo4[i] = (i < 6) ? brisk_pk_min(lo2[i], lo2[i + 2]) : brisk_pk_min_sw(lo2[i], lo2[i - 6]);
    hi4[i] = (i < 6) ? brisk_pk_max(hi2[i], hi2[i + 2]) : brisk_pk_max_sw(hi2[i], hi2[i - 6]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    lo8[i] = (i < 4) ? brisk_pk_min(lo4[i], lo4[i + 4]) : brisk_pk_min_sw(lo4[i], lo4[i - 4]);
    hi8[i] = (i < 4) ? brisk_pk_max(hi4[i], hi4[i + 4]) : brisk_pk_max_sw(hi4[i], hi4[i - 4]);
  }
  uint32_t bb = 0, bd = 0;  // running maximum of the arcs' minima / minimum of their maxima, both halves
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t lo9 = brisk_pk_min_sw(lo8[i], P[i]), hi9 = brisk_pk_max_sw(hi8[i], P[i]);  // (element i + 8)
    bb = (i == 0) ? lo9 : brisk_pk_max(bb, lo9);
    bd = (i == 0) ? hi9 : brisk_pk_min(bd, hi9);
  }
  const int best_bright = brisk_max((int)(int16_t)(bb & 0xFFFFu), (int)(int16_t)(bb >> 16));
  const int best_dark = brisk_min((int)(int16_t)(bd & 0xFFFFu), (int)(int16_t)(bd >> 16));
  return brisk_max(best_bright, -best_dark);
}

// c, n, s, w, e: centre and compass pixels (ring radius 3) of two pixels, one per 16-bit lane (values 0..255).
// Returns bit 15 of the lane (0x8000) set for a pixel that passes the pre-gate, the lane zero otherwise.
BRISK_HD uint32_t brisk_pregate_pair(uint32_t c, uint32_t n, uint32_t s, uint32_t w, uint32_t e, const BriskPregate& g) {
  // on the pixel values themselves (max(n - c, s - c) = max(n, s) - c): 18 packed operations instead of 21
  const uint32_t mxNS = brisk_pk_max(n, s), mnNS = brisk_pk_min(n, s);
  const uint32_t mxWE = brisk_pk_max(w, e), mnWE = brisk_pk_min(w, e);
  const uint32_t vb = brisk_pk_min(mxNS, mxWE);   // - c > b2: two adjacent compass points brighter
  const uint32_t vd = brisk_pk_max(mnNS, mnWE);   // c - . > b2: two adjacent compass points darker
  // range of the four compass pixels <= disc contrast (taking the centre in as well would cost two more operations
  // and lets 1.26 % instead of 1.18 % of the pixels through)
  const uint32_t t5 = brisk_pk_sub(brisk_pk_max(mxNS, mxWE), brisk_pk_min(mnNS, mnWE));
  const uint32_t upper = (uint32_t)BRISK_UPPER_THRESHOLD * 0x10001u;
  const uint32_t tc = brisk_pk_min(brisk_pk_max(t5, g.lower), upper);
  const uint32_t b2p = brisk_pk_shr(brisk_pk_mul(tc, g.K), g.shift);
  const uint32_t ev = brisk_pk_max(brisk_pk_sub(vb, c), brisk_pk_sub(c, vd));
  // passes <=> ev > b2' <=> b2' - ev < 0: the lane's sign bit (one subtraction and one 32-bit AND; a 0 / 1 result
  // would cost a compare and a select per lane)
  return brisk_pk_sub(b2p, ev) & 0x80008000u;
}

// Per-pixel detection (agast/src/oast9-16.cc:79-100 + SURVEY F5): returns D (= thrmap value) if
// (x,y) is an AGAST point at threshold thr, else 0.  Caller guarantees 3 <= x <= w-4, 3 <= y <= h-4.
BRISK_HD int brisk_detect_px(const uint8_t* p, int s, int thr, int lower = BRISK_LOWER_THRESHOLD) {
  int mn, mx;
  brisk_disc_minmax(p, s, &mn, &mx);
  const int t = mx - mn;
  const int cmp = (thr * lower) / 100;
  if (t < cmp) return 0;
  const int tc = brisk_min(brisk_max(t, lower), BRISK_UPPER_THRESHOLD);
  const int b2 = (tc * thr) / 100;
  // necessary condition: some ring pixel must differ from the centre by more than b2
  const int c = p[0];
  if (mx - c <= b2 && c - mn <= b2) return 0;
  return brisk_oast9_16_M(p, s) > b2 ? t : 0;
}

// ---------------------------------------------------------------------------------------------
// History-free score access (threshold 1): value returned by BriskLayer::GetAgastScore(x, y, 1)
// (brisk/src/brisk-layer.cc:118-132): 0 on the 3-px border, D for detections (D > 2 for
// threshold >= 20), else K'.
// ---------------------------------------------------------------------------------------------
BRISK_HD bool brisk_border3(const BriskLayerView& L, int x, int y) {
  return x < 3 || y < 3 || x >= L.w - 3 || y >= L.h - 3;
}

BRISK_HD int brisk_Kp(const BriskLayerView& L, int x, int y) {
  return brisk_Kp_from_M(brisk_oast9_16_M(L.img + (long)y * L.stride + x, L.stride));
}

BRISK_HD int brisk_V_eval(const BriskLayerView& L, int x, int y) {
  if (brisk_border3(L, x, y)) return 0;
  const int D = BRISK_SM_D(L.smap[(long)y * L.stride + x]);
  if (D > 2) return D;
  return brisk_Kp(L, x, y);
}
// ---------------------------------------------------------------------------------------------
// Literal lazy score cache (ordered path, DIRECT == 2): BriskLayer::GetAgastScore(x, y, threshold)
// (brisk/src/brisk-layer.cc:118-132) on the low byte of the score-state map, which then IS the reference's
// scores_ matrix (initial state: D at detections, 0 elsewhere).  cornerScore(b) in closed form: the bisection on
// [b, 255) (oast9-16-nms.cc:39-42, 1964-1975) never tests its lower end, so it returns max(b, min(M - 1, 254));
// for threshold 0 (b = -1) a pixel with M <= 0 therefore yields -1, stored as the uint8_t 255.
// ---------------------------------------------------------------------------------------------
BRISK_HD int brisk_S_literal(const BriskLayerView& L, int x, int y, int threshold /* uint8_t in the reference */) {
  if (x < 3 || y < 3) return 0;
  if (x >= L.w - 3 || y >= L.h - 3) return 0;
  uint16_t* sc = L.smap + (long)y * L.stride + x;
  const int m = (int)(*sc & 0xFFu);
  if (m > 2) return m;
  const int b = threshold - 1;
  const int M = brisk_oast9_16_M(L.img + (long)y * L.stride + x, L.stride);
  int score = (int)(uint8_t)brisk_max(b, brisk_min(M - 1, 254));
  if (score < threshold) score = 0;
  *sc = (uint16_t)score;
  return score;
}

// DIRECT = 1: evaluate from the image.  DIRECT = 0: read the view's pre-evaluated score block
// (kernel fast path; contains no evaluation code at all, a miss only raises the flag).
template <int DIRECT>
BRISK_HD int brisk_V(const BriskLayerView& L, int x, int y) {
  if (DIRECT == 2) return brisk_S_literal(L, x, y, 1);  // ordered path: the reference's cache, literally
  if (DIRECT) return brisk_V_eval(L, x, y);
  const unsigned ux = (unsigned)(x - L.blk.x0), uy = (unsigned)(y - L.blk.y0);
  if (ux < (unsigned)L.blk.cw && uy < (unsigned)L.blk.ch) return brisk_block_get(L.blk, (int)(uy * L.blk.cw + ux));
  L.miss = 1;
  return 0;
}

// GetAgastScore_5_8(x, y, 1) (brisk-layer.cc:134-145)
BRISK_HD int brisk_V58_eval(const BriskLayerView& L, int x, int y) {
  if (x < 2 || y < 2 || x >= L.w - 2 || y >= L.h - 2) return 0;
  return brisk_Kp_from_M(brisk_agast5_8_M(L.img + (long)y * L.stride + x, L.stride));
}
template <int DIRECT>
BRISK_HD int brisk_V58(const BriskLayerView& L, int x, int y) {
  if (DIRECT) return brisk_V58_eval(L, x, y);
  const unsigned ux = (unsigned)(x - L.blk58.x0), uy = (unsigned)(y - L.blk58.y0);
  if (L.blk58.cw && ux < 3u && uy < 3u) return brisk_block_get(L.blk58, (int)(uy * 3 + ux));
  L.miss = 1;
  return 0;
}

// Anchors of the score blocks a candidate at (x, y) on `layer` can read on the neighbouring layers
// (GetScoreMaxAbove / GetScoreMaxBelow windows incl. the 3x3 patch around their maximum and the tie rule):
// 4x4 blocks, above anchored at (int(x_1) - 1, ...), below at (int(x_1), ...).
// The window starts of brisk_score_max_other are float quotients ((float)((float)(4x - 3) / 6.0), (6x - 4) / 8.0f,
// (8x - 3) / 6.0, (6x - 2) / 4.0); only their truncations are needed here.  The exact quotients are multiples of 1/6, 1/8
// or 1/4 - never within a float rounding of the next integer - so the truncated float quotient equals the truncated
// integer quotient (checked against the float expressions for every coordinate, tests/test_emul_parity.py).
BRISK_HD void brisk_block_anchor(bool above, bool odd, int x_layer, int y_layer, int* ax, int* ay) {
  if (above) {
    if (!odd) {
      *ax = (4 * x_layer - 3) / 6 - 1;
      *ay = (4 * y_layer - 3) / 6 - 1;
    } else {
      *ax = (6 * x_layer - 4) / 8 - 1;
      *ay = (6 * y_layer - 4) / 8 - 1;
    }
  } else {
    if (!odd) {
      *ax = (8 * x_layer - 3) / 6;
      *ay = (8 * y_layer - 3) / 6;
    } else {
      *ax = (6 * x_layer - 2) / 4;
      *ay = (6 * y_layer - 2) / 4;
    }
  }
}

// Touch recorder: which pixels of the layer above a GetScoreMaxAbove call score-touches
// (4x4 block anchored at (x0, y0)); used to replay the lazy cache (event e3).
struct BriskTouch {
  int x0, y0;
  unsigned mask;
  bool on;
#ifdef CR_TIMING  // experiments (build variant): the refinement's phases on the 100 MHz clock, per lane (tools/classify_phases.py)
  long long tlast;
  int tacc[6];
#endif
};
#if defined(CR_TIMING) && defined(__HIP_DEVICE_COMPILE__)
#define BRISK_CR_T(t_, i_) { const long long now_ = (long long)wall_clock64(); (t_)->tacc[i_] += (int)(now_ - (t_)->tlast); (t_)->tlast = now_; }
#else
#define BRISK_CR_T(t_, i_)
#endif

BRISK_HD void brisk_touch(BriskTouch* t, const BriskLayerView& L, int x, int y) {
  if (!t->on || brisk_border3(L, x, y)) return;
  const int bx = x - t->x0, by = y - t->y0;
  if (bx >= 0 && bx < 4 && by >= 0 && by < 4) t->mask |= 1u << (by * 4 + bx);
}

template <int DIRECT>
BRISK_HD int brisk_Vt(const BriskLayerView& L, int x, int y, BriskTouch* t) {
  brisk_touch(t, L, x, y);
  return brisk_V<DIRECT>(L, x, y);
}

// GetAgastScore(float, float, 1) (brisk-layer.cc:147-161): bilinear blend of 4 integer scores,
// all four always evaluated (and touched), result truncated to u8.
// (the blend itself, shared with the block form of GetScoreMaxAbove below: rx1 / ry1 = fractional parts of the position)
BRISK_HD int brisk_bilinear_u8(const float rx1, const float ry1, const int s00, const int s10, const int s01, const int s11) {
  const float rx = 1.0f - rx1;
  const float ry = 1.0f - ry1;
  return (int)(uint8_t)(rx * ry * s00 + rx1 * ry * s10 + rx * ry1 * s01 + rx1 * ry1 * s11);
}
template <int DIRECT>
BRISK_HD int brisk_Vf(const BriskLayerView& L, float xf, float yf, BriskTouch* t) {
  const int x = (int)xf;
  const float rx1 = xf - (float)x;
  const int y = (int)yf;
  const float ry1 = yf - (float)y;
  const int s00 = brisk_Vt<DIRECT>(L, x, y, t);
  const int s10 = brisk_Vt<DIRECT>(L, x + 1, y, t);
  const int s01 = brisk_Vt<DIRECT>(L, x, y + 1, t);
  const int s11 = brisk_Vt<DIRECT>(L, x + 1, y + 1, t);
  return brisk_bilinear_u8(rx1, ry1, s00, s10, s01, s11);
}

// ---------------------------------------------------------------------------------------------
// Subpixel2D (brisk/src/brisk-scale-space.cc:1230-1364), including delta_y = delta_x1/2 (:1351,1355)
// ---------------------------------------------------------------------------------------------
// Least-squares quadric through the 3x3 patch, 18 q(x, y) = A x^2 + B y^2 + C x + D y + E x y + F with integer
// coefficients, maximised over [-1, 1]^2.  The coefficients are exact integers, so they are written here from the
// patch's row / column / diagonal sums; everything in floating point keeps the reference's order of operations (its
// result is the oracle), including its fall-back rules: flat Hessian -> centre; not a maximum -> best corner; maximum
// outside the square -> the better of the two edge-constrained maxima, with BOTH offsets set to that candidate's x
// offset (:1351, 1355).
struct BriskQuadric {
  int A, B, C, D, E, F;
};
BRISK_HD float brisk_quadric_at(const BriskQuadric& q, float x, float y) {
  // (/ 18.0 in double, rounded to float, in the reference: the float division gives the same float for every float
  // dividend - tools/verify_float_division.c; the dividend is a float expression there as well)
  return (q.A * x * x + q.B * y * y + q.C * x + q.D * y + q.E * x * y + q.F) / 18.0f;
}
BRISK_HD float brisk_clamp_unit(float v) { return v > 1.0f ? 1.0f : (v < -1.0f ? -1.0f : v); }

BRISK_HD_OUTLINE float brisk_subpixel2d(const int s_0_0, const int s_0_1, const int s_0_2, const int s_1_0,
                                const int s_1_1, const int s_1_2, const int s_2_0, const int s_2_1,
                                const int s_2_2, float& delta_x, float& delta_y) {
  // s_x_y: column x, row y of the patch
  const int left = s_0_0 + s_0_1 + s_0_2, right = s_2_0 + s_2_1 + s_2_2, mid_col = s_1_0 + s_1_1 + s_1_2;
  const int top = s_0_0 + s_1_0 + s_2_0, bottom = s_0_2 + s_1_2 + s_2_2, mid_row = s_0_1 + s_1_1 + s_2_1;
  BriskQuadric q;
  q.A = 3 * (left + right - 2 * mid_col);
  q.B = 3 * (top + bottom - 2 * mid_row);
  q.C = 3 * (right - left);
  q.D = 3 * (bottom - top);
  q.E = 4 * (s_0_0 - s_0_2 - s_2_0 + s_2_2);
  q.F = 2 * (5 * s_1_1 + 2 * (s_1_0 + s_0_1 + s_1_2 + s_2_1) - (s_0_0 + s_0_2 + s_2_0 + s_2_2));
  const int det = 4 * q.A * q.B - q.E * q.E;
  if (det == 0) {
    delta_x = 0.0f;
    delta_y = 0.0f;
    return (float)q.F / 18.0f;
  }
  if (!(det > 0 && q.A < 0)) {  // no maximum inside: the best of the four corners, first one wins a draw
    int best = q.C + q.D + q.E;
    delta_x = 1.0f;
    delta_y = 1.0f;
    const int sx[3] = {-1, 1, -1}, sy[3] = {1, -1, -1};
    for (int k = 0; k < 3; ++k) {
      const int v = sx[k] * q.C + sy[k] * q.D + sx[k] * sy[k] * q.E;
      if (v > best) { best = v; delta_x = (float)sx[k]; delta_y = (float)sy[k]; }
    }
    return (float)(best + q.A + q.B + q.F) / 18.0f;
  }
  delta_x = (float)(2 * q.B * q.C - q.D * q.E) / (float)(-det);
  delta_y = (float)(2 * q.A * q.D - q.C * q.E) / (float)(-det);
  // which sides of the square the unconstrained maximum violates (x: only one side is ever noted, as in the reference)
  const int out_x = delta_x > 1.0f ? 1 : (delta_x < -1.0f ? -1 : 0);
  const int out_y = delta_y > 1.0f ? 1 : (delta_y < -1.0f ? -1 : 0);
  if (out_x == 0 && out_y == 0) return brisk_quadric_at(q, delta_x, delta_y);
  float ex = 0.0f, ey = 0.0f;  // maximum along the violated vertical edge x = out_x (or the centre if none)
  if (out_x != 0) {
    ex = (float)out_x;
    ey = brisk_clamp_unit(-(float)(q.D + out_x * q.E) / (float)(2 * q.B));
  }
  float fx = 0.0f, fy = 0.0f;  // maximum along the violated horizontal edge y = out_y
  if (out_y != 0) {
    fy = (float)out_y;
    fx = brisk_clamp_unit(-(float)(q.C + out_y * q.E) / (float)(2 * q.A));
  }
  const float m_e = brisk_quadric_at(q, ex, ey), m_f = brisk_quadric_at(q, fx, fy);
  if (m_e > m_f) {
    delta_x = ex;
    delta_y = ex;  // sic (reference :1351)
    return m_e;
  }
  delta_x = fx;
  delta_y = fx;  // sic (reference :1355)
  return m_f;
}

// 3x3 patch around (x, y) with integer score access + Subpixel2D
template <int DIRECT>
BRISK_HD float brisk_patch_subpixel(const BriskLayerView& L, int x, int y, BriskTouch* t, float& dx, float& dy,
                                    int* centre) {
  const int s_0_0 = brisk_Vt<DIRECT>(L, x - 1, y - 1, t);
  const int s_1_0 = brisk_Vt<DIRECT>(L, x, y - 1, t);
  const int s_2_0 = brisk_Vt<DIRECT>(L, x + 1, y - 1, t);
  const int s_2_1 = brisk_Vt<DIRECT>(L, x + 1, y, t);
  const int s_1_1 = brisk_Vt<DIRECT>(L, x, y, t);
  const int s_0_1 = brisk_Vt<DIRECT>(L, x - 1, y, t);
  const int s_0_2 = brisk_Vt<DIRECT>(L, x - 1, y + 1, t);
  const int s_1_2 = brisk_Vt<DIRECT>(L, x, y + 1, t);
  const int s_2_2 = brisk_Vt<DIRECT>(L, x + 1, y + 1, t);
  if (centre) *centre = s_1_1;
  return brisk_subpixel2d(s_0_0, s_0_1, s_0_2, s_1_0, s_1_1, s_1_2, s_2_0, s_2_1, s_2_2, dx, dy);
}

// The same patch read the way GetKeypoints itself reads it (brisk-scale-space.cc:186-194, 232-240): the coordinates
// are floats there, so every sample is a bilinear GetAgastScore(float, float, 1) that touches a 2x2 block.  The
// values equal the integer reads for integral coordinates; the touched set differs, which only the literal cache
// (DIRECT == 2) can see.
template <int DIRECT>
BRISK_HD float brisk_patch_subpixel_f(const BriskLayerView& L, float x, float y, float& dx, float& dy) {
  BriskTouch none;
  none.on = false; none.mask = 0; none.x0 = 0; none.y0 = 0;
  const int s_0_0 = brisk_Vf<DIRECT>(L, x - 1, y - 1, &none);
  const int s_1_0 = brisk_Vf<DIRECT>(L, x, y - 1, &none);
  const int s_2_0 = brisk_Vf<DIRECT>(L, x + 1, y - 1, &none);
  const int s_2_1 = brisk_Vf<DIRECT>(L, x + 1, y, &none);
  const int s_1_1 = brisk_Vf<DIRECT>(L, x, y, &none);
  const int s_0_1 = brisk_Vf<DIRECT>(L, x - 1, y, &none);
  const int s_0_2 = brisk_Vf<DIRECT>(L, x - 1, y + 1, &none);
  const int s_1_2 = brisk_Vf<DIRECT>(L, x, y + 1, &none);
  const int s_2_2 = brisk_Vf<DIRECT>(L, x + 1, y + 1, &none);
  return brisk_subpixel2d(s_0_0, s_0_1, s_0_2, s_1_0, s_1_1, s_1_2, s_2_0, s_2_1, s_2_2, dx, dy);
}

// ---------------------------------------------------------------------------------------------
// Refine1D family (brisk-scale-space.cc:1101-1228)
// ---------------------------------------------------------------------------------------------
// Parabola through the scores of the layer below, the layer itself and the layer above, at the relative scales
// (lo, 1, hi) those layers have: octave (0.75, 1, 1.5), intra-octave (2/3, 1, 4/3), layer 0 against its virtual half-octave
// (0.7, 1, 1.5).  The integer coefficient tables are the reference's (scores x 1024, rounded), the float steps keep its
// order; the maximum's abscissa is clamped to [lo, hi].
struct BriskScaleFit {
  int a[3], b[3], c[3];      // 2nd, 1st, 0th order coefficient as combinations of (below, own, above)
  float at_lo, at_hi;        // abscissae returned for a maximum at the outer samples
  double lo, hi;             // clamp interval (compared in double, as written in the reference)
  float clamp_lo, clamp_hi;
  double denom;              // common factor of the tables x 1024
  bool float_division;       // Refine1D_2 divides in float (:1226), the other two in double
};
BRISK_HD float brisk_scale_fit(const BriskScaleFit& f, const float s_below, const float s_own, const float s_above, float& max) {
  const int v[3] = {(int)(1024.0 * s_below + 0.5), (int)(1024.0 * s_own + 0.5), (int)(1024.0 * s_above + 0.5)};
  const int a = f.a[0] * v[0] + f.a[1] * v[1] + f.a[2] * v[2];
  if (a >= 0) {  // no maximum: the largest sample (own layer first, then below, then above)
    if (s_own >= s_below && s_own >= s_above) { max = s_own; return 1.0f; }
    if (s_below >= s_own && s_below >= s_above) { max = s_below; return f.at_lo; }
    if (s_above >= s_own && s_above >= s_below) { max = s_above; return f.at_hi; }
  }
  const int b = f.b[0] * v[0] + f.b[1] * v[1] + f.b[2] * v[2];
  float r = -(float)b / (float)(2 * a);
  if (r < f.lo) r = f.clamp_lo;
  else if (r > f.hi) r = f.clamp_hi;
  const int c = f.c[0] * v[0] + f.c[1] * v[1] + f.c[2] * v[2];
  max = (float)c + (float)a * r * r + (float)b * r;
  // (Refine1D_2 divides in float, the other two by 3072.0 / 2048.0 in double and round to float: the same float, a power of
  // two exactly and 3072 by tools/verify_float_division.c)
  max = max / (float)(int)f.denom;
  return r;
}
// Refine1D (:1101-1143), Refine1D_1 (:1145-1186), Refine1D_2 (:1188-1228)
BRISK_HD_OUTLINE float brisk_refine1d(const float s_05, const float s0, const float s05, float& max) {
  const BriskScaleFit f = {{16, -24, 8}, {-40, 54, -14}, {24, -27, 6}, 0.75f, 1.5f, 0.75, 1.5, 0.75f, 1.5f, 3072.0, false};
  return brisk_scale_fit(f, s_05, s0, s05, max);
}
BRISK_HD_OUTLINE float brisk_refine1d_1(const float s_05, const float s0, const float s05, float& max) {
  const BriskScaleFit f = {{9, -18, 9}, {-21, 36, -15}, {12, -16, 6},
                           (float)0.6666666666666666666666666667, (float)1.3333333333333333333333333333,
                           0.6666666666666666666666666667, 1.33333333333333333333333333,
                           (float)0.666666666666666666666666667, (float)1.333333333333333333333333333, 2048.0, false};
  return brisk_scale_fit(f, s_05, s0, s05, max);
}
BRISK_HD_OUTLINE float brisk_refine1d_2(const float s_05, const float s0, const float s05, float& max) {
  const BriskScaleFit f = {{2, -4, 2}, {-5, 8, -3}, {3, -3, 1}, (float)0.7, 1.5f, 0.7, 1.5, (float)0.7, 1.5f, 1024.0, true};
  return brisk_scale_fit(f, s_05, s0, s05, max);
}

// ---------------------------------------------------------------------------------------------
// GetScoreMaxAbove / GetScoreMaxBelow (brisk-scale-space.cc:757-1099), history-free evaluation
// with touch recording.  `above` selects the window mapping; `odd` = (layer % 2 == 1).
// ---------------------------------------------------------------------------------------------
template <int DIRECT>
BRISK_HD float brisk_score_max_other(const BriskLayerView& Lo, const bool above, const bool odd, const int x_layer,
                                     const int y_layer, const int thr, bool& ismax, float& dx, float& dy,
                                     BriskTouch* t) {
  const int threshold = thr + BRISK_DROP_THRESHOLD;
  ismax = false;
  // (the reference divides by 6.0 and 4.0 in double and rounds the quotient to float; for a float dividend that equals the
  // float division - every finite float checked, tools/verify_float_division.c - which costs a third of the double one here)
  float x_1, x1, y_1, y1;
  if (above) {
    if (!odd) {
      x_1 = (float)(4 * (x_layer)-1 - 2) / 6.0f;
      x1 = (float)(4 * (x_layer)-1 + 2) / 6.0f;
      y_1 = (float)(4 * (y_layer)-1 - 2) / 6.0f;
      y1 = (float)(4 * (y_layer)-1 + 2) / 6.0f;
    } else {
      x_1 = (float)(6 * (x_layer)-1 - 3) / 8.0f;
      x1 = (float)(6 * (x_layer)-1 + 3) / 8.0f;
      y_1 = (float)(6 * (y_layer)-1 - 3) / 8.0f;
      y1 = (float)(6 * (y_layer)-1 + 3) / 8.0f;
    }
  } else {
    if (!odd) {
      x_1 = (float)(8 * (x_layer) + 1 - 4) / 6.0f;
      x1 = (float)(8 * (x_layer) + 1 + 4) / 6.0f;
      y_1 = (float)(8 * (y_layer) + 1 - 4) / 6.0f;
      y1 = (float)(8 * (y_layer) + 1 + 4) / 6.0f;
    } else {
      x_1 = (float)(6 * (x_layer) + 1 - 3) / 4.0f;
      x1 = (float)(6 * (x_layer) + 1 + 3) / 4.0f;
      y_1 = (float)(6 * (y_layer) + 1 - 3) / 4.0f;
      y1 = (float)(6 * (y_layer) + 1 + 3) / 4.0f;
    }
  }
  if (t->on) {
    t->x0 = (int)x_1 - 1;
    t->y0 = (int)y_1 - 1;
  }
  // The window [x_1, x1] x [y_1, y1] is sampled on a grid whose outer lines are the (fractional) window borders and
  // whose inner lines are the integer positions in between: border samples are bilinear (and touch 2x2 blocks), inner
  // samples are plain score reads.  Row-major scan; a sample above centre + drop threshold aborts - except on the
  // last grid row, which the reference does not test (:843-863, :1027-1047); the maximum's position is recorded as
  // the nearest inner line (xs / xe, ys / ye for the borders).
  const int xs = (int)(x_1 + 1), xe = (int)x1, ys = (int)(y_1 + 1), ye = (int)y1;
  const int nix = brisk_max(xe - xs + 1, 0), niy = brisk_max(ye - ys + 1, 0);  // inner lines
  int max_x = xs;
  int max_y = ys;
  float max = 0.0f;
  for (int r = 0; r < niy + 2; ++r) {
    const bool border_y = (r == 0 || r == niy + 1);
    const float yf = (r == 0) ? y_1 : (r == niy + 1) ? y1 : (float)(ys + r - 1);
    const int ylab = (r == 0) ? ys : (r == niy + 1) ? ye : ys + r - 1;
    for (int j = 0; j < nix + 2; ++j) {
      const bool border_x = (j == 0 || j == nix + 1);
      const float xf = (j == 0) ? x_1 : (j == nix + 1) ? x1 : (float)(xs + j - 1);
      const int xlab = (j == 0) ? xs : (j == nix + 1) ? xe : xs + j - 1;
      const float v = (border_x || border_y) ? (float)brisk_Vf<DIRECT>(Lo, xf, yf, t) : (float)brisk_Vt<DIRECT>(Lo, xlab, ylab, t);
      if (r != niy + 1 && v > threshold) return 0;
      if (r == 0 && j == 0) { max = v; continue; }
      if (!above && !border_x && !border_y && v == max) {  // tie rule, GetScoreMaxBelow only (:987-1010)
        const int x = xlab, y = ylab;
        const int t1 = 2 * (brisk_Vt<DIRECT>(Lo, x - 1, y, t) + brisk_Vt<DIRECT>(Lo, x + 1, y, t) + brisk_Vt<DIRECT>(Lo, x, y + 1, t) +
                            brisk_Vt<DIRECT>(Lo, x, y - 1, t)) +
                       (brisk_Vt<DIRECT>(Lo, x + 1, y + 1, t) + brisk_Vt<DIRECT>(Lo, x - 1, y + 1, t) + brisk_Vt<DIRECT>(Lo, x + 1, y - 1, t) +
                        brisk_Vt<DIRECT>(Lo, x - 1, y - 1, t));
        const int t2 = 2 * (brisk_Vt<DIRECT>(Lo, max_x - 1, max_y, t) + brisk_Vt<DIRECT>(Lo, max_x + 1, max_y, t) +
                            brisk_Vt<DIRECT>(Lo, max_x, max_y + 1, t) + brisk_Vt<DIRECT>(Lo, max_x, max_y - 1, t)) +
                       (brisk_Vt<DIRECT>(Lo, max_x + 1, max_y + 1, t) + brisk_Vt<DIRECT>(Lo, max_x - 1, max_y + 1, t) +
                        brisk_Vt<DIRECT>(Lo, max_x + 1, max_y - 1, t) + brisk_Vt<DIRECT>(Lo, max_x - 1, max_y - 1, t));
        if (t1 > t2) { max_x = x; max_y = y; }
      }
      if (v > max) {
        max = v;
        max_x = xlab;
        if (r != 0) max_y = ylab;  // (the first grid row never moves the row label: it is ys already)
      }
    }
  }

  float dx_1, dy_1;
  const float refined_max = brisk_patch_subpixel<DIRECT>(Lo, max_x, max_y, t, dx_1, dy_1, nullptr);
  const float real_x = (float)max_x + dx_1;
  const float real_y = (float)max_y + dy_1;
  bool returnrefined = true;
  if (above) {
    if (!odd) {
      dx = (real_x * 6.0f + 1.0f) / 4.0f - (float)x_layer;
      dy = (real_y * 6.0f + 1.0f) / 4.0f - (float)y_layer;
    } else {
      dx = (float)((real_x * 8.0 + 1.0) / 6.0 - (float)x_layer);
      dy = (float)((real_y * 8.0 + 1.0) / 6.0 - (float)y_layer);
    }
  } else {
    if (!odd) {
      dx = (float)((real_x * 6.0 + 1.0) / 8.0 - (float)x_layer);
      dy = (float)((real_y * 6.0 + 1.0) / 8.0 - (float)y_layer);
    } else {
      dx = (float)((real_x * 4.0 - 1.0) / 6.0 - (float)x_layer);
      dy = (float)((real_y * 4.0 - 1.0) / 6.0 - (float)y_layer);
    }
  }
  if (dx > 1.0f) { dx = 1.0f; returnrefined = false; }
  if (dx < -1.0f) { dx = -1.0f; returnrefined = false; }
  if (dy > 1.0f) { dy = 1.0f; returnrefined = false; }
  if (dy < -1.0f) { dy = -1.0f; returnrefined = false; }
  ismax = true;
  if (returnrefined) return refined_max > max ? refined_max : max;
  return max;
}

// ---------------------------------------------------------------------------------------------
// GetScoreMaxAbove on the candidate's 4 x 4 score block of the layer above (k_classify_refine: DIRECT == 0).  The same
// function as brisk_score_max_other<0>(above = true), restated for what is known there: the window is less than a pixel
// wide (4/6 or 6/8 of one), so its sample grid has two or three lines each way - border lines at the fractional window
// ends, at most one integer line between them -, every sample's taps lie in block rows / columns 1 ... 3 and the 3 x 3
// patch around the maximum inside the block.  The generic routine walks that grid with a bounds-checked, touch-recording
// block lookup per tap (about 130 instructions per bilinear sample); here a block row is a register, a tap is a bit-field
// extract, and a sample's touches are a 2 x 2 bit pattern masked with the pixels that are off the layer's 3-pixel border.
// Same scan order and abort rule (a sample above centre + drop threshold ends the scan, except on the last grid row; the
// touches of the samples before it stay), same first-wins maximum, same float expressions.  An assumption that does not
// hold raises the miss flag (the candidate is then redone by k_classify_refine_direct); tests/emul checks the two forms
// against each other on random blocks.
// ---------------------------------------------------------------------------------------------
BRISK_HD uint32_t brisk_blk_row(const BriskScoreBlock& b, int r) { return r < 2 ? (r == 0 ? b.w0 : b.w1) : (r == 2 ? b.w2 : b.w3); }
BRISK_HD float brisk_score_max_above_blk(const BriskLayerView& Lo, const bool odd, const int x_layer, const int y_layer,
                                         const int thr, bool& ismax, float& dx, float& dy, BriskTouch* t) {
  const int threshold = thr + BRISK_DROP_THRESHOLD;
  ismax = false;
  float x_1, x1, y_1, y1;
  if (!odd) {
    x_1 = (float)(4 * (x_layer)-1 - 2) / 6.0f;
    x1 = (float)(4 * (x_layer)-1 + 2) / 6.0f;
    y_1 = (float)(4 * (y_layer)-1 - 2) / 6.0f;
    y1 = (float)(4 * (y_layer)-1 + 2) / 6.0f;
  } else {
    x_1 = (float)(6 * (x_layer)-1 - 3) / 8.0f;
    x1 = (float)(6 * (x_layer)-1 + 3) / 8.0f;
    y_1 = (float)(6 * (y_layer)-1 - 3) / 8.0f;
    y1 = (float)(6 * (y_layer)-1 + 3) / 8.0f;
  }
  const int ax = (int)x_1 - 1, ay = (int)y_1 - 1;  // block / footprint anchor
  if (t->on) {
    t->x0 = ax;
    t->y0 = ay;
  }
  const int xs = (int)(x_1 + 1), xe = (int)x1, ys = (int)(y_1 + 1), ye = (int)y1;
  const int nix = brisk_max(xe - xs + 1, 0), niy = brisk_max(ye - ys + 1, 0);
  if (Lo.blk.cw != 4 || Lo.blk.ch != 4 || Lo.blk.x0 != ax || Lo.blk.y0 != ay || xs != ax + 2 || ys != ay + 2 || nix > 1 || niy > 1 ||
      xe < ax + 1 || ye < ay + 1) {
    Lo.miss = 1;
    return 0;
  }
  // pixels of the block that a touch records (off the layer's 3-pixel border), as a 4 x 4 bit mask
  unsigned okm = 0;
  if (t->on) {
    unsigned colm = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) colm |= (ax + b >= 3 && ax + b < Lo.w - 3) ? (1u << b) : 0u;
#pragma unroll
    for (int b = 0; b < 4; ++b) okm |= (ay + b >= 3 && ay + b < Lo.h - 3) ? (colm << (4 * b)) : 0u;
  }
  const uint32_t W1 = Lo.blk.w1, W2 = Lo.blk.w2, W3 = Lo.blk.w3;
  unsigned touched = 0;
  bool aborted = false, first = true;
  int max_x = xs, max_y = ys;
  float max = 0.0f;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const bool rlast = (r == niy + 1), rex = (r <= niy + 1), rborder = (r == 0 || rlast);
    const float yf = (r == 0) ? y_1 : (rlast ? y1 : (float)ys);
    const int ylab = (r == 0) ? ys : (rlast ? ye : ys);
    const int iy = (int)yf - ay;  // 1 or 2
    const uint32_t wa = (iy == 1) ? W1 : W2, wb = (iy == 1) ? W2 : W3;
    const float ry1 = yf - (float)(int)yf;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const bool jlast = (j == nix + 1), jex = (j <= nix + 1), jborder = (j == 0 || jlast);
      const float xf = (j == 0) ? x_1 : (jlast ? x1 : (float)xs);
      const int xlab = (j == 0) ? xs : (jlast ? xe : xs);
      const int ix = (int)xf - ax;  // 1 or 2
      float v;
      unsigned tb;
      if (rborder || jborder) {
        const int sh = 8 * ix;
        const int s00 = (int)((wa >> sh) & 0xFFu), s10 = (int)((wa >> (sh + 8)) & 0xFFu);
        const int s01 = (int)((wb >> sh) & 0xFFu), s11 = (int)((wb >> (sh + 8)) & 0xFFu);
        v = (float)brisk_bilinear_u8(xf - (float)(int)xf, ry1, s00, s10, s01, s11);
        tb = 0x33u << (4 * iy + ix);
      } else {  // the one inner sample (xs, ys) = block position (2, 2)
        v = (float)(int)((W2 >> 16) & 0xFFu);
        tb = 1u << 10;
      }
      if (rex && jex && !aborted) {
        touched |= tb & okm;
        if (!rlast && v > threshold) {
          aborted = true;
        } else if (first) {
          max = v;
          first = false;
        } else if (v > max) {
          max = v;
          max_x = xlab;
          if (r != 0) max_y = ylab;
        }
      }
    }
  }
  if (aborted) {
    t->mask |= touched;
    return 0;
  }
  // 3 x 3 patch around the maximum: block columns ox ... ox + 2, rows oy ... oy + 2 (ox, oy in {0, 1})
  const int ox = max_x - 1 - ax, oy = max_y - 1 - ay;
  const int psh = 8 * ox;
  const uint32_t p0 = brisk_blk_row(Lo.blk, oy) >> psh, p1 = brisk_blk_row(Lo.blk, oy + 1) >> psh, p2 = brisk_blk_row(Lo.blk, oy + 2) >> psh;
  touched |= (0x777u << (4 * oy + ox)) & okm;
  t->mask |= touched;
  float dx_1, dy_1;
  // s_x_y: column x, row y
  const float refined_max = brisk_subpixel2d((int)(p0 & 0xFFu), (int)(p1 & 0xFFu), (int)(p2 & 0xFFu), (int)((p0 >> 8) & 0xFFu),
                                             (int)((p1 >> 8) & 0xFFu), (int)((p2 >> 8) & 0xFFu), (int)((p0 >> 16) & 0xFFu),
                                             (int)((p1 >> 16) & 0xFFu), (int)((p2 >> 16) & 0xFFu), dx_1, dy_1);
  const float real_x = (float)max_x + dx_1;
  const float real_y = (float)max_y + dy_1;
  bool returnrefined = true;
  if (!odd) {
    dx = (real_x * 6.0f + 1.0f) / 4.0f - (float)x_layer;
    dy = (real_y * 6.0f + 1.0f) / 4.0f - (float)y_layer;
  } else {
    dx = (float)((real_x * 8.0 + 1.0) / 6.0 - (float)x_layer);
    dy = (float)((real_y * 8.0 + 1.0) / 6.0 - (float)y_layer);
  }
  if (dx > 1.0f) { dx = 1.0f; returnrefined = false; }
  if (dx < -1.0f) { dx = -1.0f; returnrefined = false; }
  if (dy > 1.0f) { dy = 1.0f; returnrefined = false; }
  if (dy < -1.0f) { dy = -1.0f; returnrefined = false; }
  ismax = true;
  if (returnrefined) return refined_max > max ? refined_max : max;
  return max;
}

// GetScoreMaxBelow on the candidate's 4 x 4 score block of the layer below, in the same way (tools/classify_phases.py: the
// generic walk of this window was the longest phase of k_classify_refine, 8 us of a wave's 19).  The window is 8/6 or 6/4
// of a pixel wide: border lines at its ends and one or two integer lines between them, i.e. a grid of 3 or 4 lines each way
// whose taps are the block's 16 pixels; no touches are recorded below.  Additionally to the scan rules above: an inner
// sample that EQUALS the running maximum moves the maximum's position if its 3 x 3 neighbourhood, weighted [1 2 1; 2 0 2;
// 1 2 1], outweighs that of the present position (:987-1010) - the four possible neighbourhood sums are formed up front.
BRISK_HD int brisk_blk_px(const BriskScoreBlock& b, int bx, int by) { return (int)((brisk_blk_row(b, by) >> (8 * bx)) & 0xFFu); }
BRISK_HD int brisk_blk_ring_sum(const BriskScoreBlock& b, int bx, int by) {  // (bx, by) in {1, 2}^2
  const uint32_t ra = brisk_blk_row(b, by - 1) >> (8 * (bx - 1)), rb = brisk_blk_row(b, by) >> (8 * (bx - 1)), rc = brisk_blk_row(b, by + 1) >> (8 * (bx - 1));
  const int corners = (int)(ra & 0xFFu) + (int)((ra >> 16) & 0xFFu) + (int)(rc & 0xFFu) + (int)((rc >> 16) & 0xFFu);
  const int edges = (int)((ra >> 8) & 0xFFu) + (int)((rc >> 8) & 0xFFu) + (int)(rb & 0xFFu) + (int)((rb >> 16) & 0xFFu);
  return 2 * edges + corners;
}
BRISK_HD float brisk_score_max_below_blk(const BriskLayerView& Lo, const bool odd, const int x_layer, const int y_layer,
                                         const int thr, bool& ismax, float& dx, float& dy) {
  const int threshold = thr + BRISK_DROP_THRESHOLD;
  ismax = false;
  float x_1, x1, y_1, y1;
  if (!odd) {
    x_1 = (float)(8 * (x_layer) + 1 - 4) / 6.0f;
    x1 = (float)(8 * (x_layer) + 1 + 4) / 6.0f;
    y_1 = (float)(8 * (y_layer) + 1 - 4) / 6.0f;
    y1 = (float)(8 * (y_layer) + 1 + 4) / 6.0f;
  } else {
    x_1 = (float)(6 * (x_layer) + 1 - 3) / 4.0f;
    x1 = (float)(6 * (x_layer) + 1 + 3) / 4.0f;
    y_1 = (float)(6 * (y_layer) + 1 - 3) / 4.0f;
    y1 = (float)(6 * (y_layer) + 1 + 3) / 4.0f;
  }
  const int ax = (int)x_1, ay = (int)y_1;  // block anchor
  const int xs = (int)(x_1 + 1), xe = (int)x1, ys = (int)(y_1 + 1), ye = (int)y1;
  const int nix = brisk_max(xe - xs + 1, 0), niy = brisk_max(ye - ys + 1, 0);
  if (Lo.blk.cw != 4 || Lo.blk.ch != 4 || Lo.blk.x0 != ax || Lo.blk.y0 != ay || xs != ax + 1 || ys != ay + 1 || nix < 1 || nix > 2 ||
      niy < 1 || niy > 2) {
    Lo.miss = 1;
    return 0;
  }
  const uint32_t W0 = Lo.blk.w0, W1 = Lo.blk.w1, W2 = Lo.blk.w2, W3 = Lo.blk.w3;
  // neighbourhood sums of the four inner positions (tie rule), [by - 1][bx - 1]
  const int rs11 = brisk_blk_ring_sum(Lo.blk, 1, 1), rs21 = brisk_blk_ring_sum(Lo.blk, 2, 1);
  const int rs12 = brisk_blk_ring_sum(Lo.blk, 1, 2), rs22 = brisk_blk_ring_sum(Lo.blk, 2, 2);
  bool aborted = false, first = true;
  int max_x = xs, max_y = ys;
  float max = 0.0f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool rlast = (r == niy + 1), rex = (r <= niy + 1), rborder = (r == 0 || rlast);
    const float yf = (r == 0) ? y_1 : (rlast ? y1 : (float)(ys + r - 1));
    const int ylab = (r == 0) ? ys : (rlast ? ye : ys + r - 1);
    const int iy = (int)yf - ay;  // 0 ... 2
    const uint32_t wa = (iy == 0) ? W0 : ((iy == 1) ? W1 : W2), wb = (iy == 0) ? W1 : ((iy == 1) ? W2 : W3);
    const float ry1 = yf - (float)(int)yf;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool jlast = (j == nix + 1), jex = (j <= nix + 1), jborder = (j == 0 || jlast);
      const float xf = (j == 0) ? x_1 : (jlast ? x1 : (float)(xs + j - 1));
      const int xlab = (j == 0) ? xs : (jlast ? xe : xs + j - 1);
      const int ix = brisk_min((int)xf - ax, 2);  // 0 ... 2 (a line beyond the grid is never used: keep its shifts defined)
      const int sh = 8 * ix;
      const bool border = rborder || jborder;
      float v;
      if (border) {
        const int s00 = (int)((wa >> sh) & 0xFFu), s10 = (int)((wa >> (sh + 8)) & 0xFFu);
        const int s01 = (int)((wb >> sh) & 0xFFu), s11 = (int)((wb >> (sh + 8)) & 0xFFu);
        v = (float)brisk_bilinear_u8(xf - (float)(int)xf, ry1, s00, s10, s01, s11);
      } else {
        v = (float)(int)((wa >> sh) & 0xFFu);  // plain read at (xlab, ylab) = block position (ix, iy)
      }
      if (rex && jex && !aborted) {
        if (!rlast && v > threshold) {
          aborted = true;
        } else if (first) {
          max = v;
          first = false;
        } else {
          if (!border && v == max) {  // tie rule (inner samples only)
            const int bx = xlab - ax, by = ylab - ay, mx = max_x - ax, my = max_y - ay;
            const int t1 = (by == 1) ? (bx == 1 ? rs11 : rs21) : (bx == 1 ? rs12 : rs22);
            const int t2 = (my == 1) ? (mx == 1 ? rs11 : rs21) : (mx == 1 ? rs12 : rs22);
            if (t1 > t2) { max_x = xlab; max_y = ylab; }
          }
          if (v > max) {
            max = v;
            max_x = xlab;
            if (r != 0) max_y = ylab;
          }
        }
      }
    }
  }
  if (aborted) return 0;
  const int ox = max_x - 1 - ax, oy = max_y - 1 - ay;  // patch origin in the block: {0, 1}
  const int psh = 8 * ox;
  const uint32_t p0 = brisk_blk_row(Lo.blk, oy) >> psh, p1 = brisk_blk_row(Lo.blk, oy + 1) >> psh, p2 = brisk_blk_row(Lo.blk, oy + 2) >> psh;
  float dx_1, dy_1;
  const float refined_max = brisk_subpixel2d((int)(p0 & 0xFFu), (int)(p1 & 0xFFu), (int)(p2 & 0xFFu), (int)((p0 >> 8) & 0xFFu),
                                             (int)((p1 >> 8) & 0xFFu), (int)((p2 >> 8) & 0xFFu), (int)((p0 >> 16) & 0xFFu),
                                             (int)((p1 >> 16) & 0xFFu), (int)((p2 >> 16) & 0xFFu), dx_1, dy_1);
  const float real_x = (float)max_x + dx_1;
  const float real_y = (float)max_y + dy_1;
  bool returnrefined = true;
  if (!odd) {
    dx = (float)((real_x * 6.0 + 1.0) / 8.0 - (float)x_layer);
    dy = (float)((real_y * 6.0 + 1.0) / 8.0 - (float)y_layer);
  } else {
    dx = (float)((real_x * 4.0 - 1.0) / 6.0 - (float)x_layer);
    dy = (float)((real_y * 4.0 - 1.0) / 6.0 - (float)y_layer);
  }
  if (dx > 1.0f) { dx = 1.0f; returnrefined = false; }
  if (dx < -1.0f) { dx = -1.0f; returnrefined = false; }
  if (dy > 1.0f) { dy = 1.0f; returnrefined = false; }
  if (dy < -1.0f) { dy = -1.0f; returnrefined = false; }
  ismax = true;
  if (returnrefined) return refined_max > max ? refined_max : max;
  return max;
}

// ---------------------------------------------------------------------------------------------
// IsMax2D steps 1-2 (brisk-scale-space.cc:430-498): history-free classification.
// Probe order W, E, N, S, SW, SE, NE, NW with early exit.
// ---------------------------------------------------------------------------------------------
#define BRISK_PROBE_DX(k) ((int)((0x0FF5F5u >> (2 * (k))) & 3u) - 1)
// k:        0   1   2   3   4   5   6   7
// dx:      -1  +1   0   0  -1  +1  +1  -1
// dy:       0   0  -1  +1  +1  +1  -1  -1
// (2-bit packed tables: no indexed local arrays in device code)
BRISK_HD int brisk_probe_dx(int k) { return (int)((0x2858u >> (2 * k)) & 3u) - 1; }  // {-1, 1, 0, 0, -1, 1, 1, -1}
BRISK_HD int brisk_probe_dy(int k) { return (int)((0x0A85u >> (2 * k)) & 3u) - 1; }  // {0, 0, -1, 1, 1, 1, -1, -1}
// probe index of neighbour offset (dx, dy), |dx|,|dy| <= 1, not both 0
BRISK_HD constexpr int brisk_probe_index(int dx, int dy) {  // {7, 2, 6, 0, -1, 1, 4, 3, 5}[(dy+1)*3 + (dx+1)]
  return (int)((0x5341F0627ull >> (4 * ((dy + 1) * 3 + (dx + 1)))) & 15ull) == 15 ? -1
       : (int)((0x5341F0627ull >> (4 * ((dy + 1) * 3 + (dx + 1)))) & 15ull);
}

// returns status (REJ / PASS / TIE) and the number of probes issued
template <int DIRECT>
BRISK_HD unsigned brisk_classify(const BriskLayerView& L, int x, int y, int centre, int* nprobed) {
  bool tie = false;
  for (int k = 0; k < 8; ++k) {
    const int nx = x + brisk_probe_dx(k), ny = y + brisk_probe_dy(k);
    int s = 0;
    if (!brisk_border3(L, nx, ny)) {
      if (DIRECT == 0) {
        // The block value is D for a detection (D > 2) and K' otherwise (k_score_blocks), and the probe's value only
        // matters where it reaches the centre: a detection's D below the centre changes nothing, whether it is kept or
        // zeroed.  No map read, then - the eight of them, one after the other behind early exits, were 40 % of
        // k_classify_refine's time (0.10 of 0.24 ms per 256 frames).
        const int V = brisk_V<DIRECT>(L, nx, ny);
        s = (V >= centre) ? V : 0;
      } else {
        const int D = BRISK_SM_D(L.smap[(long)ny * L.stride + nx]);
        if (D > 2) {
          s = D;
        } else {
          const int K = brisk_V<DIRECT>(L, nx, ny);  // == K' for a non-detection
          s = (K >= centre) ? K : 0;
        }
      }
    }
    if (centre < s) {
      *nprobed = k + 1;
      return BRISK_ST_REJ;
    }
    if (centre == s) tie = true;
  }
  *nprobed = 8;
  return tie ? BRISK_ST_TIE : BRISK_ST_PASS;
}

// ---------------------------------------------------------------------------------------------
// IsMax2D, literally (brisk-scale-space.cc:430-531), on the literal cache: the ordered path for AGAST thresholds
// below 20, where a detection can store a score <= 2 (which the cache treats as "not cached") and the history-free
// split of the fast path does not hold.  centre = raw map value; 8 probes with threshold = centre in the order
// W, E, N, S, SW, SE, NE, NW with early exit; equal neighbours are compared through the [1 2 1; 2 4 2; 1 2 1]
// smoothed sums of the RAW map.
// ---------------------------------------------------------------------------------------------
// RAW = true: the map reads are addressed as the reference addresses them, `data + y * cols + x` on a matrix without
// row padding - needed where the coordinates do not belong to this layer (the `at(0)` indexing of the
// suppressScaleNonmaxima = false branch, :137): a column beyond the row end then lands in the next row.  *oob is set
// when such a read would leave the matrix (undefined behaviour in the reference).
template <bool RAW>
BRISK_HD int brisk_raw_read(const BriskLayerView& L, long lin_x, long lin_y, bool* oob) {
  if (!RAW) return (int)(L.smap[lin_y * L.stride + lin_x] & 0xFFu);
  const long idx = lin_y * L.w + lin_x;
  if (idx < 0 || idx >= (long)L.w * L.h) { *oob = true; return 0; }
  return (int)(L.smap[(idx / L.w) * L.stride + (idx % L.w)] & 0xFFu);
}

template <bool RAW>
BRISK_HD bool brisk_ismax2d_literal(const BriskLayerView& L, const int x_layer, const int y_layer, bool* oob) {
  const int center = brisk_raw_read<RAW>(L, x_layer, y_layer, oob);
  const int s_10 = brisk_S_literal(L, x_layer - 1, y_layer, center);
  if (center < s_10) return false;
  const int s10 = brisk_S_literal(L, x_layer + 1, y_layer, center);
  if (center < s10) return false;
  const int s0_1 = brisk_S_literal(L, x_layer, y_layer - 1, center);
  if (center < s0_1) return false;
  const int s01 = brisk_S_literal(L, x_layer, y_layer + 1, center);
  if (center < s01) return false;
  const int s_11 = brisk_S_literal(L, x_layer - 1, y_layer + 1, center);
  if (center < s_11) return false;
  const int s11 = brisk_S_literal(L, x_layer + 1, y_layer + 1, center);
  if (center < s11) return false;
  const int s1_1 = brisk_S_literal(L, x_layer + 1, y_layer - 1, center);
  if (center < s1_1) return false;
  const int s_1_1 = brisk_S_literal(L, x_layer - 1, y_layer - 1, center);
  if (center < s_1_1) return false;
  // equal-score neighbours in the reference's list order (:478-497)
  const int ddx[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, ddy[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
  const int sv[8] = {s_1_1, s0_1, s1_1, s_10, s10, s_11, s01, s11};
  const int smoothedcenter = 4 * center + 2 * (s_10 + s10 + s0_1 + s01) + s_1_1 + s1_1 + s_11 + s11;
  for (int i = 0; i < 8; ++i) {
    if (sv[i] != center) continue;
    int other = 0;
    for (int oy = -1; oy <= 1; ++oy)
      for (int ox = -1; ox <= 1; ++ox)
        other += (ox == 0 ? 2 : 1) * (oy == 0 ? 2 : 1) * brisk_raw_read<RAW>(L, x_layer + ddx[i] + ox, y_layer + ddy[i] + oy, oob);
    if (other > smoothedcenter) return false;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// Refinement of a 2D maximum (brisk-scale-space.cc:211-287 + Refine3D :534-754).
// Returns true if a keypoint results.  e5 = the candidate reaches its own-layer patch reads;
// touch = score-touches on the layer above (event e3).
// ---------------------------------------------------------------------------------------------
// Lbelow / tl / Labove: views of layer-1, layer, layer+1 (the neighbours may be dummies where they do not exist).
template <int DIRECT>
BRISK_HD bool brisk_refine(const BriskGeom& G, const BriskLayerView& Lbelow, const BriskLayerView& tl,
                           const BriskLayerView& Labove, const int layer, const int x_layer, const int y_layer,
                           BriskKeyPoint* kp, bool* e5, BriskTouch* touch, const float* fxy = nullptr) {
  // fxy: the candidate's float coordinates where they are not integral (provided keypoints, ComputeScale; literal mode
  // only).  GetKeypoints keeps them as floats for its own patch reads and the output, and truncates them where it calls
  // Refine3D / GetScoreMaxBelow (int parameters).
  const float fx = fxy ? fxy[0] : (float)x_layer, fy = fxy ? fxy[1] : (float)y_layer;
  const float lscale = G.L[layer].scale, loffset = G.L[layer].offset;
  BriskTouch none;
  none.on = false; none.mask = 0; none.x0 = 0; none.y0 = 0;
  *e5 = false;
  touch->mask = 0;
  kp->angle = -1.0f;
  kp->class_id = -1;
  kp->octave = layer;

  if (G.single_layer) {  // :172-209 (patch via float access: 4x4 touch footprint)
    float delta_x, delta_y;
    *e5 = true;
    const float max = (DIRECT == 2) ? brisk_patch_subpixel_f<DIRECT>(tl, fx, fy, delta_x, delta_y)
                                    : brisk_patch_subpixel<DIRECT>(tl, x_layer, y_layer, &none, delta_x, delta_y, nullptr);
    kp->x = fx + delta_x;
    kp->y = fy + delta_y;
    kp->size = BRISK_BASIC_SIZE;
    kp->response = max;
    kp->octave = 0;
    return true;
  }

  if (layer == G.nlayers - 1) {  // :215-256
    bool ismax;
    float dx, dy;
    // (:227: the threshold argument is the float overload GetAgastScore(point_x, point_y, 1))
    const int centre = (DIRECT == 2) ? brisk_Vf<DIRECT>(tl, fx, fy, &none) : brisk_V<DIRECT>(tl, x_layer, y_layer);
    if (DIRECT == 0) brisk_score_max_below_blk(Lbelow, (layer & 1) != 0, x_layer, y_layer, centre, ismax, dx, dy);
    else brisk_score_max_other<DIRECT>(Lbelow, false, (layer & 1) != 0, x_layer, y_layer, centre, ismax, dx, dy, &none);
    if (!ismax) return false;
    *e5 = true;
    float delta_x, delta_y;
    const float max = (DIRECT == 2) ? brisk_patch_subpixel_f<DIRECT>(tl, fx, fy, delta_x, delta_y)
                                    : brisk_patch_subpixel<DIRECT>(tl, x_layer, y_layer, &none, delta_x, delta_y, nullptr);
    kp->x = (fx + delta_x) * lscale + loffset;
    kp->y = (fy + delta_y) * lscale + loffset;
    kp->size = BRISK_BASIC_SIZE * lscale;
    kp->response = max;
    return true;
  }

  // ---- a layer with neighbours on both sides (brisk-scale-space.cc:534-754) --------------------------------------------
  // Three layer classes share one body; what differs is tabulated:
  //   class            neighbour below                      scale fit      own-layer weight towards above | towards below
  //   0  octave 0      virtual: AGAST 5_8 scores, 3 x 3     Refine1D_2     (1.5 - s) / 0.5                | (s - 0.5) / 0.5
  //   1  octave c > 0  intra-octave below (window map 8/6)  Refine1D       (1.5 - s) / 0.5                | (s - 0.75) / 0.25
  //   2  intra-octave  octave below (window map 6/4)        Refine1D_1     4 - 3 s                        | 3 s - 2
  // with s the fitted relative scale; the weight is evaluated as (a + b s) / c in double (the reference's expressions are
  // these with the operands in another order - same roundings), the other neighbour gets 1 - weight.
  const int cls = (layer == 0) ? 0 : ((layer & 1) ? 2 : 1);
  const bool intra = (cls == 2);
  const int own_score = brisk_V<DIRECT>(tl, x_layer, y_layer);
  bool ismax = true;
  float up_dx = 0, up_dy = 0;
  touch->on = true;
  const float up_peak = (DIRECT == 0) ? brisk_score_max_above_blk(Labove, intra, x_layer, y_layer, own_score, ismax, up_dx, up_dy, touch)
                                      : brisk_score_max_other<DIRECT>(Labove, true, intra, x_layer, y_layer, own_score, ismax, up_dx, up_dy, touch);
  BRISK_CR_T(touch, 1)
  if (!ismax) return false;

  float dn_peak, dn_dx, dn_dy;
  if (cls == 0) {  // (:558-592) the layer below octave 0 does not exist: 5_8 scores of the 3 x 3 block around the point
    int q[9];
    if (DIRECT == 0 && tl.blk58.cw && tl.blk58.x0 == x_layer - 1 && tl.blk58.y0 == y_layer - 1) {
      // (the block IS this 3 x 3 neighbourhood, row-major in its bytes: no bounds-checked lookups)
      const uint32_t u0 = tl.blk58.w0, u1 = tl.blk58.w1, u2 = tl.blk58.w2;
      q[0] = (int)(u0 & 0xFFu); q[1] = (int)((u0 >> 8) & 0xFFu); q[2] = (int)((u0 >> 16) & 0xFFu); q[3] = (int)(u0 >> 24);
      q[4] = (int)(u1 & 0xFFu); q[5] = (int)((u1 >> 8) & 0xFFu); q[6] = (int)((u1 >> 16) & 0xFFu); q[7] = (int)(u1 >> 24);
      q[8] = (int)(u2 & 0xFFu);
    } else {
#pragma unroll
      for (int i = 0; i < 9; ++i) q[i] = brisk_V58<DIRECT>(tl, x_layer + i % 3 - 1, y_layer + i / 3 - 1);
    }
    int best = q[0];
#pragma unroll
    for (int i = 1; i < 9; ++i) best = brisk_max(best, q[i]);
    // (brisk_subpixel2d takes the block column by column: s_x_y)
    brisk_subpixel2d(q[0], q[3], q[6], q[1], q[4], q[7], q[2], q[5], q[8], dn_dx, dn_dy);
    dn_peak = (float)best;
  } else {
    dn_peak = (DIRECT == 0) ? brisk_score_max_below_blk(Lbelow, intra, x_layer, y_layer, own_score, ismax, dn_dx, dn_dy)
                            : brisk_score_max_other<DIRECT>(Lbelow, false, intra, x_layer, y_layer, own_score, ismax, dn_dx, dn_dy, &none);
    if (!ismax) return false;
  }
  BRISK_CR_T(touch, 2)
  *e5 = true;
  float own_dx, own_dy;
  int patch_centre;
  const float own_peak = brisk_patch_subpixel<DIRECT>(tl, x_layer, y_layer, &none, own_dx, own_dy, &patch_centre);
  BRISK_CR_T(touch, 3)

  // maximum along the scale axis?  (kMaxThreshold_ = 1, kMinDrop_ = 15; ints against floats, as the reference compares them)
  const int near_margin = patch_centre - BRISK_MAX_THRESHOLD, far_margin = patch_centre - BRISK_MIN_DROP;
  bool fit_scale = true;
  if (cls == 0) {
    if (near_margin <= (int)up_peak) fit_scale = false;  // (:600-602: octave 0 only looks up, and never rejects)
  } else if (near_margin < up_peak || near_margin < dn_peak) {
    if (far_margin > up_peak || far_margin > dn_peak) fit_scale = false;
    else return false;
  }
  const float mid = ((float)own_score < own_peak) ? own_peak : (float)own_score;  // std::max(float(center), max_layer)
  float rel_scale = 1.0f, response = own_peak;
  if (fit_scale) {
    rel_scale = cls == 0 ? brisk_refine1d_2(dn_peak, mid, up_peak, response)
              : cls == 1 ? brisk_refine1d(dn_peak, mid, up_peak, response)
                         : brisk_refine1d_1(dn_peak, mid, up_peak, response);
  }
  const bool upwards = rel_scale > 1.0;
  const double wa = upwards ? (intra ? 4.0 : 1.5) : (cls == 0 ? -0.5 : intra ? -2.0 : -0.75);
  const double wb = upwards ? (intra ? -3.0 : -1.0) : (intra ? 3.0 : 1.0);
  const double inv_wc = intra ? 1.0 : ((upwards || cls == 0) ? 2.0 : 4.0);  // (divisors 1, 0.5, 0.25: exact either way)
  const float w_own = (float)((wa + wb * rel_scale) * inv_wc);
  const float w_nb = (float)(1.0 - w_own);
  const float nb_dx = upwards ? up_dx : dn_dx, nb_dy = upwards ? up_dy : dn_dy;
  // (octave 0: scale 1, offset 0 - the reference leaves them out where it interpolates downwards, the same value)
  kp->x = (w_own * own_dx + w_nb * nb_dx + (float)x_layer) * lscale + loffset;
  kp->y = (w_own * own_dy + w_nb * nb_dy + (float)y_layer) * lscale + loffset;
  kp->size = BRISK_BASIC_SIZE * (rel_scale * lscale);
  kp->response = response;
  BRISK_CR_T(touch, 4)
  return true;
}

// ---------------------------------------------------------------------------------------------
// The ordered path as one sequential walk (k_ordered_keypoints runs it on one lane per frame; the test harness runs
// the same function): GetKeypoints (brisk-scale-space.cc:92-287) over the candidates in (layer, y, x) order on the
// literal cache.  no_scale_nms: the suppressScaleNonmaxima = false branch (:131-170) with more than one layer - layer i
// contributes as many keypoints as it has AGAST points but takes their COORDINATES from layer 0's list
// (`agastPoints.at(0)[n]`, :137) and probes layer i's maps there; only a 2-D refinement follows.  Returns true where
// the reference has no defined result (at() throws because layer i has more points than layer 0, or IsMax2D reads
// outside a score matrix).
// ---------------------------------------------------------------------------------------------
struct BriskOrderedOut {
  BriskKeyPoint* kp;
  int cap, n;            // n counts every keypoint, also those beyond cap
  const uint8_t* mask;   // optional (RemoveInvalidKeyPoints, brisk-feature-detector.cc:49-66)
  int mask_row_pitch;
};
BRISK_HD BriskLayerView brisk_view_of(const BriskGeom& G, uint8_t* pyr_frame, uint16_t* smap_frame, int l) {
  BriskLayerView v;
  v.img = pyr_frame + G.L[l].off;
  v.smap = smap_frame + G.L[l].off;
  v.w = G.L[l].w; v.h = G.L[l].h; v.stride = G.L[l].stride;
  brisk_block_clear(&v.blk);
  brisk_block_clear(&v.blk58);
  v.miss = 0;
  return v;
}
BRISK_HD void brisk_ordered_emit(BriskOrderedOut* o, const BriskKeyPoint& kp) {
  if (o->mask && o->mask[(long)(int)(kp.y + 0.5f) * o->mask_row_pitch + (int)(kp.x + 0.5f)] == 0) return;
  if (o->n < o->cap) o->kp[o->n] = kp;
  o->n++;
}
BRISK_HD bool brisk_ordered_walk(const BriskGeom& G, uint8_t* pyr_frame, uint16_t* smap_frame, const BriskCand* C,
                                 const unsigned* order, int n, bool no_scale_nms, BriskOrderedOut* out) {
  if (no_scale_nms && !G.single_layer) {
    int start[BRISK_MAX_LAYERS + 2];  // candidate ranges of the layers in the ordered list
    for (int l = 0; l <= BRISK_MAX_LAYERS + 1; ++l) start[l] = n;
    for (int r = n - 1; r >= 0; --r) start[C[order[r]].layer] = r;
    for (int l = G.nlayers - 1; l >= 0; --l)
      if (start[l] > start[l + 1]) start[l] = start[l + 1];  // empty layers
    const int n0 = start[1] - start[0];
    for (int i = 0; i < G.nlayers; ++i) {
      const int num = start[i + 1] - start[i];
      if (num > n0) return true;  // agastPoints.at(0)[n] throws std::out_of_range
      const BriskLayerView Li = brisk_view_of(G, pyr_frame, smap_frame, i);
      for (int k = 0; k < num; ++k) {
        const BriskCand& c = C[order[start[0] + k]];
        const int x = c.x, y = c.y;
        bool oob = false;
        const bool is_max = brisk_ismax2d_literal<true>(Li, x, y, &oob);
        if (oob) return true;
        if (!is_max) continue;
        float dx, dy;
        const float mx = brisk_patch_subpixel_f<2>(Li, (float)x, (float)y, dx, dy);
        BriskKeyPoint kp;
        kp.x = (float)x + dx; kp.y = (float)y + dy;
        kp.size = BRISK_BASIC_SIZE * G.L[i].scale;
        kp.angle = -1.0f; kp.response = mx; kp.octave = 0; kp.class_id = -1;
        brisk_ordered_emit(out, kp);
      }
    }
    return false;
  }
  for (int r = 0; r < n; ++r) {
    const BriskCand& c = C[order[r]];
    const int l = c.layer, x = c.x, y = c.y;
    const bool has_above = !G.single_layer && (l + 1 < G.nlayers);
    const bool has_below = !G.single_layer && (l > 0);
    const BriskLayerView Lo = brisk_view_of(G, pyr_frame, smap_frame, l);
    const BriskLayerView La = brisk_view_of(G, pyr_frame, smap_frame, has_above ? l + 1 : l);
    const BriskLayerView Lb = brisk_view_of(G, pyr_frame, smap_frame, has_below ? l - 1 : l);
    bool oob = false;
    if (!brisk_ismax2d_literal<false>(Lo, x, y, &oob)) continue;
    BriskKeyPoint kp;
    kp.x = kp.y = kp.size = kp.response = 0.f;
    BriskTouch touch;
    touch.on = false; touch.mask = 0; touch.x0 = 0; touch.y0 = 0;
    bool e5 = false;
    if (!brisk_refine<2>(G, Lb, Lo, La, l, x, y, &kp, &e5, &touch)) continue;
    brisk_ordered_emit(out, kp);
  }
  return false;
}

// ---------------------------------------------------------------------------------------------
// ComputeScale (brisk-feature-detector.cc:87-92): GetKeypoints with a non-empty keypoint list
// (brisk-scale-space.cc:104-123 and the branches behind it with perform_2d_nonMax == false) on a pyramid built with
// lowerThreshold_ = 0, walked sequentially on the literal cache like brisk_ordered_walk.
//   * per layer the provided points at (x / scale - offset) as floats, admitted inside [3, cols - 3] x [3, rows - 3];
//     each is score-touched through the bilinear overload with threshold 0 (:121);
//   * GetAgastPoints (brisk-layer.cc:99-117): a layer with no admitted point detects (lower threshold 0); otherwise
//     the maps are addressed with int(x_float + y_float * cols) - a LINEAR index into matrices without row padding -
//     and scores_[offs] = cornerScore(b = thrmap_[offs]) with the ring read at linear offsets around that pixel;
//   * then the refinement branches, without IsMax2D, with the float coordinates where GetKeypoints keeps them.
// Returns true where the reference has no defined result (the ring of such a pixel leaves the image, or
// agastPoints.at(0)[n] throws); *cap_exceeded when the scratch list of a detecting layer is too small.
// ---------------------------------------------------------------------------------------------
BRISK_HD int brisk_img_linear(const BriskLayerView& L, long idx) { return L.img[(idx / L.w) * L.stride + (idx % L.w)]; }

BRISK_HD int brisk_M_linear(const BriskLayerView& L, long offs) {
  const long s = L.w;
  const int c = brisk_img_linear(L, offs);
  int d[16];
  d[0] = brisk_img_linear(L, offs - 3) - c;          d[1] = brisk_img_linear(L, offs - 3 - s) - c;
  d[2] = brisk_img_linear(L, offs - 2 - 2 * s) - c;  d[3] = brisk_img_linear(L, offs - 1 - 3 * s) - c;
  d[4] = brisk_img_linear(L, offs - 3 * s) - c;      d[5] = brisk_img_linear(L, offs + 1 - 3 * s) - c;
  d[6] = brisk_img_linear(L, offs + 2 - 2 * s) - c;  d[7] = brisk_img_linear(L, offs + 3 - s) - c;
  d[8] = brisk_img_linear(L, offs + 3) - c;          d[9] = brisk_img_linear(L, offs + 3 + s) - c;
  d[10] = brisk_img_linear(L, offs + 2 + 2 * s) - c; d[11] = brisk_img_linear(L, offs + 1 + 3 * s) - c;
  d[12] = brisk_img_linear(L, offs + 3 * s) - c;     d[13] = brisk_img_linear(L, offs - 1 + 3 * s) - c;
  d[14] = brisk_img_linear(L, offs - 2 + 2 * s) - c; d[15] = brisk_img_linear(L, offs - 3 + s) - c;
  return brisk_oast9_16_M_from_d(d);
}

// thrmap_ value of pixel (x, y): disc contrast inside, 0 on the 3-pixel border (brisk-layer.cc:278-598)
BRISK_HD int brisk_thrmap_at(const BriskLayerView& L, int x, int y) {
  if (x < 3 || y < 3 || x >= L.w - 3 || y >= L.h - 3) return 0;
  int mn, mx;
  brisk_disc_minmax(L.img + (long)y * L.stride + x, L.stride, &mn, &mx);
  return mx - mn;
}

struct BriskScaleList {  // agastPoints[i] of the ComputeScale walk
  bool detected;         // the layer had no admitted provided point: its list are its own detections (det[start ...])
  int count, start;
};

// the k-th ... iteration helper: coordinates of provided keypoint `idx` on layer l, and whether the layer admits it
BRISK_HD bool brisk_provided_on_layer(const BriskGeom& G, int l, const BriskKeyPoint& kin, float* kx, float* ky) {
  *kx = ((float)kin.x) / G.L[l].scale - G.L[l].offset;
  *ky = ((float)kin.y) / G.L[l].scale - G.L[l].offset;
  return !(*kx < 3 || *ky < 3 || *kx > G.L[l].w - 3 || *ky > G.L[l].h - 3);
}

// ---- the items of the ComputeScale walk (below).  The walk is sequential in the reference, but its three phases are
// order-free among themselves (round 5; k_cs_* in brisk_kernels.hip run one lane per (layer, provided point)):
//   touch    GetAgastScore(float, float, 0) of every admitted point (brisk-scale-space.cc:118-121): all accesses of the phase
//            use threshold 0, so what a pixel holds afterwards does not depend on their order;
//   score    GetAgastPoints on the provided list (brisk-layer.cc:106-116): a function of the image alone, written after the
//            layer's touches;
//   refine   everything GetKeypoints does per point (:172-287): threshold-1 accesses only, whose result is the cached value
//            where that is > 2 and K' otherwise - history-free once the first two phases of ALL layers are complete, and the
//            values they store are the ones any other access would store.
BRISK_HD void brisk_cs_touch(const BriskGeom& G, uint8_t* pyr_frame, uint16_t* smap_frame, int l, float kx, float ky) {
  const BriskLayerView L = brisk_view_of(G, pyr_frame, smap_frame, l);
  // GetAgastScore(float, float, 0): the four integer accesses with threshold 0 (the blend itself is discarded)
  const int x = (int)kx, y = (int)ky;
  brisk_S_literal(L, x, y, 0); brisk_S_literal(L, x + 1, y, 0); brisk_S_literal(L, x, y + 1, 0); brisk_S_literal(L, x + 1, y + 1, 0);
}
// returns true where the reference reads beyond the image (undefined there)
BRISK_HD bool brisk_cs_score(const BriskGeom& G, uint8_t* pyr_frame, uint16_t* smap_frame, int l, float kx, float ky) {
  const BriskLayerView L = brisk_view_of(G, pyr_frame, smap_frame, l);
  const long total = (long)L.w * L.h;
  const int offs = (int)(kx + ky * (float)L.w);
  if ((long)offs - 3 * L.w - 1 < 0 || (long)offs + 3 * L.w + 1 >= total) return true;  // the ring leaves the image
  const int ox = offs % L.w, oy = offs / L.w;
  const int thr = brisk_thrmap_at(L, ox, oy);
  const int M = brisk_M_linear(L, offs);
  L.smap[(long)oy * L.stride + ox] = (uint16_t)(uint8_t)brisk_max(thr, brisk_min(M - 1, 254));
  return false;
}
// :172-209 (one layer) and :211-287: false = the point is rejected on this layer
BRISK_HD bool brisk_cs_refine(const BriskGeom& G, uint8_t* pyr_frame, uint16_t* smap_frame, int i, const BriskKeyPoint& src, float kx,
                              float ky, BriskKeyPoint* out) {
  const bool has_above = !G.single_layer && (i + 1 < G.nlayers);
  const bool has_below = !G.single_layer && (i > 0);
  const BriskLayerView Lo = brisk_view_of(G, pyr_frame, smap_frame, i);
  const BriskLayerView La = brisk_view_of(G, pyr_frame, smap_frame, has_above ? i + 1 : i);
  const BriskLayerView Lb = brisk_view_of(G, pyr_frame, smap_frame, has_below ? i - 1 : i);
  float fxy[2] = {kx, ky};
  BriskTouch touch;
  touch.on = false; touch.mask = 0; touch.x0 = 0; touch.y0 = 0;
  bool e5 = false;
  BriskKeyPoint kp = src;
  if (!brisk_refine<2>(G, Lb, Lo, La, i, (int)fxy[0], (int)fxy[1], &kp, &e5, &touch, fxy)) return false;
  kp.class_id = src.class_id;  // `agast::KeyPoint kp = keypoint;` keeps the provided class_id (:199, 244, 275)
  *out = kp;
  return true;
}
// :131-170 (suppressScaleNonmaxima == false, several layers): layer i at the coordinates of LAYER 0's entry (src0, px, py)
BRISK_HD void brisk_cs_flat(const BriskGeom& G, uint8_t* pyr_frame, uint16_t* smap_frame, int i, const BriskKeyPoint& src0, float px,
                            float py, BriskKeyPoint* out) {
  const BriskLayerView Li = brisk_view_of(G, pyr_frame, smap_frame, i);
  BriskKeyPoint kp = src0;
  float dx, dy;
  const float mx = brisk_patch_subpixel_f<2>(Li, px, py, dx, dy);
  kp.x = px + dx; kp.y = py + dy;
  kp.size = BRISK_BASIC_SIZE * G.L[i].scale;
  kp.angle = -1.0f; kp.response = mx; kp.octave = 0;
  *out = kp;
}

BRISK_HD bool brisk_compute_scale_walk(const BriskGeom& G, uint8_t* pyr_frame, uint16_t* smap_frame, const BriskKeyPoint* in,
                                       int n_in, bool suppress, uint32_t* det, int det_cap, BriskOrderedOut* out,
                                       bool* cap_exceeded) {
  BriskScaleList lists[BRISK_MAX_LAYERS];
  int det_n = 0;
  *cap_exceeded = false;
  for (int i = 0; i < G.nlayers; ++i) {  // :99-126
    const BriskLayerView L = brisk_view_of(G, pyr_frame, smap_frame, i);
    lists[i].detected = false; lists[i].count = 0; lists[i].start = det_n;
    for (int k = 0; k < n_in; ++k) {
      float kx, ky;
      if (!brisk_provided_on_layer(G, i, in[k], &kx, &ky)) continue;
      brisk_cs_touch(G, pyr_frame, smap_frame, i, kx, ky);
      lists[i].count++;
    }
    if (lists[i].count == 0) {  // GetAgastPoints on an empty list: detect (lower threshold G.lower_threshold = 0), raster order
      lists[i].detected = true;
      for (int y = 3; y <= L.h - 4; ++y)
        for (int x = 3; x <= L.w - 4; ++x) {
          const int D = brisk_detect_px(L.img + (long)y * L.stride + x, L.stride, G.threshold, G.lower_threshold);
          if (!D) continue;
          if (det_n >= det_cap) { *cap_exceeded = true; return false; }
          det[det_n++] = (uint32_t)x | ((uint32_t)y << 16);
          lists[i].count++;
        }
      for (int k = 0; k < lists[i].count; ++k) {  // scores_[offs] = cornerScore(b = thrmap_[offs]) == thrmap_[offs]
        const int x = det[lists[i].start + k] & 0xFFFF, y = det[lists[i].start + k] >> 16;
        L.smap[(long)y * L.stride + x] = (uint16_t)brisk_thrmap_at(L, x, y);
      }
    } else {  // brisk-layer.cc:106-116 with the float coordinates
      for (int k = 0; k < n_in; ++k) {
        float kx, ky;
        if (!brisk_provided_on_layer(G, i, in[k], &kx, &ky)) continue;
        if (brisk_cs_score(G, pyr_frame, smap_frame, i, kx, ky)) return true;  // the ring leaves the image
      }
    }
  }
  // entry k of agastPoints[l]: float coordinates + the keypoint whose remaining fields the output copies
  // (sequential access only: `cursor` remembers where the previous entry of a provided-point list was found)
  auto entry = [&](int l, int k, int* cursor, float* px, float* py, BriskKeyPoint* src) {
    if (lists[l].detected) {
      const uint32_t e = det[lists[l].start + k];
      *px = (float)(e & 0xFFFF); *py = (float)(e >> 16);
      src->x = *px; src->y = *py; src->size = 0.f; src->angle = -1.f; src->response = 0.f; src->octave = 0; src->class_id = -1;
      return;
    }
    for (;; ++*cursor)
      if (brisk_provided_on_layer(G, l, in[*cursor], px, py)) { *src = in[*cursor]; ++*cursor; return; }
  };
  if (!suppress && !G.single_layer) {  // :131-170, perform_2d_nonMax == false
    for (int i = 0; i < G.nlayers; ++i) {
      if (lists[i].count > lists[0].count) return true;  // agastPoints.at(0)[n] throws std::out_of_range
      int cur0 = 0;
      for (int k = 0; k < lists[i].count; ++k) {
        float px, py;
        BriskKeyPoint src0, kp;
        entry(0, k, &cur0, &px, &py, &src0);
        brisk_cs_flat(G, pyr_frame, smap_frame, i, src0, px, py, &kp);
        brisk_ordered_emit(out, kp);
      }
    }
    return false;
  }
  for (int i = 0; i < G.nlayers; ++i) {  // :172-209 (one layer) and :211-287
    int cur = 0;
    for (int k = 0; k < lists[i].count; ++k) {
      float kx, ky;
      BriskKeyPoint src, kp;
      entry(i, k, &cur, &kx, &ky, &src);
      if (!brisk_cs_refine(G, pyr_frame, smap_frame, i, src, kx, ky, &kp)) continue;
      brisk_ordered_emit(out, kp);
    }
  }
  return false;
}

// ---------------------------------------------------------------------------------------------
// Order-faithful replay of the lazy score cache for tie candidates.
//
// raw map value of pixel p as candidate c = (cx, cy) sees it; `own` = c's own 8 probes are done.
// All candidates c' < c (raster order) within distance 2 of p must carry their final status.
// `last_layer`: the layer uses the float-access touch footprints (2x2 on pass, 4x4 patch).
// ---------------------------------------------------------------------------------------------
// kp5: optional 5x5 block (row-major, origin (cx-2, cy-2)) of pre-evaluated brisk_V() values (k_score_blocks);
// for a non-detection that value is K'.  EVAL = true ignores kp5 and evaluates from the image.
template <bool EVAL>
BRISK_HD int brisk_state_at(const BriskLayerView& L, const bool float_patch, const bool pass_touch2x2, int px,
                            int py, int cx, int cy, bool own, const uint16_t* sm_local, int lx0, int ly0, int lw,
                            const uint8_t* kp5) {
  // sm_local: smap window copy [ly0..][lx0..] of width lw covering p +- 2 (values 0 outside the image)
  if (brisk_border3(L, px, py)) return 0;
  const uint16_t* wp = sm_local + (py - ly0) * lw + (px - lx0);
  const unsigned smp = wp[0];
  const int D = BRISK_SM_D(smp);
  if (D > 2) return D;
  const int Kp = EVAL ? brisk_Kp(L, px, py) : (int)kp5[(py - cy + 2) * 5 + (px - cx + 2)];
  if (Kp == 0) return 0;
  bool cached = (smp & BRISK_SM_TOUCH) != 0, any = cached;
  int t_last = cached ? 1 : 0;
  // the 24 neighbours q = p + (ox, oy) in raster order; straight-line code (all geometry tests fold at compile
  // time, the window reads are unconditional: entries outside the image are 0 = "no candidate")
#pragma unroll
  for (int i = 0; i < 25; ++i) {
    const int oy = i / 5 - 2, ox = i % 5 - 2;
    if (ox == 0 && oy == 0) continue;
    const int ddx = -ox, ddy = -oy;  // p - q
    const int qx = px + ox, qy = py + oy;
    const unsigned smq = wp[oy * lw + ox];
    const int Dq = BRISK_SM_D(smq);
    const bool self = (qx == cx) && (qy == cy);
    const bool earlier = (qy < cy) || (qy == cy && qx <= cx);  // not later in raster order than the candidate
    const bool act = earlier && (Dq != 0) && !(self && !own);
    const bool near = ddx >= -1 && ddx <= 1 && ddy >= -1 && ddy <= 1;
    if (near) {  // phase 1: IsMax2D probe of p by q
      const bool pr = act && (brisk_probe_index(ddx, ddy) < (int)BRISK_SM_NPROBED(smq));
      any = any || pr;
      t_last = pr ? Dq : t_last;
      cached = cached || (pr && Dq <= Kp);
    }
    // phase 2: threshold-1 touches of a candidate that passed IsMax2D
    bool touched = false;
    if (!float_patch) {
      touched = near && (smq & BRISK_SM_E5);
    } else {
      if (pass_touch2x2 && ddx >= 0 && ddx <= 1 && ddy >= 0 && ddy <= 1) touched = true;
      if ((smq & BRISK_SM_E5) && ddx >= -1 && ddx <= 2 && ddy >= -1 && ddy <= 2) touched = true;
    }
    const bool tc = act && !self && (BRISK_SM_STATUS(smq) == BRISK_ST_PASS) && touched;
    any = any || tc;
    t_last = tc ? 1 : t_last;
    cached = cached || tc;
  }
  if (Kp >= 3 && cached) return Kp;
  if (!any) return 0;
  return (Kp >= t_last) ? Kp : 0;
}

// ---------------------------------------------------------------------------------------------
// brisk_state_at in two steps for the tie kernel.  The only inputs of the replay that can still change while a tie
// waits are the decisions of raster-earlier tie candidates that were pending (status TIE) when its window was read;
// they enter through the `tc` events alone.  brisk_state_static does everything else before the wait and lists the
// neighbours whose tc event is still open; brisk_state_resolve finishes with the decided statuses:
//   cached / any: OR over the events; t_last: the value of the LAST event in raster order (a neighbour's probe event
//   precedes its own touch event), so an open touch event that fires overrides t_last iff no static event follows it.
// packed: bits 0-7 Kp (or the final value), 8-15 t_last, 16-21 index of the last static event + 1, 22 cached, 23 any,
// 31 final (value needs no resolve).
// ---------------------------------------------------------------------------------------------
#define BRISK_SS_FINAL 0x80000000u
BRISK_HD int brisk_clz(unsigned v) { return __builtin_clz(v); }  // v != 0
// Which of a pixel's 24 neighbours can act on it at all depends only on where the pixel lies relative to the candidate
// (raster-earlier than the candidate or the candidate itself once its own probes are done) - in the tie kernel a constant
// of the lane: bit i of `actmask` = neighbour i may act, bit i of `nselfmask` = neighbour i is not the candidate itself.
BRISK_HD void brisk_state_masks(int sx, int sy /* pixel - candidate */, bool own, unsigned* actmask, unsigned* nselfmask) {
  unsigned am = 0, ns = 0;
#pragma unroll
  for (int i = 0; i < 25; ++i) {
    const int oy = i / 5 - 2, ox = i % 5 - 2;
    if (ox == 0 && oy == 0) continue;
    const int rx = sx + ox, ry = sy + oy;  // neighbour - candidate
    const bool self = (rx == 0) && (ry == 0);
    const bool earlier = (ry < 0) || (ry == 0 && rx <= 0);
    if (earlier && !(self && !own)) am |= 1u << i;
    if (!self) ns |= 1u << i;
  }
  *actmask = am;
  *nselfmask = ns;
}
BRISK_HD unsigned brisk_state_static_m(const BriskLayerView& L, const bool float_patch, const bool pass_touch2x2, int px,
                                       int py, int cx, int cy, const unsigned actmask, const unsigned nselfmask,
                                       const uint16_t* sm_local, int lx0, int ly0, int lw, const uint8_t* kp5, unsigned* dynmask) {
  *dynmask = 0;
  if (brisk_border3(L, px, py)) return BRISK_SS_FINAL;
  const uint16_t* wp = sm_local + (py - ly0) * lw + (px - lx0);
  const unsigned smp = wp[0];
  const int D = BRISK_SM_D(smp);
  if (D > 2) return BRISK_SS_FINAL | (unsigned)D;
  const int Kp = (int)kp5[(py - cy + 2) * 5 + (px - cx + 2)];
  if (Kp == 0) return BRISK_SS_FINAL;
  bool cached = (smp & BRISK_SM_TOUCH) != 0, any = cached;
  int t_last = cached ? 1 : 0;
  int last_static = 0;  // index + 1 of the last static event
  unsigned dyn = 0;
#pragma unroll
  for (int i = 0; i < 25; ++i) {
    const int oy = i / 5 - 2, ox = i % 5 - 2;
    if (ox == 0 && oy == 0) continue;
    const int ddx = -ox, ddy = -oy;  // p - q
    const unsigned smq = wp[oy * lw + ox];
    const int Dq = BRISK_SM_D(smq);
    const bool act = ((actmask >> i) & 1u) && (Dq != 0);
    const bool near = ddx >= -1 && ddx <= 1 && ddy >= -1 && ddy <= 1;
    if (near) {
      const bool pr = act && (brisk_probe_index(ddx, ddy) < (int)BRISK_SM_NPROBED(smq));
      any = any || pr;
      t_last = pr ? Dq : t_last;
      last_static = pr ? i + 1 : last_static;
      cached = cached || (pr && Dq <= Kp);
    }
    bool touched = false;
    if (!float_patch) {
      touched = near && (smq & BRISK_SM_E5);
    } else {
      if (pass_touch2x2 && ddx >= 0 && ddx <= 1 && ddy >= 0 && ddy <= 1) touched = true;
      if ((smq & BRISK_SM_E5) && ddx >= -1 && ddx <= 2 && ddy >= -1 && ddy <= 2) touched = true;
    }
    const bool open = act && ((nselfmask >> i) & 1u) && touched;
    const unsigned stq = BRISK_SM_STATUS(smq);
    const bool tc = open && (stq == BRISK_ST_PASS);
    any = any || tc;
    t_last = tc ? 1 : t_last;
    last_static = tc ? i + 1 : last_static;
    cached = cached || tc;
    dyn |= (open && stq == BRISK_ST_TIE) ? (1u << i) : 0u;
  }
  *dynmask = dyn;
  return (unsigned)Kp | ((unsigned)t_last << 8) | ((unsigned)last_static << 16) | (cached ? 1u << 22 : 0u) | (any ? 1u << 23 : 0u);
}
BRISK_HD unsigned brisk_state_static(const BriskLayerView& L, const bool float_patch, const bool pass_touch2x2, int px,
                                     int py, int cx, int cy, bool own, const uint16_t* sm_local, int lx0, int ly0, int lw,
                                     const uint8_t* kp5, unsigned* dynmask) {
  *dynmask = 0;
  if (brisk_border3(L, px, py)) return BRISK_SS_FINAL;
  const uint16_t* wp = sm_local + (py - ly0) * lw + (px - lx0);
  const unsigned smp = wp[0];
  const int D = BRISK_SM_D(smp);
  if (D > 2) return BRISK_SS_FINAL | (unsigned)D;
  const int Kp = (int)kp5[(py - cy + 2) * 5 + (px - cx + 2)];
  if (Kp == 0) return BRISK_SS_FINAL;
  bool cached = (smp & BRISK_SM_TOUCH) != 0, any = cached;
  int t_last = cached ? 1 : 0;
  int last_static = 0;  // index + 1 of the last static event
  unsigned dyn = 0;
#pragma unroll
  for (int i = 0; i < 25; ++i) {
    const int oy = i / 5 - 2, ox = i % 5 - 2;
    if (ox == 0 && oy == 0) continue;
    const int ddx = -ox, ddy = -oy;  // p - q
    const int qx = px + ox, qy = py + oy;
    const unsigned smq = wp[oy * lw + ox];
    const int Dq = BRISK_SM_D(smq);
    const bool self = (qx == cx) && (qy == cy);
    const bool earlier = (qy < cy) || (qy == cy && qx <= cx);
    const bool act = earlier && (Dq != 0) && !(self && !own);
    const bool near = ddx >= -1 && ddx <= 1 && ddy >= -1 && ddy <= 1;
    if (near) {
      const bool pr = act && (brisk_probe_index(ddx, ddy) < (int)BRISK_SM_NPROBED(smq));
      any = any || pr;
      t_last = pr ? Dq : t_last;
      last_static = pr ? i + 1 : last_static;
      cached = cached || (pr && Dq <= Kp);
    }
    bool touched = false;
    if (!float_patch) {
      touched = near && (smq & BRISK_SM_E5);
    } else {
      if (pass_touch2x2 && ddx >= 0 && ddx <= 1 && ddy >= 0 && ddy <= 1) touched = true;
      if ((smq & BRISK_SM_E5) && ddx >= -1 && ddx <= 2 && ddy >= -1 && ddy <= 2) touched = true;
    }
    const bool open = act && !self && touched;
    const unsigned stq = BRISK_SM_STATUS(smq);
    const bool tc = open && (stq == BRISK_ST_PASS);
    any = any || tc;
    t_last = tc ? 1 : t_last;
    last_static = tc ? i + 1 : last_static;
    cached = cached || tc;
    dyn |= (open && stq == BRISK_ST_TIE) ? (1u << i) : 0u;
  }
  *dynmask = dyn;
  return (unsigned)Kp | ((unsigned)t_last << 8) | ((unsigned)last_static << 16) | (cached ? 1u << 22 : 0u) | (any ? 1u << 23 : 0u);
}

// sm_local now carries the decided statuses of the neighbours listed in dynmask
BRISK_HD int brisk_state_resolve(unsigned packed, unsigned dynmask, int px, int py, const uint16_t* sm_local, int lx0,
                                 int ly0, int lw) {
  if (packed & BRISK_SS_FINAL) return (int)(packed & 0xFFu);
  const int Kp = (int)(packed & 0xFFu);
  int t_last = (int)((packed >> 8) & 0xFFu);
  const int last_static = (int)((packed >> 16) & 0x3Fu);
  bool cached = (packed >> 22) & 1u, any = (packed >> 23) & 1u;
  const uint16_t* wp = sm_local + (py - ly0) * lw + (px - lx0);
  int hi_fired = 0;  // index + 1 of the last open event that fires
  for (unsigned m = dynmask; m;) {
    const int i = 31 - brisk_clz(m);
    m &= ~(1u << i);
    const int oy = i / 5 - 2, ox = i % 5 - 2;
    if (BRISK_SM_STATUS((unsigned)wp[oy * lw + ox]) == BRISK_ST_PASS) { hi_fired = i + 1; break; }  // highest index first
  }
  if (hi_fired) {
    cached = true; any = true;
    if (hi_fired >= last_static) t_last = 1;
  }
  if (Kp >= 3 && cached) return Kp;
  if (!any) return 0;
  return (Kp >= t_last) ? Kp : 0;
}

// slot 0..7 = probe values, slot 8..32 = raw values (the tie kernel's lanes): static step / resolve step
BRISK_HD unsigned brisk_tie_slot_static(const BriskLayerView& L, const bool float_patch, const bool pass_touch2x2, int cx,
                                        int cy, int slot, const uint16_t* sm_local, int lx0, int ly0, int lw,
                                        const uint8_t* kp5, unsigned* dynmask) {
  const bool probe = slot < 8;
  const int q = probe ? 0 : slot - 8;
  const int px = cx + (probe ? brisk_probe_dx(slot) : (q % 5) - 2), py = cy + (probe ? brisk_probe_dy(slot) : (q / 5) - 2);
  *dynmask = 0;
  if (!probe && px == cx && py == cy) return BRISK_SS_FINAL;  // (the centre: the caller substitutes it)
  return brisk_state_static(L, float_patch, pass_touch2x2, px, py, cx, cy, !probe, sm_local, lx0, ly0, lw, kp5, dynmask);
}
// the static step of slot 0..32 with the lane's masks (brisk_state_masks of the slot's offset: computed once per lane)
BRISK_HD void brisk_tie_slot_offset(int slot, int* sx, int* sy, bool* own) {
  const bool probe = slot < 8;
  const int q = probe ? 0 : slot - 8;
  *sx = probe ? brisk_probe_dx(slot) : (q % 5) - 2;
  *sy = probe ? brisk_probe_dy(slot) : (q / 5) - 2;
  *own = !probe;
}
BRISK_HD unsigned brisk_tie_slot_static_m(const BriskLayerView& L, const bool float_patch, const bool pass_touch2x2, int cx, int cy,
                                          int slot, int sx, int sy, unsigned actmask, unsigned nselfmask, const uint16_t* sm_local,
                                          int lx0, int ly0, int lw, const uint8_t* kp5, unsigned* dynmask) {
  *dynmask = 0;
  if (slot >= 8 && sx == 0 && sy == 0) return BRISK_SS_FINAL;  // (the centre: the caller substitutes it)
  return brisk_state_static_m(L, float_patch, pass_touch2x2, cx + sx, cy + sy, cx, cy, actmask, nselfmask, sm_local, lx0, ly0, lw, kp5, dynmask);
}
BRISK_HD int brisk_tie_slot_resolve(const BriskLayerView& L, unsigned packed, unsigned dynmask, int cx, int cy, int centre,
                                    int slot, const uint16_t* sm_local, int lx0, int ly0, int lw, const uint8_t* kp5) {
  const bool probe = slot < 8;
  const int q = probe ? 0 : slot - 8;
  const int px = cx + (probe ? brisk_probe_dx(slot) : (q % 5) - 2), py = cy + (probe ? brisk_probe_dy(slot) : (q / 5) - 2);
  if (!probe && px == cx && py == cy) return centre;
  const int m = brisk_state_resolve(packed, dynmask, px, py, sm_local, lx0, ly0, lw);
  if (!probe || m > 2) return m;
  if (brisk_border3(L, px, py)) return 0;
  const int K = (int)kp5[(py - cy + 2) * 5 + (px - cx + 2)];
  return (K >= centre) ? K : 0;
}

// k_tie_resolve runs the layers of a frame as a pipeline: rows of the layer below that must be complete before a tie in
// row cy of this layer can be decided.  A tie of the layer below in row y touches rows y0 .. y0 + 3 of this layer with
// y0 = (int)((4y - 3) / 6) - 1 (below is an octave) or (int)((6y - 4) / 8) - 1 (below is an intra-octave)
// (brisk_score_max_other); the tie reads touches in rows cy - 2 .. cy + 2.  All rows < the returned value must be done.
BRISK_HD int brisk_tie_rows_needed(int cy, bool below_is_octave) {
  return below_is_octave ? ((cy + 5) * 3) / 2 + 2 : ((cy + 5) * 4) / 3 + 2;
}

// IsMax2D steps 3-4 (brisk-scale-space.cc:499-530) for a tie candidate, split so that the per-pixel
// cache replays can run one lane per pixel:
//   ret[k]   (k = 0..7, probe order)  value the candidate's k-th probe returns
//   raw[25]  raw map values of the 5x5 block around the candidate after its 8 probes
template <bool EVAL>
BRISK_HD int brisk_tie_probe_value(const BriskLayerView& L, const bool float_patch, const bool pass_touch2x2, int cx,
                                   int cy, int centre, int k, const uint16_t* sm_local, int lx0, int ly0, int lw,
                                   const uint8_t* kp5) {
  const int nx = cx + brisk_probe_dx(k), ny = cy + brisk_probe_dy(k);
  const int m = brisk_state_at<EVAL>(L, float_patch, pass_touch2x2, nx, ny, cx, cy, false, sm_local, lx0, ly0, lw, kp5);
  if (m > 2) return m;
  if (brisk_border3(L, nx, ny)) return 0;
  const int K = EVAL ? brisk_Kp(L, nx, ny) : (int)kp5[(ny - cy + 2) * 5 + (nx - cx + 2)];
  return (K >= centre) ? K : 0;
}

template <bool EVAL>
BRISK_HD int brisk_tie_raw_value(const BriskLayerView& L, const bool float_patch, const bool pass_touch2x2, int cx,
                                 int cy, int centre, int q /* 0..24, row-major 5x5 */, const uint16_t* sm_local,
                                 int lx0, int ly0, int lw, const uint8_t* kp5) {
  const int qx = cx + (q % 5) - 2, qy = cy + (q / 5) - 2;
  if (qx == cx && qy == cy) return centre;
  return brisk_state_at<EVAL>(L, float_patch, pass_touch2x2, qx, qy, cx, cy, true, sm_local, lx0, ly0, lw, kp5);
}

// slot 0..7 = probe values, slot 8..32 = raw values: one code path for all lanes of the tie kernel
template <bool EVAL>
BRISK_HD int brisk_tie_slot_value(const BriskLayerView& L, const bool float_patch, const bool pass_touch2x2, int cx,
                                  int cy, int centre, int slot, const uint16_t* sm_local, int lx0, int ly0, int lw,
                                  const uint8_t* kp5) {
  const bool probe = slot < 8;
  const int q = probe ? 0 : slot - 8;
  const int px = cx + (probe ? brisk_probe_dx(slot) : (q % 5) - 2), py = cy + (probe ? brisk_probe_dy(slot) : (q / 5) - 2);
  if (!probe && px == cx && py == cy) return centre;
  const int m = brisk_state_at<EVAL>(L, float_patch, pass_touch2x2, px, py, cx, cy, !probe, sm_local, lx0, ly0, lw, kp5);
  if (!probe || m > 2) return m;
  if (brisk_border3(L, px, py)) return 0;
  const int K = EVAL ? brisk_Kp(L, px, py) : (int)kp5[(py - cy + 2) * 5 + (px - cx + 2)];
  return (K >= centre) ? K : 0;
}

// one neighbour of the tie list (o = 0..7 in list order): false iff it is an equal-score neighbour whose smoothed
// 3x3 sum exceeds the candidate's.  brisk_tie_decide() == all eight of these (the kernel evaluates them on 8 lanes).
BRISK_HD bool brisk_tie_neighbour_ok(int centre, const int* s /* 8 probe values */, const int* raw /* 25 */, int o) {
  const int smoothedcenter = 4 * centre + 2 * (s[0] + s[1] + s[2] + s[3]) + s[7] + s[6] + s[4] + s[5];
  const int k = (int)((0x53410627u >> (4 * o)) & 7u);  // tie list order {7, 2, 6, 0, 1, 4, 3, 5}
  if (s[k] != centre) return true;
  const int nx = brisk_probe_dx(k) + 2, ny = brisk_probe_dy(k) + 2;  // position inside the 5x5 block
  int other = 0;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) other += (dx == 0 ? 2 : 1) * (dy == 0 ? 2 : 1) * raw[(ny + dy) * 5 + nx + dx];
  return !(other > smoothedcenter);
}

BRISK_HD bool brisk_tie_decide(int centre, const int* s /* 8 probe values */, const int* raw /* 25 */) {
  // s: W,E,N,S,SW,SE,NE,NW
  const int smoothedcenter = 4 * centre + 2 * (s[0] + s[1] + s[2] + s[3]) + s[7] + s[6] + s[4] + s[5];
  // tie list order: (-1,-1),(0,-1),(1,-1),(-1,0),(1,0),(-1,1),(0,1),(1,1)
  const int order[8] = {7, 2, 6, 0, 1, 4, 3, 5};
  for (int o = 0; o < 8; ++o) {
    const int k = order[o];
    if (s[k] != centre) continue;
    const int nx = brisk_probe_dx(k) + 2, ny = brisk_probe_dy(k) + 2;  // position inside the 5x5 block
    int other = 0;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) other += (dx == 0 ? 2 : 1) * (dy == 0 ? 2 : 1) * raw[(ny + dy) * 5 + nx + dx];
    if (other > smoothedcenter) return false;
  }
  return true;
}

template <bool EVAL>
BRISK_HD bool brisk_tie_eval(const BriskLayerView& L, const bool float_patch, const bool pass_touch2x2, int cx,
                             int cy, const uint16_t* sm_local, int lx0, int ly0, int lw, const uint8_t* kp5) {
  const int centre = BRISK_SM_D(sm_local[(cy - ly0) * lw + (cx - lx0)]);
  int s[8], raw[25];
  for (int k = 0; k < 8; ++k) s[k] = brisk_tie_probe_value<EVAL>(L, float_patch, pass_touch2x2, cx, cy, centre, k, sm_local, lx0, ly0, lw, kp5);
  for (int q = 0; q < 25; ++q) raw[q] = brisk_tie_raw_value<EVAL>(L, float_patch, pass_touch2x2, cx, cy, centre, q, sm_local, lx0, ly0, lw, kp5);
  return brisk_tie_decide(centre, s, raw);
}
