// brisk_device_describe.h - per-item device logic of the BRISK descriptor
// (brisk/src/brisk-descriptor-extractor.cc).  `__host__ __device__` for the same reason as
// brisk_device_detect.h.  Compile with -ffp-contract=off.
#pragma once
#include <math.h>

#include "brisk_common.h"

#ifndef BRISK_PI
#define BRISK_PI 3.14159265358979323846
#endif

// Scale index of a keypoint (:636-650).  The reference evaluates
//   max(int(64 / lb * (log(size / 7.2f) / log2f) + 0.5), 0) saturated to 63
// with host libm; the host pre-computes, by bisection over that exact expression, the smallest
// float size reaching each index, so the device needs no transcendental and cannot disagree.
BRISK_HD int brisk_scale_index(const BriskPatternDev& P, float size) {
  if (!P.scale_invariant) return P.basicscale;
  int s = 0;
#pragma unroll 8
  for (int i = 1; i < BRISK_SCALES; ++i) s += (size >= P.size_thresh[i]) ? 1 : 0;
  return s;
}

// RoiPredicate (:532-536) negated: keypoint stays iff border <= x < cols-border (same for y)
BRISK_HD bool brisk_inside_border(const BriskPatternDev& P, int scale, float x, float y, int cols, int rows) {
  const int border = P.size_list[scale];
  const float minX = (float)border, minY = (float)border;
  const float maxX = (float)(cols - border), maxY = (float)(rows - border);
  return !((x < minX) || (x >= maxX) || (y < minY) || (y >= maxY));
}

// Pattern point (scale, rot, i): the reference tabulates
//   x = float(scaleList[s] * (u_x cos(theta) - u_y sin(theta)))      (:231-234, V2)
//   x = float(scaleList[s] * r * cos(alpha + theta))                 (:105-108, V1)
// for all 64 x 1024 x points (51.9 MB).  Both are float(double(m[s][i]) * U[rot][i]) with a
// scale-independent double U, so the engine keeps U (1 MB, L2-resident) and m, sigma (34 KB).
struct BriskSamplePoint {
  float x, y, sigma;
  int scaling, scaling2;  // int(4194304.0 / area), int(float(scaling) * area / 1024.0) with area = float(4 sigma^2)
};

BRISK_HD BriskSamplePoint brisk_pattern_point(const BriskPatternDev& P, int scale, int rot, int i) {
  BriskSamplePoint sp;
  const int si = scale * P.npoints + i;
  const double m = (double)P.mult[si];
  const double* uv = P.uv + ((long)rot * P.npoints + i) * 2;
  sp.x = (float)(m * uv[0]);
  sp.y = (float)(m * uv[1]);
  sp.sigma = P.sigma[si];
  sp.scaling = P.scaling[2 * si];
  sp.scaling2 = P.scaling[2 * si + 1];
  return sp;
}

// SmoothedIntensity<uchar,int> (:370-530).  Weighted box sum: 4 corner pixels (weights A..D), 4 edge
// strips and the interior, the strips/interior taken from the integral image (12 samples).  The
// loop branch (:497-529, dx+dy <= 2) is the same integer with the true corners.  The integral branch
// (:445-495, dx+dy > 2) steps `ptr += dy*imagecols + 1` (:453), so its C and D weights multiply the
// pixels (x_right+1, y_bottom-1) and (x_left+1, y_bottom-1) instead of the bottom corners; the
// golden vectors contain this behaviour, so it is reproduced.
// integral: exclusive prefix sums, (rows+1) x (cols+1), row stride istride (u32, wrap-around).
// Pixel the reference reads at data[y * cols + x] for x that may reach cols (the displaced corner of :453 sits one
// column right of the box: for a box that ends in the last column the linear address is the first pixel of the next
// row).  The engine's rows are padded (stride >= cols), so the wrap is made explicit.
BRISK_HD unsigned brisk_linear_px(const uint8_t* img, int stride, int cols, int x, int y) {
  if (x >= cols) { x -= cols; y += 1; }
  return img[(long)y * stride + x];
}

BRISK_HD int brisk_smoothed_intensity(const uint8_t* img, int stride, int cols, const uint32_t* integral, int istride,
                                      float key_x, float key_y, const BriskSamplePoint& sp) {
  const float sigma_half = sp.sigma;
  const float xf = sp.x + key_x;
  const float yf = sp.y + key_y;
  if (sigma_half < 0.5) {  // :391-408
    const int x = (int)xf, y = (int)yf;
    const int r_x = (int)((xf - x) * 1024);
    const int r_y = (int)((yf - y) * 1024);
    const int r_x_1 = (1024 - r_x);
    const int r_y_1 = (1024 - r_y);
    const uint8_t* ptr = img + x + (long)y * stride;
    int ret_val = (r_x_1 * r_y_1 * (int)ptr[0]);
    ret_val += (r_x * r_y_1 * (int)ptr[1]);
    ret_val += (r_x * r_y * (int)ptr[stride + 1]);
    ret_val += (r_x_1 * r_y * (int)ptr[stride]);
    return (ret_val) / 1024;
  }
  const int scaling = sp.scaling;
  const int scaling2 = sp.scaling2;
  const float x_1 = xf - sigma_half;
  const float x1 = xf + sigma_half;
  const float y_1 = yf - sigma_half;
  const float y1 = yf + sigma_half;
  const int x_left = (int)(x_1 + 0.5);
  const int y_top = (int)(y_1 + 0.5);
  const int x_right = (int)(x1 + 0.5);
  const int y_bottom = (int)(y1 + 0.5);
  const float r_x_1 = (float)((float)x_left - x_1 + 0.5);
  const float r_y_1 = (float)((float)y_top - y_1 + 0.5);
  const float r_x1 = (float)(x1 - (float)x_right + 0.5);
  const float r_y1 = (float)(y1 - (float)y_bottom + 0.5);
  const int dx = x_right - x_left - 1;
  const int dy = y_bottom - y_top - 1;
  const unsigned A = (unsigned)(int)((r_x_1 * r_y_1) * scaling);
  const unsigned B = (unsigned)(int)((r_x1 * r_y_1) * scaling);
  const unsigned C = (unsigned)(int)((r_x1 * r_y1) * scaling);
  const unsigned D = (unsigned)(int)((r_x_1 * r_y1) * scaling);
  const unsigned r_x_1_i = (unsigned)(int)(r_x_1 * scaling);
  const unsigned r_y_1_i = (unsigned)(int)(r_y_1 * scaling);
  const unsigned r_x1_i = (unsigned)(int)(r_x1 * scaling);
  const unsigned r_y1_i = (unsigned)(int)(r_y1 * scaling);
  // Integral samples at columns {xl, xl+1, xr, xr+1} x rows {yt, yt+1, yb, yb+1}: 8 gathers of two adjacent
  // columns each.  The corner PIXELS are differences of those same samples (exact in wrap-around arithmetic), so
  // the image itself is only read for the two displaced bottom corners of the reference quirk.
  const bool quirk = (dx + dy > 2);
  const uint32_t* r0 = integral + (long)y_top * istride;
  const uint32_t* r1 = r0 + istride;
  const uint32_t* r2 = integral + (long)y_bottom * istride;
  const uint32_t* r3 = r2 + istride;
  const int c0 = x_left, c2 = x_right;
#if defined(__HIP_DEVICE_COMPILE__)
  // adjacent column pairs as one 8-byte gather each (4-byte aligned is enough for global_load_dwordx2)
  typedef uint32_t __attribute__((ext_vector_type(2), aligned(4))) u32x2_t;
  const u32x2_t p00 = *reinterpret_cast<const u32x2_t*>(r0 + c0), p02 = *reinterpret_cast<const u32x2_t*>(r0 + c2);
  const u32x2_t p10 = *reinterpret_cast<const u32x2_t*>(r1 + c0), p12 = *reinterpret_cast<const u32x2_t*>(r1 + c2);
  const u32x2_t p20 = *reinterpret_cast<const u32x2_t*>(r2 + c0), p22 = *reinterpret_cast<const u32x2_t*>(r2 + c2);
  const u32x2_t p30 = *reinterpret_cast<const u32x2_t*>(r3 + c0), p32 = *reinterpret_cast<const u32x2_t*>(r3 + c2);
  const uint32_t i00 = p00.x, i01 = p00.y, i02 = p02.x, i03 = p02.y;
  const uint32_t i10 = p10.x, i11 = p10.y, i12 = p12.x, i13 = p12.y;
  const uint32_t i20 = p20.x, i21 = p20.y, i22 = p22.x, i23 = p22.y;
  const uint32_t i30 = p30.x, i31 = p30.y, i32 = p32.x, i33 = p32.y;
#else
  const uint32_t i00 = r0[c0], i01 = r0[c0 + 1], i02 = r0[c2], i03 = r0[c2 + 1];
  const uint32_t i10 = r1[c0], i11 = r1[c0 + 1], i12 = r1[c2], i13 = r1[c2 + 1];
  const uint32_t i20 = r2[c0], i21 = r2[c0 + 1], i22 = r2[c2], i23 = r2[c2 + 1];
  const uint32_t i30 = r3[c0], i31 = r3[c0 + 1], i32 = r3[c2], i33 = r3[c2 + 1];
#endif
  const unsigned tl = i11 - i01 - i10 + i00;  // pixel (x_left, y_top)
  const unsigned tr = i13 - i03 - i12 + i02;  // pixel (x_right, y_top)
  unsigned br, bl;
  if (quirk) {
    br = brisk_linear_px(img, stride, cols, x_right + 1, y_bottom - 1);
    bl = brisk_linear_px(img, stride, cols, x_left + 1, y_bottom - 1);
  } else {
    br = i33 - i23 - i32 + i22;  // pixel (x_right, y_bottom)
    bl = i31 - i21 - i30 + i20;  // pixel (x_left, y_bottom)
  }
  const uint32_t top = i12 - i11 - i02 + i01;     // first row, interior columns
  const uint32_t bottom = i32 - i31 - i22 + i21;  // last row, interior columns
  const uint32_t left = i21 - i20 - i11 + i10;    // first column, interior rows
  const uint32_t right = i23 - i22 - i13 + i12;   // last column, interior rows
  const uint32_t middle = i22 - i21 - i12 + i11;  // interior
  const uint32_t acc = A * tl + B * tr + C * br + D * bl + r_y_1_i * top + r_y1_i * bottom + r_x_1_i * left +
                       r_x1_i * right + (unsigned)scaling * middle;
  return (int)acc / scaling2;
}

// Exact signed division by an invariant divisor d >= 2 (Granlund / Montgomery, "Hacker's Delight" 10-1): multiplier and
// shift on the host, three or four integer operations per division on the device.  n / d == brisk_div_by_magic(n, M, sh)
// for EVERY 32-bit n (C semantics: truncation toward zero), checked in tests/test_emul_parity.py.
BRISK_HD void brisk_div_magic(int d, int* M, int* sh) {
  const unsigned two31 = 0x80000000u;
  const unsigned ad = (unsigned)d;
  const unsigned t = two31;
  const unsigned anc = t - 1 - t % ad;
  int p = 31;
  unsigned q1 = two31 / anc, r1 = two31 - q1 * anc, q2 = two31 / ad, r2 = two31 - q2 * ad, delta;
  do {
    p = p + 1;
    q1 = 2 * q1; r1 = 2 * r1;
    if (r1 >= anc) { q1 = q1 + 1; r1 = r1 - anc; }
    q2 = 2 * q2; r2 = 2 * r2;
    if (r2 >= ad) { q2 = q2 + 1; r2 = r2 - ad; }
    delta = ad - r2;
  } while (q1 < delta || (q1 == delta && r1 == 0));
  *M = (int)(q2 + 1);
  *sh = p - 32;
}
BRISK_HD int brisk_div_by_magic(int n, int M, int sh) {
  int q = (int)(((long long)M * (long long)n) >> 32);
  if (M < 0) q += n;
  q >>= sh;
  return q + (int)((unsigned)q >> 31);
}

// ---- SmoothedIntensity, box branch (:410-530), split into the stages k_describe runs: address / weights, the 4 x 4
// integral samples + the two displaced corner pixels (loaded by the caller), combination.  Same arithmetic as
// brisk_smoothed_intensity above; tests/emul runs these on the CPU against the oracle.
// tab_z / tab_w: BriskPatternDev::tab words 2 and 3 of the (scale, point): scaling | shift << 24 | plain << 30, and the
// magic multiplier of scaling2 (plain: scaling2 itself) - brisk_pack_tab.
BRISK_HD void brisk_pack_tab(int scaling, int scaling2, int* tab_z, int* tab_w) {
  if (scaling2 >= 2 && scaling >= 0 && scaling < (1 << 24)) {
    int M, sh;
    brisk_div_magic(scaling2, &M, &sh);
    *tab_z = scaling | (sh << 24);
    *tab_w = M;
  } else {  // degenerate boxes (never sampled through the box branch in practice): plain division on the device
    *tab_z = (scaling & 0xFFFFFF) | (1 << 30);
    *tab_w = scaling2;
  }
}
struct BriskBoxPrep {
  int x_left, y_top, x_right, y_bottom;
  unsigned A, B, C, D, r_x_1_i, r_y_1_i, r_x1_i, r_y1_i;
  int scaling, magic, shift;  // acc / scaling2 as a multiplication (brisk_div_by_magic); shift < 0: plain division by `magic`
  bool quirk;                 // dx + dy > 2: the integral branch with its displaced bottom corners (:453)
};
BRISK_HD BriskBoxPrep brisk_box_prep(float xf, float yf, float sigma_half, int tab_z, int tab_w) {
  BriskBoxPrep p;
  const int scaling = tab_z & 0xFFFFFF;
  p.scaling = scaling; p.magic = tab_w; p.shift = (tab_z & (1 << 30)) ? -1 : ((tab_z >> 24) & 31);
  const float x_1 = xf - sigma_half, x1 = xf + sigma_half, y_1 = yf - sigma_half, y1 = yf + sigma_half;
  p.x_left = (int)(x_1 + 0.5); p.y_top = (int)(y_1 + 0.5); p.x_right = (int)(x1 + 0.5); p.y_bottom = (int)(y1 + 0.5);
  const float r_x_1 = (float)((float)p.x_left - x_1 + 0.5);
  const float r_y_1 = (float)((float)p.y_top - y_1 + 0.5);
  const float r_x1 = (float)(x1 - (float)p.x_right + 0.5);
  const float r_y1 = (float)(y1 - (float)p.y_bottom + 0.5);
  const int dx = p.x_right - p.x_left - 1, dy = p.y_bottom - p.y_top - 1;
  p.A = (unsigned)(int)((r_x_1 * r_y_1) * scaling);
  p.B = (unsigned)(int)((r_x1 * r_y_1) * scaling);
  p.C = (unsigned)(int)((r_x1 * r_y1) * scaling);
  p.D = (unsigned)(int)((r_x_1 * r_y1) * scaling);
  p.r_x_1_i = (unsigned)(int)(r_x_1 * scaling);
  p.r_y_1_i = (unsigned)(int)(r_y_1 * scaling);
  p.r_x1_i = (unsigned)(int)(r_x1 * scaling);
  p.r_y1_i = (unsigned)(int)(r_y1 * scaling);
  p.quirk = (dx + dy > 2);
  return p;
}
// i[r][c]: integral samples at rows {y_top, y_top + 1, y_bottom, y_bottom + 1} x columns {x_left, x_left + 1, x_right,
// x_right + 1}; br / bl: image pixels (x_right + 1, y_bottom - 1) / (x_left + 1, y_bottom - 1) (used when p.quirk).
// Returns the weighted sum before the division by scaling2.
// mask: 0xFFFFFFFF for the 32-bit integral image; 0xFFFFFF when the samples are integral values modulo 2^24 (every region
// sum of the box - at most (2 sigma + 3)^2 pixels - is below 2^24, so the difference modulo 2^24 IS the region sum).
BRISK_HD uint32_t brisk_box_acc(const BriskBoxPrep& p, uint32_t i00, uint32_t i01, uint32_t i02, uint32_t i03, uint32_t i10,
                                uint32_t i11, uint32_t i12, uint32_t i13, uint32_t i20, uint32_t i21, uint32_t i22, uint32_t i23,
                                uint32_t i30, uint32_t i31, uint32_t i32, uint32_t i33, unsigned qbr, unsigned qbl,
                                uint32_t mask = 0xFFFFFFFFu) {
  const unsigned tl = (i11 - i01 - i10 + i00) & mask;  // pixel (x_left, y_top)
  const unsigned tr = (i13 - i03 - i12 + i02) & mask;  // pixel (x_right, y_top)
  const unsigned br = p.quirk ? qbr : ((i33 - i23 - i32 + i22) & mask);
  const unsigned bl = p.quirk ? qbl : ((i31 - i21 - i30 + i20) & mask);
  const uint32_t top = (i12 - i11 - i02 + i01) & mask;
  const uint32_t bottom = (i32 - i31 - i22 + i21) & mask;
  const uint32_t left = (i21 - i20 - i11 + i10) & mask;
  const uint32_t right = (i23 - i22 - i13 + i12) & mask;
  const uint32_t middle = (i22 - i21 - i12 + i11) & mask;
  return p.A * tl + p.B * tr + p.C * br + p.D * bl + p.r_y_1_i * top + p.r_y1_i * bottom + p.r_x_1_i * left + p.r_x1_i * right +
         (unsigned)p.scaling * middle;
}
// The same sum from the two SIDES of the box (round 5: k_describe reads a box's left and right column pairs on the two lanes
// of a lane pair, brisk_describe.hip).  A side = the integral samples of one column pair: t0 / t1 = rows y_top, y_top + 1 at
// columns (c, c + 1); q = row y_bottom - 1 at (c + 1, c + 2); b0 = row y_bottom at (c, c + 1, c + 2); b1 = row y_bottom + 1 at
// (c, c + 1), with c = x_left or x_right.  It reduces to six numbers: its corner pixels (top; bottom - the displaced one of the
// reference quirk when p.quirk), its column strip, and three differences down its INNER column (c + 1 on the left, c on the
// right), from which top / bottom / middle of brisk_box_acc follow as right minus left.  brisk_box_acc_pair(own, partner) is
// brisk_box_acc for the lane that holds `own` (odd: the right side); tests/test_emul_parity.py checks the identity.
struct BriskBoxSide {
  uint32_t ct, cb, st, dt, db, dm;
};
BRISK_HD BriskBoxSide brisk_box_side(uint32_t t0x, uint32_t t0y, uint32_t t1x, uint32_t t1y, uint32_t qx, uint32_t qy, uint32_t b0x,
                                     uint32_t b0y, uint32_t b0z, uint32_t b1x, uint32_t b1y, bool quirk, bool right, uint32_t mask) {
  BriskBoxSide r;
  r.ct = (t1y - t0y - t1x + t0x) & mask;
  r.cb = (quirk ? (b0z - b0y - qy + qx) : (b1y - b0y - b1x + b0x)) & mask;
  r.st = (b0y - b0x - t1y + t1x) & mask;
  const uint32_t t0i = right ? t0x : t0y, t1i = right ? t1x : t1y, b0i = right ? b0x : b0y, b1i = right ? b1x : b1y;
  r.dt = t1i - t0i;
  r.db = b1i - b0i;
  r.dm = b0i - t1i;
  return r;
}
BRISK_HD uint32_t brisk_box_acc_pair(const BriskBoxPrep& p, const BriskBoxSide& own, const BriskBoxSide& par, bool odd, uint32_t mask) {
  const uint32_t tt = par.dt - own.dt, tb = par.db - own.db, tm = par.dm - own.dm;  // right minus left on the even lane
  const uint32_t top = (odd ? 0u - tt : tt) & mask, bottom = (odd ? 0u - tb : tb) & mask, middle = (odd ? 0u - tm : tm) & mask;
  const unsigned w_own_t = odd ? p.B : p.A, w_par_t = odd ? p.A : p.B, w_own_b = odd ? p.C : p.D, w_par_b = odd ? p.D : p.C;
  const unsigned w_own_s = odd ? p.r_x1_i : p.r_x_1_i, w_par_s = odd ? p.r_x_1_i : p.r_x1_i;
  return w_own_t * own.ct + w_par_t * par.ct + w_own_b * own.cb + w_par_b * par.cb + p.r_y_1_i * top + p.r_y1_i * bottom +
         w_own_s * own.st + w_par_s * par.st + (unsigned)p.scaling * middle;
}
BRISK_HD int brisk_box_divide(const BriskBoxPrep& p, uint32_t acc) {
  return p.shift < 0 ? (int)acc / p.magic : brisk_div_by_magic((int)acc, p.magic, p.shift);
}

// long-pair contribution (:721-730): C integer division truncates toward zero
BRISK_HD void brisk_long_pair(const int* values, const int* lp /* i, j, wdx, wdy */, int* d0, int* d1) {
  const int delta_t = values[lp[0]] - values[lp[1]];
  *d0 = delta_t * lp[2] / 1024;
  *d1 = delta_t * lp[3] / 1024;
}

// orientation (:732-739)
BRISK_HD float brisk_angle_from_direction(int direction0, int direction1) {
  return (float)(atan2((double)(float)direction1, (double)(float)direction0) / BRISK_PI * 180.0);
}

BRISK_HD int brisk_theta_from_angle(float angle, bool estimated) {
  int theta;
  if (estimated) theta = (int)((BRISK_NROT * angle) / (360.0) + 0.5);       // :734-735 (float product)
  else theta = (int)(BRISK_NROT * (angle / (360.0)) + 0.5);                  // :746-747 (double product)
  if (theta < 0) theta += BRISK_NROT;
  if (theta >= BRISK_NROT) theta -= BRISK_NROT;
  return theta;
}
