/*
 * brisk_oracle_detect.c - CPU restatement of the BRISK scale-space detector (oracle; TEST ONLY).
 * See brisk_oracle.h for the rules.  Scalar C99, no SIMD, literal lazy score cache, literal
 * sequential candidate order.  All float/double mixing follows the reference expressions
 * (un-suffixed literals are double).  Build with -ffp-contract=off (oracle/Makefile).
 */
#include "brisk_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* Down-sampling: brisk/src/image-down-sampling.cc                                             */
/* ------------------------------------------------------------------------------------------ */

static inline int avg_u8(int a, int b) { return (a + b + 1) >> 1; } /* pavgb */

/* image-down-sampling.cc:142-392.  Three column classes per output row:
 *   full SIMD pairs of 16-blocks  -> avg(avg(t,b)) twice-rounded-up      (:296-339)
 *   one odd trailing 16-block     -> (v0 + v1) / 2 truncating            (:341-365)
 *   (w % 16) / 2 leftover columns -> (a + b + c + d + 2) / 4             (:376-382)
 * Rows: while the second source row pointer is below end - leftover (:286-295). */
void bo_halfsample8(const uint8_t* src, int w, int h, uint8_t* dst) {
  const int leftover = (w % 16) / 2;
  const int hsize = w / 16;
  const int end = hsize / 2;
  const int half_end = hsize % 2;
  const int dw = w / 2;
  const long total = (long)w * h - leftover;
  int row = 0;
  while ((long)(2 * row + 1) * w < total) {
    const uint8_t* p1 = src + (long)(2 * row) * w;
    const uint8_t* p2 = p1 + w;
    uint8_t* d = dst + (long)row * dw;
    int c = 0; /* output column */
    for (int i = 0; i < end; ++i) {
      for (int k = 0; k < 16; ++k, ++c) {
        int v0 = avg_u8(p1[2 * c], p2[2 * c]);
        int v1 = avg_u8(p1[2 * c + 1], p2[2 * c + 1]);
        d[c] = (uint8_t)avg_u8(v0, v1);
      }
    }
    if (half_end) {
      for (int j = 0; j < 8; ++j, ++c) {
        int v0 = avg_u8(p1[2 * c], p2[2 * c]);
        int v1 = avg_u8(p1[2 * c + 1], p2[2 * c + 1]);
        d[c] = (uint8_t)((v0 + v1) / 2);
      }
    }
    for (int k = 0; k < leftover; ++k, ++c) {
      unsigned tmp = (unsigned)p1[2 * c] + p1[2 * c + 1] + p2[2 * c] + p2[2 * c + 1];
      d[c] = (uint8_t)((tmp + 2) / 4);
    }
    ++row;
  }
}

/* image-down-sampling.cc:550-787.  Per 3 source rows two output rows; 15-column SIMD blocks
 * (:712-751: u = avg(avg(A,B),A), l = avg(avg(C,B),C); per triple o0 = avg(avg(p0,p1),p0),
 * o1 = avg(avg(p2,p1),p2)), then ((w/3)*3) % 15 leftover columns with the /9 formula (:754-773). */
void bo_twothirdsample8(const uint8_t* src, int w, int h, uint8_t* dst) {
  const int leftover = ((w / 3) * 3) % 15;
  const int hsize = w / 15;
  const int dw = (w / 3) * 2;
  int row = 0, row_dest = 0;
  while (row + 2 < h) {
    const uint8_t* p1 = src + (long)row * w;
    const uint8_t* p2 = p1 + w;
    const uint8_t* p3 = p2 + w;
    uint8_t* d1 = dst + (long)row_dest * dw;
    uint8_t* d2 = d1 + dw;
    for (int i = 0; i < hsize; ++i) {
      for (int t = 0; t < 5; ++t) {
        int u[3], l[3];
        for (int k = 0; k < 3; ++k) {
          int a = p1[3 * t + k], b = p2[3 * t + k], c = p3[3 * t + k];
          u[k] = avg_u8(avg_u8(a, b), a);
          l[k] = avg_u8(avg_u8(c, b), c);
        }
        d1[2 * t] = (uint8_t)avg_u8(avg_u8(u[0], u[1]), u[0]);
        d1[2 * t + 1] = (uint8_t)avg_u8(avg_u8(u[2], u[1]), u[2]);
        d2[2 * t] = (uint8_t)avg_u8(avg_u8(l[0], l[1]), l[0]);
        d2[2 * t + 1] = (uint8_t)avg_u8(avg_u8(l[2], l[1]), l[2]);
      }
      p1 += 15; p2 += 15; p3 += 15; d1 += 10; d2 += 10;
    }
    for (int j = 0; j < leftover; j += 3) {
      const unsigned A1 = p1[0], A2 = p1[1], A3 = p1[2];
      const unsigned B1 = p2[0], B2 = p2[1], B3 = p2[2];
      const unsigned C1 = p3[0], C2 = p3[1], C3 = p3[2];
      p1 += 3; p2 += 3; p3 += 3;
      *d1++ = (uint8_t)(((4 * A1 + 2 * (A2 + B1 + 1) + B2 + 1) / 9) & 0xFF);
      *d1++ = (uint8_t)(((4 * A3 + 2 * (A2 + B3 + 1) + B2 + 1) / 9) & 0xFF);
      *d2++ = (uint8_t)(((4 * C1 + 2 * (C2 + B1 + 1) + B2 + 1) / 9) & 0xFF);
      *d2++ = (uint8_t)(((4 * C3 + 2 * (C2 + B3 + 1) + B2 + 1) / 9) & 0xFF);
    }
    row += 3;
    row_dest += 2;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Threshold map: brisk/src/brisk-layer.cc:278-598, pass order kept literally                  */
/* ------------------------------------------------------------------------------------------ */

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }


static inline void thr_pass1_px(const uint8_t* img, int w, int x, int y, uint8_t* tmax, uint8_t* tmin) {
  const uint8_t* r0 = img + (long)(y - 1) * w + x - 1;
  const uint8_t* r1 = r0 + w;
  const uint8_t* r2 = r1 + w;
  int mx = r0[0], mn = r0[0];
#define BO_MM(v) do { int v_ = (v); if (v_ > mx) mx = v_; if (v_ < mn) mn = v_; } while (0)
  BO_MM(r0[1]); BO_MM(r0[2]); BO_MM(r1[0]); BO_MM(r1[1]); BO_MM(r1[2]); BO_MM(r2[0]); BO_MM(r2[1]); BO_MM(r2[2]);
  tmax[(long)y * w + x] = (uint8_t)mx;
  tmin[(long)y * w + x] = (uint8_t)mn;
}

static inline void thr_pass2_px(const uint8_t* img, int w, int x, int y, const uint8_t* tmax,
                                const uint8_t* tmin, uint8_t* thr) {
  const long o = (long)y * w + x;
  const long w2 = 2L * w;
  int mx = img[o], mn = img[o];
  BO_MM(img[o - 2 - w2]); BO_MM(img[o + 2 - w2]); BO_MM(img[o + 2 + w2]); BO_MM(img[o - 2 + w2]);
  mx = imax(imax(mx, tmax[o - w2]), imax(tmax[o + w2], imax(tmax[o - 2], tmax[o + 2])));
  mn = imin(imin(mn, tmin[o - w2]), imin(tmin[o + w2], imin(tmin[o - 2], tmin[o + 2])));
  thr[o] = (uint8_t)(mx - mn);
}

void bo_threshold_map(const uint8_t* img, int w, int h, uint8_t* thrmap) {
  uint8_t* tmax = (uint8_t*)calloc((size_t)w * h + 64, 1);
  uint8_t* tmin = (uint8_t*)calloc((size_t)w * h + 64, 1);
  memset(thrmap, 0, (size_t)w * h);
  /* SIMD pass 1 (:286-379): 16-wide blocks while x + 16 < w - 1 */
  for (int y = 1; y < h - 1; y++) {
    int x = 1;
    while (x + 16 < w - 1) {
      for (int k = 0; k < 16; ++k) thr_pass1_px(img, w, x + k, y, tmax, tmin);
      x += 16;
    }
  }
  /* SIMD pass 2 (:381-494): may read tmp columns still zero; overwritten by the scalar tail */
  for (int y = 3; y < h - 3; y++) {
    int x = 3;
    while (x + 16 < w - 3) {
      for (int k = 0; k < 16; ++k) thr_pass2_px(img, w, x + k, y, tmax, tmin, thrmap);
      x += 16;
    }
  }
  /* scalar tail of pass 1 (:496-541) */
  for (int x = imax(1, 16 * ((w - 2) / 16) - 16); x < w - 1; x++)
    for (int y = 1; y < h - 1; y++) thr_pass1_px(img, w, x, y, tmax, tmin);
  /* scalar tail of pass 2 (:543-597) */
  for (int x = imax(3, 16 * ((w - 6) / 16) - 16); x < w - 3; x++)
    for (int y = 3; y < h - 3; y++) thr_pass2_px(img, w, x, y, tmax, tmin, thrmap);
  free(tmax);
  free(tmin);
}

/* ------------------------------------------------------------------------------------------ */
/* AGAST: segment tests + bisection                                                            */
/* ------------------------------------------------------------------------------------------ */

/* Ring order agast/include/agast/oast9-16.h:99-116 (contiguous around the radius-3 circle). */
static const int kRing16[16][2] = {{-3, 0}, {-3, -1}, {-2, -2}, {-1, -3}, {0, -3}, {1, -3},
                                   {2, -2}, {3, -1},  {3, 0},   {3, 1},   {2, 2},  {1, 3},
                                   {0, 3},  {-1, 3},  {-2, 2},  {-3, 1}};
/* agast/include/agast/agast5-8.h:66-75 */
static const int kRing8[8][2] = {{-1, 0}, {-1, -1}, {0, -1}, {1, -1}, {1, 0}, {1, 1}, {0, 1}, {-1, 1}};

/* The machine-generated decision trees (oast9-16.cc:100-1843, oast9-16-nms.cc:64-1962,
 * agast5-8-nms.cc:59-336) are the plain segment test: corner at b iff some run of
 * `arc` contiguous ring pixels is entirely > c + b or entirely < c - b. */
static int is_corner(const uint8_t* p, int stride, int b, const int (*ring)[2], int n, int arc) {
  const int cb = *p + b, c_b = *p - b;
  unsigned bright = 0, dark = 0;
  for (int i = 0; i < n; ++i) {
    int v = p[ring[i][0] + ring[i][1] * stride];
    if (v > cb) bright |= 1u << i;
    if (v < c_b) dark |= 1u << i;
  }
  for (int pass = 0; pass < 2; ++pass) {
    unsigned m = pass ? dark : bright;
    m |= m << n; /* unroll the circle */
    unsigned run = m;
    for (int k = 1; k < arc; ++k) run &= m >> k;
    if (run & ((1u << n) - 1)) return 1;
  }
  return 0;
}

static int corner_score(const uint8_t* p, int stride, int b, const int (*ring)[2], int n, int arc) {
  /* oast9-16-nms.cc:39-42,1964-1975 / agast5-8-nms.cc:39-42,338-356 */
  int bmin = b, bmax = 255, b_test = (bmax + bmin) / 2;
  while (1) {
    if (is_corner(p, stride, b_test, ring, n, arc)) bmin = b_test;
    else bmax = b_test;
    if (bmin == bmax - 1 || bmin == bmax) return bmin;
    b_test = (bmin + bmax) / 2;
  }
}

int bo_oast9_16_corner_score(const uint8_t* p, int stride, int b) {
  return corner_score(p, stride, b, kRing16, 16, 9);
}
int bo_agast5_8_corner_score(const uint8_t* p, int stride, int b) {
  return corner_score(p, stride, b, kRing8, 8, 5);
}

/* oast9-16.cc:43-100,1844-1856 with AstDetector::set_threshold (ast-detector.h:62-68). */
int bo_oast9_16_detect(const uint8_t* img, int w, int h, const uint8_t* thrmap, int b, int upper,
                       int lower, int* xy, int cap) {
  const int cmp = (b * lower) / 100;
  const int xsizeB = w - 4, ysizeB = h - 3;
  int total = 0;
  for (int y = 3; y < ysizeB; y++) {
    for (int x = 3; x <= xsizeB; x++) {
      int b2;
      if (thrmap) {
        int t = thrmap[x + (long)y * w];
        if (t < cmp) continue;
        if (t < lower) t = lower;
        if (t > upper) t = upper;
        b2 = (t * b) / 100;
      } else {
        b2 = b;
      }
      {
        /* early out, same result: 9 contiguous ring pixels contain two adjacent compass points
         * (the generated tree likewise rejects most pixels after 2-3 comparisons) */
        const uint8_t* p = img + (long)y * w + x;
        const int cb = *p + b2, c_b = *p - b2;
        const int pw = p[-3], pe = p[3], pn = p[-3 * w], ps = p[3 * w];
        const int bw = pw > cb, be = pe > cb, bn = pn > cb, bs = ps > cb;
        const int dw = pw < c_b, de = pe < c_b, dn = pn < c_b, ds = ps < c_b;
        if (!((bw & bn) | (bn & be) | (be & bs) | (bs & bw) | (dw & dn) | (dn & de) | (de & ds) | (ds & dw))) continue;
      }
      if (!is_corner(img + (long)y * w + x, w, b2, kRing16, 16, 9)) continue;
      if (total < cap) { xy[2 * total] = x; xy[2 * total + 1] = y; }
      total++;
    }
  }
  return total;
}

/* ------------------------------------------------------------------------------------------ */
/* BriskLayer: brisk/src/brisk-layer.cc                                                        */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
  int w, h;
  uint8_t* img;
  uint8_t* scores; /* lazy score cache (scores_) */
  uint8_t* thrmap;
  float scale, offset;
  int upper, lower;
  int* pts; /* agast points (x,y) */
  int npts;
} bo_layer;

struct bo_scale_space {
  int layers;
  int threshold;
  bo_layer* l;
};

static void layer_finish(bo_layer* L, int upper, int lower) {
  L->scores = (uint8_t*)calloc((size_t)L->w * L->h + 64, 1);
  L->thrmap = (uint8_t*)malloc((size_t)L->w * L->h + 64);
  L->upper = upper;
  L->lower = lower;
  L->pts = NULL;
  L->npts = 0;
  bo_threshold_map(L->img, L->w, L->h, L->thrmap);
}

/* brisk-layer.cc:99-117 */
static void layer_get_agast_points(bo_layer* L, int threshold) {
  int cap = 4096;
  L->pts = (int*)malloc(sizeof(int) * 2 * cap);
  int n = bo_oast9_16_detect(L->img, L->w, L->h, L->thrmap, threshold, L->upper, L->lower, L->pts, cap);
  if (n > cap) {
    cap = n;
    L->pts = (int*)realloc(L->pts, sizeof(int) * 2 * cap);
    bo_oast9_16_detect(L->img, L->w, L->h, L->thrmap, threshold, L->upper, L->lower, L->pts, cap);
  }
  L->npts = n;
  for (int i = 0; i < n; i++) {
    const int offs = L->pts[2 * i] + L->pts[2 * i + 1] * L->w;
    int thr = L->thrmap[offs];
    L->scores[offs] = (uint8_t)bo_oast9_16_corner_score(L->img + offs, L->w, thr);
  }
}

/* brisk-layer.cc:118-132 */
static uint8_t S(bo_layer* L, int x, int y, uint8_t threshold) {
  if (x < 3 || y < 3) return 0;
  if (x >= L->w - 3 || y >= L->h - 3) return 0;
  uint8_t* score = L->scores + x + (long)y * L->w;
  if (*score > 2) return *score;
  *score = (uint8_t)bo_oast9_16_corner_score(L->img + x + (long)y * L->w, L->w, (int)threshold - 1);
  if (*score < threshold) *score = 0;
  return *score;
}

/* brisk-layer.cc:134-145 */
static uint8_t S58(bo_layer* L, int x, int y, uint8_t threshold) {
  if (x < 2 || y < 2) return 0;
  if (x >= L->w - 2 || y >= L->h - 2) return 0;
  uint8_t score = (uint8_t)bo_agast5_8_corner_score(L->img + x + (long)y * L->w, L->w, (int)threshold - 1);
  if (score < threshold) score = 0;
  return score;
}

/* brisk-layer.cc:147-161 (scale <= 1 branch; the scale > 1 branch is unreachable from the path) */
static uint8_t Sf(bo_layer* L, float xf, float yf, uint8_t threshold) {
  const int x = (int)xf;
  const float rx1 = xf - (float)x;
  const float rx = 1.0f - rx1;
  const int y = (int)yf;
  const float ry1 = yf - (float)y;
  const float ry = 1.0f - ry1;
  const int s00 = S(L, x, y, threshold);
  const int s10 = S(L, x + 1, y, threshold);
  const int s01 = S(L, x, y + 1, threshold);
  const int s11 = S(L, x + 1, y + 1, threshold);
  return (uint8_t)(rx * ry * s00 + rx1 * ry * s10 + rx * ry1 * s01 + rx1 * ry1 * s11);
}

/* ------------------------------------------------------------------------------------------ */
/* BriskScaleSpace: brisk/src/brisk-scale-space.cc                                             */
/* ------------------------------------------------------------------------------------------ */

static const float kBasicSize = 12.0f; /* :45 */
static const int kMaxThreshold = 1;    /* :47 */
static const int kDropThreshold = 5;   /* :48 */
static const int kMinDrop = 15;        /* :49 */

/* :54-90 with BriskLayer ctors brisk-layer.cc:53-95 */
bo_scale_space* bo_scale_space_create(const uint8_t* img, int w, int h, int threshold, int octaves) {
  return bo_scale_space_create_ex(img, w, h, threshold, octaves, 10); /* kDefaultLowerThreshold :50-51 */
}

/* ConstructPyramid(image, threshold, overwrite_lower_thres) (:64-90) */
bo_scale_space* bo_scale_space_create_ex(const uint8_t* img, int w, int h, int threshold, int octaves, int lower_threshold) {
  bo_scale_space* s = (bo_scale_space*)calloc(1, sizeof(*s));
  s->layers = (octaves == 0) ? 1 : 2 * octaves;
  s->threshold = threshold;
  s->l = (bo_layer*)calloc((size_t)s->layers, sizeof(bo_layer));
  const int upper = 230, lower = lower_threshold; /* :50-51 */
  bo_layer* L0 = &s->l[0];
  L0->w = w; L0->h = h;
  L0->img = (uint8_t*)malloc((size_t)w * h + 64);
  memcpy(L0->img, img, (size_t)w * h);
  L0->scale = 1.0f; L0->offset = 0.0f;
  layer_finish(L0, upper, lower);
  for (int i = 1; i < s->layers; ++i) {
    bo_layer* L = &s->l[i];
    const bo_layer* P = (i == 1) ? &s->l[0] : &s->l[i - 2];
    if (i == 1) { /* TWOTHIRDSAMPLE brisk-layer.cc:82-87 */
      L->w = 2 * (P->w / 3); L->h = 2 * (P->h / 3);
      L->img = (uint8_t*)calloc((size_t)L->w * L->h + 64, 1);
      bo_twothirdsample8(P->img, P->w, P->h, L->img);
      L->scale = (float)(P->scale * 1.5);
    } else { /* HALFSAMPLE brisk-layer.cc:77-81 */
      L->w = P->w / 2; L->h = P->h / 2;
      L->img = (uint8_t*)calloc((size_t)L->w * L->h + 64, 1);
      bo_halfsample8(P->img, P->w, P->h, L->img);
      L->scale = P->scale * 2;
    }
    L->offset = (float)(0.5 * L->scale - 0.5);
    layer_finish(L, upper, lower);
  }
  return s;
}

void bo_scale_space_destroy(bo_scale_space* s) {
  if (!s) return;
  for (int i = 0; i < s->layers; ++i) {
    free(s->l[i].img); free(s->l[i].scores); free(s->l[i].thrmap); free(s->l[i].pts);
  }
  free(s->l);
  free(s);
}

int bo_scale_space_layers(const bo_scale_space* s) { return s->layers; }

const uint8_t* bo_scale_space_map(const bo_scale_space* s, int layer, int which, int* w, int* h) {
  const bo_layer* L = &s->l[layer];
  *w = L->w; *h = L->h;
  return which == 0 ? L->img : which == 1 ? L->scores : L->thrmap;
}

/* :1230-1364 */
static float subpixel2d(const int s_0_0, const int s_0_1, const int s_0_2, const int s_1_0,
                        const int s_1_1, const int s_1_2, const int s_2_0, const int s_2_1,
                        const int s_2_2, float* delta_x_, float* delta_y_) {
  float delta_x, delta_y;
  int tmp1 = s_0_0 + s_0_2 - 2 * s_1_1 + s_2_0 + s_2_2;
  int coeff1 = 3 * (tmp1 + s_0_1 - ((s_1_0 + s_1_2) << 1) + s_2_1);
  int coeff2 = 3 * (tmp1 - ((s_0_1 + s_2_1) << 1) + s_1_0 + s_1_2);
  int tmp2 = s_0_2 - s_2_0;
  int tmp3 = (s_0_0 + tmp2 - s_2_2);
  int tmp4 = tmp3 - 2 * tmp2;
  int coeff3 = -3 * (tmp3 + s_0_1 - s_2_1);
  int coeff4 = -3 * (tmp4 + s_1_0 - s_1_2);
  int coeff5 = (s_0_0 - s_0_2 - s_2_0 + s_2_2) * 4;
  int coeff6 = -(s_0_0 + s_0_2 - ((s_1_0 + s_0_1 + s_1_2 + s_2_1) << 1) - 5 * s_1_1 + s_2_0 + s_2_2) * 2;

  int H_det = 4 * coeff1 * coeff2 - coeff5 * coeff5;

  if (H_det == 0) {
    *delta_x_ = 0.0f; *delta_y_ = 0.0f;
    return (float)((float)coeff6 / 18.0);
  }
  if (!(H_det > 0 && coeff1 < 0)) {
    int tmp_max = coeff3 + coeff4 + coeff5;
    delta_x = 1.0f; delta_y = 1.0f;
    int tmp = -coeff3 + coeff4 - coeff5;
    if (tmp > tmp_max) { tmp_max = tmp; delta_x = -1.0f; delta_y = 1.0f; }
    tmp = coeff3 - coeff4 - coeff5;
    if (tmp > tmp_max) { tmp_max = tmp; delta_x = 1.0f; delta_y = -1.0f; }
    tmp = -coeff3 - coeff4 + coeff5;
    if (tmp > tmp_max) { tmp_max = tmp; delta_x = -1.0f; delta_y = -1.0f; }
    *delta_x_ = delta_x; *delta_y_ = delta_y;
    return (float)((float)(tmp_max + coeff1 + coeff2 + coeff6) / 18.0);
  }
  delta_x = (float)(2 * coeff2 * coeff3 - coeff4 * coeff5) / (float)(-H_det);
  delta_y = (float)(2 * coeff1 * coeff4 - coeff3 * coeff5) / (float)(-H_det);
  int tx = 0, tx_ = 0, ty = 0, ty_ = 0;
  if (delta_x > 1.0) tx = 1;
  else if (delta_x < -1.0) tx_ = 1;
  if (delta_y > 1.0) ty = 1;
  if (delta_y < -1.0) ty_ = 1;

  if (tx || tx_ || ty || ty_) {
    float delta_x1 = 0.0f, delta_x2 = 0.0f, delta_y1 = 0.0f, delta_y2 = 0.0f;
    if (tx) {
      delta_x1 = 1.0f;
      delta_y1 = -(float)(coeff4 + coeff5) / (float)(2 * coeff2);
      if (delta_y1 > 1.0) delta_y1 = 1.0f; else if (delta_y1 < -1.0) delta_y1 = -1.0f;
    } else if (tx_) {
      delta_x1 = -1.0f;
      delta_y1 = -(float)(coeff4 - coeff5) / (float)(2 * coeff2);
      if (delta_y1 > 1.0) delta_y1 = 1.0f; else if (delta_y1 < -1.0) delta_y1 = -1.0f;
    }
    if (ty) {
      delta_y2 = 1.0f;
      delta_x2 = -(float)(coeff3 + coeff5) / (float)(2 * coeff1);
      if (delta_x2 > 1.0) delta_x2 = 1.0f; else if (delta_x2 < -1.0) delta_x2 = -1.0f;
    } else if (ty_) {
      delta_y2 = -1.0f;
      delta_x2 = -(float)(coeff3 - coeff5) / (float)(2 * coeff1);
      if (delta_x2 > 1.0) delta_x2 = 1.0f; else if (delta_x2 < -1.0) delta_x2 = -1.0f;
    }
    float max1 = (float)((coeff1 * delta_x1 * delta_x1 + coeff2 * delta_y1 * delta_y1 + coeff3 * delta_x1 +
                          coeff4 * delta_y1 + coeff5 * delta_x1 * delta_y1 + coeff6) / 18.0);
    float max2 = (float)((coeff1 * delta_x2 * delta_x2 + coeff2 * delta_y2 * delta_y2 + coeff3 * delta_x2 +
                          coeff4 * delta_y2 + coeff5 * delta_x2 * delta_y2 + coeff6) / 18.0);
    if (max1 > max2) { /* :1349-1357: delta_y = delta_x1/2 is the reference's behaviour (kept) */
      *delta_x_ = delta_x1; *delta_y_ = delta_x1;
      return max1;
    } else {
      *delta_x_ = delta_x2; *delta_y_ = delta_x2;
      return max2;
    }
  }
  *delta_x_ = delta_x; *delta_y_ = delta_y;
  return (float)((coeff1 * delta_x * delta_x + coeff2 * delta_y * delta_y + coeff3 * delta_x +
                  coeff4 * delta_y + coeff5 * delta_x * delta_y + coeff6) / 18.0);
}

/* :1101-1143 */
static float refine1d(const float s_05, const float s0, const float s05, float* max) {
  int i_05 = (int)(1024.0 * s_05 + 0.5);
  int i0 = (int)(1024.0 * s0 + 0.5);
  int i05 = (int)(1024.0 * s05 + 0.5);
  int three_a = 16 * i_05 - 24 * i0 + 8 * i05;
  if (three_a >= 0) {
    if (s0 >= s_05 && s0 >= s05) { *max = s0; return 1.0f; }
    if (s_05 >= s0 && s_05 >= s05) { *max = s_05; return 0.75f; }
    if (s05 >= s0 && s05 >= s_05) { *max = s05; return 1.5f; }
  }
  int three_b = -40 * i_05 + 54 * i0 - 14 * i05;
  float ret_val = -(float)three_b / (float)(2 * three_a);
  if (ret_val < 0.75) ret_val = 0.75f;
  else if (ret_val > 1.5) ret_val = 1.5f;
  int three_c = +24 * i_05 - 27 * i0 + 6 * i05;
  float m = (float)three_c + (float)three_a * ret_val * ret_val + (float)three_b * ret_val;
  m = (float)(m / 3072.0);
  *max = m;
  return ret_val;
}

/* :1145-1186 */
static float refine1d_1(const float s_05, const float s0, const float s05, float* max) {
  int i_05 = (int)(1024.0 * s_05 + 0.5);
  int i0 = (int)(1024.0 * s0 + 0.5);
  int i05 = (int)(1024.0 * s05 + 0.5);
  int two_a = 9 * i_05 - 18 * i0 + 9 * i05;
  if (two_a >= 0) {
    if (s0 >= s_05 && s0 >= s05) { *max = s0; return 1.0f; }
    if (s_05 >= s0 && s_05 >= s05) { *max = s_05; return (float)0.6666666666666666666666666667; }
    if (s05 >= s0 && s05 >= s_05) { *max = s05; return (float)1.3333333333333333333333333333; }
  }
  int two_b = -21 * i_05 + 36 * i0 - 15 * i05;
  float ret_val = -(float)two_b / (float)(2 * two_a);
  if (ret_val < 0.6666666666666666666666666667) ret_val = (float)0.666666666666666666666666667;
  else if (ret_val > 1.33333333333333333333333333) ret_val = (float)1.333333333333333333333333333;
  int two_c = +12 * i_05 - 16 * i0 + 6 * i05;
  float m = (float)two_c + (float)two_a * ret_val * ret_val + (float)two_b * ret_val;
  m = (float)(m / 2048.0);
  *max = m;
  return ret_val;
}

/* :1188-1228 */
static float refine1d_2(const float s_05, const float s0, const float s05, float* max) {
  int i_05 = (int)(1024.0 * s_05 + 0.5);
  int i0 = (int)(1024.0 * s0 + 0.5);
  int i05 = (int)(1024.0 * s05 + 0.5);
  int a = 2 * i_05 - 4 * i0 + 2 * i05;
  if (a >= 0) {
    if (s0 >= s_05 && s0 >= s05) { *max = s0; return 1.0f; }
    if (s_05 >= s0 && s_05 >= s05) { *max = s_05; return (float)0.7; }
    if (s05 >= s0 && s05 >= s_05) { *max = s05; return 1.5f; }
  }
  int b = -5 * i_05 + 8 * i0 - 3 * i05;
  float ret_val = -(float)b / (float)(2 * a);
  if (ret_val < 0.7) ret_val = (float)0.7;
  else if (ret_val > 1.5) ret_val = 1.5f;
  int c = +3 * i_05 - 3 * i0 + 1 * i05;
  float m = (float)c + (float)a * ret_val * ret_val + (float)b * ret_val;
  m = m / 1024;
  *max = m;
  return ret_val;
}

/* :430-531 */
static int g_oob; /* set when is_max_2d would read outside the score matrix (only possible with the at(0) indexing, :137) */
static int is_max_2d(bo_scale_space* s, int layer, const int x_layer, const int y_layer) {
  bo_layer* l = &s->l[layer];
  const int scorescols = l->w;
  const long total = (long)l->w * l->h;
  {
    const long lo = (long)(y_layer - 2) * scorescols + x_layer - 2, hi = (long)(y_layer + 2) * scorescols + x_layer + 2;
    if (lo < 0 || hi >= total) { /* a read of this call could leave the matrix: check each one exactly */
      const long c = (long)y_layer * scorescols + x_layer;
      if (c < 0 || c >= total) { g_oob = 1; return 0; }
    }
  }
  const uint8_t* data = l->scores + (long)y_layer * scorescols + x_layer;
  const uint8_t center = *data;
  const uint8_t s_10 = S(l, x_layer - 1, y_layer, center);
  if (center < s_10) return 0;
  const uint8_t s10 = S(l, x_layer + 1, y_layer, center);
  if (center < s10) return 0;
  const uint8_t s0_1 = S(l, x_layer, y_layer - 1, center);
  if (center < s0_1) return 0;
  const uint8_t s01 = S(l, x_layer, y_layer + 1, center);
  if (center < s01) return 0;
  const uint8_t s_11 = S(l, x_layer - 1, y_layer + 1, center);
  if (center < s_11) return 0;
  const uint8_t s11 = S(l, x_layer + 1, y_layer + 1, center);
  if (center < s11) return 0;
  const uint8_t s1_1 = S(l, x_layer + 1, y_layer - 1, center);
  if (center < s1_1) return 0;
  const uint8_t s_1_1 = S(l, x_layer - 1, y_layer - 1, center);
  if (center < s_1_1) return 0;

  int delta[16];
  int nd = 0;
  if (center == s_1_1) { delta[nd++] = -1; delta[nd++] = -1; }
  if (center == s0_1) { delta[nd++] = 0; delta[nd++] = -1; }
  if (center == s1_1) { delta[nd++] = 1; delta[nd++] = -1; }
  if (center == s_10) { delta[nd++] = -1; delta[nd++] = 0; }
  if (center == s10) { delta[nd++] = 1; delta[nd++] = 0; }
  if (center == s_11) { delta[nd++] = -1; delta[nd++] = 1; }
  if (center == s01) { delta[nd++] = 0; delta[nd++] = 1; }
  if (center == s11) { delta[nd++] = 1; delta[nd++] = 1; }
  if (nd != 0) {
    int smoothedcenter = 4 * center + 2 * (s_10 + s10 + s0_1 + s01) + s_1_1 + s1_1 + s_11 + s11;
    for (int i = 0; i < nd; i += 2) {
      data = l->scores + (long)(y_layer - 1 + delta[i + 1]) * scorescols + x_layer + delta[i] - 1;
      if (data < l->scores || data + 2 * scorescols + 2 >= l->scores + total) { g_oob = 1; return 0; }
      int othercenter = *data;
      data++; othercenter += 2 * (*data);
      data++; othercenter += *data;
      data += scorescols; othercenter += 2 * (*data);
      data--; othercenter += 4 * (*data);
      data--; othercenter += 2 * (*data);
      data += scorescols; othercenter += *data;
      data++; othercenter += 2 * (*data);
      data++; othercenter += *data;
      if (othercenter > smoothedcenter) return 0;
    }
  }
  return 1;
}

/* :757-915 */
static float get_score_max_above(bo_scale_space* s, const int layer, const int x_layer,
                                 const int y_layer, const int thr, int* ismax, float* dx_, float* dy_) {
  int threshold = thr + kDropThreshold;
  *ismax = 0;
  float x_1, x1, y_1, y1;
  bo_layer* la = &s->l[layer + 1];
  if (layer % 2 == 0) {
    x_1 = (float)((float)(4 * (x_layer)-1 - 2) / 6.0);
    x1 = (float)((float)(4 * (x_layer)-1 + 2) / 6.0);
    y_1 = (float)((float)(4 * (y_layer)-1 - 2) / 6.0);
    y1 = (float)((float)(4 * (y_layer)-1 + 2) / 6.0);
  } else {
    x_1 = (float)(6 * (x_layer)-1 - 3) / 8.0f;
    x1 = (float)(6 * (x_layer)-1 + 3) / 8.0f;
    y_1 = (float)(6 * (y_layer)-1 - 3) / 8.0f;
    y1 = (float)(6 * (y_layer)-1 + 3) / 8.0f;
  }
  int max_x = (int)(x_1 + 1);
  int max_y = (int)(y_1 + 1);
  float tmp_max;
  float max = Sf(la, x_1, y_1, 1);
  if (max > threshold) return 0;
  for (int x = (int)(x_1 + 1); x <= (int)x1; x++) {
    tmp_max = Sf(la, (float)x, y_1, 1);
    if (tmp_max > threshold) return 0;
    if (tmp_max > max) { max = tmp_max; max_x = x; }
  }
  tmp_max = Sf(la, x1, y_1, 1);
  if (tmp_max > threshold) return 0;
  if (tmp_max > max) { max = tmp_max; max_x = (int)x1; }

  for (int y = (int)(y_1 + 1); y <= (int)y1; y++) {
    tmp_max = Sf(la, x_1, (float)y, 1);
    if (tmp_max > threshold) return 0;
    if (tmp_max > max) { max = tmp_max; max_x = (int)(x_1 + 1); max_y = y; }
    for (int x = (int)(x_1 + 1); x <= (int)x1; x++) {
      tmp_max = S(la, x, y, 1);
      if (tmp_max > threshold) return 0;
      if (tmp_max > max) { max = tmp_max; max_x = x; max_y = y; }
    }
    tmp_max = Sf(la, x1, (float)y, 1);
    if (tmp_max > threshold) return 0;
    if (tmp_max > max) { max = tmp_max; max_x = (int)x1; max_y = y; }
  }

  tmp_max = Sf(la, x_1, y1, 1);
  if (tmp_max > max) { max = tmp_max; max_x = (int)(x_1 + 1); max_y = (int)y1; }
  for (int x = (int)(x_1 + 1); x <= (int)x1; x++) {
    tmp_max = Sf(la, (float)x, y1, 1);
    if (tmp_max > max) { max = tmp_max; max_x = x; max_y = (int)y1; }
  }
  tmp_max = Sf(la, x1, y1, 1);
  if (tmp_max > max) { max = tmp_max; max_x = (int)x1; max_y = (int)y1; }

  int s_0_0 = S(la, max_x - 1, max_y - 1, 1);
  int s_1_0 = S(la, max_x, max_y - 1, 1);
  int s_2_0 = S(la, max_x + 1, max_y - 1, 1);
  int s_2_1 = S(la, max_x + 1, max_y, 1);
  int s_1_1 = S(la, max_x, max_y, 1);
  int s_0_1 = S(la, max_x - 1, max_y, 1);
  int s_0_2 = S(la, max_x - 1, max_y + 1, 1);
  int s_1_2 = S(la, max_x, max_y + 1, 1);
  int s_2_2 = S(la, max_x + 1, max_y + 1, 1);
  float dx_1, dy_1;
  float refined_max = subpixel2d(s_0_0, s_0_1, s_0_2, s_1_0, s_1_1, s_1_2, s_2_0, s_2_1, s_2_2, &dx_1, &dy_1);

  float real_x = (float)max_x + dx_1;
  float real_y = (float)max_y + dy_1;
  int returnrefined = 1;
  float dx, dy;
  if (layer % 2 == 0) {
    dx = (real_x * 6.0f + 1.0f) / 4.0f - (float)x_layer;
    dy = (real_y * 6.0f + 1.0f) / 4.0f - (float)y_layer;
  } else {
    dx = (float)((real_x * 8.0 + 1.0) / 6.0 - (float)x_layer);
    dy = (float)((real_y * 8.0 + 1.0) / 6.0 - (float)y_layer);
  }
  if (dx > 1.0f) { dx = 1.0f; returnrefined = 0; }
  if (dx < -1.0f) { dx = -1.0f; returnrefined = 0; }
  if (dy > 1.0f) { dy = 1.0f; returnrefined = 0; }
  if (dy < -1.0f) { dy = -1.0f; returnrefined = 0; }
  *dx_ = dx; *dy_ = dy;
  *ismax = 1;
  if (returnrefined) return refined_max > max ? refined_max : max; /* std::max(refined_max, max) */
  return max;
}

/* :917-1099 */
static float get_score_max_below(bo_scale_space* s, const int layer, const int x_layer,
                                 const int y_layer, const int thr, int* ismax, float* dx_, float* dy_) {
  int threshold = thr + kDropThreshold;
  *ismax = 0;
  float x_1, x1, y_1, y1;
  if (layer % 2 == 0) {
    x_1 = (float)((float)(8 * (x_layer) + 1 - 4) / 6.0);
    x1 = (float)((float)(8 * (x_layer) + 1 + 4) / 6.0);
    y_1 = (float)((float)(8 * (y_layer) + 1 - 4) / 6.0);
    y1 = (float)((float)(8 * (y_layer) + 1 + 4) / 6.0);
  } else {
    x_1 = (float)((float)(6 * (x_layer) + 1 - 3) / 4.0);
    x1 = (float)((float)(6 * (x_layer) + 1 + 3) / 4.0);
    y_1 = (float)((float)(6 * (y_layer) + 1 - 3) / 4.0);
    y1 = (float)((float)(6 * (y_layer) + 1 + 3) / 4.0);
  }
  bo_layer* lb = &s->l[layer - 1];

  int max_x = (int)(x_1 + 1);
  int max_y = (int)(y_1 + 1);
  float tmp_max;
  float max = Sf(lb, x_1, y_1, 1);
  if (max > threshold) return 0;
  for (int x = (int)(x_1 + 1); x <= (int)x1; x++) {
    tmp_max = Sf(lb, (float)x, y_1, 1);
    if (tmp_max > threshold) return 0;
    if (tmp_max > max) { max = tmp_max; max_x = x; }
  }
  tmp_max = Sf(lb, x1, y_1, 1);
  if (tmp_max > threshold) return 0;
  if (tmp_max > max) { max = tmp_max; max_x = (int)x1; }

  for (int y = (int)(y_1 + 1); y <= (int)y1; y++) {
    tmp_max = Sf(lb, x_1, (float)y, 1);
    if (tmp_max > threshold) return 0;
    if (tmp_max > max) { max = tmp_max; max_x = (int)(x_1 + 1); max_y = y; }
    for (int x = (int)(x_1 + 1); x <= (int)x1; x++) {
      tmp_max = S(lb, x, y, 1);
      if (tmp_max > threshold) return 0;
      if (tmp_max == max) {
        const int t1 = 2 * (S(lb, x - 1, y, 1) + S(lb, x + 1, y, 1) + S(lb, x, y + 1, 1) + S(lb, x, y - 1, 1)) +
                       (S(lb, x + 1, y + 1, 1) + S(lb, x - 1, y + 1, 1) + S(lb, x + 1, y - 1, 1) + S(lb, x - 1, y - 1, 1));
        const int t2 = 2 * (S(lb, max_x - 1, max_y, 1) + S(lb, max_x + 1, max_y, 1) + S(lb, max_x, max_y + 1, 1) +
                            S(lb, max_x, max_y - 1, 1)) +
                       (S(lb, max_x + 1, max_y + 1, 1) + S(lb, max_x - 1, max_y + 1, 1) +
                        S(lb, max_x + 1, max_y - 1, 1) + S(lb, max_x - 1, max_y - 1, 1));
        if (t1 > t2) { max_x = x; max_y = y; }
      }
      if (tmp_max > max) { max = tmp_max; max_x = x; max_y = y; }
    }
    tmp_max = Sf(lb, x1, (float)y, 1);
    if (tmp_max > threshold) return 0;
    if (tmp_max > max) { max = tmp_max; max_x = (int)x1; max_y = y; }
  }

  tmp_max = Sf(lb, x_1, y1, 1);
  if (tmp_max > max) { max = tmp_max; max_x = (int)(x_1 + 1); max_y = (int)y1; }
  for (int x = (int)(x_1 + 1); x <= (int)x1; x++) {
    tmp_max = Sf(lb, (float)x, y1, 1);
    if (tmp_max > max) { max = tmp_max; max_x = x; max_y = (int)y1; }
  }
  tmp_max = Sf(lb, x1, y1, 1);
  if (tmp_max > max) { max = tmp_max; max_x = (int)x1; max_y = (int)y1; }

  int s_0_0 = S(lb, max_x - 1, max_y - 1, 1);
  int s_1_0 = S(lb, max_x, max_y - 1, 1);
  int s_2_0 = S(lb, max_x + 1, max_y - 1, 1);
  int s_2_1 = S(lb, max_x + 1, max_y, 1);
  int s_1_1 = S(lb, max_x, max_y, 1);
  int s_0_1 = S(lb, max_x - 1, max_y, 1);
  int s_0_2 = S(lb, max_x - 1, max_y + 1, 1);
  int s_1_2 = S(lb, max_x, max_y + 1, 1);
  int s_2_2 = S(lb, max_x + 1, max_y + 1, 1);
  float dx_1, dy_1;
  float refined_max = subpixel2d(s_0_0, s_0_1, s_0_2, s_1_0, s_1_1, s_1_2, s_2_0, s_2_1, s_2_2, &dx_1, &dy_1);

  float real_x = (float)max_x + dx_1;
  float real_y = (float)max_y + dy_1;
  int returnrefined = 1;
  float dx, dy;
  if (layer % 2 == 0) {
    dx = (float)((real_x * 6.0 + 1.0) / 8.0 - (float)x_layer);
    dy = (float)((real_y * 6.0 + 1.0) / 8.0 - (float)y_layer);
  } else {
    dx = (float)((real_x * 4.0 - 1.0) / 6.0 - (float)x_layer);
    dy = (float)((real_y * 4.0 - 1.0) / 6.0 - (float)y_layer);
  }
  if (dx > 1.0) { dx = 1.0f; returnrefined = 0; }
  if (dx < -1.0) { dx = -1.0f; returnrefined = 0; }
  if (dy > 1.0) { dy = 1.0f; returnrefined = 0; }
  if (dy < -1.0) { dy = -1.0f; returnrefined = 0; }
  *dx_ = dx; *dy_ = dy;
  *ismax = 1;
  if (returnrefined) return refined_max > max ? refined_max : max;
  return max;
}

static inline float fmax_std(float a, float b) { return (a < b) ? b : a; } /* std::max(a,b) */

/* :534-754 */
static float refine3d(bo_scale_space* s, const int layer, const int x_layer, const int y_layer,
                      float* x_, float* y_, float* scale_, int* ismax) {
  *ismax = 1;
  bo_layer* tl = &s->l[layer];
  const int center = S(tl, x_layer, y_layer, 1);
  float x = 0, y = 0, scale = 0;

  float delta_x_above = 0, delta_y_above = 0;
  float max_above = get_score_max_above(s, layer, x_layer, y_layer, center, ismax, &delta_x_above, &delta_y_above);
  if (!*ismax) return 0.0f;

  float max = 0; /* to be returned */
  int doScaleRefinement = 1;

  if (layer % 2 == 0) { /* on octave */
    float delta_x_below, delta_y_below;
    float max_below_float;
    if (layer == 0) {
      unsigned char max_below_uchar = 0;
      bo_layer* l = &s->l[0];
      int s_0_0 = S58(l, x_layer - 1, y_layer - 1, 1);
      max_below_uchar = (unsigned char)s_0_0;
      int s_1_0 = S58(l, x_layer, y_layer - 1, 1);
      if (s_1_0 > max_below_uchar) max_below_uchar = (unsigned char)s_1_0;
      int s_2_0 = S58(l, x_layer + 1, y_layer - 1, 1);
      if (s_2_0 > max_below_uchar) max_below_uchar = (unsigned char)s_2_0;
      int s_2_1 = S58(l, x_layer + 1, y_layer, 1);
      if (s_2_1 > max_below_uchar) max_below_uchar = (unsigned char)s_2_1;
      int s_1_1 = S58(l, x_layer, y_layer, 1);
      if (s_1_1 > max_below_uchar) max_below_uchar = (unsigned char)s_1_1;
      int s_0_1 = S58(l, x_layer - 1, y_layer, 1);
      if (s_0_1 > max_below_uchar) max_below_uchar = (unsigned char)s_0_1;
      int s_0_2 = S58(l, x_layer - 1, y_layer + 1, 1);
      if (s_0_2 > max_below_uchar) max_below_uchar = (unsigned char)s_0_2;
      int s_1_2 = S58(l, x_layer, y_layer + 1, 1);
      if (s_1_2 > max_below_uchar) max_below_uchar = (unsigned char)s_1_2;
      int s_2_2 = S58(l, x_layer + 1, y_layer + 1, 1);
      if (s_2_2 > max_below_uchar) max_below_uchar = (unsigned char)s_2_2;
      max_below_float = subpixel2d(s_0_0, s_0_1, s_0_2, s_1_0, s_1_1, s_1_2, s_2_0, s_2_1, s_2_2,
                                   &delta_x_below, &delta_y_below);
      max_below_float = max_below_uchar;
    } else {
      max_below_float = get_score_max_below(s, layer, x_layer, y_layer, center, ismax, &delta_x_below, &delta_y_below);
      if (!*ismax) return 0;
    }

    int s_0_0 = S(tl, x_layer - 1, y_layer - 1, 1);
    int s_1_0 = S(tl, x_layer, y_layer - 1, 1);
    int s_2_0 = S(tl, x_layer + 1, y_layer - 1, 1);
    int s_2_1 = S(tl, x_layer + 1, y_layer, 1);
    int s_1_1 = S(tl, x_layer, y_layer, 1);
    int s_0_1 = S(tl, x_layer - 1, y_layer, 1);
    int s_0_2 = S(tl, x_layer - 1, y_layer + 1, 1);
    int s_1_2 = S(tl, x_layer, y_layer + 1, 1);
    int s_2_2 = S(tl, x_layer + 1, y_layer + 1, 1);

    if (layer == 0) {
      if (s_1_1 - kMaxThreshold <= (int)max_above) doScaleRefinement = 0;
    } else {
      if ((s_1_1 - kMaxThreshold < (max_above)) || (s_1_1 - kMaxThreshold < (max_below_float))) {
        if ((s_1_1 - kMinDrop > (max_above)) || (s_1_1 - kMinDrop > (max_below_float))) {
          doScaleRefinement = 0;
        } else {
          *ismax = 0;
          return 0.0f;
        }
      }
    }

    float delta_x_layer, delta_y_layer;
    float max_layer = subpixel2d(s_0_0, s_0_1, s_0_2, s_1_0, s_1_1, s_1_2, s_2_0, s_2_1, s_2_2,
                                 &delta_x_layer, &delta_y_layer);

    if (doScaleRefinement) {
      if (layer == 0) scale = refine1d_2(max_below_float, fmax_std((float)center, max_layer), max_above, &max);
      else scale = refine1d(max_below_float, fmax_std((float)center, max_layer), max_above, &max);
    } else {
      scale = 1.0f;
      max = max_layer;
    }

    if (scale > 1.0) {
      const float r0 = (float)((1.5 - scale) / .5);
      const float r1 = (float)(1.0 - r0);
      x = (r0 * delta_x_layer + r1 * delta_x_above + (float)x_layer) * tl->scale + tl->offset;
      y = (r0 * delta_y_layer + r1 * delta_y_above + (float)y_layer) * tl->scale + tl->offset;
    } else {
      if (layer == 0) {
        const float r0 = (float)((scale - 0.5) / 0.5);
        const float r_1 = (float)(1.0 - r0);
        x = r0 * delta_x_layer + r_1 * delta_x_below + (float)x_layer;
        y = r0 * delta_y_layer + r_1 * delta_y_below + (float)y_layer;
      } else {
        const float r0 = (float)((scale - 0.75) / 0.25);
        const float r_1 = (float)(1.0 - r0);
        x = (r0 * delta_x_layer + r_1 * delta_x_below + (float)x_layer) * tl->scale + tl->offset;
        y = (r0 * delta_y_layer + r_1 * delta_y_below + (float)y_layer) * tl->scale + tl->offset;
      }
    }
  } else { /* on intra */
    float delta_x_below, delta_y_below;
    float max_below = get_score_max_below(s, layer, x_layer, y_layer, center, ismax, &delta_x_below, &delta_y_below);
    if (!*ismax) return 0.0f;

    int s_0_0 = S(tl, x_layer - 1, y_layer - 1, 1);
    int s_1_0 = S(tl, x_layer, y_layer - 1, 1);
    int s_2_0 = S(tl, x_layer + 1, y_layer - 1, 1);
    int s_2_1 = S(tl, x_layer + 1, y_layer, 1);
    int s_1_1 = S(tl, x_layer, y_layer, 1);
    int s_0_1 = S(tl, x_layer - 1, y_layer, 1);
    int s_0_2 = S(tl, x_layer - 1, y_layer + 1, 1);
    int s_1_2 = S(tl, x_layer, y_layer + 1, 1);
    int s_2_2 = S(tl, x_layer + 1, y_layer + 1, 1);

    if ((s_1_1 - kMaxThreshold < (max_above)) || (s_1_1 - kMaxThreshold < (max_below))) {
      if ((s_1_1 - kMinDrop > (max_above)) || (s_1_1 - kMinDrop > (max_below))) {
        doScaleRefinement = 0;
      } else {
        *ismax = 0;
        return 0.0f;
      }
    }

    float delta_x_layer, delta_y_layer;
    float max_layer = subpixel2d(s_0_0, s_0_1, s_0_2, s_1_0, s_1_1, s_1_2, s_2_0, s_2_1, s_2_2,
                                 &delta_x_layer, &delta_y_layer);
    if (doScaleRefinement) {
      scale = refine1d_1(max_below, fmax_std((float)center, max_layer), max_above, &max);
    } else {
      scale = 1.0f;
      max = max_layer;
    }
    if (scale > 1.0) {
      const float r0 = (float)(4.0 - scale * 3.0);
      const float r1 = (float)(1.0 - r0);
      x = (r0 * delta_x_layer + r1 * delta_x_above + (float)x_layer) * tl->scale + tl->offset;
      y = (r0 * delta_y_layer + r1 * delta_y_above + (float)y_layer) * tl->scale + tl->offset;
    } else {
      const float r0 = (float)(scale * 3.0 - 2.0);
      const float r_1 = (float)(1.0 - r0);
      x = (r0 * delta_x_layer + r_1 * delta_x_below + (float)x_layer) * tl->scale + tl->offset;
      y = (r0 * delta_y_layer + r_1 * delta_y_below + (float)y_layer) * tl->scale + tl->offset;
    }
  }
  scale *= tl->scale;
  *x_ = x; *y_ = y; *scale_ = scale;
  return max;
}

/* :92-287 (detection path: empty input keypoints, suppressScaleNonmaxima = true) */
int bo_scale_space_get_keypoints(bo_scale_space* s, bo_keypoint** out) {
  return bo_scale_space_get_keypoints_ex(s, 1, out);
}

/* GetKeypoints with suppressScaleNonmaxima_ given (:92-287).  suppress == 0 with more than one layer is the branch
 * :131-170 with its `agastPoints.at(0)[n]` indexing (:137); returns -1 where the reference has no defined result (at()
 * throws because layer i has more points than layer 0, or IsMax2D reads outside a score matrix).  PARITY UNPINNED for
 * that branch: no test or golden vector of the reference exercises it. */
int bo_scale_space_get_keypoints_ex(bo_scale_space* s, int suppress, bo_keypoint** out) {
  for (int i = 0; i < s->layers; ++i) layer_get_agast_points(&s->l[i], s->threshold);
  int cap = 1024, n = 0;
  bo_keypoint* kps = (bo_keypoint*)malloc(sizeof(bo_keypoint) * cap);
#define PUSH(KP) do { if (n == cap) { cap *= 2; kps = (bo_keypoint*)realloc(kps, sizeof(bo_keypoint) * cap); } kps[n++] = (KP); } while (0)

  if (!suppress && s->layers > 1) { /* :131-170 */
    g_oob = 0;
    for (int i = 0; i < s->layers; i++) {
      bo_layer* l = &s->l[i];
      const int num = l->npts;
      for (int k = 0; k < num; k++) {
        if (k >= s->l[0].npts) { free(kps); *out = NULL; return -1; } /* agastPoints.at(0)[n]: std::out_of_range */
        const float point_x = (float)s->l[0].pts[2 * k], point_y = (float)s->l[0].pts[2 * k + 1];
        const int is_max = is_max_2d(s, i, (int)point_x, (int)point_y);
        if (g_oob) { free(kps); *out = NULL; return -1; }
        if (!is_max) continue;
        int s_0_0 = Sf(l, point_x - 1, point_y - 1, 1);
        int s_1_0 = Sf(l, point_x, point_y - 1, 1);
        int s_2_0 = Sf(l, point_x + 1, point_y - 1, 1);
        int s_2_1 = Sf(l, point_x + 1, point_y, 1);
        int s_1_1 = Sf(l, point_x, point_y, 1);
        int s_0_1 = Sf(l, point_x - 1, point_y, 1);
        int s_0_2 = Sf(l, point_x - 1, point_y + 1, 1);
        int s_1_2 = Sf(l, point_x, point_y + 1, 1);
        int s_2_2 = Sf(l, point_x + 1, point_y + 1, 1);
        float delta_x, delta_y;
        float max = subpixel2d(s_0_0, s_0_1, s_0_2, s_1_0, s_1_1, s_1_2, s_2_0, s_2_1, s_2_2, &delta_x, &delta_y);
        bo_keypoint kp;
        kp.x = point_x + delta_x; kp.y = point_y + delta_y;
        kp.size = kBasicSize * l->scale; kp.angle = -1; kp.response = max; kp.octave = 0; kp.class_id = -1;
        PUSH(kp);
      }
    }
    *out = kps;
    return n;
  }

  if (s->layers == 1) { /* :172-209 */
    bo_layer* l = &s->l[0];
    for (int k = 0; k < l->npts; k++) {
      const float point_x = (float)l->pts[2 * k], point_y = (float)l->pts[2 * k + 1];
      if (!is_max_2d(s, 0, (int)point_x, (int)point_y)) continue;
      int s_0_0 = Sf(l, point_x - 1, point_y - 1, 1);
      int s_1_0 = Sf(l, point_x, point_y - 1, 1);
      int s_2_0 = Sf(l, point_x + 1, point_y - 1, 1);
      int s_2_1 = Sf(l, point_x + 1, point_y, 1);
      int s_1_1 = Sf(l, point_x, point_y, 1);
      int s_0_1 = Sf(l, point_x - 1, point_y, 1);
      int s_0_2 = Sf(l, point_x - 1, point_y + 1, 1);
      int s_1_2 = Sf(l, point_x, point_y + 1, 1);
      int s_2_2 = Sf(l, point_x + 1, point_y + 1, 1);
      float delta_x, delta_y;
      float max = subpixel2d(s_0_0, s_0_1, s_0_2, s_1_0, s_1_1, s_1_2, s_2_0, s_2_1, s_2_2, &delta_x, &delta_y);
      bo_keypoint kp;
      kp.x = point_x + delta_x; kp.y = point_y + delta_y;
      kp.size = kBasicSize; kp.angle = -1; kp.response = max; kp.octave = 0; kp.class_id = -1;
      PUSH(kp);
    }
    *out = kps;
    return n;
  }

  for (int i = 0; i < s->layers; i++) {
    bo_layer* l = &s->l[i];
    const int num = l->npts;
    if (i == s->layers - 1) { /* :215-256 */
      for (int k = 0; k < num; k++) {
        const float point_x = (float)l->pts[2 * k], point_y = (float)l->pts[2 * k + 1];
        if (!is_max_2d(s, i, (int)point_x, (int)point_y)) continue;
        int ismax;
        float dx, dy;
        get_score_max_below(s, i, (int)point_x, (int)point_y, Sf(l, point_x, point_y, 1), &ismax, &dx, &dy);
        if (!ismax) continue;
        int s_0_0 = Sf(l, point_x - 1, point_y - 1, 1);
        int s_1_0 = Sf(l, point_x, point_y - 1, 1);
        int s_2_0 = Sf(l, point_x + 1, point_y - 1, 1);
        int s_2_1 = Sf(l, point_x + 1, point_y, 1);
        int s_1_1 = Sf(l, point_x, point_y, 1);
        int s_0_1 = Sf(l, point_x - 1, point_y, 1);
        int s_0_2 = Sf(l, point_x - 1, point_y + 1, 1);
        int s_1_2 = Sf(l, point_x, point_y + 1, 1);
        int s_2_2 = Sf(l, point_x + 1, point_y + 1, 1);
        float delta_x, delta_y;
        float max = subpixel2d(s_0_0, s_0_1, s_0_2, s_1_0, s_1_1, s_1_2, s_2_0, s_2_1, s_2_2, &delta_x, &delta_y);
        bo_keypoint kp;
        kp.x = (point_x + delta_x) * l->scale + l->offset;
        kp.y = (point_y + delta_y) * l->scale + l->offset;
        kp.size = kBasicSize * l->scale; kp.angle = -1; kp.response = max; kp.octave = i; kp.class_id = -1;
        PUSH(kp);
      }
    } else { /* :257-285 */
      for (int k = 0; k < num; k++) {
        const float point_x = (float)l->pts[2 * k], point_y = (float)l->pts[2 * k + 1];
        if (!is_max_2d(s, i, (int)point_x, (int)point_y)) continue;
        int ismax;
        float x, y, scale;
        float score = refine3d(s, i, (int)point_x, (int)point_y, &x, &y, &scale, &ismax);
        if (!ismax) continue;
        bo_keypoint kp;
        kp.x = x; kp.y = y; kp.size = kBasicSize * scale; kp.angle = -1; kp.response = score;
        kp.octave = i; kp.class_id = -1;
        PUSH(kp);
      }
    }
  }
#undef PUSH
  *out = kps;
  return n;
}

/* brisk-feature-detector.cc:49-66,77-85 */
int bo_detect(const uint8_t* img, int w, int h, int threshold, int octaves, const uint8_t* mask,
              bo_keypoint** out) {
  return bo_detect_ex(img, w, h, threshold, octaves, 1, mask, out);
}

int bo_detect_ex(const uint8_t* img, int w, int h, int threshold, int octaves, int suppress_scale_nonmaxima,
                 const uint8_t* mask, bo_keypoint** out) {
  bo_scale_space* s = bo_scale_space_create(img, w, h, threshold, octaves);
  int n = bo_scale_space_get_keypoints_ex(s, suppress_scale_nonmaxima, out);
  bo_scale_space_destroy(s);
  if (n < 0) return n;
  if (mask) {
    bo_keypoint* k = *out;
    int m = 0;
    for (int i = 0; i < n; ++i) {
      if (mask[(long)(int)(k[i].y + 0.5f) * w + (int)(k[i].x + 0.5f)] == 0) continue;
      k[m++] = k[i];
    }
    n = m;
  }
  return n;
}

/* ------------------------------------------------------------------------------------------ */
/* BriskFeatureDetector::ComputeScale (brisk-feature-detector.cc:87-92): ConstructPyramid(image, threshold, 0) +   */
/* GetKeypoints with a non-empty keypoint list (brisk-scale-space.cc:104-123 and the branches that follow with      */
/* perform_2d_nonMax == false).  PARITY UNPINNED: nothing in the reference calls or tests this entry.                */
/* Returns -1 where the reference has no defined result: GetAgastPoints addresses the maps with                      */
/* int(x_float + y_float * cols) (brisk-layer.cc:110-115) and cornerScore then reads the ring around that pixel by   */
/* linear offsets - past the image for points of a layer's last rows; or agastPoints.at(0)[n] throws (:137).         */
/* ------------------------------------------------------------------------------------------ */
typedef struct { bo_keypoint* k; int n, cap; } kp_list;
static void kp_push(kp_list* L, bo_keypoint kp) {
  if (L->n == L->cap) { L->cap = L->cap ? 2 * L->cap : 64; L->k = (bo_keypoint*)realloc(L->k, sizeof(bo_keypoint) * L->cap); }
  L->k[L->n++] = kp;
}

int bo_compute_scale(const uint8_t* img, int w, int h, int threshold, int octaves, int suppress_scale_nonmaxima,
                     const bo_keypoint* in, int n_in, bo_keypoint** out) {
  bo_scale_space* s = bo_scale_space_create_ex(img, w, h, threshold, octaves, 0);
  if (n_in == 0) { /* empty list: plain detection on the lower-threshold-0 pyramid (no mask filter in ComputeScale) */
    int n = bo_scale_space_get_keypoints_ex(s, suppress_scale_nonmaxima, out);
    bo_scale_space_destroy(s);
    return n;
  }
  kp_list* lists = (kp_list*)calloc((size_t)s->layers, sizeof(kp_list));
  kp_list res = {0, 0, 0};
  int undefined = 0;
  for (int i = 0; i < s->layers && !undefined; ++i) { /* :99-126 */
    bo_layer* l = &s->l[i];
    for (int k = 0; k < n_in; ++k) {
      bo_keypoint kp = in[k];
      kp.x = ((float)in[k].x) / l->scale - l->offset;
      kp.y = ((float)in[k].y) / l->scale - l->offset;
      if (kp.x < 3 || kp.y < 3 || kp.x > l->w - 3 || kp.y > l->h - 3) continue;
      Sf(l, kp.x, kp.y, 0); /* "calculates and stores the score of this keypoint in the score map" */
      kp_push(&lists[i], kp);
    }
    if (lists[i].n == 0) { /* GetAgastPoints on an empty list detects (brisk-layer.cc:103-105), lower threshold 0 */
      layer_get_agast_points(l, threshold);
      for (int k = 0; k < l->npts; ++k) {
        bo_keypoint kp;
        kp.x = (float)l->pts[2 * k]; kp.y = (float)l->pts[2 * k + 1];
        kp.size = 0; kp.angle = -1; kp.response = 0; kp.octave = 0; kp.class_id = -1; /* agast::KeyPoint h; (oast9-16.cc:48) */
        kp_push(&lists[i], kp);
      }
    } else { /* brisk-layer.cc:106-116 with float coordinates */
      const long total = (long)l->w * l->h;
      for (int k = 0; k < lists[i].n; ++k) {
        const int offs = lists[i].k[k].x + lists[i].k[k].y * l->w; /* float + float * int -> float -> int */
        if (offs - 3 * l->w - 1 < 0 || offs + 3 * l->w + 1 >= total) { undefined = 1; break; } /* ring leaves the image */
        const int thr = l->thrmap[offs];
        l->scores[offs] = (uint8_t)bo_oast9_16_corner_score(l->img + offs, l->w, thr);
      }
    }
  }
#define EMIT(KP) kp_push(&res, (KP))
  if (!undefined && !suppress_scale_nonmaxima && s->layers > 1) { /* :131-170, perform_2d_nonMax == false */
    for (int i = 0; i < s->layers && !undefined; i++) {
      bo_layer* l = &s->l[i];
      for (int k = 0; k < lists[i].n; k++) {
        if (k >= lists[0].n) { undefined = 1; break; }
        const bo_keypoint keypoint = lists[0].k[k];
        const float point_x = keypoint.x, point_y = keypoint.y;
        int s_0_0 = Sf(l, point_x - 1, point_y - 1, 1);
        int s_1_0 = Sf(l, point_x, point_y - 1, 1);
        int s_2_0 = Sf(l, point_x + 1, point_y - 1, 1);
        int s_2_1 = Sf(l, point_x + 1, point_y, 1);
        int s_1_1 = Sf(l, point_x, point_y, 1);
        int s_0_1 = Sf(l, point_x - 1, point_y, 1);
        int s_0_2 = Sf(l, point_x - 1, point_y + 1, 1);
        int s_1_2 = Sf(l, point_x, point_y + 1, 1);
        int s_2_2 = Sf(l, point_x + 1, point_y + 1, 1);
        float delta_x, delta_y;
        float max = subpixel2d(s_0_0, s_0_1, s_0_2, s_1_0, s_1_1, s_1_2, s_2_0, s_2_1, s_2_2, &delta_x, &delta_y);
        bo_keypoint kp = keypoint;
        kp.x = point_x + delta_x; kp.y = point_y + delta_y;
        kp.size = kBasicSize * l->scale; kp.angle = -1; kp.response = max; kp.octave = 0;
        EMIT(kp);
      }
    }
  } else if (!undefined && s->layers == 1) { /* :172-209 */
    bo_layer* l = &s->l[0];
    for (int k = 0; k < lists[0].n; k++) {
      const bo_keypoint keypoint = lists[0].k[k];
      const float point_x = keypoint.x, point_y = keypoint.y;
      int s_0_0 = Sf(l, point_x - 1, point_y - 1, 1);
      int s_1_0 = Sf(l, point_x, point_y - 1, 1);
      int s_2_0 = Sf(l, point_x + 1, point_y - 1, 1);
      int s_2_1 = Sf(l, point_x + 1, point_y, 1);
      int s_1_1 = Sf(l, point_x, point_y, 1);
      int s_0_1 = Sf(l, point_x - 1, point_y, 1);
      int s_0_2 = Sf(l, point_x - 1, point_y + 1, 1);
      int s_1_2 = Sf(l, point_x, point_y + 1, 1);
      int s_2_2 = Sf(l, point_x + 1, point_y + 1, 1);
      float delta_x, delta_y;
      float max = subpixel2d(s_0_0, s_0_1, s_0_2, s_1_0, s_1_1, s_1_2, s_2_0, s_2_1, s_2_2, &delta_x, &delta_y);
      bo_keypoint kp = keypoint;
      kp.x = point_x + delta_x; kp.y = point_y + delta_y;
      kp.size = kBasicSize; kp.angle = -1; kp.response = max; kp.octave = 0;
      EMIT(kp);
    }
  } else if (!undefined) { /* :211-287 */
    for (int i = 0; i < s->layers; i++) {
      bo_layer* l = &s->l[i];
      for (int k = 0; k < lists[i].n; k++) {
        const bo_keypoint keypoint = lists[i].k[k];
        const float point_x = keypoint.x, point_y = keypoint.y;
        if (i == s->layers - 1) {
          int ismax;
          float dx, dy;
          get_score_max_below(s, i, (int)point_x, (int)point_y, Sf(l, point_x, point_y, 1), &ismax, &dx, &dy);
          if (!ismax) continue;
          int s_0_0 = Sf(l, point_x - 1, point_y - 1, 1);
          int s_1_0 = Sf(l, point_x, point_y - 1, 1);
          int s_2_0 = Sf(l, point_x + 1, point_y - 1, 1);
          int s_2_1 = Sf(l, point_x + 1, point_y, 1);
          int s_1_1 = Sf(l, point_x, point_y, 1);
          int s_0_1 = Sf(l, point_x - 1, point_y, 1);
          int s_0_2 = Sf(l, point_x - 1, point_y + 1, 1);
          int s_1_2 = Sf(l, point_x, point_y + 1, 1);
          int s_2_2 = Sf(l, point_x + 1, point_y + 1, 1);
          float delta_x, delta_y;
          float max = subpixel2d(s_0_0, s_0_1, s_0_2, s_1_0, s_1_1, s_1_2, s_2_0, s_2_1, s_2_2, &delta_x, &delta_y);
          bo_keypoint kp = keypoint;
          kp.x = (point_x + delta_x) * l->scale + l->offset;
          kp.y = (point_y + delta_y) * l->scale + l->offset;
          kp.size = kBasicSize * l->scale; kp.angle = -1; kp.response = max; kp.octave = i;
          EMIT(kp);
        } else {
          int ismax;
          float x, y, scale;
          float score = refine3d(s, i, (int)point_x, (int)point_y, &x, &y, &scale, &ismax);
          if (!ismax) continue;
          bo_keypoint kp = keypoint;
          kp.x = x; kp.y = y; kp.size = kBasicSize * scale; kp.angle = -1; kp.response = score; kp.octave = i;
          EMIT(kp);
        }
      }
    }
  }
#undef EMIT
  for (int i = 0; i < s->layers; ++i) free(lists[i].k);
  free(lists);
  bo_scale_space_destroy(s);
  if (undefined) { free(res.k); *out = NULL; return -1; }
  if (!res.k) res.k = (bo_keypoint*)malloc(sizeof(bo_keypoint));
  *out = res.k;
  return res.n;
}

void bo_free(void* p) { free(p); }
