/*
 * brisk_oracle_describe.c - CPU restatement of BriskDescriptorExtractor (oracle; TEST ONLY).
 * See brisk_oracle.h.  Follows brisk/src/brisk-descriptor-extractor.cc and
 * brisk/include/brisk/internal/integral-image.h.  libm calls are the double versions (the
 * reference calls unqualified log/pow/cos/sin/sqrt/atan2/ceil with only <cmath>-style headers in
 * scope, i.e. ::f(double)).
 */
#include "brisk_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "default_pattern.inc"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define BO_SCALES 64u     /* scales_      :58 */
#define BO_NROT 1024u     /* n_rot_       :62 */
static const float kScaleRange = 30.0f; /* scalerange_ :60 */
static const float kBasicSizeD = 12.0f; /* basicSize_  :57 */

typedef struct { float x, y, sigma; } bo_pattern_point;               /* helper-structures.h:46-50 */
typedef struct { unsigned i, j; } bo_short_pair;                       /* :51-54 */
typedef struct { unsigned i, j; int weighted_dx, weighted_dy; } bo_long_pair; /* :55-60 */

struct bo_extractor {
  int rotation_invariance, scale_invariance;
  unsigned points;
  bo_pattern_point* pattern; /* [scale][rot][point] */
  float* scale_list;
  unsigned* size_list;
  bo_short_pair* short_pairs;
  bo_long_pair* long_pairs;
  unsigned n_short, n_long;
  int strings;
};

/* integral-image.h:56-161: exclusive 2-D prefix sum in int32 (the 2-row/4-column blocking of the
 * reference does not change any value; plain int arithmetic, same wrap-around). */
void bo_integral_image8(const uint8_t* img, int w, int h, int32_t* out) {
  const int sw = w + 1;
  memset(out, 0, sizeof(int32_t) * (size_t)sw);
  for (int y = 0; y < h; ++y) {
    int32_t* sum = out + (size_t)(y + 1) * sw;
    const int32_t* prev = sum - sw;
    int32_t s = 0;
    sum[0] = 0;
    for (int x = 0; x < w; ++x) {
      s += img[(size_t)y * w + x];
      sum[x + 1] = (int32_t)((uint32_t)prev[x + 1] + (uint32_t)s);
    }
  }
}

/* :180-291 (InitFromStream); tokens come either from the built-in table or a .ptn text. */
static int init_from_tables(bo_extractor* e, unsigned npoints, const float* pts /* [n][3] */,
                            unsigned n_short, const unsigned* sp, unsigned n_long, const unsigned* lp,
                            float patternScale) {
  e->points = npoints;
  e->pattern = (bo_pattern_point*)malloc(sizeof(bo_pattern_point) * (size_t)npoints * BO_SCALES * BO_NROT);
  bo_pattern_point* it = e->pattern;
  const float lb_scale = (float)(log(kScaleRange) / log(2.0));
  const float lb_scale_step = lb_scale / (BO_SCALES);
  e->scale_list = (float*)malloc(sizeof(float) * BO_SCALES);
  e->size_list = (unsigned*)malloc(sizeof(unsigned) * BO_SCALES);
  const float sigma_scale = 1.3f;
  float* u_x = (float*)malloc(sizeof(float) * npoints);
  float* u_y = (float*)malloc(sizeof(float) * npoints);
  float* sigma = (float*)malloc(sizeof(float) * npoints);
  for (unsigned i = 0; i < npoints; i++) {
    u_x[i] = pts[3 * i + 0]; u_x[i] *= patternScale;
    u_y[i] = pts[3 * i + 1]; u_y[i] *= patternScale;
    sigma[i] = pts[3 * i + 2]; sigma[i] *= patternScale;
  }
  for (unsigned scale = 0; scale < BO_SCALES; ++scale) {
    e->scale_list[scale] = (float)pow(2.0, (double)(scale * lb_scale_step));
    e->size_list[scale] = 0;
    double theta;
    for (size_t rot = 0; rot < BO_NROT; ++rot) {
      for (unsigned i = 0; i < npoints; i++) {
        theta = (double)rot * 2 * M_PI / (double)BO_NROT;
        it->x = (float)(e->scale_list[scale] * (u_x[i] * cos(theta) - u_y[i] * sin(theta)));
        it->y = (float)(e->scale_list[scale] * (u_x[i] * sin(theta) + u_y[i] * cos(theta)));
        it->sigma = sigma_scale * e->scale_list[scale] * sigma[i];
        const unsigned size = (unsigned)(ceil(((sqrt((double)(it->x * it->x + it->y * it->y))) + it->sigma)) + 1);
        if (e->size_list[scale] < size) e->size_list[scale] = size;
        ++it;
      }
    }
  }
  e->n_short = n_short;
  e->short_pairs = (bo_short_pair*)malloc(sizeof(bo_short_pair) * n_short);
  for (unsigned p = 0; p < n_short; p++) { e->short_pairs[p].i = sp[2 * p]; e->short_pairs[p].j = sp[2 * p + 1]; }
  e->n_long = n_long;
  e->long_pairs = (bo_long_pair*)malloc(sizeof(bo_long_pair) * n_long);
  for (unsigned p = 0; p < n_long; p++) {
    unsigned i = lp[2 * p], j = lp[2 * p + 1];
    e->long_pairs[p].i = i; e->long_pairs[p].j = j;
    float dx = (u_x[j] - u_x[i]);
    float dy = (u_y[j] - u_y[i]);
    float norm_sq = dx * dx + dy * dy;
    e->long_pairs[p].weighted_dx = (int)((dx / (norm_sq)) * 2048.0 + 0.5);
    e->long_pairs[p].weighted_dy = (int)((dy / (norm_sq)) * 2048.0 + 0.5);
  }
  e->strings = (int)ceil(((float)n_short) / 128.0) * 4 * 4;
  free(u_x); free(u_y); free(sigma);
  return (n_short == 384) ? 0 : -1; /* CHECK_EQ(noShortPairs_, kDescriptorLength) :286 */
}

/* :65-178 (generateKernel), called as in :316-339 */
static void generate_kernel(bo_extractor* e, const float* radiusList, const int* numberList, int rings,
                            float dMax, float dMin) {
  unsigned points = 0;
  for (int ring = 0; ring < rings; ring++) points += (unsigned)numberList[ring];
  e->points = points;
  e->pattern = (bo_pattern_point*)malloc(sizeof(bo_pattern_point) * (size_t)points * BO_SCALES * BO_NROT);
  bo_pattern_point* it = e->pattern;
  const float lb_scale = (float)(log(kScaleRange) / log(2.0));
  const float lb_scale_step = lb_scale / (BO_SCALES);
  e->scale_list = (float*)malloc(sizeof(float) * BO_SCALES);
  e->size_list = (unsigned*)malloc(sizeof(unsigned) * BO_SCALES);
  const float sigma_scale = 1.3f;
  for (unsigned scale = 0; scale < BO_SCALES; ++scale) {
    e->scale_list[scale] = (float)pow((double)2.0, (double)(scale * lb_scale_step));
    e->size_list[scale] = 0;
    double alpha, theta;
    for (size_t rot = 0; rot < BO_NROT; ++rot) {
      theta = (double)rot * 2 * M_PI / (double)BO_NROT;
      for (int ring = 0; ring < rings; ++ring) {
        for (int num = 0; num < numberList[ring]; ++num) {
          alpha = ((double)num) * 2 * M_PI / (double)numberList[ring];
          it->x = (float)(e->scale_list[scale] * radiusList[ring] * cos(alpha + theta));
          it->y = (float)(e->scale_list[scale] * radiusList[ring] * sin(alpha + theta));
          if (ring == 0) {
            it->sigma = (float)(sigma_scale * e->scale_list[scale] * 0.5);
          } else {
            it->sigma = (float)(sigma_scale * e->scale_list[scale] * ((double)radiusList[ring]) *
                                sin(M_PI / numberList[ring]));
          }
          const unsigned size = (unsigned)(ceil(((e->scale_list[scale] * radiusList[ring]) + it->sigma)) + 1);
          if (e->size_list[scale] < size) e->size_list[scale] = size;
          ++it;
        }
      }
    }
  }
  const unsigned npairs = points * (points - 1) / 2;
  e->short_pairs = (bo_short_pair*)malloc(sizeof(bo_short_pair) * npairs);
  e->long_pairs = (bo_long_pair*)malloc(sizeof(bo_long_pair) * npairs);
  e->n_short = 0;
  e->n_long = 0;
  const float dMin_sq = dMin * dMin;
  const float dMax_sq = dMax * dMax;
  for (unsigned i = 1; i < points; i++) {
    for (unsigned j = 0; j < i; j++) {
      const float dx = e->pattern[j].x - e->pattern[i].x;
      const float dy = e->pattern[j].y - e->pattern[i].y;
      const float norm_sq = (dx * dx + dy * dy);
      if (norm_sq > dMin_sq) {
        bo_long_pair* lp = &e->long_pairs[e->n_long];
        lp->weighted_dx = (int)((dx / (norm_sq)) * 2048.0 + 0.5);
        lp->weighted_dy = (int)((dy / (norm_sq)) * 2048.0 + 0.5);
        lp->i = i; lp->j = j;
        ++e->n_long;
      }
      if (norm_sq < dMax_sq) {
        bo_short_pair* sp = &e->short_pairs[e->n_short]; /* indexChange = identity */
        sp->j = j; sp->i = i;
        ++e->n_short;
      }
    }
  }
  e->strings = (int)ceil(((float)e->n_short) / 128.0) * 4 * 4;
}

bo_extractor* bo_extractor_create(int rotation_invariant, int scale_invariant, int version,
                                  float pattern_scale, const char* pattern_text) {
  bo_extractor* e = (bo_extractor*)calloc(1, sizeof(*e));
  e->rotation_invariance = rotation_invariant;
  e->scale_invariance = scale_invariant;
  if (version == 2) {
    if (pattern_text == NULL) {
      unsigned sp[2 * BRISK_DEFAULT_NSHORT], lp[2 * BRISK_DEFAULT_NLONG];
      for (int i = 0; i < BRISK_DEFAULT_NSHORT; ++i) { sp[2 * i] = brisk_default_short_pairs[i][0]; sp[2 * i + 1] = brisk_default_short_pairs[i][1]; }
      for (int i = 0; i < BRISK_DEFAULT_NLONG; ++i) { lp[2 * i] = brisk_default_long_pairs[i][0]; lp[2 * i + 1] = brisk_default_long_pairs[i][1]; }
      init_from_tables(e, BRISK_DEFAULT_NPOINTS, &brisk_default_points[0][0], BRISK_DEFAULT_NSHORT, sp,
                       BRISK_DEFAULT_NLONG, lp, pattern_scale);
    } else { /* .ptn text: N, N x {x y sigma}, S, S x {i j}, L, L x {i j} */
      char* txt = strdup(pattern_text);
      char* save = NULL;
      char* tok = strtok_r(txt, " \t\r\n", &save);
#define NEXT() (tok ? (cur = tok, tok = strtok_r(NULL, " \t\r\n", &save), cur) : "0")
      const char* cur;
      unsigned n = (unsigned)strtoul(NEXT(), NULL, 10);
      float* pts = (float*)malloc(sizeof(float) * 3 * n);
      for (unsigned i = 0; i < 3 * n; ++i) pts[i] = strtof(NEXT(), NULL);
      unsigned ns = (unsigned)strtoul(NEXT(), NULL, 10);
      unsigned* sp = (unsigned*)malloc(sizeof(unsigned) * 2 * ns);
      for (unsigned i = 0; i < 2 * ns; ++i) sp[i] = (unsigned)strtoul(NEXT(), NULL, 10);
      unsigned nl = (unsigned)strtoul(NEXT(), NULL, 10);
      unsigned* lp = (unsigned*)malloc(sizeof(unsigned) * 2 * nl);
      for (unsigned i = 0; i < 2 * nl; ++i) lp[i] = (unsigned)strtoul(NEXT(), NULL, 10);
#undef NEXT
      init_from_tables(e, n, pts, ns, sp, nl, lp, pattern_scale);
      free(pts); free(sp); free(lp); free(txt);
    }
  } else if (version == 1) { /* :316-339 */
    float rList[5];
    int nList[5] = {1, 10, 14, 15, 20};
    const double f = 0.85 * pattern_scale;
    rList[0] = (float)(f * 0);
    rList[1] = (float)(f * 2.9);
    rList[2] = (float)(f * 4.9);
    rList[3] = (float)(f * 7.4);
    rList[4] = (float)(f * 10.8);
    generate_kernel(e, rList, nList, 5, 5.85f, 8.2f);
  } else {
    free(e);
    return NULL;
  }
  return e;
}

void bo_extractor_destroy(bo_extractor* e) {
  if (!e) return;
  free(e->pattern); free(e->scale_list); free(e->size_list); free(e->short_pairs); free(e->long_pairs);
  free(e);
}

int bo_extractor_descriptor_size(const bo_extractor* e) { return e->strings; }
int bo_extractor_points(const bo_extractor* e) { return (int)e->points; }
const float* bo_extractor_scale_list(const bo_extractor* e) { return e->scale_list; }
const unsigned* bo_extractor_size_list(const bo_extractor* e) { return e->size_list; }
const float* bo_extractor_pattern(const bo_extractor* e) { return (const float*)e->pattern; }

/* :370-530, ImgPixel_T = unsigned char, IntegralPixel_T = int */
static int smoothed_intensity(const bo_extractor* e, const uint8_t* image, int imagecols,
                              const int32_t* integral, const float key_x, const float key_y,
                              const unsigned scale, const unsigned rot, const unsigned point) {
  const bo_pattern_point* bp = &e->pattern[(size_t)scale * BO_NROT * e->points + (size_t)rot * e->points + point];
  const float xf = bp->x + key_x;
  const float yf = bp->y + key_y;
  const int x = (int)xf;
  const int y = (int)yf;
  const float sigma_half = bp->sigma;
  const float area = (float)(4.0 * sigma_half * sigma_half);
  int ret_val;
  if (sigma_half < 0.5) {
    const int r_x = (int)((xf - x) * 1024);
    const int r_y = (int)((yf - y) * 1024);
    const int r_x_1 = (1024 - r_x);
    const int r_y_1 = (1024 - r_y);
    const uint8_t* ptr = image + x + (long)y * imagecols;
    ret_val = (r_x_1 * r_y_1 * (int)(*ptr));
    ptr++;
    ret_val += (r_x * r_y_1 * (int)(*ptr));
    ptr += imagecols;
    ret_val += (r_x * r_y * (int)(*ptr));
    ptr--;
    ret_val += (r_x_1 * r_y * (int)(*ptr));
    return (ret_val) / 1024;
  }
  const int scaling = (int)(4194304.0 / area);
  const int scaling2 = (int)((float)scaling * area / 1024.0);
  const int integralcols = imagecols + 1;
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
  const int A = (int)((r_x_1 * r_y_1) * scaling);
  const int B = (int)((r_x1 * r_y_1) * scaling);
  const int C = (int)((r_x1 * r_y1) * scaling);
  const int D = (int)((r_x_1 * r_y1) * scaling);
  const int r_x_1_i = (int)(r_x_1 * scaling);
  const int r_y_1_i = (int)(r_y_1 * scaling);
  const int r_x1_i = (int)(r_x1 * scaling);
  const int r_y1_i = (int)(r_y1 * scaling);

  if (dx + dy > 2) {
    const uint8_t* ptr = image + x_left + (long)imagecols * y_top;
    ret_val = A * (int)(*ptr);
    ptr += dx + 1;
    ret_val += B * (int)(*ptr);
    ptr += (long)dy * imagecols + 1;
    ret_val += C * (int)(*ptr);
    ptr -= dx + 1;
    ret_val += D * (int)(*ptr);

    const int32_t* pi = integral + x_left + (long)integralcols * y_top + 1;
    const int tmp1 = (*pi);
    pi += dx;
    const int tmp2 = (*pi);
    pi += integralcols;
    const int tmp3 = (*pi);
    pi++;
    const int tmp4 = (*pi);
    pi += (long)dy * integralcols;
    const int tmp5 = (*pi);
    pi--;
    const int tmp6 = (*pi);
    pi += integralcols;
    const int tmp7 = (*pi);
    pi -= dx;
    const int tmp8 = (*pi);
    pi -= integralcols;
    const int tmp9 = (*pi);
    pi--;
    const int tmp10 = (*pi);
    pi -= (long)dy * integralcols;
    const int tmp11 = (*pi);
    pi++;
    const int tmp12 = (*pi);

    /* unsigned arithmetic == the reference's wrapping int arithmetic */
    const int upper = (int)((uint32_t)(tmp3 - tmp2 + tmp1 - tmp12) * (uint32_t)r_y_1_i);
    const int middle = (int)((uint32_t)(tmp6 - tmp3 + tmp12 - tmp9) * (uint32_t)scaling);
    const int left = (int)((uint32_t)(tmp9 - tmp12 + tmp11 - tmp10) * (uint32_t)r_x_1_i);
    const int right = (int)((uint32_t)(tmp5 - tmp4 + tmp3 - tmp6) * (uint32_t)r_x1_i);
    const int bottom = (int)((uint32_t)(tmp7 - tmp6 + tmp9 - tmp8) * (uint32_t)r_y1_i);
    return (int)((ret_val + upper + middle + left + right + bottom) / scaling2);
  }

  const uint8_t* ptr = image + x_left + (long)imagecols * y_top;
  ret_val = A * (int)(*ptr);
  ptr++;
  const uint8_t* end1 = ptr + dx;
  for (; ptr < end1; ptr++) ret_val += r_y_1_i * (int)(*ptr);
  ret_val += B * (int)(*ptr);
  ptr += imagecols - dx - 1;
  const uint8_t* end_j = ptr + (long)dy * imagecols;
  for (; ptr < end_j; ptr += imagecols - dx - 1) {
    ret_val += r_x_1_i * (int)(*ptr);
    ptr++;
    const uint8_t* end2 = ptr + dx;
    for (; ptr < end2; ptr++) ret_val += (int)(*ptr) * scaling;
    ret_val += r_x1_i * (int)(*ptr);
  }
  ret_val += D * (int)(*ptr);
  ptr++;
  const uint8_t* end3 = ptr + dx;
  for (; ptr < end3; ptr++) ret_val += r_y1_i * (int)(*ptr);
  ret_val += C * (int)(*ptr);
  return (int)((ret_val) / scaling2);
}

/* :618-650: scale index of a keypoint */
int bo_extractor_scale_index(const bo_extractor* e, float size) {
  static const float log2 = (float)0.693147180559945;
  const float lb_scalerange = (float)(log(kScaleRange) / (log2));
  const float basicSize06 = (float)(kBasicSizeD * 0.6);
  unsigned scale;
  if (e->scale_invariance) {
    int v = (int)(BO_SCALES / lb_scalerange * (log(size / (basicSize06)) / log2) + 0.5);
    scale = (unsigned)(v > 0 ? v : 0);
    if (scale >= BO_SCALES) scale = BO_SCALES - 1;
  } else {
    int v = (int)(BO_SCALES / lb_scalerange * (log(1.45 * kBasicSizeD / (basicSize06)) / log2) + 0.5);
    scale = (unsigned)(v > 0 ? v : 0);
  }
  return (int)scale;
}

/* :612-778 doDescriptorComputation (cv::Mat descriptor container, 8-bit image) */
int bo_extractor_compute(const bo_extractor* e, const uint8_t* img, int w, int h, bo_keypoint* kps,
                         int n, uint8_t* desc) {
  int* kscales = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  int m = 0;
  for (int k = 0; k < n; k++) {
    const int scale = bo_extractor_scale_index(e, kps[k].size);
    const int border = (int)e->size_list[scale];
    const int border_x = w - border;
    const int border_y = h - border;
    /* RoiPredicate :532-536 with float(minX..maxY) */
    const float minX = (float)border, minY = (float)border, maxX = (float)border_x, maxY = (float)border_y;
    if (!((kps[k].x < minX) || (kps[k].x >= maxX) || (kps[k].y < minY) || (kps[k].y >= maxY))) {
      kps[m] = kps[k];
      kscales[m] = scale;
      m++;
    }
  }
  const int ksize = m;
  memset(desc, 0, (size_t)ksize * e->strings);
  int32_t* integral = (int32_t*)malloc(sizeof(int32_t) * (size_t)(w + 1) * (h + 1));
  bo_integral_image8(img, w, h, integral);
  int* values = (int*)malloc(sizeof(int) * e->points);
  for (int k = 0; k < ksize; ++k) {
    int theta;
    bo_keypoint* kp = &kps[k];
    const int scale = kscales[k];
    const float x = kp->x, y = kp->y;
    if (kp->angle == -1) {
      if (!e->rotation_invariance) {
        theta = 0;
      } else {
        for (unsigned i = 0; i < e->points; i++)
          values[i] = smoothed_intensity(e, img, w, integral, x, y, (unsigned)scale, 0, i);
        int direction0 = 0, direction1 = 0;
        for (unsigned p = 0; p < e->n_long; ++p) {
          const bo_long_pair* it = &e->long_pairs[p];
          int t1 = values[it->i];
          int t2 = values[it->j];
          const int delta_t = (t1 - t2);
          const int tmp0 = delta_t * (it->weighted_dx) / 1024;
          const int tmp1 = delta_t * (it->weighted_dy) / 1024;
          direction0 += tmp0;
          direction1 += tmp1;
        }
        kp->angle = (float)(atan2((double)(float)direction1, (double)(float)direction0) / M_PI * 180.0);
        theta = (int)((BO_NROT * kp->angle) / (360.0) + 0.5);
        if (theta < 0) theta += BO_NROT;
        if (theta >= (int)BO_NROT) theta -= BO_NROT;
      }
    } else {
      if (!e->rotation_invariance) {
        theta = 0;
      } else {
        theta = (int)(BO_NROT * (kp->angle / (360.0)) + 0.5);
        if (theta < 0) theta += BO_NROT;
        if (theta >= (int)BO_NROT) theta -= BO_NROT;
      }
    }
    for (unsigned i = 0; i < e->points; i++)
      values[i] = smoothed_intensity(e, img, w, integral, x, y, (unsigned)scale, (unsigned)theta, i);
    /* setDescriptorBits :538-564 */
    uint32_t* ptr2 = (uint32_t*)(desc + (size_t)e->strings * k);
    int shifter = 0;
    for (unsigned p = 0; p < e->n_short; ++p) {
      int t1 = values[e->short_pairs[p].i];
      int t2 = values[e->short_pairs[p].j];
      if (t1 > t2) *ptr2 |= ((1u) << shifter);
      ++shifter;
      if (shifter == 32) { shifter = 0; ++ptr2; }
    }
  }
  free(values); free(integral); free(kscales);
  return ksize;
}
