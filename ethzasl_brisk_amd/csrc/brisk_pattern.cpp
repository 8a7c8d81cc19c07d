// brisk_pattern.cpp - see brisk_pattern.h.  Reference: brisk/src/brisk-descriptor-extractor.cc.
// NOTE: every libm call takes an explicit double argument.  The reference calls unqualified
// log/sqrt/... on floats with only ::f(double) in scope, i.e. the double versions; in this
// translation unit <math.h> also exposes the float overloads, which must not be picked.
#include "brisk_pattern.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <sstream>

#include "brisk_default_pattern.inc"

namespace {

const float kScaleRange = 30.0f;  // scalerange_ (:60)
const float kBasicSize = 12.0f;   // basicSize_  (:57)
const float kSigmaScale = 1.3f;   // sigma_scale (:91, :206)

void fill_scale_list(BriskPatternHost* P) {
  // :85-86/:200-201 and :94/:220
  const float lb_scale = (float)(log((double)kScaleRange) / log(2.0));
  const float lb_scale_step = lb_scale / (unsigned)BRISK_SCALES;
  P->scale_list.resize(BRISK_SCALES);
  for (unsigned s = 0; s < BRISK_SCALES; ++s) P->scale_list[s] = (float)pow(2.0, (double)(s * lb_scale_step));
}

// sizeList_[s] = max over all rotations / points of ceil(radius + sigma) + 1 (:118-123, :239-246)
void fill_size_list_v2(BriskPatternHost* P) {
  const int n = P->npoints;
  P->size_list.assign(BRISK_SCALES, 0);
  for (int s = 0; s < BRISK_SCALES; ++s) {
    unsigned best = 0;
    for (int rot = 0; rot < BRISK_NROT; ++rot) {
      for (int i = 0; i < n; ++i) {
        const double m = (double)P->mult[s * n + i];
        const float x = (float)(m * P->uv[((size_t)rot * n + i) * 2 + 0]);
        const float y = (float)(m * P->uv[((size_t)rot * n + i) * 2 + 1]);
        const float sg = P->sigma[s * n + i];
        const unsigned size = (unsigned)(ceil(((sqrt((double)(x * x + y * y))) + sg)) + 1);
        if (best < size) best = size;
      }
    }
    P->size_list[s] = (int)best;
  }
}

// scaling / scaling2 of SmoothedIntensity (:387, :412-413) depend only on sigma: tabulated per (scale, point)
void fill_scaling(BriskPatternHost* P) {
  const size_t n = P->sigma.size();
  P->scaling.resize(2 * n);
  for (size_t i = 0; i < n; ++i) {
    const float sigma_half = P->sigma[i];
    const float area = (float)(4.0 * sigma_half * sigma_half);
    const int scaling = (int)(4194304.0 / area);
    const int scaling2 = (int)((float)scaling * area / 1024.0);
    P->scaling[2 * i] = scaling;
    P->scaling[2 * i + 1] = scaling2;
  }
}

void fill_thresholds(BriskPatternHost* P) {
  fill_scaling(P);
  // smallest float size whose scale index (before saturation) reaches s; bisection on the bit pattern
  P->size_thresh.assign(BRISK_SCALES, 0.0f);
  for (int s = 1; s < BRISK_SCALES; ++s) {
    uint32_t lo = 0x00800000u;  // smallest normal float: index 0
    uint32_t hi = 0x7F000000u;  // huge: index saturates
    while (hi - lo > 1) {
      const uint32_t mid = lo + (hi - lo) / 2;
      float f;
      memcpy(&f, &mid, 4);
      if (brisk_pattern_scale_index_host(f) >= s) hi = mid; else lo = mid;
    }
    memcpy(&P->size_thresh[s], &hi, 4);
  }
  // scale used when scale invariance is off (:631-635)
  static const float log2f_ = (float)0.693147180559945;
  const float lb_scalerange = (float)(log((double)kScaleRange) / (log2f_));
  const float basicSize06 = (float)(kBasicSize * 0.6);
  const int v = (int)((unsigned)BRISK_SCALES / lb_scalerange * (log(1.45 * kBasicSize / (basicSize06)) / log2f_) + 0.5);
  P->basicscale = v > 0 ? v : 0;
}

bool finish_v2(BriskPatternHost* P, const std::vector<float>& pts, const std::vector<unsigned>& sp,
               const std::vector<unsigned>& lp, float patternScale, std::string* err) {
  const int n = (int)pts.size() / 3;
  if (n <= 0 || n > BRISK_MAX_POINTS) { *err = "pattern: bad point count"; return false; }
  if ((int)sp.size() / 2 != 384) {  // CHECK_EQ(noShortPairs_, kDescriptorLength) (:286)
    *err = "pattern: number of short pairs must be 384";
    return false;
  }
  if ((int)lp.size() / 2 > BRISK_MAX_LONG) { *err = "pattern: too many long pairs"; return false; }
  for (unsigned v : sp) if (v >= (unsigned)n) { *err = "pattern: short pair index out of range"; return false; }
  for (unsigned v : lp) if (v >= (unsigned)n) { *err = "pattern: long pair index out of range"; return false; }
  P->npoints = n;
  std::vector<float> u_x(n), u_y(n), sg(n);
  for (int i = 0; i < n; ++i) {  // :212-216
    u_x[i] = pts[3 * i + 0]; u_x[i] *= patternScale;
    u_y[i] = pts[3 * i + 1]; u_y[i] *= patternScale;
    sg[i] = pts[3 * i + 2]; sg[i] *= patternScale;
  }
  fill_scale_list(P);
  P->mult.resize((size_t)BRISK_SCALES * n);
  P->sigma.resize((size_t)BRISK_SCALES * n);
  for (int s = 0; s < BRISK_SCALES; ++s)
    for (int i = 0; i < n; ++i) {
      P->mult[s * n + i] = P->scale_list[s];
      P->sigma[s * n + i] = kSigmaScale * P->scale_list[s] * sg[i];  // :236
    }
  P->uv.resize((size_t)BRISK_NROT * n * 2);
  for (int rot = 0; rot < BRISK_NROT; ++rot) {
    const double theta = (double)rot * 2 * M_PI / (double)BRISK_NROT;  // :228-229
    for (int i = 0; i < n; ++i) {
      P->uv[((size_t)rot * n + i) * 2 + 0] = (u_x[i] * cos(theta) - u_y[i] * sin(theta));  // :231-232
      P->uv[((size_t)rot * n + i) * 2 + 1] = (u_x[i] * sin(theta) + u_y[i] * cos(theta));  // :233-234
    }
  }
  fill_size_list_v2(P);
  P->nshort = (int)sp.size() / 2;
  P->short_pairs.resize(sp.size());
  for (size_t k = 0; k < sp.size(); ++k) P->short_pairs[k] = (uint16_t)sp[k];
  P->nlong = (int)lp.size() / 2;
  P->long_pairs.resize((size_t)P->nlong * 4);
  for (int p = 0; p < P->nlong; ++p) {  // :267-280
    const unsigned i = lp[2 * p], j = lp[2 * p + 1];
    const float dx = (u_x[j] - u_x[i]);
    const float dy = (u_y[j] - u_y[i]);
    const float norm_sq = dx * dx + dy * dy;
    P->long_pairs[4 * p + 0] = (int)i;
    P->long_pairs[4 * p + 1] = (int)j;
    P->long_pairs[4 * p + 2] = (int)((dx / (norm_sq)) * 2048.0 + 0.5);
    P->long_pairs[4 * p + 3] = (int)((dy / (norm_sq)) * 2048.0 + 0.5);
  }
  P->strings = (int)ceil(((float)P->nshort) / 128.0) * 4 * 4;  // :283-284
  fill_thresholds(P);
  return true;
}

// BRISK 1.0 kernel: generateKernel(rList, nList, 5.85, 8.2) (:65-178, :316-339)
bool build_v1(BriskPatternHost* P, float patternScale, std::string* err) {
  const int rings = 5;
  float rList[5];
  const int nList[5] = {1, 10, 14, 15, 20};
  const double f = 0.85 * patternScale;
  rList[0] = (float)(f * 0);
  rList[1] = (float)(f * 2.9);
  rList[2] = (float)(f * 4.9);
  rList[3] = (float)(f * 7.4);
  rList[4] = (float)(f * 10.8);
  const float dMax = 5.85f, dMin = 8.2f;
  int n = 0;
  for (int r = 0; r < rings; ++r) n += nList[r];
  P->npoints = n;
  fill_scale_list(P);
  P->mult.resize((size_t)BRISK_SCALES * n);
  P->sigma.resize((size_t)BRISK_SCALES * n);
  P->uv.resize((size_t)BRISK_NROT * n * 2);
  P->size_list.assign(BRISK_SCALES, 0);
  for (int s = 0; s < BRISK_SCALES; ++s) {
    int i = 0;
    for (int ring = 0; ring < rings; ++ring)
      for (int num = 0; num < nList[ring]; ++num, ++i) {
        const float sl = P->scale_list[s];
        P->mult[s * n + i] = sl * rList[ring];  // scaleList_[scale] * radiusList[ring] (float product, :105)
        float sg;
        if (ring == 0) sg = (float)(kSigmaScale * sl * 0.5);  // :111
        else sg = (float)(kSigmaScale * sl * ((double)rList[ring]) * sin(M_PI / nList[ring]));  // :113-114
        P->sigma[s * n + i] = sg;
        const unsigned size = (unsigned)(ceil((double)((sl * rList[ring]) + sg)) + 1);  // :118-120
        if ((unsigned)P->size_list[s] < size) P->size_list[s] = (int)size;
      }
  }
  for (int rot = 0; rot < BRISK_NROT; ++rot) {
    const double theta = (double)rot * 2 * M_PI / (double)BRISK_NROT;  // :100
    int i = 0;
    for (int ring = 0; ring < rings; ++ring)
      for (int num = 0; num < nList[ring]; ++num, ++i) {
        const double alpha = ((double)num) * 2 * M_PI / (double)nList[ring];  // :104
        P->uv[((size_t)rot * n + i) * 2 + 0] = cos(alpha + theta);
        P->uv[((size_t)rot * n + i) * 2 + 1] = sin(alpha + theta);
      }
  }
  // pairs from the scale-0 / rotation-0 points (:147-173), indexChange = identity
  std::vector<float> px(n), py(n);
  for (int i = 0; i < n; ++i) {
    px[i] = (float)((double)P->mult[i] * P->uv[(size_t)i * 2 + 0]);
    py[i] = (float)((double)P->mult[i] * P->uv[(size_t)i * 2 + 1]);
  }
  const float dMin_sq = dMin * dMin, dMax_sq = dMax * dMax;
  for (int i = 1; i < n; i++)
    for (int j = 0; j < i; j++) {
      const float dx = px[j] - px[i];
      const float dy = py[j] - py[i];
      const float norm_sq = (dx * dx + dy * dy);
      if (norm_sq > dMin_sq) {
        P->long_pairs.push_back(i);
        P->long_pairs.push_back(j);
        P->long_pairs.push_back((int)((dx / (norm_sq)) * 2048.0 + 0.5));
        P->long_pairs.push_back((int)((dy / (norm_sq)) * 2048.0 + 0.5));
      }
      if (norm_sq < dMax_sq) {
        P->short_pairs.push_back((uint16_t)i);
        P->short_pairs.push_back((uint16_t)j);
      }
    }
  P->nshort = (int)P->short_pairs.size() / 2;
  P->nlong = (int)P->long_pairs.size() / 4;
  if (P->nshort > BRISK_MAX_SHORT || P->nlong > BRISK_MAX_LONG) { *err = "pattern: generated kernel too large"; return false; }
  P->strings = (int)ceil(((float)P->nshort) / 128.0) * 4 * 4;  // :176
  fill_thresholds(P);
  return true;
}

}  // namespace

int brisk_pattern_scale_index_host(float size) {
  static const float log2f_ = (float)0.693147180559945;                   // :621
  static const float lb_scalerange = (float)(log((double)kScaleRange) / (log2f_));  // :622
  static const float basicSize06 = (float)(kBasicSize * 0.6);             // :629
  const int v = (int)((unsigned)BRISK_SCALES / lb_scalerange * (log((double)(size / (basicSize06))) / log2f_) + 0.5);  // :640-641
  return v > 0 ? v : 0;
}

bool brisk_pattern_build_default(int version, float pattern_scale, BriskPatternHost* out, std::string* err) {
  *out = BriskPatternHost();
  if (version == 2) {
    std::vector<float> pts(&brisk_default_points[0][0], &brisk_default_points[0][0] + 3 * BRISK_DEFAULT_NPOINTS);
    std::vector<unsigned> sp, lp;
    for (int i = 0; i < BRISK_DEFAULT_NSHORT; ++i) { sp.push_back(brisk_default_short_pairs[i][0]); sp.push_back(brisk_default_short_pairs[i][1]); }
    for (int i = 0; i < BRISK_DEFAULT_NLONG; ++i) { lp.push_back(brisk_default_long_pairs[i][0]); lp.push_back(brisk_default_long_pairs[i][1]); }
    return finish_v2(out, pts, sp, lp, pattern_scale, err);
  }
  if (version == 1) return build_v1(out, pattern_scale, err);
  *err = "only Version::briskV1 or Version::briskV2 supported!";  // :341
  return false;
}

bool brisk_pattern_build_from_text(const char* text, float pattern_scale, BriskPatternHost* out, std::string* err) {
  *out = BriskPatternHost();
  if (!text) { *err = "pattern: null text"; return false; }
  std::istringstream ss(text);
  std::string tok;
  auto next = [&](std::string* t) { return (bool)(ss >> *t); };
  if (!next(&tok)) { *err = "pattern: empty"; return false; }
  const long n = strtol(tok.c_str(), nullptr, 10);
  if (n <= 0 || n > BRISK_MAX_POINTS) { *err = "pattern: bad point count"; return false; }
  std::vector<float> pts(3 * n);
  for (long i = 0; i < 3 * n; ++i) {
    if (!next(&tok)) { *err = "pattern: truncated points"; return false; }
    pts[i] = strtof(tok.c_str(), nullptr);
  }
  std::vector<unsigned> sp, lp;
  if (!next(&tok)) { *err = "pattern: missing short pairs"; return false; }
  const long ns = strtol(tok.c_str(), nullptr, 10);
  if (ns < 0 || ns > BRISK_MAX_SHORT) { *err = "pattern: bad short pair count"; return false; }
  for (long i = 0; i < 2 * ns; ++i) {
    if (!next(&tok)) { *err = "pattern: truncated short pairs"; return false; }
    sp.push_back((unsigned)strtoul(tok.c_str(), nullptr, 10));
  }
  if (!next(&tok)) { *err = "pattern: missing long pairs"; return false; }
  const long nl = strtol(tok.c_str(), nullptr, 10);
  if (nl < 0 || nl > BRISK_MAX_LONG) { *err = "pattern: bad long pair count"; return false; }
  for (long i = 0; i < 2 * nl; ++i) {
    if (!next(&tok)) { *err = "pattern: truncated long pairs"; return false; }
    lp.push_back((unsigned)strtoul(tok.c_str(), nullptr, 10));
  }
  return finish_v2(out, pts, sp, lp, pattern_scale, err);
}
