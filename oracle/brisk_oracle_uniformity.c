/* brisk_oracle_uniformity.c - CPU restatement of the reference's keypoint uniformity enforcement.
 *
 * TEST INFRASTRUCTURE ONLY (see brisk_oracle.h): never linked into or loaded by the product library.
 *
 * Follows EnforceKeyPointUniformity (brisk/include/brisk/internal/uniformity-enforcement-inl.h:44-194) with the
 * occupancy mask LUT of ScaleSpaceLayer (brisk/include/brisk/internal/scale-space-layer-inl.h:88-97).
 *
 * PARITY UNPINNED.  In the reference this filter is only reachable through the Harris ScaleSpaceFeatureDetector
 * (scale-space-layer-inl.h:372-375), which is outside the AGAST hot path, and the only golden data that exercises it
 * (brisk_verification_harris.set) does so through that detector; nothing in the reference pins it for AGAST
 * keypoints.  The engine offers it as an optional post-filter of BriskFeatureDetector output (BASELINE config 4:
 * "uniformity-enforced"); this file is the literal arithmetic of the reference applied to (x, y, response):
 *   - points sorted by score, descending (reference: std::sort on operator<, which is not stable; here equal scores
 *     keep their input order);
 *   - scaling = 15 / radius (float); occupancy u8 image of (rows * ceil(scaling) + 32) x (cols * ceil(scaling) + 32);
 *   - a point at occupancy cell (int(y * scaling + 16), int(x * scaling + 16)) is accepted iff
 *     sqrtf(sqrtf(score / maxScore)) * 255 >= occupancy there; an accepted point adds, saturating,
 *     uint8(ceil(LUT[y][x] * 0.99f * that value)) over the 31 x 31 cells around it;
 *   - at most max_keypoints points are kept; output order = acceptance order (descending score).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "brisk_oracle.h"

typedef struct { float score; int index; } scored;

static int cmp_scored(const void* a, const void* b) {
  const scored* p = (const scored*)a;
  const scored* q = (const scored*)b;
  if (p->score != q->score) return p->score > q->score ? -1 : 1;
  return (p->index > q->index) - (p->index < q->index);
}

/* kps: n keypoints (x, y, response used); writes the kept keypoints to out (capacity n) and returns their number */
int bo_enforce_uniformity(const bo_keypoint* kps, int n, int rows, int cols, double radius, int max_keypoints,
                          bo_keypoint* out) {
  if (n <= 0) return 0;
  float lut[31][31];
  for (int x = 0; x < 31; ++x)
    for (int y = 0; y < 31; ++y) {
      const double v = 1 - (double)((15 - x) * (15 - x) + (15 - y) * (15 - y)) / (double)(15 * 15);
      lut[y][x] = (float)(v > 0.0 ? v : 0.0);
    }
  scored* order = (scored*)malloc(sizeof(scored) * (size_t)n);
  for (int i = 0; i < n; ++i) { order[i].score = kps[i].response; order[i].index = i; }
  qsort(order, (size_t)n, sizeof(scored), cmp_scored);
  const float maxScore = order[0].score;
  const float scaling = (float)(15.0 / (float)radius);
  const int oh = (int)(rows * ceil(scaling) + 32), ow = (int)(cols * ceil(scaling) + 32);
  uint8_t* occ = (uint8_t*)calloc((size_t)oh * ow, 1);
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    const bo_keypoint* p = &kps[order[i].index];
    const int cy = (int)(p->y * scaling + 16);
    const int cx = (int)(p->x * scaling + 16);
    const double s0 = (double)occ[(size_t)cy * ow + cx];
    const float nsc1 = sqrtf(sqrtf(p->response / maxScore)) * 255.0f;
    if (nsc1 < s0) continue;
    const float nsc = 0.99f * nsc1;
    for (int y = 0; y < 31; ++y)
      for (int x = 0; x < 31; ++x) {
        uint8_t* c = &occ[(size_t)(cy + y - 15) * ow + (cx + x - 15)];
        const int add = (int)(uint8_t)(int)ceilf(lut[y][x] * nsc);
        const int s = *c + add;
        *c = (uint8_t)(s > 255 ? 255 : s);
      }
    out[kept++] = *p;
    if (kept == max_keypoints) break;
  }
  free(occ);
  free(order);
  return kept;
}

/* KeyPointBucketing (brisk/include/brisk/internal/key-point-bucketing-inl.h:74-112) + KeyPointBuckets::filterKeyPoints
 * (:40-72) + KeyPointBuckets' constructor (key-point-bucketing.h:50-67), applied to (x, y, response) of AGAST keypoints
 * (x, y truncated to the integer pixel as PointWithScore holds them, score-calculator.h:66-85).  PARITY UNPINNED for the
 * same reason as the uniformity filter: the reference only reaches it through the Harris ScaleSpaceLayer.
 *   - points sorted by score, descending (std::sort / std::partial_sort on `score > other.score`: not stable; here equal
 *     scores keep their input order);
 *   - one bucket in either direction: the first max_keypoints points are kept (:87-98);
 *   - otherwise max per bucket = max_keypoints / (nbu * nbv), step_u = 1 + (cols - 1) / nbu, step_v = 1 + (rows - 1) / nbv,
 *     a point is kept while the counter of bucket (x / step_u, y / step_v) is below the maximum (:54-63).
 * Returns the number of kept points (written to out in descending score order), -1 for arguments the reference CHECKs. */
int bo_key_point_bucketing(const bo_keypoint* kps, int n, int rows, int cols, int max_keypoints, int nbu, int nbv,
                           bo_keypoint* out) {
  if (rows <= 0 || cols <= 0 || nbu <= 0 || nbv <= 0 || nbu >= cols || nbv >= rows || max_keypoints <= 0) return -1;
  if (n <= 0) return 0;
  if ((nbu == 1 || nbv == 1) && n <= max_keypoints) {  /* :87-88: sorted and cut only when there are too many - else untouched */
    for (int i = 0; i < n; ++i) out[i] = kps[i];
    return n;
  }
  scored* order = (scored*)malloc(sizeof(scored) * (size_t)n);
  for (int i = 0; i < n; ++i) { order[i].score = kps[i].response; order[i].index = i; }
  qsort(order, (size_t)n, sizeof(scored), cmp_scored);
  int kept = 0;
  if (nbu == 1 || nbv == 1) {
    for (int i = 0; i < n && kept < max_keypoints; ++i) out[kept++] = kps[order[i].index];
  } else {
    const unsigned max_per_bucket = (unsigned)max_keypoints / ((unsigned)nbu * (unsigned)nbv);
    const unsigned step_u = 1u + ((unsigned)cols - 1u) / (unsigned)nbu, step_v = 1u + ((unsigned)rows - 1u) / (unsigned)nbv;
    unsigned* counter = (unsigned*)calloc((size_t)nbu * nbv, sizeof(unsigned));
    for (int i = 0; i < n; ++i) {
      const bo_keypoint* p = &kps[order[i].index];
      const unsigned cu = (unsigned)(int)p->x / step_u, cv = (unsigned)(int)p->y / step_v;
      unsigned* c = &counter[(size_t)cu * nbv + cv];
      if (*c < max_per_bucket) { ++*c; out[kept++] = *p; }
    }
    free(counter);
  }
  free(order);
  return kept;
}
