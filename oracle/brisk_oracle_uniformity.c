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
