/* brisk_oracle_match.c - CPU restatement of the reference's Hamming distance and brute-force matcher.
 *
 * TEST INFRASTRUCTURE ONLY (see brisk_oracle.h): never linked into or loaded by the product library.
 *
 * Follows
 *   brisk::Hamming::operator()            brisk/include/brisk/internal/hamming.h:98-112
 *   Hamming::SSSE3PopcntofXORed           brisk/include/brisk/internal/hamming-inl.h:85-134
 *   BruteForceMatcher::commonKnnMatchImpl brisk/src/brute-force-matcher.cc:80-162
 *   BruteForceMatcher::commonRadiusMatchImpl                     :164-213
 * Pinned by the reference's known-answer test (brisk/src/test/test-popcount.cc:60-105, vectors restated in
 * tests/test_oracle_golden.py) and by its homography inlier test (brisk/src/test/test-match.cc:49-126).
 *
 * Third-party behaviour outside /root/reference (OpenCV 3, un-pinned: brisk/package.xml:22), restated from the
 * published cv::DescriptorMatcher source:
 *   isPossibleMatch(mask, q, t) = mask.empty() || mask(q, t) != 0
 *   isMaskedOut(masks, q)       = !masks.empty() && every mask is non-empty and has an all-zero row q
 *                                 (`outCount == masks.size()`: the query can match nothing in any image; an image
 *                                 without a mask, or an empty mask - no train descriptors -, keeps the query alive)
 * Order of equal distances: the reference selects k times the first minimum (image order, then train index) and
 * then calls std::sort on DMatch::operator< (distance only), which is not stable.  The oracle (and the engine)
 * return the lexicographic (distance, imgIdx, trainIdx) order: what the selection produces, and what std::sort
 * leaves untouched for lists of up to 16 entries (insertion sort).
 */
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include "brisk_oracle.h"

/* popcount of a ^ b over size / 16 128-bit words (hamming.h:98-112: `size / 16`, trailing bytes are ignored);
 * the SSSE3 routine is a do-while: a size below 16 still reads one word - callers never do that. */
int bo_hamming(const uint8_t* a, const uint8_t* b, int size) {
  static const uint8_t pop4[16] = {0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4};
  const int words = size / 16;
  int result = 0;
  for (int i = 0; i < words * 16; ++i) {
    const uint8_t x = a[i] ^ b[i];
    result += pop4[x & 0xF] + pop4[x >> 4];
  }
  return result;
}

static int possible(const uint8_t* mask, int mask_pitch, int q, int t) { return mask == NULL || mask[(size_t)q * mask_pitch + t] != 0; }

static int masked_out(int nimg, const uint8_t* const* masks, const int* mask_pitch, const int* ntrain, int q) {
  if (!masks || nimg <= 0) return 0;
  for (int i = 0; i < nimg; ++i) {
    if (!masks[i] || ntrain[i] <= 0) return 0;   /* cv::Mat::empty(): not counted, so outCount < masks.size() */
    for (int t = 0; t < ntrain[i]; ++t)
      if (masks[i][(size_t)q * mask_pitch[i] + t] != 0) return 0;
  }
  return 1;
}

static int cmp_match(const void* pa, const void* pb) {
  const bo_dmatch* a = (const bo_dmatch*)pa;
  const bo_dmatch* b = (const bo_dmatch*)pb;
  if (a->distance != b->distance) return a->distance < b->distance ? -1 : 1;
  if (a->imgIdx != b->imgIdx) return a->imgIdx < b->imgIdx ? -1 : 1;
  return (a->trainIdx > b->trainIdx) - (a->trainIdx < b->trainIdx);
}

/* out: nq * k entries (row q at out + q * k), out_count[q] = matches of query q (0 for a masked-out query) */
void bo_match_knn(const uint8_t* query, int nq, int q_pitch, int dim, int nimg, const uint8_t* const* train,
                  const int* ntrain, const int* t_pitch, const uint8_t* const* masks, const int* mask_pitch, int k,
                  bo_dmatch* out, int* out_count) {
  int** all = (int**)malloc(sizeof(int*) * (size_t)(nimg > 0 ? nimg : 1));
  for (int i = 0; i < nimg; ++i) all[i] = (int*)malloc(sizeof(int) * (size_t)(ntrain[i] > 0 ? ntrain[i] : 1));
  for (int q = 0; q < nq; ++q) {
    out_count[q] = 0;
    if (masked_out(nimg, masks, mask_pitch, ntrain, q)) continue;
    const uint8_t* d1 = query + (size_t)q * q_pitch;
    for (int i = 0; i < nimg; ++i)
      for (int t = 0; t < ntrain[i]; ++t)
        all[i][t] = possible(masks ? masks[i] : NULL, masks ? mask_pitch[i] : 0, q, t)
                        ? bo_hamming(d1, train[i] + (size_t)t * t_pitch[i], dim) : INT_MAX;
    for (int kk = 0; kk < k; ++kk) {
      bo_dmatch best;
      best.queryIdx = q; best.trainIdx = -1; best.imgIdx = -1; best.distance = 3.402823466e+38f;
      for (int i = 0; i < nimg; ++i) {
        if (ntrain[i] <= 0) continue;
        int mv = all[i][0], ml = 0;  /* minMaxLoc: first minimum */
        for (int t = 1; t < ntrain[i]; ++t)
          if (all[i][t] < mv) { mv = all[i][t]; ml = t; }
        if ((double)mv < (double)best.distance) { best.trainIdx = ml; best.imgIdx = i; best.distance = (float)mv; }
      }
      if (best.trainIdx == -1) break;
      /* an exhausted / fully masked row has minimum INT_MAX: float(INT_MAX) = 2147483648 < FLT_MAX, so the
       * reference does push such a "match" (brute-force-matcher.cc:139-153); restated literally */
      all[best.imgIdx][best.trainIdx] = INT_MAX;
      out[(size_t)q * k + out_count[q]++] = best;
    }
    qsort(out + (size_t)q * k, (size_t)out_count[q], sizeof(bo_dmatch), cmp_match);
  }
  for (int i = 0; i < nimg; ++i) free(all[i]);
  free(all);
}

/* returns the matches of all queries concatenated (malloc'd, bo_free), out_count[q] per query */
bo_dmatch* bo_match_radius(const uint8_t* query, int nq, int q_pitch, int dim, int nimg, const uint8_t* const* train,
                           const int* ntrain, const int* t_pitch, const uint8_t* const* masks, const int* mask_pitch,
                           float max_distance, int* out_count) {
  size_t cap = 1024, n = 0;
  bo_dmatch* out = (bo_dmatch*)malloc(sizeof(bo_dmatch) * cap);
  for (int q = 0; q < nq; ++q) {
    out_count[q] = 0;
    if (masked_out(nimg, masks, mask_pitch, ntrain, q)) continue;
    const uint8_t* d1 = query + (size_t)q * q_pitch;
    const size_t first = n;
    for (int i = 0; i < nimg; ++i)
      for (int t = 0; t < ntrain[i]; ++t) {
        if (!possible(masks ? masks[i] : NULL, masks ? mask_pitch[i] : 0, q, t)) continue;
        const int d = bo_hamming(d1, train[i] + (size_t)t * t_pitch[i], dim);
        if ((float)d < max_distance) {
          if (n == cap) { cap *= 2; out = (bo_dmatch*)realloc(out, sizeof(bo_dmatch) * cap); }
          out[n].queryIdx = q; out[n].trainIdx = t; out[n].imgIdx = i; out[n].distance = (float)d;
          ++n;
        }
      }
    out_count[q] = (int)(n - first);
    qsort(out + first, n - first, sizeof(bo_dmatch), cmp_match);
  }
  return out;
}
