/*
 * brisk_oracle.h - CPU restatement ("oracle") of the ethzasl_brisk detect+describe hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library (ethzasl_brisk_amd/) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / reported CPU baseline.
 *
 * Parity pin: this restatement reproduces the reference's own golden vectors
 * (brisk/src/test/test_data/brisk_verification_{ast,harris}.set) bit-exactly - see
 * tests/test_oracle_golden.py.  The reference itself is unbuildable in this image (it needs
 * OpenCV or NestorCV plus glog headers: agast/include/agast/wrap-opencv.h:41-53), so there is no
 * oracle/_ref build.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef BRISK_ORACLE_H_
#define BRISK_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Binary-identical to cv::KeyPoint (7 x 4 bytes). */
typedef struct {
  float x, y, size, angle, response;
  int octave, class_id;
} bo_keypoint;

/* ---- stage-level entry points (used by per-stage parity tests) ---- */
/* brisk/src/image-down-sampling.cc:142-392 ; dst is (w/2) x (h/2) */
void bo_halfsample8(const uint8_t* src, int w, int h, uint8_t* dst);
/* brisk/src/image-down-sampling.cc:550-787 ; dst is 2*(w/3) x 2*(h/3) */
void bo_twothirdsample8(const uint8_t* src, int w, int h, uint8_t* dst);
/* brisk/src/brisk-layer.cc:278-598 (pass order emulated literally) */
void bo_threshold_map(const uint8_t* img, int w, int h, uint8_t* thrmap);
/* agast/src/oast9-16-nms.cc:39-1976 (bisection + segment test) */
int bo_oast9_16_corner_score(const uint8_t* p, int stride, int b);
/* agast/src/agast5-8-nms.cc:39-357 */
int bo_agast5_8_corner_score(const uint8_t* p, int stride, int b);
/* agast/src/oast9-16.cc:43-1859 ; writes (x,y) pairs in raster order, returns count (<= cap) */
int bo_oast9_16_detect(const uint8_t* img, int w, int h, const uint8_t* thrmap, int b, int upper,
                       int lower, int* xy, int cap);
/* brisk/include/brisk/internal/integral-image.h:56-161 ; out is (h+1) x (w+1) int32 */
void bo_integral_image8(const uint8_t* img, int w, int h, int32_t* out);

/* 16-bit image functions (brisk_oracle_16bit.c): image-down-sampling.cc:56-139, :394-548, integral-image.h:163-218.
 * The down-samplers return -1 where the reference's loops write nothing (fewer than 16 / 12 usable columns). */
int bo_halfsample16(const uint16_t* src, int w, int h, uint16_t* dst);
int bo_twothirdsample16(const uint16_t* src, int w, int h, uint16_t* dst);
void bo_integral_image16(const uint16_t* src, int w, int h, float* out);

/* ---- detector: brisk/src/brisk-feature-detector.cc:77-85 ---- */
typedef struct bo_scale_space bo_scale_space;
/* BriskScaleSpace(octaves, suppress) + ConstructPyramid(image, threshold) */
bo_scale_space* bo_scale_space_create(const uint8_t* img, int w, int h, int threshold, int octaves);
/* ConstructPyramid(image, threshold, overwrite_lower_thres) */
bo_scale_space* bo_scale_space_create_ex(const uint8_t* img, int w, int h, int threshold, int octaves, int lower_threshold);
void bo_scale_space_destroy(bo_scale_space* s);
int bo_scale_space_layers(const bo_scale_space* s);
/* which: 0 image, 1 score map (lazy cache state), 2 threshold map */
const uint8_t* bo_scale_space_map(const bo_scale_space* s, int layer, int which, int* w, int* h);
/* GetKeypoints (suppressScaleNonmaxima = true paths, octaves >= 0); returns count; *out malloc'd */
int bo_scale_space_get_keypoints(bo_scale_space* s, bo_keypoint** out);
/* the same with suppressScaleNonmaxima_ given; suppress == 0 with several layers is brisk-scale-space.cc:131-170 incl. the
 * agastPoints.at(0) indexing (PARITY UNPINNED: nothing in the reference's tests reaches that branch); -1 = the
 * reference has no defined result on this input */
int bo_scale_space_get_keypoints_ex(bo_scale_space* s, int suppress, bo_keypoint** out);
int bo_detect_ex(const uint8_t* img, int w, int h, int threshold, int octaves, int suppress_scale_nonmaxima,
                 const uint8_t* mask, bo_keypoint** out);
/* Convenience: whole detectImpl incl. optional mask (h x w, u8, may be NULL). */
int bo_detect(const uint8_t* img, int w, int h, int threshold, int octaves, const uint8_t* mask,
              bo_keypoint** out);
/* BriskFeatureDetector::ComputeScale (brisk-feature-detector.cc:87-92, brisk-scale-space.cc:104-123): scores / scales
 * for provided keypoints; PARITY UNPINNED (nothing in the reference exercises it); -1 = the reference has no defined
 * result on this input (see the function).  *out malloc'd. */
int bo_compute_scale(const uint8_t* img, int w, int h, int threshold, int octaves, int suppress_scale_nonmaxima,
                     const bo_keypoint* in, int n_in, bo_keypoint** out);
void bo_free(void* p);

/* ---- extractor: brisk/src/brisk-descriptor-extractor.cc ---- */
typedef struct bo_extractor bo_extractor;
/* version 2 = default 66-point pattern (or pattern_text if non-NULL, .ptn syntax);
 * version 1 = generated 60-point legacy kernel. */
bo_extractor* bo_extractor_create(int rotation_invariant, int scale_invariant, int version,
                                  float pattern_scale, const char* pattern_text);
void bo_extractor_destroy(bo_extractor* e);
int bo_extractor_descriptor_size(const bo_extractor* e); /* strings_ (48 or 64) */
int bo_extractor_points(const bo_extractor* e);
const float* bo_extractor_scale_list(const bo_extractor* e);        /* 64 */
const unsigned* bo_extractor_size_list(const bo_extractor* e);      /* 64 */
const float* bo_extractor_pattern(const bo_extractor* e);           /* [64][1024][points][3] */
/* scale index of a keypoint size (brisk-descriptor-extractor.cc:636-650) */
int bo_extractor_scale_index(const bo_extractor* e, float size);
/* compute(): filters kps in place (count returned), writes angle, desc = count x strings_ (zeroed) */
int bo_extractor_compute(const bo_extractor* e, const uint8_t* img, int w, int h, bo_keypoint* kps,
                         int n, uint8_t* desc);

/* ---- matcher: brisk/include/brisk/internal/hamming.h, brisk/src/brute-force-matcher.cc (brisk_oracle_match.c) ---- */
typedef struct { int queryIdx, trainIdx, imgIdx; float distance; } bo_dmatch; /* == cv::DMatch */
int bo_hamming(const uint8_t* a, const uint8_t* b, int size_bytes);
void bo_match_knn(const uint8_t* query, int nq, int q_pitch, int dim, int nimg, const uint8_t* const* train,
                  const int* ntrain, const int* t_pitch, const uint8_t* const* masks, const int* mask_pitch, int k,
                  bo_dmatch* out, int* out_count);
bo_dmatch* bo_match_radius(const uint8_t* query, int nq, int q_pitch, int dim, int nimg, const uint8_t* const* train,
                           const int* ntrain, const int* t_pitch, const uint8_t* const* masks, const int* mask_pitch,
                           float max_distance, int* out_count);

/* ---- uniformity enforcement (brisk_oracle_uniformity.c; PARITY UNPINNED for AGAST keypoints, see there) ---- */
int bo_enforce_uniformity(const bo_keypoint* kps, int n, int rows, int cols, double radius, int max_keypoints,
                          bo_keypoint* out);
/* KeyPointBucketing (key-point-bucketing-inl.h:40-112): kept points in descending score order; -1 = arguments the reference CHECKs */
int bo_key_point_bucketing(const bo_keypoint* kps, int n, int rows, int cols, int max_keypoints, int nbu, int nbv,
                           bo_keypoint* out);

#ifdef __cplusplus
}
#endif
#endif  /* BRISK_ORACLE_H_ */
