/*
 * brisk_hip_debug.h - test and tuning entry points of libbrisk_hip.so, NOT part of the drop-in boundary.
 *
 * They exist only in libraries built with -DBRISK_HIP_TUNING (ethzasl_brisk_amd/build.py: build(), the library the test
 * suite, tools/ and bench.py load).  The release library (build_release(): libbrisk_hip_release.so, what INTEGRATION.md
 * links) exports none of them, reads no tuning variable from the environment and compiles every debug bit to 0.
 */
#ifndef BRISK_HIP_DEBUG_H_
#define BRISK_HIP_DEBUG_H_

#include "brisk_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-stage device entry points (parity tests of individual kernels) ---------------------- */
/* which: 0 pyramid image, 1 score-state map low byte (D), after the last detect on frame slot 0.
 * Copies layer `layer` (w x h, tightly packed u8) to the host buffer. */
int brisk_hip_debug_layer(brisk_hip_ctx* ctx, int frame, int layer, int which, uint8_t* out, int* w, int* h);
/* test knobs: bit0 = route every AGAST candidate through the direct-evaluation safety-net kernel; bit 25 = the pinned
 * result buffer of the one-frame host calls holds 16 KB only (results beyond it take the staged copies) */
int brisk_hip_debug_set_flags(brisk_hip_ctx* ctx, int flags);
/* integral image of frame slot `frame` after the last describe: (h+1) x (w+1) u32, tightly packed.  The engine keeps it
 * modulo 2^24 in 3-byte elements where the pattern's boxes are small enough for that (every built-in pattern; the values
 * come back zero-extended) and as u32 otherwise - brisk_hip_debug_integral_bits tells which (24 / 32); debug flag bit 18
 * forces the 32-bit form. */
int brisk_hip_debug_integral(brisk_hip_ctx* ctx, int frame, uint32_t* out);
int brisk_hip_debug_integral_bits(brisk_hip_ctx* ctx, int frame);
/* test knob: overwrites the device a pattern handle believes its tables live on (the engine refuses a pattern / context
 * pair of different devices with BRISK_HIP_ERR_ARG; a one-GPU box can only test the refusal by forging the field).
 * device < 0 restores the true one. */
int brisk_hip_debug_forge_pattern_device(brisk_hip_pattern* p, int device);
/* number of describe calls that reused the image a detect call had left on the device (brisk_hip_describe_same_image, or
 * brisk_hip_describe under BRISK_HIP_IMAGE_CACHE=1) and skipped the upload and the layer-0 pass */
int brisk_hip_debug_image_reuse(brisk_hip_ctx* ctx);
/* the uniformity filter of brisk_hip_set_uniformity alone, on a GIVEN keypoint list of a rows x cols image (tests of the
 * filter kernels on lists no detector produces: tight clusters, the smallest radii); out holds n_in keypoints */
int brisk_hip_debug_filter_keypoints(brisk_hip_ctx* ctx, const brisk_hip_keypoint* in, int n_in, int rows, int cols, double radius,
                                     int max_keypoints, brisk_hip_keypoint* out, int* n);
/* wall seconds per phase of the pool's calls, summed over all calls so far (tools): 0 until the call has joined a group, 1 its
 * staging copies, 2 leader: until a context is free and all members have staged, 3 leader: copies queued, 4 leader: batch queued,
 * 5 leader: until the results are in host memory, 6 member: until its group is done, 7 the whole call; out holds 8 doubles */
int brisk_hip_debug_pool_phases(brisk_hip_pool* pool, double* out);
/* per-frame work counts of the last batch (tools only): out[0] candidates, out[1] keypoints, out[2] described
 * keypoints, out[3] overflow flags, out[4 + l] tie candidates of layer l, out[20 .. 27] experiment words; out holds 28 ints. */
int brisk_hip_debug_counters(brisk_hip_ctx* ctx, int frame, int* out, int* nlayers);
/* the raw counter record of a frame (tools only; instrumented build variants append fields): returns its size in bytes,
 * -1 on error; `bytes` = size of out */
int brisk_hip_debug_counters_raw(brisk_hip_ctx* ctx, int frame, void* out, int bytes);

#ifdef __cplusplus
}
#endif
#endif /* BRISK_HIP_DEBUG_H_ */
