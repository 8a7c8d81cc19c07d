// brisk_common.h - shared types of the MI355X BRISK engine (host + device).
//
// Data layout in HBM (per frame slot):
//   pyr   : u8  pyramid, layer l at byte offset L[l].off, row stride L[l].stride (multiple of 64)
//   smap  : u16 "score-state map", same offsets/strides as pyr (in elements):
//             bits 0-7   D   = contrast score of a detected pixel (initial scores_ value,
//                              brisk/src/brisk-layer.cc:110-116), 0 = not a detection
//             bits 8-11  number of IsMax2D probes the candidate issued before its early exit
//             bits 12-13 candidate status (BRISK_ST_*)
//             bit  14    candidate reaches its own-layer 3x3 / 4x4 patch reads (if it is a 2D max)
//             bit  15    pixel was score-touched (threshold 1) from the layer below
//   cand  : BriskCand[cand_cap] candidate records (unordered, atomic append)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BRISK_HD __host__ __device__ inline
#else
#define BRISK_HD inline
#endif

// BRISK_HIP_TUNING (the build the tests, tools and bench.py load): environment knobs, debug bits and the brisk_hip_debug_*
// entry points (include/brisk_hip_debug.h) exist.  Without it - the release library a maintainer links,
// libbrisk_hip_release.so - every knob is its default at compile time: no getenv, no debug bit read by any kernel.
#ifdef BRISK_HIP_TUNING
#define BRISK_DBG_FLAGS(G) ((G).debug_flags)
#else
BRISK_HD int brisk_no_debug_flags() { return 0; }  // (a call, not a literal: no constant-operand warnings; folded all the same)
#define BRISK_DBG_FLAGS(G) brisk_no_debug_flags()
#endif

// experiment (build variant -DBRISK_CHAIN_PRIO=n, tools/r06_chainprio.sh): the latency-bound kernels that run beside the integral kernel -
// tie resolution, k_finalize, k_desc_prepare - raise their waves' priority
#ifdef BRISK_CHAIN_PRIO
#define BRISK_CHAIN_SETPRIO() __builtin_amdgcn_s_setprio(BRISK_CHAIN_PRIO)
#else
#define BRISK_CHAIN_SETPRIO() do { } while (0)
#endif

#define BRISK_MAX_LAYERS 16
// Below this AGAST threshold a detection may store a score <= 2, which the reference's lazy cache treats as "not cached"
// (brisk-layer.cc:118-132): a FRAME in which k_detect stores such a score runs the ordered path (k_ordered_keypoints); all
// other frames - thresholds 10 ... 19 on ordinary images: all of them - stay on the fast path (BriskFrameCounters::low_score)
#define BRISK_FAST_PATH_MIN_THRESHOLD 20
#define BRISK_STRIDE_ALIGN 64

// candidate status (smap bits 12-13)
#define BRISK_ST_REJ 0u   // failed the 8-neighbour comparison (still issued probes)
#define BRISK_ST_PASS 1u  // 2D maximum (no tie, or tie resolved in its favour)
#define BRISK_ST_TIE 2u   // 2D maximum with equal-score neighbours, not yet resolved
#define BRISK_ST_FAIL 3u  // tie resolved against it

#define BRISK_SM_D(v) ((v) & 0xFFu)
#define BRISK_SM_NPROBED(v) (((v) >> 8) & 0xFu)
#define BRISK_SM_STATUS(v) (((v) >> 12) & 0x3u)
#define BRISK_SM_E5 0x4000u
#define BRISK_SM_TOUCH 0x8000u

// reference constants: brisk/src/brisk-scale-space.cc:45-51
#define BRISK_BASIC_SIZE 12.0f
#define BRISK_MAX_THRESHOLD 1
#define BRISK_DROP_THRESHOLD 5
#define BRISK_MIN_DROP 15
#define BRISK_UPPER_THRESHOLD 230
#define BRISK_LOWER_THRESHOLD 10

// descriptor constants: brisk/src/brisk-descriptor-extractor.cc:57-62
#define BRISK_SCALES 64
#define BRISK_NROT 1024
#define BRISK_MAX_POINTS 128
#define BRISK_MAX_SHORT 1792  // generateKernel with 60 points: up to 1770 pairs (pattern scales < 1 make most of them short: descriptors of up to 224 bytes)
#define BRISK_MAX_LONG 2048

struct BriskLayerGeom {
  int w, h, stride;
  int off;  // element offset of this layer inside the per-frame pyramid / smap buffers
  float scale, offset;
};

struct BriskGeom {
  int nlayers;
  int w, h;            // layer-0 size
  int pyr_elems;       // total elements per frame (sum of stride*h, 256-aligned per layer)
  int threshold;       // AGAST threshold (20..255)
  int single_layer;    // octaves == 0
  int lower_threshold; // lowerThreshold_ of the layers (10; 0 in ComputeScale, brisk-feature-detector.cc:90): ordered path if != 10
  int no_scale_nms;    // suppressScaleNonmaxima == false with more than one layer (ordered path, brisk-scale-space.cc:131-170)
  int debug_flags;     // test knobs: bit0 send every candidate through k_classify_refine_direct (safety-net test);
                       // bits 8-15 k_describe blocks per frame / 8; bit16 integral image not overlapped; bit17 matcher
                       // always through the distance matrix
  BriskLayerGeom L[BRISK_MAX_LAYERS];
  // Layer 0 read in place: when the caller's frames already have the pyramid's layer-0 layout (row pitch = L[0].stride,
  // 16-byte aligned) the detector's kernels read them directly and no layer-0 copy is written (2 MB per 1080p frame).
  // l0_ext = first frame of the launch (null: layer 0 lives in the pyramid buffer), l0_pitch = bytes between frames.
  const uint8_t* l0_ext;
  long l0_pitch;
};

// binary-identical to cv::KeyPoint
struct BriskKeyPoint {
  float x, y, size, angle, response;
  int octave, class_id;
};

struct BriskCand {
  uint16_t x, y;
  uint8_t layer, D, status, flags;  // flags bit0: refined keypoint valid (3-D max), bit1: e5
  int16_t fp_x0, fp_y0;             // e3 footprint anchor on layer+1 (4x4 block)
  uint16_t fp_mask;
  uint16_t pad;
  float kx, ky, ksize, kresp;       // refined keypoint
  uint32_t key;                     // (layer << 26) | (y << 13) | x : output order
};

#define BRISK_TIE_MAX_BANDS 8
// per-frame counters (device)
struct BriskFrameCounters {
  int ncand;                        // appended candidates (may exceed cap -> overflow)
  int ntie[BRISK_MAX_LAYERS];       // tie candidates per layer
  int nkp;                          // final keypoints (detect order)
  int ndesc;                        // keypoints surviving the descriptor border filter
  int overflow;                     // bit0 cand overflow, bit1 tie overflow, bit2 keypoint overflow
  int nredo;                        // candidates deferred to k_classify_refine_direct
  int nvalid_large;                 // > 0: k_finalize left the ordering of this many keypoints to k_finalize_large
  int full_clear;                   // the ordered path wrote the map outside the candidates' footprints: clear all of it
  int tie_ticket;                   // k_tie_resolve: work tickets (entry xcd of the batch, or entry 0 for fewer than 8 frames)
  int tie_prog[BRISK_MAX_LAYERS];   // k_tie_resolve: rows of layer l whose ties are decided and whose touches are performed
  int orient_ticket, desc_ticket;   // k_describe (stage 0 / 1): next run of keypoints (in processing order) to be handed out
  int nestimate;                    // kept keypoints that came without an angle (k_describe stage 0 skips frames that have none)
  int tie_sorted;                   // k_tie_resolve: bit l = layer l's tie list has been rewritten in raster order (layers beyond the on-chip capacity)
  int low_score;                    // k_detect stored a detection score <= 2 (possible below threshold 20 only): the frame takes the ordered path
  int i24;                          // k_integral_final wrote this frame's integral image in 3-byte elements (k_describe reads it accordingly)
  int pad[2];
#ifdef TR_TIMING  // experiments (build variant): per-phase time of k_tie_resolve's decision loop (tools/tie_phases.py)
  int tphase[8];
#endif
#ifdef SB_TIMING  // experiments (build variant): per-phase wave time of k_score_blocks (tools/score_block_phases.py)
  int sphase[8];
#endif
#ifdef CR_TIMING  // experiments (build variant): per-phase lane time of k_classify_refine (tools/classify_phases.py)
  int cphase[8];
#endif
#ifdef DT_TIMING  // experiments (build variant): per-phase wave time of k_detect (tools/detect_phases.py)
  int tdet[8];
#endif
#ifdef DS_TIMING  // experiments (build variant): per-phase wave time of k_describe (tools/describe_phases.py)
  int dphase[8];
#endif
#ifdef TR_TIMELINE  // experiments (build variant): wall-clock stamps of k_tie_resolve per layer (tools/tie_timeline.py)
  int tl[BRISK_MAX_LAYERS * 8];
#endif
  // k_tie_resolve_pair with a layer cut into row bands (small calls): progress of band b = 1 .. BRISK_TIE_MAX_BANDS - 1 of layer l at
  // [l * (BRISK_TIE_MAX_BANDS - 1) + b - 1] (band 0 uses tie_prog[l]); at the end of the struct: the experiment arrays above keep their offsets
  int tie_prog_b[BRISK_MAX_LAYERS * (BRISK_TIE_MAX_BANDS - 1)];
};

// descriptor pattern tables (device pointers or host pointers, same layout)
struct BriskPatternDev {
  int npoints, nshort, nlong, strings;   // strings = descriptor bytes (48 / 64)
  int rotation_invariant, scale_invariant, basicscale;
  int reg_tables;         // at most 896 long pairs with 16-bit weights, at most 512 short pairs, at most 128 points: k_describe keeps the pair tables in registers
  int has_bilinear;       // some (scale, point) has sigma < 0.5: SmoothedIntensity's bilinear branch (:391-408) is reachable
  int int24_ok;           // every box of the pattern covers fewer than 2^24 / 255 pixels: the 24-bit integral image suffices
  const float* mult;      // [64][npoints]  multiplier m so that x = (float)((double)m * U)
  const float* sigma;     // [64][npoints]  box half side
  const int* scaling;     // [64][npoints][2] {scaling, scaling2} of the box (functions of sigma only, :412-413)
  const int* tab;         // [64][npoints][4] {mult, sigma (float bits), scaling, scaling2}: the three tables above in one
                          // 16-byte record per (scale, point) - what k_describe loads (device copy only, may be null on the host)
  const double* uv;       // [1024][npoints][2] unit-scale rotated offsets (x, y)
  const float* size_thresh;  // [64] size_thresh[s] = smallest keypoint size with scale index >= s
  const int* size_list;   // [64] border per scale index
  const uint16_t* short_pairs;  // [nshort][2] (i, j)
  const int* long_pairs;        // [nlong][4]  (i, j, weighted_dx, weighted_dy)
};

BRISK_HD int brisk_align_up(int v, int a) { return (v + a - 1) / a * a; }
// image of layer l of frame `frame` (frame index inside the launch)
BRISK_HD const uint8_t* brisk_layer_img(const BriskGeom& G, const uint8_t* pyr, int frame, int l) {
  if (l == 0 && G.l0_ext) return G.l0_ext + (long)frame * G.l0_pitch;
  return pyr + (long)frame * G.pyr_elems + G.L[l].off;
}
