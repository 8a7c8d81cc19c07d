// brisk_kernels.h - host-visible launch interface of brisk_kernels.hip
#pragma once
#include <hip/hip_runtime.h>

#include "brisk_common.h"

#define BRISK_DETECT_TILE_W 64
#ifndef BRISK_DETECT_ROWS_PER_THREAD
#define BRISK_DETECT_ROWS_PER_THREAD 4
#endif
#define BRISK_DETECT_TILE_H (16 * BRISK_DETECT_ROWS_PER_THREAD)

struct BriskTileTable {
  int first_tile[BRISK_MAX_LAYERS + 1];
  int tiles_x[BRISK_MAX_LAYERS];
  int total_tiles;
};

// per-batch device buffers of the detector (arrays over frame slots)
struct BriskDetectBuffers {
  uint8_t* pyr;                  // [slots][pyr_elems]
  uint16_t* smap;                // [slots][pyr_elems]
  BriskCand* cand;               // [slots][cand_cap]
  uint8_t* blocks;               // [slots][cand_cap][64] score blocks of the candidates
  int* tie_idx;                  // [slots][BRISK_MAX_LAYERS][tie_cap]
  unsigned* keys;                // [slots][2 * cand_cap]
  BriskFrameCounters* counters;  // [slots]
  BriskKeyPoint* kp_out;         // [slots][kp_cap]
  uint32_t* bandsum;             // [slots][nbands][istride] column sums of band_h-row bands of layer 0 (for the integral)
  int band_h;                    // 96 after the detector's pyramid (k_pyramid_fused), 64 after the descriptor-only layer-0 kernel
  int istride;                   // integral / bandsum row stride (elements)
  int cand_cap, tie_cap, kp_cap;
};

struct BriskDescribeBuffers {
  uint32_t* integral;  // [slots][iframe_elems] (frame pitch in 4-byte units whatever the element size)
  int istride;         // integral row stride (elements)
  int ibits;           // element size of the integral image of this call, every frame alike: 32, or 24 = 3-byte elements (values
                       // modulo 2^24, row pitch istride * 3 bytes: a quarter less to write and to fetch; BriskPatternDev::int24_ok).
                       // k_describe is instantiated for one of the two per launch: the choice is per CALL (brisk_capi.hip,
                       // integral_format), never per frame
  long iframe_elems;
  BriskKeyPoint* dkp;  // [slots][kp_cap] filtered keypoints (angle filled in)
  int* dscale;         // [slots][kp_cap]
  int* dperm;          // [slots][kp_cap] processing order of the keypoints (spatially sorted, L2 locality)
  uint4* drec;         // [slots][kp_cap] the keypoints in processing order: {x, y, angle (float bits), scale | index << 8}
  uint8_t* desc;       // [slots][kp_cap][desc_pitch]
  int desc_pitch;
  int* dp_work;        // [slots][dp_work_stride] work area of the multi-workgroup keypoint preparation (brisk_dp_work_ints)
  long dp_work_stride;
};
long brisk_dp_work_ints(int kp_cap);

// Optional per-stage timing with HIP events on the launch stream (bench.py roofline leg).
#define BRISK_PROF_STAGES 9
#define BRISK_PROF_MAX_CALLS 64
struct BriskProfiler {
  bool on = false;
  int calls = 0;                                                   // calls recorded since the last reset
  hipEvent_t ev[BRISK_PROF_MAX_CALLS][BRISK_PROF_STAGES + 1] = {};  // created lazily
  bool used[BRISK_PROF_MAX_CALLS][BRISK_PROF_STAGES + 1] = {};
  // the integral image kernel of a detect + describe batch runs on the side stream: its own event pair
  hipEvent_t side_ev[BRISK_PROF_MAX_CALLS][2] = {};
  bool side_used[BRISK_PROF_MAX_CALLS] = {};
  bool created = false;
};
// stage ids
enum { BRISK_STG_PYRAMID = 0, BRISK_STG_DETECT, BRISK_STG_CLASSIFY, BRISK_STG_TIES, BRISK_STG_FINALIZE,
       BRISK_STG_POSTFILTER /* uniformity enforcement / bucketing (empty interval when both are off) */,
       BRISK_STG_INTEGRAL, BRISK_STG_DESC_PREPARE, BRISK_STG_DESCRIBE };
const char* brisk_stage_name(int i);
void brisk_prof_begin_call(BriskProfiler* P);
void brisk_prof_mark(BriskProfiler* P, int slot, hipStream_t s);  // slot k = start of stage k (k == stages: end)
void brisk_prof_mark_side(BriskProfiler* P, int which, hipStream_t side);  // 0 / 1 = before / after the side-stream kernel
void brisk_prof_destroy(BriskProfiler* P);

// detect + describe in one batch: the integral image runs on `side` beside the detector's latency-bound tail
struct BriskOverlap {
  hipStream_t side;
  hipEvent_t fork, join;
  const BriskDescribeBuffers* Dd;
};
int brisk_device_cus();  // compute units of the current device (cached)
// frames: u8 images, frame f at frames + f*frame_pitch, row pitch row_pitch (device memory)
void brisk_launch_detect(const BriskGeom& G, const BriskTileTable& T, const BriskDetectBuffers& B, int nframes,
                         const uint8_t* frames, long frame_pitch, int row_pitch, const uint8_t* mask,
                         long mask_frame_pitch, int mask_row_pitch, hipStream_t s, BriskProfiler* prof,
                         const BriskOverlap* ov = nullptr);
// zeroes what the previous batch (geometry Gprev, nframes frames) left in the score-state map
void brisk_launch_smap_clear(const BriskGeom& Gprev, const BriskDetectBuffers& B, int nframes, hipStream_t s);
// ComputeScale: pyramid + the provided-keypoint walk for one frame (d_in: n_in keypoints in device memory)
void brisk_launch_compute_scale(const BriskGeom& G, const BriskDetectBuffers& B, const uint8_t* frame, int row_pitch,
                                const BriskKeyPoint* d_in, int n_in, int suppress, hipStream_t s);
// only stages layer 0 (descriptor-only calls)
void brisk_launch_layer0_only(const BriskGeom& G, const BriskDetectBuffers& B, int nframes, const uint8_t* frames,
                              long frame_pitch, int row_pitch, hipStream_t s);
// integral image of layer 0 from the band sums the pyramid kernel left (brisk_kernels.hip)
void brisk_launch_integral(const BriskGeom& G, const uint8_t* pyr, const uint32_t* bandsum, uint32_t* integral, int istride,
                           long iframe_elems, int band_h, int nframes, hipStream_t s, int ibits = 32,
                           BriskFrameCounters* counters = nullptr);
// sum of the batch's candidate counts -> *host_word (pinned, mapped): candidates in bits 0-39, frames in bits 40-63
void brisk_launch_publish_single(const BriskFrameCounters* counters, const BriskKeyPoint* kps, const uint8_t* desc, int which, int max_kp,
                                 int dev_pitch, int expect, uint8_t* host, unsigned o_cnt, unsigned o_kp, unsigned o_desc, int* done,
                                 unsigned seq, hipStream_t s);
void brisk_launch_batch_density(const BriskFrameCounters* counters, int nframes, int cand_cap, long long* host_word, hipStream_t s);
// kp_in: [slots][kp_cap]; n_in: per-frame counts at byte stride n_in_stride
void brisk_launch_describe(const BriskGeom& G, const BriskPatternDev& P, const BriskDetectBuffers& B,
                           const BriskDescribeBuffers& Dd, int nframes, const BriskKeyPoint* kp_in, const int* n_in,
                           long n_in_stride, hipStream_t s, BriskProfiler* prof, const BriskOverlap* ov = nullptr,
                           int n_in_max = -1 /* largest per-frame count if the host knows it, else -1 */);

// ---- the batch path's exit to host memory (brisk_export.hip) ----
struct BriskExportSlab {  // device memory: what k_export_egress writes to the host, in the host's layout
  int* counts;            // [frames]
  int* flags;             // [frames]
  long long* offsets;     // [frames + 1]
  uint32_t* kps;          // [rows_cap][7]
  uint32_t* desc;         // [rows_cap][desc_stride / 4]
};
// counts / flags / exact prefix sums of the last batch's rows (which: 0 detected, 1 described) and the rows themselves
// -> slab, on the batch's stream
void brisk_launch_export_pack(const BriskFrameCounters* counters, const BriskKeyPoint* kps, const uint8_t* desc, int kp_cap, int dev_pitch,
                              int strings, int nframes, int which, long long rows_cap, int desc_stride, int cut_flag,
                              const BriskExportSlab& S, hipStream_t s);
// slab -> host memory the device can write (pinned / registered): the stored rows only
void brisk_launch_export_egress(const BriskExportSlab& S, int nframes, int desc_stride, int* h_counts, int* h_flags, long long* h_offsets,
                                void* h_kps, void* h_desc, hipStream_t s);

// streaming probe (brisk_hip_stream_ceiling): mode 0 copies `bytes` from a to b with 16-byte loads/stores, mode 1 only reads a
void brisk_launch_stream_probe(const void* a, void* b, size_t bytes, int mode, hipStream_t s);

// ---- Hamming brute-force matcher (brisk_match.hip) ----
struct BriskDMatch {  // binary-identical to cv::DMatch
  int queryIdx, trainIdx, imgIdx;
  float distance;
};
void brisk_launch_match_dist(const uint8_t* query, int q_pitch, int q0, int nqb, const uint8_t* train, int t_pitch, int nt,
                             int words, const uint8_t* mask, long mask_pitch, uint16_t* dist, long dist_pitch,
                             hipStream_t s);
void brisk_launch_match_masked_out(const uint8_t* mask, long mask_pitch, int q0, int nqb, const int* img_start,
                                   const int* has_mask, int nimg, int* masked, hipStream_t s);
void brisk_launch_match_knn(const uint16_t* dist, long dist_pitch, int q0, int nqb, int nt, const int* img_start, int nimg,
                            const int* masked, int k, BriskDMatch* out, int* out_count, hipStream_t s);
void brisk_launch_match_radius(const uint16_t* dist, long dist_pitch, int q0, int nqb, int nt, const int* img_start,
                               int nimg, const int* masked, float max_distance, int cap, BriskDMatch* out, int* out_count,
                               int dim_bytes, hipStream_t s);
// k <= 2, one train set, no masks: fused distance + top-2 kernel; false = not covered, use the matrix path
bool brisk_launch_match_knn_fused(const uint8_t* query, int q_pitch, int nq, const uint8_t* train, int t_pitch, int nt,
                                  int words32, int k, BriskDMatch* out, int* out_count, hipStream_t s);

// ---- uniformity enforcement / keypoint bucketing (brisk_uniformity.hip): optional post-filters of the detector's keypoints ----
void brisk_launch_bucketing(BriskKeyPoint* kp, BriskFrameCounters* counters, int* order, BriskKeyPoint* tmp, int kp_cap, int rows,
                            int cols, int nbu, int nbv, int max_keypoints, int nframes, hipStream_t s);
void brisk_launch_uniformity(BriskKeyPoint* kp, BriskFrameCounters* counters, int* order, BriskKeyPoint* tmp, uint8_t* occ,
                             long occ_frame, int ow, int kp_cap, float scaling, int max_keypoints, int nframes, hipStream_t s);

// ---- 16-bit image functions (brisk_image16.hip); strides in elements ----
void brisk_launch_halfsample16(const uint16_t* src, int sstride, int w, int h, uint16_t* dst, int dstride, hipStream_t s);
void brisk_launch_twothirdsample16(const uint16_t* src, int sstride, int w, int h, uint16_t* dst, int dstride, hipStream_t s);
// rowsum: w x h floats of scratch; out: (h + 1) x (w + 1) floats at row stride ostride
void brisk_launch_integral16(const uint16_t* src, int sstride, int w, int h, float* rowsum, float* out, int ostride, hipStream_t s);
