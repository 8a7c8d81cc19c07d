// brisk_capi.hip - implementation of the C ABI declared in include/brisk_hip.h.
// Host side only: context / workspace management, pyramid geometry, H2D/D2H staging, kernel launches.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sched.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/brisk_hip.h"
#include "../../include/brisk_hip_debug.h"
#include "brisk_common.h"
#include "brisk_kernels.h"
#include "brisk_pattern.h"
#include "brisk_device_describe.h"

static_assert(sizeof(brisk_hip_keypoint) == 28 && sizeof(BriskKeyPoint) == 28, "cv::KeyPoint layout");

struct brisk_hip_pattern {
  int device;  // device the tables live on (usable from every context of that device)
  int true_device;  // (device can be forged by a test: brisk_hip_debug_forge_pattern_device)
  BriskPatternHost host;
  BriskPatternDev dev;  // device pointers
  void* blob;           // single device allocation backing dev.*
};

struct brisk_hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;
  std::string err;
  int cand_cap = 65536, kp_cap = 16384, tie_cap = 8192;
  int desc_pitch = 64;  // bytes per descriptor row on the device: grows with the longest descriptor a pattern of this context produced (generateKernel at pattern scales < 1: up to 224 bytes)
  // current allocation
  int slots = 0;
  long pyr_elems_alloc = 0;
  long iframe_elems_alloc = 0;
  BriskDetectBuffers B{};
  BriskDescribeBuffers D{};
  uint8_t* d_stage = nullptr;  // staging for host-buffer calls (image + mask)
  size_t stage_bytes = 0;
  BriskKeyPoint* d_kp_in = nullptr;  // host-provided keypoints (describe-only)
  int* d_n_in = nullptr;
  // last geometry
  BriskGeom G{};
  BriskTileTable T{};
  int last_nframes = 0;
  bool last_has_desc = false;
  int last_desc_pitch = 0;  // device pitch of the descriptor rows the last call wrote (a host describe call packs them at the caller's pitch)
  const uint8_t* last_l0_ext = nullptr;  // layer 0 of the last batch was read in place from the caller's frames (debug_layer)
  long last_l0_pitch = 0;
  BriskProfiler prof;
  int debug_flags = 0;
  // sub-batch streams: a batch is split into nsub slices that run on their own streams, so that the latency-bound
  // kernels of one slice (tie resolution: one workgroup per frame) overlap with the throughput-bound kernels of the
  // others instead of leaving most CUs idle
  int nsub = 1;
  hipStream_t sub[8] = {};
  hipEvent_t fork_ev = nullptr, join_ev[8] = {};
  bool sub_created = false;
  int last_frames_per_launch = 0;
  // second stream of a detect + describe batch (integral image beside the detector's tail)
  hipStream_t side = nullptr;
  hipEvent_t side_fork = nullptr, side_join = nullptr;
  int overlap = 1;
  // what the last detect batch left in the score-state map (geometry + frame count), cleared by the next one
  BriskGeom dirtyG{};
  int dirty_frames = 0;
  // optional uniformity enforcement after the detector (brisk_hip_set_uniformity / brisk_hip_detect_uniform)
  double uni_radius = 0.0;
  int uni_max = 0x7FFFFFFF;
  // the other post-filter of the reference (KeyPointBucketing; used when uniformity enforcement is off): 0 buckets = off
  int bk_u = 0, bk_v = 0, bk_max = 0;
  uint8_t* d_occ = nullptr;
  size_t occ_bytes = 0;
  BriskKeyPoint* d_uni_tmp = nullptr;
  int* d_uni_order = nullptr;
  size_t uni_items = 0;
  // candidate density of the last detect + describe batch (k_batch_density writes it into pinned host memory; read without
  // synchronisation by the next batch: integral_format)
  long long* h_density = nullptr;
  int integral_fmt = 0;  // brisk_hip_set_integral_format: 0 auto, 24, 32
  // results of a one-frame host-buffer call land here (pinned) behind the kernels: one wait per call (download_single)
  uint8_t* h_res = nullptr;
  int* d_pub_done = nullptr;  // k_publish_single: workgroups that have written their share
  unsigned pub_seq = 0;       // sequence word of the last call (the host polls for it)
  int spec_nkp = 1024;        // keypoints the next detect call is expected to return (sizes the publishing kernel's grid)
  double density_mpx = 0.0;  // megapixels per frame of the batch the word belongs to
  void* d_img16[3] = {nullptr, nullptr, nullptr};  // scratch of the 16-bit image functions (source, destination, row sums): grown, never shrunk
  size_t img16_bytes[3] = {0, 0, 0};
  void* d_match = nullptr;  // workspace of brisk_hip_match_knn_device
  size_t match_bytes = 0;
  // Calls share one workspace but may be issued on different streams: every call that uses the workspace first makes
  // its stream wait for the end of the previous one (event recorded at the end of each call).
  hipEvent_t done_ev = nullptr;
  bool done_valid = false;      // some call has queued work on the workspace (on last_stream)
  bool done_recorded = false;   // done_ev has been recorded behind that work (ensure_done_event)
  hipStream_t last_stream = nullptr;
  hipEvent_t block_ev = nullptr;  // blocking-sync event of the one-frame calls (download_single: when polling would starve other threads)
  // host-fed batches: two device staging buffers filled over a copy stream while the previous slice computes
  uint8_t* d_hstage[2] = {nullptr, nullptr};
  size_t hstage_bytes = 0;
  uint8_t* d_imgs = nullptr;  // brisk_hip_detect_images / _describe_images: all frames of a call, resident until the next such call
  size_t imgs_bytes = 0;
  // pageable images of the multi-image calls: copied into these pinned buffers by a few host threads (a pageable hipMemcpy is a
  // staging copy on the CALLING thread: 95 us per 2 MB image), then moved by one DMA per slice
  uint8_t* h_pin[2] = {nullptr, nullptr};
  size_t pin_bytes = 0;
  hipEvent_t pin_ev[2] = {nullptr, nullptr};
  bool pin_used[2] = {false, false};
  uint8_t* h_kin_pin = nullptr;  // brisk_hip_describe_images: the provided keypoint lists, packed (pinned)
  size_t kin_pin_bytes = 0;
  hipEvent_t kin_pin_ev = nullptr;
  bool kin_pin_used = false;
  int image_reuse_multi = 0;  // describe_images calls that took the frames of the last multi-image call from the device
  struct {                    // what d_imgs holds (a describe call that names the same buffers as unchanged skips its uploads)
    int n = 0, w = 0, h = 0, stride = 0;
    std::vector<const uint8_t*> ptrs;
  } imgs;
  hipStream_t copy_stream = nullptr;
  hipEvent_t copied_ev[2] = {nullptr, nullptr}, consumed_ev[2] = {nullptr, nullptr};
  // The image of the last host-buffer detect call is still on the device (staging buffer, layer 0 and the pyramid
  // kernel's 96-row band sums of slot 0).  The reference API forces detect() and compute() to be two calls on the same
  // cv::Mat (test-binary-equal.cc:215,237); a caller who KNOWS that the pixels did not change in between says so
  // (brisk_hip_describe_same_image) and the second upload and layer-0 pass are skipped.  brisk_hip_describe itself always
  // uses the pixels it is given, like the reference's compute() - unless BRISK_HIP_IMAGE_CACHE=1 opts in to recognising
  // the buffer by pointer, size, stride and a 64-bit hash over a 1/16 sample of its pixels (a change that avoids every
  // sampled word goes unnoticed: off by default).
  struct {
    bool valid = false;
    const uint8_t* ptr = nullptr;
    int w = 0, h = 0, stride = 0;
    uint64_t hash = 0;
    const uint8_t* l0_ext = nullptr;  // layer 0 was read in place from the staging buffer (width a multiple of 64)
    int hits = 0;                     // describe calls that reused the device copy (brisk_hip_debug_image_reuse)
  } img_cache;
  // brisk_hip_batch_download_all: two slots of {device slab, pinned bounce buffer, events, the transfer in flight}
  struct ExportSlot {
    void* slab = nullptr;
    size_t slab_bytes = 0;
    uint8_t* bounce = nullptr;  // pinned staging for destinations the device cannot write (pageable memory)
    size_t bounce_bytes = 0;
    hipEvent_t packed = nullptr, done = nullptr;
    bool done_valid = false;  // `done` has been recorded: the slab is in use until it fires
    bool pending = false;     // the transfer has not been completed by a wait yet
    bool use_bounce = false;
    unsigned ticket = 0;
    int nframes = 0;
    brisk_hip_batch_host_results dst{};  // the caller's destinations
    brisk_hip_batch_host_results wr{};   // where k_export_egress writes (the caller's arrays, or the bounce buffer)
    int rc = BRISK_HIP_OK, flagged = 0;
    std::string msg;
  } ex[2];
  hipStream_t egress = nullptr;
  unsigned ex_seq = 0;
  int last_strings = 0;  // descriptor bytes of the pattern the last describing call used
};

// debug bits / environment knobs exist in BRISK_HIP_TUNING builds only (brisk_common.h); the release library sees none
#ifdef BRISK_HIP_TUNING
static inline int dbg_flags(const brisk_hip_ctx* c) { return c->debug_flags; }
static inline const char* tuning_env(const char* name) { return getenv(name); }
#else
static inline int dbg_flags(const brisk_hip_ctx*) { return 0; }
static inline const char* tuning_env(const char*) { return nullptr; }
#endif

// 64-bit hash over one 8-byte word of every 128 bytes of every row (row-dependent phase): ~130 KB of a 1080p frame, a few
// microseconds.  A caller that changes the image between detect() and compute() is noticed unless the change avoids all
// sampled words (BRISK_HIP_IMAGE_CACHE=0 turns the reuse off).
static uint64_t image_sample_hash(const uint8_t* img, int w, int h, int stride) {
  uint64_t hsh = 0x9E3779B97F4A7C15ull ^ ((uint64_t)w << 32) ^ (uint64_t)h;
  for (int y = 0; y < h; ++y) {
    const uint8_t* row = img + (size_t)y * stride;
    for (int x = (y * 24) & 127; x + 8 <= w; x += 128) {
      uint64_t v;
      memcpy(&v, row + x, 8);
      hsh = (hsh ^ v) * 0xFF51AFD7ED558CCDull;
      hsh ^= hsh >> 32;
    }
    if (w < 8) for (int x = 0; x < w; ++x) hsh = (hsh ^ row[x]) * 0x100000001B3ull;
  }
  return hsh;
}
// the sampled-hash recognition of an unchanged buffer: opt-in (a change that misses every sampled word is not seen)
static bool image_hash_reuse_enabled() {
  static const bool on = getenv("BRISK_HIP_IMAGE_CACHE") && atoi(getenv("BRISK_HIP_IMAGE_CACHE")) == 1;
  return on;
}

// The workspace is reused by every call: a call orders its stream behind the previous call's work.  The previous work is known by
// the stream it was queued on; an event behind it is recorded only when somebody needs one - a call on ANOTHER stream, a copy
// stream that must wait for it - and never for the usual case, call after call on the context's own stream (stream order does
// it; round 6: three hipEventRecord + three hipStreamWaitEvent per detect() + compute() pair, each a serialised runtime call).
// Work on a caller's stream gets its event at once: the stream may not exist any more when the next call comes.
static int ensure_done_event(brisk_hip_ctx* c) {
  if (!c->done_valid || c->done_recorded) return BRISK_HIP_OK;
  if (!c->done_ev && hipEventCreateWithFlags(&c->done_ev, hipEventDisableTiming) != hipSuccess) return BRISK_HIP_ERR_HIP;
  if (hipEventRecord(c->done_ev, c->last_stream) != hipSuccess) return BRISK_HIP_ERR_HIP;
  c->done_recorded = true;
  return BRISK_HIP_OK;
}
static int workspace_acquire(brisk_hip_ctx* c, hipStream_t s) {
  if (!c->done_valid || s == c->last_stream) return BRISK_HIP_OK;
  if (ensure_done_event(c) || hipStreamWaitEvent(s, c->done_ev, 0) != hipSuccess) return BRISK_HIP_ERR_HIP;
  return BRISK_HIP_OK;
}
static int workspace_release(brisk_hip_ctx* c, hipStream_t s) {
  c->last_stream = s;
  c->done_valid = true;
  c->done_recorded = false;
  return s == c->stream ? BRISK_HIP_OK : ensure_done_event(c);
}

// Every exit after the workspace was acquired must leave done_ev behind the work already queued (kernels of earlier
// slices, the side-stream integral, the score-state map clear): the next call, possibly on another stream, orders itself
// behind that event.  The guard records it on the error paths too.
struct WorkspaceGuard {
  brisk_hip_ctx* c;
  hipStream_t s;
  bool armed;
  WorkspaceGuard(brisk_hip_ctx* c_, hipStream_t s_) : c(c_), s(s_), armed(true) {}
  ~WorkspaceGuard() { if (armed) (void)workspace_release(c, s); }
  int release() { armed = false; return workspace_release(c, s); }
};
// waits for THIS context's queued work only (other contexts - one per host thread in the C++ classes - keep running)
static hipError_t wait_own_work(brisk_hip_ctx* c) {
  if (!c->done_valid) return hipSuccess;
  return c->done_recorded ? hipEventSynchronize(c->done_ev) : hipStreamSynchronize(c->last_stream);
}

// ---- what brisk_comm.hip needs from a context (not part of the C ABI) -----------------------------------------------
int brisk_hip_internal_fail(brisk_hip_ctx* ctx, int code, const char* msg) {
  if (ctx) { std::lock_guard<std::mutex> lk(ctx->mu); ctx->err = msg; }
  return code;
}
// device pointers / geometry of the last batch's described results (null pointers when no described batch exists)
int brisk_hip_internal_batch_view(brisk_hip_ctx* ctx, const BriskFrameCounters** counters, const BriskKeyPoint** dkp,
                                  const uint8_t** desc, int* kp_cap, int* desc_pitch, int* nframes, int* device, hipStream_t* stream) {
  if (!ctx) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const bool have = ctx->last_has_desc && ctx->last_nframes > 0;
  if (counters) *counters = have ? ctx->B.counters : nullptr;
  if (dkp) *dkp = have ? ctx->D.dkp : nullptr;
  if (desc) *desc = have ? ctx->D.desc : nullptr;
  if (kp_cap) *kp_cap = ctx->B.kp_cap;
  if (desc_pitch) *desc_pitch = ctx->last_desc_pitch ? ctx->last_desc_pitch : ctx->D.desc_pitch;
  if (nframes) *nframes = ctx->last_nframes;
  if (device) *device = ctx->device;
  if (stream) *stream = ctx->stream;
  return BRISK_HIP_OK;
}

#define HIPCHK(ctx, call)                                                                       \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                           \
      return BRISK_HIP_ERR_HIP;                                                                 \
    }                                                                                           \
  } while (0)

static int fail(brisk_hip_ctx* ctx, int code, const char* msg) {
  if (ctx) ctx->err = msg;
  return code;
}

// Pyramid geometry: brisk/src/brisk-scale-space.cc:54-90, brisk/src/brisk-layer.cc:72-95
static void make_geometry(int w, int h, int threshold, int octaves, BriskGeom* G, BriskTileTable* T) {
  memset(G, 0, sizeof(*G));
  G->nlayers = (octaves == 0) ? 1 : 2 * octaves;
  G->single_layer = (octaves == 0);
  G->w = w;
  G->h = h;
  G->threshold = threshold;
  G->lower_threshold = BRISK_LOWER_THRESHOLD;
  int off = 0;
  for (int l = 0; l < G->nlayers; ++l) {
    BriskLayerGeom& L = G->L[l];
    if (l == 0) {
      L.w = w; L.h = h; L.scale = 1.0f; L.offset = 0.0f;
    } else if (l == 1) {
      L.w = 2 * (G->L[0].w / 3); L.h = 2 * (G->L[0].h / 3);
      L.scale = (float)(G->L[0].scale * 1.5);
      L.offset = (float)(0.5 * L.scale - 0.5);
    } else {
      L.w = G->L[l - 2].w / 2; L.h = G->L[l - 2].h / 2;
      L.scale = G->L[l - 2].scale * 2;
      L.offset = (float)(0.5 * L.scale - 0.5);
    }
    L.stride = brisk_align_up(L.w > 0 ? L.w : 1, BRISK_STRIDE_ALIGN);
    L.off = off;
    off += brisk_align_up(L.stride * (L.h > 0 ? L.h : 1), 256);
  }
  G->pyr_elems = off;
  int t = 0;
  for (int l = 0; l < G->nlayers; ++l) {
    T->first_tile[l] = t;
    T->tiles_x[l] = (G->L[l].stride + BRISK_DETECT_TILE_W - 1) / BRISK_DETECT_TILE_W;
    const int ty = (G->L[l].h + BRISK_DETECT_TILE_H - 1) / BRISK_DETECT_TILE_H;
    t += T->tiles_x[l] * ty;
  }
  for (int l = G->nlayers; l <= BRISK_MAX_LAYERS; ++l) T->first_tile[l] = t;
  T->total_tiles = t;
}

static void free_buffers(brisk_hip_ctx* c) {
  hipFree(c->B.pyr); hipFree(c->B.smap); hipFree(c->B.cand); hipFree(c->B.blocks); hipFree(c->B.tie_idx); hipFree(c->B.keys);
  hipFree(c->B.counters); hipFree(c->B.kp_out); hipFree(c->D.integral); hipFree(c->B.bandsum); hipFree(c->D.dkp); hipFree(c->D.dscale); hipFree(c->D.dperm); hipFree(c->D.drec); hipFree(c->D.dp_work);
  hipFree(c->D.desc); hipFree(c->d_kp_in); hipFree(c->d_n_in);
  c->B = BriskDetectBuffers{};
  c->D = BriskDescribeBuffers{};
  c->d_kp_in = nullptr; c->d_n_in = nullptr;
  c->slots = 0; c->pyr_elems_alloc = 0; c->iframe_elems_alloc = 0;
}

static int ensure_buffers(brisk_hip_ctx* c, int nframes, const BriskGeom& G) {
  const int istride = brisk_align_up(G.w + 1, 16);
  const long iframe = (long)istride * (G.h + 1);
  if (nframes <= c->slots && G.pyr_elems <= c->pyr_elems_alloc && iframe <= c->iframe_elems_alloc &&
      c->B.cand_cap == c->cand_cap && c->B.kp_cap == c->kp_cap && c->D.desc_pitch >= c->desc_pitch) {
    c->D.istride = istride;
    c->B.istride = istride;
    c->D.iframe_elems = iframe;
    c->D.ibits = 32;
    return BRISK_HIP_OK;
  }
  HIPCHK(c, hipDeviceSynchronize());
  c->img_cache.valid = false;
  const int slots = nframes > c->slots ? nframes : c->slots;
  const long pyr = G.pyr_elems > c->pyr_elems_alloc ? G.pyr_elems : c->pyr_elems_alloc;
  const long ifr = iframe > c->iframe_elems_alloc ? iframe : c->iframe_elems_alloc;
  free_buffers(c);
  c->B.cand_cap = c->cand_cap; c->B.kp_cap = c->kp_cap; c->B.tie_cap = c->tie_cap;
  c->B.band_h = 96;  // band height of the detector's pyramid kernel (the descriptor-only call uses its own)
  HIPCHK(c, hipMalloc(&c->B.pyr, (size_t)slots * pyr + 256));
  HIPCHK(c, hipMalloc(&c->B.smap, ((size_t)slots * pyr + 256) * sizeof(uint16_t)));
  HIPCHK(c, hipMalloc(&c->B.cand, (size_t)slots * c->cand_cap * sizeof(BriskCand)));
  HIPCHK(c, hipMalloc(&c->B.blocks, (size_t)slots * c->cand_cap * 64));
  HIPCHK(c, hipMalloc(&c->B.tie_idx, (size_t)slots * BRISK_MAX_LAYERS * c->tie_cap * sizeof(int)));
  HIPCHK(c, hipMalloc(&c->B.keys, (size_t)slots * c->cand_cap * 2 * sizeof(unsigned)));
  HIPCHK(c, hipMalloc(&c->B.counters, (size_t)slots * sizeof(BriskFrameCounters)));
  HIPCHK(c, hipMalloc(&c->B.kp_out, (size_t)slots * c->kp_cap * sizeof(BriskKeyPoint)));
  HIPCHK(c, hipMalloc(&c->D.integral, (size_t)slots * ifr * sizeof(uint32_t)));
  HIPCHK(c, hipMalloc(&c->B.bandsum, (size_t)slots * ((ifr / 64) + 4 * 8192 + 64) * sizeof(uint32_t)));
  HIPCHK(c, hipMalloc(&c->D.dkp, (size_t)slots * c->kp_cap * sizeof(BriskKeyPoint)));
  HIPCHK(c, hipMalloc(&c->D.dscale, (size_t)slots * c->kp_cap * sizeof(int)));
  HIPCHK(c, hipMalloc(&c->D.dperm, (size_t)slots * c->kp_cap * sizeof(int)));
  HIPCHK(c, hipMalloc(&c->D.drec, ((size_t)slots * c->kp_cap + 4) * sizeof(uint4)));
  c->D.dp_work_stride = brisk_dp_work_ints(c->kp_cap);
  HIPCHK(c, hipMalloc(&c->D.dp_work, (size_t)slots * c->D.dp_work_stride * sizeof(int)));
  c->D.desc_pitch = c->desc_pitch;
  HIPCHK(c, hipMalloc(&c->D.desc, (size_t)slots * c->kp_cap * c->D.desc_pitch));
  HIPCHK(c, hipMalloc(&c->d_kp_in, (size_t)slots * c->kp_cap * sizeof(BriskKeyPoint)));
  HIPCHK(c, hipMalloc(&c->d_n_in, (size_t)slots * sizeof(int)));
  HIPCHK(c, hipMemset(c->B.pyr, 0, (size_t)slots * pyr + 256));
  // the score-state map is kept all-zero between batches: k_detect writes detections only, the next detect batch
  // first clears what the previous one left (k_smap_clear)
  HIPCHK(c, hipMemset(c->B.smap, 0, ((size_t)slots * pyr + 256) * sizeof(uint16_t)));
  // hipMemset on device memory runs on the null stream and may return before it is done; the engine's streams are
  // non-blocking (not ordered against the null stream): without this wait the first batch on fresh buffers could start
  // writing detections into a map that is still being zeroed (seen once the pyramid kernel got fast enough)
  HIPCHK(c, hipDeviceSynchronize());
  c->dirty_frames = 0;
  c->slots = slots; c->pyr_elems_alloc = pyr; c->iframe_elems_alloc = ifr;
  c->D.istride = istride;
  c->B.istride = istride;
  c->D.iframe_elems = iframe;
  c->D.ibits = 32;
  return BRISK_HIP_OK;
}

static int ensure_stage(brisk_hip_ctx* c, size_t bytes) {
  if (bytes <= c->stage_bytes) return BRISK_HIP_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->img_cache.valid = false;
  hipFree(c->d_stage);
  c->d_stage = nullptr;
  c->stage_bytes = 0;
  HIPCHK(c, hipMalloc(&c->d_stage, bytes + 256));
  c->stage_bytes = bytes;
  return BRISK_HIP_OK;
}

// Element size of a call's integral image: 32 bits, or 24 (3-byte elements, values modulo 2^24; where the pattern's boxes
// allow it: BriskPatternDev::int24_ok).  The integral image is written and fetched once per frame; a quarter fewer bytes
// shorten the window in which k_integral_final runs beside the tie chain (0.70 -> 0.63 ms per 256 frames; bench line + 1.4 %),
// but k_describe itself is 2 - 10 % slower on 3-byte elements (byte-unaligned gathers straddle more sectors, one v_alignbit
// per column): sparse frames gain, dense frames and descriptor-only calls with many keypoints lose (threshold 30: - 11 %,
// config 5: 0.27 vs 0.31 ms).  So: 32 bits in descriptor-only calls; in detect + describe batches 24 bits while the
// PREVIOUS batch of the context had at most 3 000 AGAST candidates per megapixel (BASELINE configs 2 and 4: 1 100; threshold
// 50: 12 000) - a stream's batches look alike, the count comes back through pinned memory without a synchronisation, and it
// only steers speed.  (Both forms in ONE k_describe, chosen per frame on the device, cost more registers - 224 instead of
// 195 - than the smaller image saves.)  Debug bit 18 forces 32, bit 24 forces 24 (stage parity tests of both forms);
// BRISK_INTEGRAL_BITS=32 / 24 for A / B runs.
static void integral_format(const brisk_hip_ctx* ctx, const brisk_hip_pattern* pat, bool batch_with_detect, int* ibits) {
  static const int env = tuning_env("BRISK_INTEGRAL_BITS") ? atoi(tuning_env("BRISK_INTEGRAL_BITS")) : 0;
  *ibits = 32;
  if (!pat || !pat->dev.int24_ok) return;
  if (env == 32 || (dbg_flags(ctx) & (1 << 18))) return;
  if (env == 24 || (dbg_flags(ctx) & (1 << 24))) { *ibits = 24; return; }
  if (ctx->integral_fmt == BRISK_HIP_INTEGRAL_U32) return;                     // brisk_hip_set_integral_format
  if (ctx->integral_fmt == BRISK_HIP_INTEGRAL_U24) { *ibits = 24; return; }
  if (!batch_with_detect) return;
  double density = 0.0;  // candidates per megapixel of the last batch (none yet: sparse is the common case)
  if (ctx->h_density && ctx->density_mpx > 0.0) {
    const long long wv = __atomic_load_n(ctx->h_density, __ATOMIC_RELAXED);
    const long long frames = wv >> 40, cands = wv & 0xFFFFFFFFFFll;
    if (frames > 0) density = (double)cands / ((double)frames * ctx->density_mpx);
  }
  if (density <= 3000.0) *ibits = 24;
}
// a pattern's tables live on ONE device: a context of another device would dereference that device's memory from its
// kernels (a fault, or silent peer reads on every sample) - the mistake a one-thread-per-GPU host makes first
static int check_pattern_device(brisk_hip_ctx* ctx, const brisk_hip_pattern* pat) {
  if (pat && pat->device != ctx->device) {
    char msg[160];
    snprintf(msg, sizeof msg, "pattern handle was created on device %d, the context runs on device %d: create one extractor per GPU",
             pat->device, ctx->device);
    return fail(ctx, BRISK_HIP_ERR_ARG, msg);
  }
  return BRISK_HIP_OK;
}

// host image -> device staging (rows at `pitch`): one linear copy when neither side has row padding (a pitched copy is
// issued row by row: 1080 short rows take several times as long as 2 MB in one piece, profiles/r04_microbench_copy.json)
static hipError_t upload_rows(uint8_t* dst, int pitch, const uint8_t* src, int stride, int w, int h, hipStream_t s) {
  if (pitch == w && stride == w) return hipMemcpyAsync(dst, src, (size_t)w * h, hipMemcpyHostToDevice, s);
  return hipMemcpy2DAsync(dst, pitch, src, stride, w, h, hipMemcpyHostToDevice, s);
}

static int check_detect_args(brisk_hip_ctx* ctx, int w, int h, int threshold, int octaves) {
  if (w <= 0 || h <= 0 || w > 8191 || h > 8191) return fail(ctx, BRISK_HIP_ERR_ARG, "image size must be in [1, 8191]");
  if (octaves < 0 || 2 * octaves > BRISK_MAX_LAYERS) return fail(ctx, BRISK_HIP_ERR_ARG, "octaves must be in [0, 8]");
  if (threshold < 1 || threshold > 255)
    return fail(ctx, BRISK_HIP_ERR_THRESHOLD, "AGAST threshold must be in [1, 255]");
  return BRISK_HIP_OK;
}

static int usable_cpus();

extern "C" {

int brisk_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int brisk_hip_usable_cpus(void) { return usable_cpus(); }

int brisk_hip_host_register(void* ptr, size_t bytes) {
  if (!ptr || !bytes) return BRISK_HIP_ERR_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return BRISK_HIP_ERR_NO_DEVICE;
  const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
  if (e == hipSuccess) return BRISK_HIP_OK;
  (void)hipGetLastError();
  return e == hipErrorHostMemoryAlreadyRegistered ? BRISK_HIP_ERR_ARG : BRISK_HIP_ERR_HIP;
}

int brisk_hip_host_unregister(void* ptr) {
  if (!ptr) return BRISK_HIP_ERR_ARG;
  if (hipHostUnregister(ptr) == hipSuccess) return BRISK_HIP_OK;
  (void)hipGetLastError();
  return BRISK_HIP_ERR_ARG;
}

int brisk_hip_create(int device, brisk_hip_ctx** out) {
  if (!out) return BRISK_HIP_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return BRISK_HIP_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return BRISK_HIP_ERR_NO_DEVICE;
  brisk_hip_ctx* c = new brisk_hip_ctx();
  c->device = device;
  c->B.band_h = 96;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return BRISK_HIP_ERR_HIP;
  }
  *out = c;
  return BRISK_HIP_OK;
}

void brisk_hip_destroy(brisk_hip_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  free_buffers(c);
  hipFree(c->d_stage);
  if (c->d_match) hipFree(c->d_match);
  for (int i = 0; i < 3; ++i) if (c->d_img16[i]) hipFree(c->d_img16[i]);
  if (c->h_density) hipHostFree(c->h_density);
  if (c->h_res) hipHostFree(c->h_res);
  if (c->d_pub_done) hipFree(c->d_pub_done);
  if (c->done_ev) hipEventDestroy(c->done_ev);
  if (c->block_ev) hipEventDestroy(c->block_ev);
  for (int i = 0; i < 2; ++i) {
    if (c->d_hstage[i]) hipFree(c->d_hstage[i]);
    if (c->copied_ev[i]) hipEventDestroy(c->copied_ev[i]);
    if (c->consumed_ev[i]) hipEventDestroy(c->consumed_ev[i]);
  }
  if (c->d_imgs) hipFree(c->d_imgs);
  if (c->h_kin_pin) hipHostFree(c->h_kin_pin);
  if (c->kin_pin_ev) hipEventDestroy(c->kin_pin_ev);
  for (int i = 0; i < 2; ++i) {
    if (c->h_pin[i]) hipHostFree(c->h_pin[i]);
    if (c->pin_ev[i]) hipEventDestroy(c->pin_ev[i]);
  }
  if (c->copy_stream) hipStreamDestroy(c->copy_stream);
  for (auto& E : c->ex) {
    if (E.slab) hipFree(E.slab);
    if (E.bounce) hipHostFree(E.bounce);
    if (E.packed) hipEventDestroy(E.packed);
    if (E.done) hipEventDestroy(E.done);
  }
  if (c->egress) hipStreamDestroy(c->egress);
  if (c->d_occ) hipFree(c->d_occ);
  if (c->d_uni_tmp) hipFree(c->d_uni_tmp);
  if (c->d_uni_order) hipFree(c->d_uni_order);
  brisk_prof_destroy(&c->prof);
  if (c->side) { hipStreamDestroy(c->side); hipEventDestroy(c->side_fork); hipEventDestroy(c->side_join); }
  if (c->sub_created) {
    for (int i = 0; i < 8; ++i) { hipStreamDestroy(c->sub[i]); hipEventDestroy(c->join_ev[i]); }
    hipEventDestroy(c->fork_ev);
  }
  hipStreamDestroy(c->stream);
  delete c;
}

const char* brisk_hip_last_error(const brisk_hip_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int brisk_hip_set_capacity(brisk_hip_ctx* ctx, int max_candidates, int max_keypoints) {
  if (!ctx) return BRISK_HIP_ERR_ARG;
  if (max_candidates < 256 || max_keypoints < 16) return fail(ctx, BRISK_HIP_ERR_ARG, "capacity too small");
  // (k_describe's keypoint records carry the keypoint index in 23 bits beside the scale index)
  if (max_keypoints >= (1 << 23)) return fail(ctx, BRISK_HIP_ERR_ARG, "keypoint capacity must be below 2^23");
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->cand_cap = max_candidates;
  ctx->kp_cap = max_keypoints;
  ctx->tie_cap = max_candidates / 4 > 1024 ? max_candidates / 4 : 1024;
  return BRISK_HIP_OK;
}

int brisk_hip_reserve(brisk_hip_ctx* ctx, int min_candidates, int min_keypoints) {
  if (!ctx) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (min_keypoints >= (1 << 23)) return fail(ctx, BRISK_HIP_ERR_ARG, "keypoint capacity must be below 2^23");
  if (min_candidates > ctx->cand_cap) {
    ctx->cand_cap = min_candidates;
    if (min_candidates / 4 > ctx->tie_cap) ctx->tie_cap = min_candidates / 4;
  }
  if (min_keypoints > ctx->kp_cap) ctx->kp_cap = min_keypoints;
  return BRISK_HIP_OK;
}

// ---- pattern -----------------------------------------------------------------------------------
static int upload_pattern(brisk_hip_ctx* ctx, brisk_hip_pattern* p) {
  const BriskPatternHost& H = p->host;
  const size_t n = (size_t)H.npoints;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t o_mult = take(64 * n * 4), o_sigma = take(64 * n * 4), o_uv = take((size_t)BRISK_NROT * n * 16);
  const size_t o_scl = take(64 * n * 8), o_tab = take(64 * n * 16);
  const size_t o_thr = take(64 * 4), o_size = take(64 * 4), o_sp = take((size_t)H.nshort * 4), o_lp = take((size_t)H.nlong * 16);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMalloc(&p->blob, off));
  char* b = (char*)p->blob;
  HIPCHK(ctx, hipMemcpy(b + o_mult, H.mult.data(), 64 * n * 4, hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemcpy(b + o_sigma, H.sigma.data(), 64 * n * 4, hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemcpy(b + o_scl, H.scaling.data(), 64 * n * 8, hipMemcpyHostToDevice));
  {
    // {mult, sigma, scaling | shift << 24 | plain << 30, magic multiplier of scaling2 (plain: scaling2 itself)}
    std::vector<int> tab(64 * n * 4);
    for (size_t i = 0; i < 64 * n; ++i) {
      memcpy(&tab[4 * i], &H.mult[i], 4);
      memcpy(&tab[4 * i + 1], &H.sigma[i], 4);
      brisk_pack_tab(H.scaling[2 * i], H.scaling[2 * i + 1], &tab[4 * i + 2], &tab[4 * i + 3]);
    }
    HIPCHK(ctx, hipMemcpy(b + o_tab, tab.data(), 64 * n * 16, hipMemcpyHostToDevice));
  }
  HIPCHK(ctx, hipMemcpy(b + o_uv, H.uv.data(), (size_t)BRISK_NROT * n * 16, hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemcpy(b + o_thr, H.size_thresh.data(), 64 * 4, hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemcpy(b + o_size, H.size_list.data(), 64 * 4, hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemcpy(b + o_sp, H.short_pairs.data(), (size_t)H.nshort * 4, hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemcpy(b + o_lp, H.long_pairs.data(), (size_t)H.nlong * 16, hipMemcpyHostToDevice));
  BriskPatternDev& d = p->dev;
  d.npoints = H.npoints; d.nshort = H.nshort; d.nlong = H.nlong; d.strings = H.strings;
  d.rotation_invariant = 1; d.scale_invariant = 1; d.basicscale = H.basicscale;
  d.has_bilinear = 0;
  d.reg_tables = (H.nlong <= 896 && H.nshort <= 512 && H.npoints <= 128) ? 1 : 0;
  for (int i = 0; i < H.nlong; ++i)
    if (H.long_pairs[4 * i + 2] < -32768 || H.long_pairs[4 * i + 2] > 32767 || H.long_pairs[4 * i + 3] < -32768 || H.long_pairs[4 * i + 3] > 32767) d.reg_tables = 0;
  float sigma_max = 0.f;
  for (float sg : H.sigma) { d.has_bilinear |= (sg < 0.5f) ? 1 : 0; sigma_max = sg > sigma_max ? sg : sigma_max; }
  // 24-bit integral image: a box spans at most 2 sigma + 3 pixels a side (sigma = the half side SmoothedIntensity works with),
  // every region sum it takes must stay below 2^24: (2 sigma + 3)^2 * 255 < 2^24  <=>  sigma < 126.7
  d.int24_ok = (!d.has_bilinear && sigma_max < 120.0f) ? 1 : 0;
  d.mult = (const float*)(b + o_mult); d.sigma = (const float*)(b + o_sigma); d.uv = (const double*)(b + o_uv);
  d.scaling = (const int*)(b + o_scl);
  d.tab = (const int*)(b + o_tab);
  d.size_thresh = (const float*)(b + o_thr); d.size_list = (const int*)(b + o_size);
  d.short_pairs = (const uint16_t*)(b + o_sp); d.long_pairs = (const int*)(b + o_lp);
  return BRISK_HIP_OK;
}

static int pattern_finish(brisk_hip_ctx* ctx, brisk_hip_pattern* p, bool ok, const std::string& err,
                          brisk_hip_pattern** out) {
  if (!ok) {
    ctx->err = err;
    delete p;
    return BRISK_HIP_ERR_PATTERN;
  }
  p->device = p->true_device = ctx->device;
  p->blob = nullptr;
  const int rc = upload_pattern(ctx, p);
  if (rc != BRISK_HIP_OK) {
    if (p->blob) hipFree(p->blob);
    delete p;
    return rc;
  }
  *out = p;
  return BRISK_HIP_OK;
}

int brisk_hip_pattern_create(brisk_hip_ctx* ctx, int version, float pattern_scale, brisk_hip_pattern** out) {
  if (!ctx || !out) return BRISK_HIP_ERR_ARG;
  *out = nullptr;
  std::lock_guard<std::mutex> lk(ctx->mu);
  brisk_hip_pattern* p = new brisk_hip_pattern();
  std::string err;
  const bool ok = brisk_pattern_build_default(version, pattern_scale, &p->host, &err);
  return pattern_finish(ctx, p, ok, err, out);
}

int brisk_hip_pattern_create_from_text(brisk_hip_ctx* ctx, const char* ptn_text, float pattern_scale,
                                       brisk_hip_pattern** out) {
  if (!ctx || !out || !ptn_text) return BRISK_HIP_ERR_ARG;
  *out = nullptr;
  std::lock_guard<std::mutex> lk(ctx->mu);
  brisk_hip_pattern* p = new brisk_hip_pattern();
  std::string err;
  const bool ok = brisk_pattern_build_from_text(ptn_text, pattern_scale, &p->host, &err);
  return pattern_finish(ctx, p, ok, err, out);
}

void brisk_hip_pattern_destroy(brisk_hip_pattern* p) {
  if (!p) return;
  hipSetDevice(p->true_device);  // (p->device may be forged by a test: only the guard comparison reads it)
  if (p->blob) hipFree(p->blob);
  delete p;
}

int brisk_hip_pattern_descriptor_size(const brisk_hip_pattern* p) { return p ? p->host.strings : 0; }
int brisk_hip_pattern_points(const brisk_hip_pattern* p) { return p ? p->host.npoints : 0; }

int brisk_hip_pattern_tables(const brisk_hip_pattern* p, float* scale_list, int* size_list, float* size_thresh) {
  if (!p) return BRISK_HIP_ERR_ARG;
  if (scale_list) memcpy(scale_list, p->host.scale_list.data(), 64 * 4);
  if (size_list) memcpy(size_list, p->host.size_list.data(), 64 * 4);
  if (size_thresh) memcpy(size_thresh, p->host.size_thresh.data(), 64 * 4);
  return BRISK_HIP_OK;
}

// ---- batch (device-resident) path ------------------------------------------------------------------
struct BatchArgs {
  const brisk_hip_pattern* pat;
  int w, h, threshold, octaves;
  long frame_pitch;
  int row_pitch;
  const uint8_t* d_mask;
  long mask_frame_pitch;
  int mask_row_pitch;
  bool do_detect, do_describe;
  double uni_radius;  // uniformity enforcement of this call (0 = off)
  int uni_max;
  int bk_u = 0, bk_v = 0, bk_max = 0;  // KeyPointBucketing of this call (0 buckets = off)
  bool no_scale_nms = false;  // suppressScaleNonmaxima == false with octaves > 0
  int lower_threshold = BRISK_LOWER_THRESHOLD;  // 0: ComputeScale's pyramid (brisk-feature-detector.cc:90)
  bool format_as_full_batch = false;  // timing experiments (debug bit 27): the descriptor half alone, on the integral format the whole batch would use
  bool inplace_ok = true;  // layer 0 may be read from the frame buffer (not for the host-fed path's recycled staging buffers)
};

// scratch of the post-filters (uniformity enforcement: occupancy images for the frames beyond the on-chip capacity; both:
// order / keypoint scratch per slot)
static int ensure_filter_buffers(brisk_hip_ctx* ctx, int w, int h, int nframes, double uni_radius) {
  size_t need = 0;
  if (uni_radius > 0.0) {
    const float scaling = (float)(15.0 / (float)uni_radius);
    const int oh = (int)(h * ceil(scaling) + 32), ow = (int)(w * ceil(scaling) + 32);
    need = (size_t)(((long)oh * ow + 255) / 256 * 256) * nframes + 64;
  }
  const size_t items = (size_t)ctx->slots * ctx->kp_cap;
  if (need > ctx->occ_bytes || items > ctx->uni_items) {
    HIPCHK(ctx, hipDeviceSynchronize());
    if (need > ctx->occ_bytes) {
      if (ctx->d_occ) (void)hipFree(ctx->d_occ);
      ctx->d_occ = nullptr; ctx->occ_bytes = 0;
      HIPCHK(ctx, hipMalloc(&ctx->d_occ, need));
      ctx->occ_bytes = need;
    }
    if (items > ctx->uni_items) {
      if (ctx->d_uni_tmp) (void)hipFree(ctx->d_uni_tmp);
      if (ctx->d_uni_order) (void)hipFree(ctx->d_uni_order);
      ctx->d_uni_tmp = nullptr; ctx->d_uni_order = nullptr; ctx->uni_items = 0;
      HIPCHK(ctx, hipMalloc(&ctx->d_uni_tmp, items * sizeof(BriskKeyPoint)));
      HIPCHK(ctx, hipMalloc(&ctx->d_uni_order, items * sizeof(int)));
      ctx->uni_items = items;
    }
  }
  return BRISK_HIP_OK;
}

// geometry, workspace, restoring the all-zero score-state map, profiler bookkeeping: once per batch, on stream s
static int batch_begin(brisk_hip_ctx* ctx, const BatchArgs& A, int nframes, hipStream_t s) {
  int rc = check_detect_args(ctx, A.w, A.h, A.threshold, A.octaves);
  if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  make_geometry(A.w, A.h, A.threshold, A.octaves, &ctx->G, &ctx->T);
  ctx->G.debug_flags = dbg_flags(ctx);
  ctx->G.no_scale_nms = (A.no_scale_nms && A.octaves != 0) ? 1 : 0;
  ctx->G.lower_threshold = A.lower_threshold;
  if (A.do_describe && A.pat && A.pat->host.strings > ctx->desc_pitch) ctx->desc_pitch = brisk_align_up(A.pat->host.strings, 16);
  rc = ensure_buffers(ctx, nframes, ctx->G);
  if (rc) return rc;
  integral_format(ctx, A.pat, A.format_as_full_batch || (A.do_detect && A.do_describe), &ctx->D.ibits);
  const bool bucketing = A.do_detect && !(A.uni_radius > 0.0) && A.bk_u > 0;
  if ((A.do_detect && A.uni_radius > 0.0) || bucketing) {
    rc = ensure_filter_buffers(ctx, A.w, A.h, nframes, bucketing ? 0.0 : A.uni_radius);
    if (rc) return rc;
  }
  if (bucketing && (A.bk_u >= A.w || A.bk_v >= A.h))
    return fail(ctx, BRISK_HIP_ERR_ARG, "bucketing: more buckets than pixels (key-point-bucketing-inl.h:82-83)");
  rc = workspace_acquire(ctx, s);
  if (rc) return fail(ctx, rc, "hipStreamWaitEvent failed");
  ctx->img_cache.valid = false;  // slot 0 is rewritten (detect_host sets it again for its own image)
  brisk_prof_begin_call(&ctx->prof);
  if (A.do_detect) {
    if (ctx->dirty_frames > 0) brisk_launch_smap_clear(ctx->dirtyG, ctx->B, ctx->dirty_frames, s);
    ctx->dirtyG = ctx->G;
    ctx->dirty_frames = nframes;
  }
  return BRISK_HIP_OK;
}

// frames [f0, f0 + nf) of the batch: d_frames points at frame f0's image, results go to frame slots f0...
static int batch_slice(brisk_hip_ctx* ctx, const BatchArgs& A, const uint8_t* d_frames, long f0, int nf, hipStream_t si,
                       BriskProfiler* prof, bool overlap_integral) {
  const int nbands = (A.h + ctx->B.band_h - 1) / ctx->B.band_h;
  BriskDetectBuffers Bi = ctx->B;
  Bi.pyr += f0 * ctx->G.pyr_elems;
  Bi.smap += f0 * ctx->G.pyr_elems;
  Bi.cand += f0 * Bi.cand_cap;
  Bi.blocks += f0 * Bi.cand_cap * 64;
  Bi.tie_idx += f0 * BRISK_MAX_LAYERS * Bi.tie_cap;
  Bi.keys += f0 * Bi.cand_cap * 2;
  Bi.counters += f0;
  Bi.kp_out += f0 * Bi.kp_cap;
  Bi.bandsum += f0 * nbands * Bi.istride;
  BriskDescribeBuffers Di = ctx->D;
  Di.ibits = ctx->D.ibits;  // (chosen once per call: batch_begin)
  Di.integral += f0 * Di.iframe_elems;
  Di.dkp += f0 * Bi.kp_cap;
  Di.dscale += f0 * Bi.kp_cap;
  Di.dperm += f0 * Bi.kp_cap;
  Di.drec += f0 * Bi.kp_cap;
  Di.dp_work += f0 * Di.dp_work_stride;
  Di.desc += f0 * Bi.kp_cap * Di.desc_pitch;
  BriskOverlap ov{};
  const BriskOverlap* ovp = nullptr;
  if (A.do_detect && A.do_describe && overlap_integral && ctx->overlap && !(dbg_flags(ctx) & 0x10000)) {
    if (!ctx->side) {
      // lowest priority: the detector's latency-bound kernels get their workgroups placed first, the
      // bandwidth-bound integral kernel fills what they leave
      int least = 0, greatest = 0;
      HIPCHK(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
      const char* pe = tuning_env("BRISK_SIDE_PRIO");  // tuning experiments: 0 = default priority for the side stream
      HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->side, hipStreamNonBlocking, (pe && atoi(pe) == 0) ? 0 : least));
      HIPCHK(ctx, hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming));
      HIPCHK(ctx, hipEventCreateWithFlags(&ctx->side_join, hipEventDisableTiming));
    }
    ov.side = ctx->side; ov.fork = ctx->side_fork; ov.join = ctx->side_join; ov.Dd = &Di;
    ovp = &ov;
  }
  // Layer 0 in place: frames that already have the pyramid's layer-0 layout are read where they are (fast path only;
  // the ordered path and ComputeScale keep their private copy)
  BriskGeom Gs = ctx->G;
  {
    static const bool inplace_on = !(tuning_env("BRISK_L0_INPLACE") && atoi(tuning_env("BRISK_L0_INPLACE")) == 0);
    const bool ordered = Gs.threshold < BRISK_FAST_PATH_MIN_THRESHOLD || Gs.no_scale_nms || Gs.lower_threshold != BRISK_LOWER_THRESHOLD;
    // (width == stride: the caller's rows have no pad columns, so nothing is read that the argument check did not
    // cover - a pitch of align_up(width, 64) with width % 64 != 0 would make the clamped 16-byte loads read the caller's
    // pad bytes, up to 63 of them beyond the last frame's last row)
    if (A.do_detect && A.inplace_ok && inplace_on && !ordered && A.row_pitch == Gs.L[0].stride && A.w == Gs.L[0].stride &&
        ((((uintptr_t)d_frames) | (uintptr_t)A.frame_pitch) & 15) == 0) {
      Gs.l0_ext = d_frames;
      Gs.l0_pitch = A.frame_pitch;
      if (f0 == 0) { ctx->last_l0_ext = d_frames; ctx->last_l0_pitch = A.frame_pitch; }
    } else if (f0 == 0) {
      ctx->last_l0_ext = nullptr;
    }
  }
  if (A.do_detect) {
    brisk_launch_detect(Gs, ctx->T, Bi, nf, d_frames, A.frame_pitch, A.row_pitch,
                        A.d_mask ? A.d_mask + f0 * A.mask_frame_pitch : nullptr, A.mask_frame_pitch, A.mask_row_pitch, si, prof, ovp);
  }
  if (A.do_detect && A.uni_radius > 0.0) {
    // EnforceKeyPointUniformity as a post-filter of the detected keypoints (brisk_uniformity.hip)
    const float scaling = (float)(15.0 / (float)A.uni_radius);
    const int oh = (int)(A.h * ceil(scaling) + 32), ow = (int)(A.w * ceil(scaling) + 32);
    const long occ_frame = ((long)oh * ow + 255) / 256 * 256;
    brisk_launch_uniformity(Bi.kp_out, Bi.counters, ctx->d_uni_order + f0 * Bi.kp_cap, ctx->d_uni_tmp + f0 * Bi.kp_cap,
                            ctx->d_occ + f0 * occ_frame, occ_frame, ow, Bi.kp_cap, scaling, A.uni_max, nf, si);
  }
  if (A.do_detect && !(A.uni_radius > 0.0) && A.bk_u > 0) {
    // KeyPointBucketing as a post-filter of the detected keypoints (brisk_uniformity.hip)
    brisk_launch_bucketing(Bi.kp_out, Bi.counters, ctx->d_uni_order + f0 * Bi.kp_cap, ctx->d_uni_tmp + f0 * Bi.kp_cap, Bi.kp_cap,
                           A.h, A.w, A.bk_u, A.bk_v, A.bk_max, nf, si);
  }
  if (A.do_detect) brisk_prof_mark(prof, BRISK_STG_INTEGRAL, si);  // end of the post-filter interval
  if (A.do_describe) {
    BriskPatternDev P = A.pat->dev;
    brisk_launch_describe(Gs, P, Bi, Di, nf, Bi.kp_out, &Bi.counters[0].nkp, sizeof(BriskFrameCounters), si, prof, ovp);
  }
  return BRISK_HIP_OK;
}

static int batch_end(brisk_hip_ctx* ctx, const BatchArgs& A, int nframes, hipStream_t s) {
  if (A.do_detect && A.do_describe) {  // candidate density of this batch for the next one's choice of integral format
    if (!ctx->h_density) {
      HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_density, sizeof(long long), hipHostMallocMapped));
      *ctx->h_density = 0;
    }
    ctx->density_mpx = (double)A.w * (double)A.h / 1e6;
    brisk_launch_batch_density(ctx->B.counters, nframes, ctx->B.cand_cap, ctx->h_density, s);
  }
  if (ctx->prof.on) ctx->prof.calls++;
  HIPCHK(ctx, hipGetLastError());
  ctx->last_nframes = nframes;
  ctx->last_has_desc = A.do_describe;
  ctx->last_desc_pitch = ctx->D.desc_pitch;
  if (A.do_describe && A.pat) ctx->last_strings = A.pat->host.strings;
  const int rc = workspace_release(ctx, s);
  if (rc) return fail(ctx, rc, "hipEventRecord failed");
  return BRISK_HIP_OK;
}

static int run_batch(brisk_hip_ctx* ctx, const brisk_hip_pattern* pat, const uint8_t* d_frames, int nframes, int w, int h,
                     long frame_pitch, int row_pitch, int threshold, int octaves, const uint8_t* d_mask,
                     long mask_frame_pitch, int mask_row_pitch, hipStream_t s, bool do_detect, bool do_describe,
                     double uni_radius = -1.0, int uni_max = 0, bool no_scale_nms = false,
                     int lower_threshold = BRISK_LOWER_THRESHOLD, const int* bucketing = nullptr) {
  if (!d_frames || nframes <= 0 || row_pitch < w || frame_pitch < (long)row_pitch * (h - 1) + w)
    return fail(ctx, BRISK_HIP_ERR_ARG, "bad frame buffer description");
  if (int rcp = check_pattern_device(ctx, pat)) return rcp;
  // timing experiments only (debug bit 27): the descriptor half of a batch alone, on the keypoints the previous batch left
  const bool full_batch = do_detect && do_describe;
  if (do_detect && do_describe && (dbg_flags(ctx) & (1 << 27)) && ctx->last_nframes >= nframes) do_detect = false;
  BatchArgs A{pat, w, h, threshold, octaves, frame_pitch, row_pitch, d_mask, mask_frame_pitch, mask_row_pitch, do_detect,
              do_describe, uni_radius < 0.0 ? ctx->uni_radius : uni_radius, uni_radius < 0.0 ? ctx->uni_max : uni_max};
  A.no_scale_nms = no_scale_nms;
  A.lower_threshold = lower_threshold;
  A.format_as_full_batch = full_batch;
  // post-filters given per call (uni_radius >= 0 / bucketing != null) never read or write the context's settings
  if (bucketing) { A.bk_u = bucketing[0]; A.bk_v = bucketing[1]; A.bk_max = bucketing[2]; }
  else if (uni_radius < 0.0) { A.bk_u = ctx->bk_u; A.bk_v = ctx->bk_v; A.bk_max = ctx->bk_max; }
  int rc = batch_begin(ctx, A, nframes, s);
  if (rc) return rc;
  WorkspaceGuard guard(ctx, s);
  int nsub = ctx->nsub;
  if (nsub > 8) nsub = 8;
  if (nframes < 16 * nsub) nsub = nframes >= 32 ? 2 : 1;
  if (nsub > 1 && !ctx->sub_created) {
    for (int i = 0; i < 8; ++i) {
      HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->sub[i], hipStreamNonBlocking));
      HIPCHK(ctx, hipEventCreateWithFlags(&ctx->join_ev[i], hipEventDisableTiming));
    }
    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming));
    ctx->sub_created = true;
  }
  // timing experiments only (debug bits 20-23 = g with bit 27): the descriptor half in groups of 8 g frames, one after the
  // other on `s` (integral image of a group -> its descriptors: is the integral still in the Infinity Cache?); bit 19: the
  // integral images of all frames first, then the groups (the same launches with a stale cache)
  const int grp = 8 * ((dbg_flags(ctx) >> 20) & 0xF);
  if (grp > 0 && !do_detect && do_describe) {
    if (dbg_flags(ctx) & (1 << 19))
      brisk_launch_integral(ctx->G, ctx->B.pyr, ctx->B.bandsum, ctx->D.integral, ctx->D.istride, ctx->D.iframe_elems, ctx->B.band_h, nframes, s);
    for (long f0 = 0; f0 < nframes; f0 += grp) {
      const int nf = (int)((nframes - f0) < grp ? (nframes - f0) : grp);
      if (f0 == 0) ctx->last_frames_per_launch = nf;
      rc = batch_slice(ctx, A, d_frames + f0 * frame_pitch, f0, nf, s, (f0 == 0) ? &ctx->prof : nullptr, false);
      if (rc) return rc;
    }
    guard.armed = false;
    return batch_end(ctx, A, nframes, s);
  }
  if (nsub > 1) HIPCHK(ctx, hipEventRecord(ctx->fork_ev, s));
  for (int i = 0; i < nsub; ++i) {
    const long f0 = (long)nframes * i / nsub, f1 = (long)nframes * (i + 1) / nsub;
    const int nf = (int)(f1 - f0);
    if (nf <= 0) continue;
    hipStream_t si = s;
    if (nsub > 1) {
      si = ctx->sub[i];
      HIPCHK(ctx, hipStreamWaitEvent(si, ctx->fork_ev, 0));
    }
    if (i == 0) ctx->last_frames_per_launch = nf;
    rc = batch_slice(ctx, A, d_frames + f0 * frame_pitch, f0, nf, si, (i == 0) ? &ctx->prof : nullptr, nsub == 1);
    if (rc) return rc;
    if (nsub > 1) {
      HIPCHK(ctx, hipEventRecord(ctx->join_ev[i], si));
      HIPCHK(ctx, hipStreamWaitEvent(s, ctx->join_ev[i], 0));
    }
  }
  guard.armed = false;  // batch_end records the event itself
  return batch_end(ctx, A, nframes, s);
}

// ---- host-fed batch: frames in (pinned) host memory, H2D on a copy stream overlapped with compute ---------------
static int host_slice_frames() {
  static const int v = [] { const char* e = tuning_env("BRISK_HOST_SLICE"); const int n = e ? atoi(e) : 0; return n > 0 ? n : 64; }();
  return v;
}

static bool device_can_write(const void* p, void** dev);
#define BRISK_STAGED_SLICE 32  // frames per pinned staging buffer of the multi-image calls
// host threads that copy pageable images into the pinned staging buffers (the caller's thread is one of them)
static int copy_threads() {
  static const int n = [] {
    const char* e = tuning_env("BRISK_COPY_THREADS");
    int t = e ? atoi(e) : usable_cpus() / 2;
    return t < 1 ? 1 : (t > 8 ? 8 : t);
  }();
  return n;
}
// Frames ptrs[0 .. nf) (host memory, row pitch `stride`) -> d_dst (rows at dpitch, frames dframe apart) on stream cs.
// staged: the sources are pageable - `copy_threads()` host threads copy them into pinned buffer b of the context, one DMA moves
// the slice (nf <= BRISK_STAGED_SLICE); otherwise one asynchronous 2-D copy per frame straight from the caller's pinned memory.
static int upload_frames(brisk_hip_ctx* ctx, uint8_t* d_dst, size_t dframe, int dpitch, const uint8_t* const* ptrs, int nf, int w, int h, int stride,
                         bool staged, int b, hipStream_t cs) {
  if (!staged) {
    for (int f = 0; f < nf; ++f)
      HIPCHK(ctx, hipMemcpy2DAsync(d_dst + (size_t)f * dframe, dpitch, ptrs[f], stride, w, h, hipMemcpyHostToDevice, cs));
    return BRISK_HIP_OK;
  }
  if (ctx->pin_bytes < dframe * BRISK_STAGED_SLICE) {
    for (int i = 0; i < 2; ++i) {
      if (ctx->pin_used[i]) HIPCHK(ctx, hipEventSynchronize(ctx->pin_ev[i]));
      ctx->pin_used[i] = false;
      if (ctx->h_pin[i]) (void)hipHostFree(ctx->h_pin[i]);
      ctx->h_pin[i] = nullptr;
    }
    ctx->pin_bytes = 0;
    for (int i = 0; i < 2; ++i) {
      HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_pin[i], dframe * BRISK_STAGED_SLICE + 256, hipHostMallocDefault));
      if (!ctx->pin_ev[i]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->pin_ev[i], hipEventDisableTiming));
    }
    ctx->pin_bytes = dframe * BRISK_STAGED_SLICE;
  }
  if (ctx->pin_used[b]) HIPCHK(ctx, hipEventSynchronize(ctx->pin_ev[b]));  // the DMA that last read this buffer (two slices ago)
  uint8_t* const pin = ctx->h_pin[b];
  // work items = quarter frames (row bands): a few threads stay busy to the end of the slice
  std::atomic<int> next{0};
  const int items = nf * 4;
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= items) break;
      const int f = i >> 2, part = i & 3;
      const int y0 = (int)((long)h * part / 4), y1 = (int)((long)h * (part + 1) / 4);
      if (y1 <= y0) continue;  // (images of fewer than four rows)
      uint8_t* d = pin + (size_t)f * dframe;
      const uint8_t* sp = ptrs[f];
      if (stride == dpitch) memcpy(d + (size_t)y0 * dpitch, sp + (size_t)y0 * stride, (size_t)(y1 - y0 - 1) * dpitch + (size_t)w);
      else for (int y = y0; y < y1; ++y) memcpy(d + (size_t)y * dpitch, sp + (size_t)y * stride, (size_t)w);
    }
  };
  const int T = copy_threads() < items ? copy_threads() : items;
  std::vector<std::thread> helpers;
  for (int t = 1; t < T; ++t) helpers.emplace_back(work);
  work();
  for (std::thread& t : helpers) t.join();
  HIPCHK(ctx, hipMemcpyAsync(d_dst, pin, dframe * (size_t)nf, hipMemcpyHostToDevice, cs));
  HIPCHK(ctx, hipEventRecord(ctx->pin_ev[b], cs));
  ctx->pin_used[b] = true;
  return BRISK_HIP_OK;
}
// pageable host memory?  (what a cv::Mat's buffer is unless the caller page-locked it)
static bool is_pageable(const void* p) {
  void* dev = nullptr;
  return !device_can_write(p, &dev);
}

// frame_ptrs: null (the frames are h_frames + f * frame_pitch) or one host pointer per frame (the multi-image calls: separate
// cv::Mat buffers); pat == null: detection only
static int batch_host_locked(brisk_hip_ctx* ctx, const brisk_hip_pattern* pat, const uint8_t* h_frames, int nframes, int w, int h,
                             long frame_pitch, int row_pitch, int threshold, int octaves, const uint8_t* const* frame_ptrs = nullptr,
                             uint8_t* d_resident = nullptr /* the frames land here (f * dframe) and stay, instead of in the recycled staging */) {
  if ((!h_frames && !frame_ptrs) || nframes <= 0 || row_pitch < w || (!frame_ptrs && frame_pitch < (long)row_pitch * (h - 1) + w))
    return fail(ctx, BRISK_HIP_ERR_ARG, "bad frame buffer description");
  if (frame_ptrs)
    for (int f = 0; f < nframes; ++f)
      if (!frame_ptrs[f]) return fail(ctx, BRISK_HIP_ERR_ARG, "null image in the list");
  if (int rcp = check_pattern_device(ctx, pat)) return rcp;
  BatchArgs A{pat, w, h, threshold, octaves, 0, 0, nullptr, 0, 0, true, pat != nullptr, ctx->uni_radius, ctx->uni_max};
  A.bk_u = ctx->bk_u; A.bk_v = ctx->bk_v; A.bk_max = ctx->bk_max;
  A.inplace_ok = d_resident != nullptr;  // the staging buffers are recycled slice after slice: the engine keeps its own layer-0 copy (the link, not the engine, bounds this path)
  hipStream_t s = ctx->stream;
  int rc = batch_begin(ctx, A, nframes, s);
  if (rc) return rc;
  WorkspaceGuard guard(ctx, s);
  const bool staged = frame_ptrs && is_pageable(frame_ptrs[0]);
  const int slice_max = staged ? BRISK_STAGED_SLICE : host_slice_frames();
  const int slice = slice_max < nframes ? slice_max : nframes;
  const int dpitch = brisk_align_up(w, 64);           // device staging: rows at a 64-byte aligned pitch
  const size_t dframe = (size_t)dpitch * h;
  bool fresh = false;
  if (!ctx->copy_stream) {
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      HIPCHK(ctx, hipEventCreateWithFlags(&ctx->copied_ev[i], hipEventDisableTiming));
      HIPCHK(ctx, hipEventCreateWithFlags(&ctx->consumed_ev[i], hipEventDisableTiming));
    }
    fresh = true;
  }
  if (!d_resident && ctx->hstage_bytes < dframe * slice) {
    HIPCHK(ctx, hipDeviceSynchronize());
    for (int i = 0; i < 2; ++i) {
      if (ctx->d_hstage[i]) (void)hipFree(ctx->d_hstage[i]);
      ctx->d_hstage[i] = nullptr;
    }
    ctx->hstage_bytes = 0;
    for (int i = 0; i < 2; ++i) HIPCHK(ctx, hipMalloc(&ctx->d_hstage[i], dframe * slice + 256));
    ctx->hstage_bytes = dframe * slice;
    fresh = true;
  }
  A.frame_pitch = (long)dframe;
  A.row_pitch = dpitch;
  // Only host-fed calls touch the staging buffers, and every use records consumed_ev[b] behind the slice that read
  // buffer b: the first copies of this call wait for exactly that, i.e. they overlap the tail of the previous call's
  // compute.  (New buffers / events: nothing to wait for.)
  if (fresh) {
    HIPCHK(ctx, hipEventRecord(ctx->consumed_ev[0], s));
    HIPCHK(ctx, hipEventRecord(ctx->consumed_ev[1], s));
  }
  int k = 0;
  for (long f0 = 0; f0 < nframes; f0 += slice, ++k) {
    const int nf = (int)((nframes - f0 < slice) ? nframes - f0 : slice);
    const int b = k & 1;
    uint8_t* const dslice = d_resident ? d_resident + (size_t)f0 * dframe : ctx->d_hstage[b];
    if (!d_resident) HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->consumed_ev[b], 0));
    else if (k == 0 && ctx->done_valid && !ensure_done_event(ctx)) HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->done_ev, 0));  // (the previous call may still read the resident frames)
    const uint8_t* src = frame_ptrs ? nullptr : h_frames + f0 * frame_pitch;
    if (frame_ptrs) {
      rc = upload_frames(ctx, dslice, dframe, dpitch, frame_ptrs + f0, nf, w, h, row_pitch, staged, b, ctx->copy_stream);
      if (rc) return rc;
    } else if (frame_pitch == (long)row_pitch * h) {          // rows of consecutive frames at one pitch: a single 2-D copy
      HIPCHK(ctx, hipMemcpy2DAsync(dslice, dpitch, src, row_pitch, w, (size_t)h * nf, hipMemcpyHostToDevice,
                                   ctx->copy_stream));
    } else {
      for (int f = 0; f < nf; ++f)
        HIPCHK(ctx, hipMemcpy2DAsync(dslice + (size_t)f * dframe, dpitch, src + (long)f * frame_pitch, row_pitch, w,
                                     h, hipMemcpyHostToDevice, ctx->copy_stream));
    }
    HIPCHK(ctx, hipEventRecord(ctx->copied_ev[b], ctx->copy_stream));
    HIPCHK(ctx, hipStreamWaitEvent(s, ctx->copied_ev[b], 0));
    if (k == 0) ctx->last_frames_per_launch = nf;
    rc = batch_slice(ctx, A, dslice, f0, nf, s, (k == 0) ? &ctx->prof : nullptr, true);
    if (rc) return rc;
    if (!d_resident) HIPCHK(ctx, hipEventRecord(ctx->consumed_ev[b], s));
  }
  guard.armed = false;  // batch_end records the event itself
  return batch_end(ctx, A, nframes, s);
}

int brisk_hip_detect_describe_batch_host(brisk_hip_ctx* ctx, const brisk_hip_pattern* pat, const uint8_t* h_frames,
                                         int nframes, int w, int h, long frame_pitch, int row_pitch, int threshold,
                                         int octaves) {
  if (!ctx || !pat) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return batch_host_locked(ctx, pat, h_frames, nframes, w, h, frame_pitch, row_pitch, threshold, octaves);
}

int brisk_hip_detect_describe_batch(brisk_hip_ctx* ctx, const brisk_hip_pattern* pat, const uint8_t* d_frames, int nframes,
                                    int w, int h, long frame_pitch, int row_pitch, int threshold, int octaves,
                                    void* stream) {
  if (!ctx || !pat) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  return run_batch(ctx, pat, d_frames, nframes, w, h, frame_pitch, row_pitch, threshold, octaves, nullptr, 0, 0, s, true,
                   true);
}

int brisk_hip_detect_batch(brisk_hip_ctx* ctx, const uint8_t* d_frames, int nframes, int w, int h, long frame_pitch,
                           int row_pitch, int threshold, int octaves, void* stream) {
  if (!ctx) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  return run_batch(ctx, nullptr, d_frames, nframes, w, h, frame_pitch, row_pitch, threshold, octaves, nullptr, 0, 0, s, true,
                   false);
}

int brisk_hip_batch_results(brisk_hip_ctx* ctx, const int** d_detected, const int** d_described, int* count_stride,
                            const brisk_hip_keypoint** d_detected_kps, const brisk_hip_keypoint** d_described_kps,
                            const uint8_t** d_desc, int* kp_cap, int* desc_pitch) {
  if (!ctx) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!ctx->B.counters) return fail(ctx, BRISK_HIP_ERR_ARG, "no batch has run");
  if (d_detected) *d_detected = &ctx->B.counters[0].nkp;
  if (d_described) *d_described = &ctx->B.counters[0].ndesc;
  if (count_stride) *count_stride = (int)sizeof(BriskFrameCounters);
  if (d_detected_kps) *d_detected_kps = (const brisk_hip_keypoint*)ctx->B.kp_out;
  if (d_described_kps) *d_described_kps = (const brisk_hip_keypoint*)ctx->D.dkp;
  if (d_desc) *d_desc = ctx->D.desc;
  if (kp_cap) *kp_cap = ctx->B.kp_cap;
  // (a host describe call with packed destination rows wrote slot 0's rows at that pitch: what the rows HAVE, not what the
  // workspace was allocated for)
  if (desc_pitch) *desc_pitch = ctx->last_desc_pitch ? ctx->last_desc_pitch : ctx->D.desc_pitch;
  return BRISK_HIP_OK;
}

static int overflow_to_rc(brisk_hip_ctx* ctx, int flags) {
  if (flags & 1) return fail(ctx, BRISK_HIP_ERR_CAPACITY, "candidate capacity exceeded (brisk_hip_set_capacity)");
  if (flags & 2) return fail(ctx, BRISK_HIP_ERR_CAPACITY, "tie-candidate capacity exceeded (brisk_hip_set_capacity)");
  if (flags & 4) return fail(ctx, BRISK_HIP_ERR_CAPACITY, "keypoint capacity exceeded (brisk_hip_set_capacity)");
  if (flags & 8) {
    static const char* const site[8] = {"", " (the layer below)", " (a pending tie outside its chunk)", " (a neighbour's decision)",
                                        " (the writer wave)", " (the sorted tie list of the layer below)", "", ""};
    char msg[128];
    snprintf(msg, sizeof(msg), "tie resolution gave up waiting for a decision%s (internal error)", site[(flags >> 8) & 7]);
    return fail(ctx, BRISK_HIP_ERR_HIP, msg);
  }
  if (flags & 16)
    return fail(ctx, BRISK_HIP_ERR_UNSUPPORTED,
                "no defined result in the reference on this input: suppressScaleNonmaxima=false indexes layer 0's point list "
                "past its end or reads outside a score matrix (brisk-scale-space.cc:137), or ComputeScale addresses the maps of "
                "a point in a layer's last rows beyond the image (brisk-layer.cc:110-115)");
  return BRISK_HIP_OK;
}

int brisk_hip_batch_status(brisk_hip_ctx* ctx, int nframes, int* overflow_flags) {
  if (!ctx) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (nframes <= 0 || nframes > ctx->slots) return fail(ctx, BRISK_HIP_ERR_ARG, "bad frame count");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, wait_own_work(ctx));
  std::vector<BriskFrameCounters> c(nframes);
  HIPCHK(ctx, hipMemcpy(c.data(), ctx->B.counters, sizeof(BriskFrameCounters) * nframes, hipMemcpyDeviceToHost));
  int f = 0;
  for (auto& x : c) f |= x.overflow;
  if (overflow_flags) *overflow_flags = f;
  return overflow_to_rc(ctx, f);
}

static int download_locked(brisk_hip_ctx* ctx, int frame, int which, brisk_hip_keypoint* kps, int cap, int* n,
                           uint8_t* desc, int desc_stride, int strings, int dev_pitch = 0) {
  if (!dev_pitch) dev_pitch = ctx->D.desc_pitch;
  if (frame < 0 || frame >= ctx->slots) return fail(ctx, BRISK_HIP_ERR_ARG, "bad frame index");
  HIPCHK(ctx, wait_own_work(ctx));
  BriskFrameCounters c;
  HIPCHK(ctx, hipMemcpy(&c, ctx->B.counters + frame, sizeof(c), hipMemcpyDeviceToHost));
  int rc = overflow_to_rc(ctx, c.overflow);
  if (rc) return rc;
  const int cnt = which ? c.ndesc : c.nkp;
  if (n) *n = cnt;
  if (kps) {
    if (cnt > cap) return fail(ctx, BRISK_HIP_ERR_CAPACITY, "output keypoint buffer too small");
    const BriskKeyPoint* src = (which ? ctx->D.dkp : ctx->B.kp_out) + (size_t)frame * ctx->B.kp_cap;
    if (cnt) HIPCHK(ctx, hipMemcpy(kps, src, sizeof(BriskKeyPoint) * (size_t)cnt, hipMemcpyDeviceToHost));
  }
  if (desc && which && cnt) {
    if (cnt > cap) return fail(ctx, BRISK_HIP_ERR_CAPACITY, "output descriptor buffer too small");
    const uint8_t* src = ctx->D.desc + (size_t)frame * ctx->B.kp_cap * ctx->D.desc_pitch;
    if (dev_pitch == desc_stride && strings == desc_stride)  // rows packed on both sides: one linear copy (a pitched one of 75 k rows takes 4 x as long)
      HIPCHK(ctx, hipMemcpy(desc, src, (size_t)cnt * strings, hipMemcpyDeviceToHost));
    else
      HIPCHK(ctx, hipMemcpy2D(desc, desc_stride, src, dev_pitch, strings, cnt, hipMemcpyDeviceToHost));
  }
  return BRISK_HIP_OK;
}

// CPUs this process may use: the affinity mask, cut by a cgroup CPU quota (containers) - what decides whether a thread that
// waits for its results may poll or has to sleep
static int usable_cpus() {
  static const int n = [] {
    int c = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) c = CPU_COUNT(&set);
    if (c <= 0) c = (int)sysconf(_SC_NPROCESSORS_ONLN);
    long quota = -1, period = -1;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota | max> <period>"
      char q[64];
      if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atol(q);
      fclose(f);
    } else {
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%ld", &quota) != 1) quota = -1; fclose(g); }
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%ld", &period) != 1) period = -1; fclose(g); }
    }
    if (quota > 0 && period > 0) {
      const int lim = (int)((quota + period - 1) / period);
      if (lim > 0 && lim < c) c = lim;
    }
    return c > 0 ? c : 1;
  }();
  return n;
}
// host threads that are polling for a one-frame call's results right now (all contexts of the process)
static std::atomic<int> g_pollers{0};
struct PollerScope {
  int n;
  PollerScope() : n(g_pollers.fetch_add(1, std::memory_order_relaxed) + 1) {}
  ~PollerScope() { g_pollers.fetch_sub(1, std::memory_order_relaxed); }
};

// Results of a ONE-frame host-buffer call (frame slot 0).  download_locked costs three blocking copies - the count, then
// the keypoints, then the descriptor rows: 20 us each, a third of such a call on a 640 x 480 frame.  Here a small kernel
// behind the call's kernels (k_publish_single) writes the counter record and the rows it announces straight into pinned
// host memory and a sequence word behind them; the host polls that word (it changes a microsecond after the last store;
// waking up from a stream wait takes ten) and copies out.  Results that do not fit the pinned buffer take download_locked.
#define BRISK_SINGLE_BYTES (1u << 20)
// the pinned block of the one-frame calls: [0] sequence word (device -> host), [16] the provided-keypoint count of a describe call
// (host -> device: the kernels read it where it is - no copy, no runtime call), [64] counter record, then the rows
static int ensure_single_buffer(brisk_hip_ctx* ctx) {
  if (ctx->h_res) return BRISK_HIP_OK;
  HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_res, BRISK_SINGLE_BYTES, hipHostMallocCoherent));
  memset(ctx->h_res, 0, 64);
  HIPCHK(ctx, hipMalloc((void**)&ctx->d_pub_done, 64));
  HIPCHK(ctx, hipMemset(ctx->d_pub_done, 0, 64));
  HIPCHK(ctx, hipDeviceSynchronize());
  return BRISK_HIP_OK;
}
static int download_single(brisk_hip_ctx* ctx, int which, brisk_hip_keypoint* kps, int cap, int* n, uint8_t* desc,
                           int desc_stride, int strings, int dev_pitch, int expect) {
  const bool want_desc = desc && which;
  if (int rcb = ensure_single_buffer(ctx)) return rcb;
  const unsigned o_cnt = 64, o_kp = (unsigned)((o_cnt + sizeof(BriskFrameCounters) + 255) & ~(size_t)255);
  const size_t row = sizeof(BriskKeyPoint) + (want_desc ? (size_t)dev_pitch : 0);
  // (debug bit 25: a 16 KB limit, so that tests reach the staged-copy path with a few hundred keypoints)
  const size_t limit = (dbg_flags(ctx) & (1 << 25)) ? (size_t)o_kp + 256 + 16384 : (size_t)BRISK_SINGLE_BYTES;
  long max_kp = kps ? (long)((limit - o_kp - 256) / row) : 0;
  if (max_kp > cap) max_kp = cap;
  if (max_kp > ctx->B.kp_cap) max_kp = ctx->B.kp_cap;
  const unsigned o_desc = (unsigned)((o_kp + (size_t)max_kp * sizeof(BriskKeyPoint) + 63) & ~(size_t)63);
  ctx->pub_seq = (ctx->pub_seq + 1) & 0x7FFFFFFFu;
  if (!ctx->pub_seq) ctx->pub_seq = 1;
  const unsigned seq = ctx->pub_seq;
  brisk_launch_publish_single(ctx->B.counters, which ? ctx->D.dkp : ctx->B.kp_out, want_desc ? ctx->D.desc : nullptr, which, (int)max_kp,
                              dev_pitch, expect < max_kp ? expect : (int)max_kp, ctx->h_res, o_cnt, o_kp, o_desc, ctx->d_pub_done, seq, ctx->stream);
  HIPCHK(ctx, hipGetLastError());
  // The host polls the sequence word (a one-frame call lasts 0.1 ... 0.7 ms and the word arrives microseconds after the
  // last kernel; a blocking wait costs 12 us of wake-up) - but only for a bounded time: after 2 ms (a stream held up behind
  // foreign work, a 4K frame at a low threshold) the thread stops burning its core and sleeps in hipStreamSynchronize;
  // a kernel that failed never writes the word, which the synchronisation reports.  (Round-4 advisor finding: the poll had
  // no deadline.)
  volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(ctx->h_res);
  unsigned v = 0;
  // Polling needs a core per waiting thread: with more callers than the process has CPUs (round 6: 32 threads on 16 CPUs fell from
  // 9 k to 4 k frames/s, every thread burning its time slice on the flag while the threads that had work waited for a core) the
  // thread sleeps on a blocking event instead - an interrupt and ~20 us of wake-up, but the core goes to someone who needs it.
  const PollerScope poller;
  if (poller.n >= usable_cpus()) {
    if (!ctx->block_ev) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->block_ev, hipEventBlockingSync | hipEventDisableTiming));
    HIPCHK(ctx, hipEventRecord(ctx->block_ev, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->block_ev));
  }
  const auto t_poll = std::chrono::steady_clock::now();
  for (unsigned spin = 1;; ++spin) {
    v = __atomic_load_n(flag, __ATOMIC_ACQUIRE);
    if ((v & 0x7FFFFFFFu) == seq) break;
    if ((spin & 1023) == 0 && std::chrono::steady_clock::now() - t_poll > std::chrono::milliseconds(2)) {
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      v = __atomic_load_n(flag, __ATOMIC_ACQUIRE);
      if ((v & 0x7FFFFFFFu) == seq) break;
      return fail(ctx, BRISK_HIP_ERR_HIP, "the result kernel finished without publishing");
    }
    __builtin_ia32_pause();
  }
  if (v & 0x80000000u) return download_locked(ctx, 0, which, kps, cap, n, desc, desc_stride, strings, dev_pitch);  // more than the pinned buffer holds
  BriskFrameCounters c;
  memcpy(&c, ctx->h_res + o_cnt, sizeof(c));
  int rc = overflow_to_rc(ctx, c.overflow);
  if (rc) return rc;
  const int cnt = which ? c.ndesc : c.nkp;
  if (n) *n = cnt;
  if (!which) ctx->spec_nkp = cnt + cnt / 4 + 64;
  if (!kps || cnt == 0) return BRISK_HIP_OK;
  if (cnt > cap) return fail(ctx, BRISK_HIP_ERR_CAPACITY, want_desc ? "output descriptor buffer too small" : "output keypoint buffer too small");
  memcpy(kps, ctx->h_res + o_kp, (size_t)cnt * sizeof(BriskKeyPoint));
  if (want_desc) {
    const uint8_t* src = ctx->h_res + o_desc;
    if (dev_pitch == desc_stride && strings == desc_stride) memcpy(desc, src, (size_t)cnt * strings);
    else for (int i = 0; i < cnt; ++i) memcpy(desc + (size_t)i * desc_stride, src + (size_t)i * dev_pitch, strings);
  }
  return BRISK_HIP_OK;
}

int brisk_hip_batch_download(brisk_hip_ctx* ctx, int frame, int which, brisk_hip_keypoint* kps, int cap, int* n,
                             uint8_t* desc, int desc_stride) {
  if (!ctx) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // descriptor width of the last describe is not stored per call: copy the full pitch-limited row
  const int pitch = ctx->last_desc_pitch ? ctx->last_desc_pitch : ctx->D.desc_pitch;
  return download_locked(ctx, frame, which, kps, cap, n, desc, desc_stride, desc_stride < pitch ? desc_stride : pitch, pitch);
}

// ---- the batch path's exit to host memory (brisk_hip_batch_download_all; kernels: brisk_export.hip) ------------------
// Can the device write to this address?  Pinned / registered host memory, managed and device memory: yes (through the
// device-side alias the runtime reports); pageable host memory: no - the transfer then lands in the context's pinned bounce
// buffer and brisk_hip_batch_download_wait copies it out.
static bool device_can_write(const void* p, void** dev) {
  hipPointerAttribute_t a;
  memset(&a, 0, sizeof a);
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // (an unregistered pointer is an error on older runtimes, hipMemoryTypeUnregistered on newer ones)
    return false;
  }
  if (a.type != hipMemoryTypeHost && a.type != hipMemoryTypeDevice && a.type != hipMemoryTypeManaged && a.type != hipMemoryTypeUnified)
    return false;
  *dev = a.devicePointer ? a.devicePointer : const_cast<void*>(p);
  return true;
}

// byte offsets of the five arrays inside a slab / bounce buffer holding `frames` frames and `rows` rows
struct ExportLayout {
  size_t counts, flags, offsets, kps, desc, bytes;
  ExportLayout(int frames, long long rows, int desc_stride) {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    counts = 0;
    flags = up(counts + sizeof(int) * (size_t)frames);
    offsets = up(flags + sizeof(int) * (size_t)frames);
    kps = up(offsets + sizeof(long long) * ((size_t)frames + 1));
    desc = up(kps + sizeof(BriskKeyPoint) * (size_t)rows);
    bytes = up(desc + (size_t)rows * (size_t)desc_stride) + 256;
  }
};

// the egress kernel of slot E has finished (ctx->mu held): status of the transfer, and - for a pageable destination - the
// copy out of the bounce buffer
static void export_finish(brisk_hip_ctx* ctx, brisk_hip_ctx::ExportSlot& E) {
  const brisk_hip_batch_host_results& W = E.wr;
  int flagged = 0, orf = 0;
  for (int f = 0; f < E.nframes; ++f)
    if (W.flags[f]) { ++flagged; orf |= W.flags[f]; }
  if (E.use_bounce) {
    const long long rows = W.offsets[E.nframes];
    memcpy(E.dst.counts, W.counts, sizeof(int) * (size_t)E.nframes);
    memcpy(E.dst.flags, W.flags, sizeof(int) * (size_t)E.nframes);
    memcpy(E.dst.offsets, W.offsets, sizeof(long long) * ((size_t)E.nframes + 1));
    if (rows > 0) memcpy(E.dst.kps, W.kps, sizeof(BriskKeyPoint) * (size_t)rows);
    if (rows > 0 && E.dst.desc) memcpy(E.dst.desc, W.desc, (size_t)rows * (size_t)E.dst.desc_stride);
  }
  E.pending = false;
  E.flagged = flagged;
  E.rc = BRISK_HIP_OK;
  E.msg.clear();
  if (flagged) {
    char msg[200];
    if (orf & 7) {
      E.rc = overflow_to_rc(ctx, orf);
      E.msg = ctx->err;
    } else if (orf & BRISK_HIP_ROWS_CUT) {
      snprintf(msg, sizeof msg, "%d frame(s) did not fit the destination's rows_cap (flags[f] & BRISK_HIP_ROWS_CUT); their counts are reported",
               flagged);
      E.rc = BRISK_HIP_ERR_CAPACITY;
      E.msg = msg;
    } else {
      E.rc = overflow_to_rc(ctx, orf);
      E.msg = ctx->err;
    }
  }
}

static int download_all_locked(brisk_hip_ctx* ctx, int which, const brisk_hip_batch_host_results* dst, hipStream_t s, unsigned* ticket,
                               bool known_pinned = false /* the arrays come from hipHostMalloc (the pool's own): no pointer queries */,
                               bool egress_on_s = false /* the transfer on the batch's own stream (the pool: its contexts run one group at a time) */) {
  if (!dst || !ticket || (which != 0 && which != 1)) return fail(ctx, BRISK_HIP_ERR_ARG, "download_all: null destination / ticket, or which not 0 / 1");
  *ticket = 0;
  const int nframes = ctx->last_nframes;
  if (nframes <= 0 || !ctx->B.counters) return fail(ctx, BRISK_HIP_ERR_ARG, "download_all: no batch has run on this context");
  if (which && !ctx->last_has_desc) return fail(ctx, BRISK_HIP_ERR_ARG, "download_all: the last batch described nothing (which = 1)");
  const int strings = which ? ctx->last_strings : 0;
  const bool want_desc = which && dst->desc;
  if (dst->frames_cap < nframes || dst->rows_cap < 0 || !dst->counts || !dst->flags || !dst->offsets || (dst->rows_cap > 0 && !dst->kps))
    return fail(ctx, BRISK_HIP_ERR_ARG, "download_all: frames_cap below the batch's frames, or a null counts / flags / offsets / kps array");
  if (want_desc && (dst->desc_stride < strings || dst->desc_stride % 4 || strings % 4 || strings <= 0))
    return fail(ctx, BRISK_HIP_ERR_ARG, "download_all: desc_stride must be a multiple of 4 and at least the descriptor size");
  if ((((uintptr_t)dst->counts | (uintptr_t)dst->flags | (uintptr_t)dst->kps | (uintptr_t)dst->desc) & 3) || ((uintptr_t)dst->offsets & 7))
    return fail(ctx, BRISK_HIP_ERR_ARG, "download_all: destination arrays must be 4-byte aligned (offsets: 8)");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int dstride = want_desc ? dst->desc_stride : 4;
  brisk_hip_ctx::ExportSlot& E = ctx->ex[(ctx->ex_seq + 1) & 1];
  // The transfer's stream.  A context that runs detect + describe batches already owns a second stream - `side`, where the integral
  // kernel runs beside the detector's tail - and the transfer goes there: a process gets 4 hardware queues by default, and with the
  // caller's own stream, the context's main, side and copy streams a FIFTH active stream shares a queue with one of them (measured:
  // host-to-host 26.3 k -> 21.1 k frames/s after batches on a caller's stream).  Behind the transfer the next batch's integral kernel
  // starts up to half a millisecond late, inside its window beside the tie chain.  Contexts without a side stream get an egress stream.
  static const bool own_egress = tuning_env("BRISK_EXPORT_OWN_STREAM") && atoi(tuning_env("BRISK_EXPORT_OWN_STREAM")) == 1;  // A / B runs
  hipStream_t es = egress_on_s ? s : ((ctx->side && !own_egress) ? ctx->side : ctx->egress);
  if (!es) {
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->egress, hipStreamNonBlocking));
    es = ctx->egress;
  }
  if (!E.packed) {
    HIPCHK(ctx, hipEventCreateWithFlags(&E.packed, hipEventDisableTiming));
    HIPCHK(ctx, hipEventCreateWithFlags(&E.done, hipEventDisableTiming));
  }
  if (E.pending) {  // a third transfer in flight: complete the oldest first
    HIPCHK(ctx, hipEventSynchronize(E.done));
    export_finish(ctx, E);
  }
  const ExportLayout LY(dst->frames_cap, dst->rows_cap, dstride);
  if (LY.bytes > E.slab_bytes) {
    if (E.done_valid) HIPCHK(ctx, hipEventSynchronize(E.done));
    if (E.slab) (void)hipFree(E.slab);
    E.slab = nullptr; E.slab_bytes = 0;
    HIPCHK(ctx, hipMalloc(&E.slab, LY.bytes));
    E.slab_bytes = LY.bytes;
  }
  // where the egress kernel writes: the caller's arrays when the device can reach all of them, else the bounce buffer
  brisk_hip_batch_host_results W = *dst;
  void* dv[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  if (known_pinned) { dv[0] = dst->counts; dv[1] = dst->flags; dv[2] = dst->offsets; dv[3] = dst->kps; dv[4] = dst->desc; }
  const bool direct = known_pinned || (device_can_write(dst->counts, &dv[0]) && device_can_write(dst->flags, &dv[1]) && device_can_write(dst->offsets, &dv[2]) &&
                (dst->rows_cap == 0 || device_can_write(dst->kps, &dv[3])) && (!want_desc || device_can_write(dst->desc, &dv[4])));
  if (direct) {
    W.counts = static_cast<int*>(dv[0]); W.flags = static_cast<int*>(dv[1]); W.offsets = static_cast<long long*>(dv[2]);
    W.kps = static_cast<brisk_hip_keypoint*>(dv[3]); W.desc = want_desc ? static_cast<uint8_t*>(dv[4]) : nullptr;
  } else {
    if (LY.bytes > E.bounce_bytes) {
      if (E.bounce) (void)hipHostFree(E.bounce);
      E.bounce = nullptr; E.bounce_bytes = 0;
      HIPCHK(ctx, hipHostMalloc((void**)&E.bounce, LY.bytes, hipHostMallocDefault));
      E.bounce_bytes = LY.bytes;
    }
    W.counts = reinterpret_cast<int*>(E.bounce + LY.counts); W.flags = reinterpret_cast<int*>(E.bounce + LY.flags);
    W.offsets = reinterpret_cast<long long*>(E.bounce + LY.offsets); W.kps = reinterpret_cast<brisk_hip_keypoint*>(E.bounce + LY.kps);
    W.desc = want_desc ? E.bounce + LY.desc : nullptr;
  }
  uint8_t* sl = static_cast<uint8_t*>(E.slab);
  const BriskExportSlab S{reinterpret_cast<int*>(sl + LY.counts), reinterpret_cast<int*>(sl + LY.flags),
                          reinterpret_cast<long long*>(sl + LY.offsets), reinterpret_cast<uint32_t*>(sl + LY.kps),
                          reinterpret_cast<uint32_t*>(sl + LY.desc)};
  if (workspace_acquire(ctx, s)) return fail(ctx, BRISK_HIP_ERR_HIP, "hipStreamWaitEvent failed");
  WorkspaceGuard guard(ctx, s);
  if (E.done_valid) HIPCHK(ctx, hipStreamWaitEvent(s, E.done, 0));  // the slab's previous transfer
  const int dev_pitch = ctx->last_desc_pitch ? ctx->last_desc_pitch : ctx->D.desc_pitch;
  brisk_launch_export_pack(ctx->B.counters, which ? ctx->D.dkp : ctx->B.kp_out, want_desc ? ctx->D.desc : nullptr, ctx->B.kp_cap, dev_pitch,
                           want_desc ? strings : 4, nframes, which, dst->rows_cap, dstride, BRISK_HIP_ROWS_CUT, S, s);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipEventRecord(E.packed, s));
  if (guard.release()) return fail(ctx, BRISK_HIP_ERR_HIP, "hipEventRecord failed");
  HIPCHK(ctx, hipStreamWaitEvent(es, E.packed, 0));
  static const bool egress_off = tuning_env("BRISK_EXPORT_EGRESS") && atoi(tuning_env("BRISK_EXPORT_EGRESS")) == 0;  // timing experiments: pack only (the destination stays unwritten)
  if (!egress_off) brisk_launch_export_egress(S, nframes, dstride, W.counts, W.flags, W.offsets, W.kps, W.desc, es);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipEventRecord(E.done, es));
  E.done_valid = true;
  E.pending = true;
  E.use_bounce = !direct;
  E.nframes = nframes;
  E.dst = *dst;
  if (!want_desc) E.dst.desc = nullptr;
  if (direct) {  // the host reads the caller's own arrays
    W = *dst;
    if (!want_desc) W.desc = nullptr;
  }
  E.wr = W;
  E.ticket = ++ctx->ex_seq;
  if (!E.ticket) E.ticket = ++ctx->ex_seq;  // (0 is never a ticket)
  *ticket = E.ticket;
  return BRISK_HIP_OK;
}

int brisk_hip_batch_download_all(brisk_hip_ctx* ctx, int which, const brisk_hip_batch_host_results* dst, void* stream,
                                 unsigned* ticket) {
  if (!ctx) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return download_all_locked(ctx, which, dst, stream ? (hipStream_t)stream : ctx->stream, ticket);
}

int brisk_hip_batch_download_wait(brisk_hip_ctx* ctx, unsigned ticket, int* frames_flagged) {
  if (!ctx) return BRISK_HIP_ERR_ARG;
  std::unique_lock<std::mutex> lk(ctx->mu);
  if (frames_flagged) *frames_flagged = 0;
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, BRISK_HIP_ERR_HIP, "hipSetDevice failed");
  // the transfers up to `ticket`, oldest first; the lock is released while the host waits for the device
  for (int pass = 0; pass < 2; ++pass) {
    brisk_hip_ctx::ExportSlot* E = nullptr;
    for (auto& X : ctx->ex)
      if (X.pending && (int)(X.ticket - ticket) <= 0 && (!E || (int)(X.ticket - E->ticket) < 0)) E = &X;
    if (!E) break;
    const unsigned t = E->ticket;
    hipEvent_t ev = E->done;
    lk.unlock();
    const hipError_t e = hipEventSynchronize(ev);
    lk.lock();
    if (e != hipSuccess) {
      ctx->err = std::string("brisk_hip_batch_download_wait: ") + hipGetErrorString(e);
      return BRISK_HIP_ERR_HIP;
    }
    if (E->pending && E->ticket == t) export_finish(ctx, *E);  // (unless another thread completed it meanwhile)
  }
  for (auto& X : ctx->ex)
    if (X.ticket == ticket && ticket != 0 && !X.pending) {
      if (frames_flagged) *frames_flagged = X.flagged;
      if (X.rc) ctx->err = X.msg;
      return X.rc;
    }
  return fail(ctx, BRISK_HIP_ERR_ARG, "download_wait: unknown ticket (never issued on this context, or two later transfers have replaced it)");
}

int brisk_hip_detect_describe_batch_host_results(brisk_hip_ctx* ctx, const brisk_hip_pattern* pat, const uint8_t* h_frames,
                                                 int nframes, int w, int h, long frame_pitch, int row_pitch, int threshold,
                                                 int octaves, const brisk_hip_batch_host_results* dst, unsigned* ticket) {
  if (!ctx || !pat) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!dst || !ticket) return fail(ctx, BRISK_HIP_ERR_ARG, "null destination / ticket");
  *ticket = 0;
  if (dst->frames_cap < nframes) return fail(ctx, BRISK_HIP_ERR_ARG, "download_all: frames_cap below the batch's frames");
  const int rc = batch_host_locked(ctx, pat, h_frames, nframes, w, h, frame_pitch, row_pitch, threshold, octaves);
  if (rc) return rc;
  return download_all_locked(ctx, 1, dst, ctx->stream, ticket);
}

// ---- the multi-image overloads of the reference's base classes as batches -------------------------------------------------------
static int ensure_images(brisk_hip_ctx* ctx, int nimages, int w, int h) {
  const size_t dframe = (size_t)brisk_align_up(w, 64) * h;
  if (ctx->imgs_bytes < dframe * (size_t)nimages) {
    HIPCHK(ctx, hipDeviceSynchronize());
    if (ctx->d_imgs) (void)hipFree(ctx->d_imgs);
    ctx->d_imgs = nullptr; ctx->imgs_bytes = 0;
    ctx->imgs.n = 0;
    HIPCHK(ctx, hipMalloc(&ctx->d_imgs, dframe * (size_t)nimages + 256));
    ctx->imgs_bytes = dframe * (size_t)nimages;
  }
  return BRISK_HIP_OK;
}
static void remember_images(brisk_hip_ctx* ctx, const uint8_t* const* images, int nimages, int w, int h, int stride) {
  ctx->imgs.n = nimages; ctx->imgs.w = w; ctx->imgs.h = h; ctx->imgs.stride = stride;
  ctx->imgs.ptrs.assign(images, images + nimages);
}

int brisk_hip_detect_images(brisk_hip_ctx* ctx, const uint8_t* const* images, int nimages, int w, int h, int stride, int threshold,
                            int octaves, const brisk_hip_batch_host_results* dst, unsigned* ticket) {
  if (!ctx) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!images || !dst || !ticket) return fail(ctx, BRISK_HIP_ERR_ARG, "null image list / destination / ticket");
  *ticket = 0;
  if (dst->frames_cap < nimages) return fail(ctx, BRISK_HIP_ERR_ARG, "download_all: frames_cap below the batch's frames");
  if (nimages <= 0 || w <= 0 || h <= 0 || w > 8191 || h > 8191 || stride < w) return fail(ctx, BRISK_HIP_ERR_ARG, "bad image description");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // the frames stay on the device until the next multi-image call: a describe call on the same, unchanged buffers takes them from there
  int rc = ensure_images(ctx, nimages, w, h);
  if (rc) return rc;
  ctx->imgs.n = 0;
  rc = batch_host_locked(ctx, nullptr, nullptr, nimages, w, h, 0, stride, threshold, octaves, images, ctx->d_imgs);
  if (rc) return rc;
  remember_images(ctx, images, nimages, w, h, stride);
  return download_all_locked(ctx, 0, dst, ctx->stream, ticket);
}

static int describe_batch_locked(brisk_hip_ctx* ctx, const brisk_hip_pattern* pat, const uint8_t* d_frames, int nframes, int w, int h,
                                 long frame_pitch, int row_pitch, int rot, int scl, int n_in_max, hipStream_t s);

int brisk_hip_describe_images(brisk_hip_ctx* ctx, const brisk_hip_pattern* pat, const uint8_t* const* images, int nimages, int w, int h,
                              int stride, const brisk_hip_keypoint* const* kps, const int* nkps, int rotation_invariant,
                              int scale_invariant, int same_images, const brisk_hip_batch_host_results* dst, unsigned* ticket) {
  if (!ctx || !pat) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!images || !kps || !nkps || !dst || !ticket || nimages <= 0) return fail(ctx, BRISK_HIP_ERR_ARG, "null image / keypoint list, destination or ticket");
  *ticket = 0;
  if (w <= 0 || h <= 0 || w > 8191 || h > 8191 || stride < w) return fail(ctx, BRISK_HIP_ERR_ARG, "bad image description");
  if (dst->frames_cap < nimages) return fail(ctx, BRISK_HIP_ERR_ARG, "download_all: frames_cap below the batch's frames");
  if (int rcp = check_pattern_device(ctx, pat)) return rcp;
  int nmax = 0;
  for (int f = 0; f < nimages; ++f) {
    if (!images[f] || nkps[f] < 0 || (nkps[f] > 0 && !kps[f])) return fail(ctx, BRISK_HIP_ERR_ARG, "null image or keypoint list in the batch");
    nmax = nkps[f] > nmax ? nkps[f] : nmax;
  }
  if (nmax > ctx->kp_cap) return fail(ctx, BRISK_HIP_ERR_CAPACITY, "more keypoints than the configured capacity");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  BriskGeom Gd;
  BriskTileTable Td;
  make_geometry(w, h, 20, 0, &Gd, &Td);
  if (pat->host.strings > ctx->desc_pitch) ctx->desc_pitch = brisk_align_up(pat->host.strings, 16);
  int rc = ensure_buffers(ctx, nimages, Gd);
  if (rc) return rc;
  // all frames of the call resident at once (the descriptor-only batch is one launch sequence): a device buffer of the context
  const int dpitch = brisk_align_up(w, 64);
  const size_t dframe = (size_t)dpitch * h;
  // same_images: the caller's word (as brisk_hip_describe_same_image) that images[] are the very buffers of the context's last
  // multi-image call, unchanged - honoured when the list is that list
  bool resident = same_images && ctx->imgs.n == nimages && ctx->imgs.w == w && ctx->imgs.h == h && ctx->imgs.stride == stride;
  for (int f = 0; f < nimages && resident; ++f) resident = ctx->imgs.ptrs[(size_t)f] == images[f];
  if (!resident) {
    rc = ensure_images(ctx, nimages, w, h);
    if (rc) return rc;
  }
  hipStream_t s = ctx->stream;
  if (workspace_acquire(ctx, s)) return fail(ctx, BRISK_HIP_ERR_HIP, "hipStreamWaitEvent failed");
  WorkspaceGuard guard(ctx, s);
  if (!resident) {
    ctx->imgs.n = 0;
    const bool staged = is_pageable(images[0]);
    const int step = staged ? BRISK_STAGED_SLICE : nimages;
    for (int f0 = 0, k = 0; f0 < nimages; f0 += step, ++k) {
      const int nf = nimages - f0 < step ? nimages - f0 : step;
      rc = upload_frames(ctx, ctx->d_imgs + (size_t)f0 * dframe, dframe, dpitch, images + f0, nf, w, h, stride, staged, k & 1, s);
      if (rc) return rc;
    }
    remember_images(ctx, images, nimages, w, h, stride);
  } else {
    ctx->image_reuse_multi++;
  }
  // the provided lists: packed into one pinned block at a common pitch by this thread, moved by ONE 2-D copy (a small copy per image is
  // 10 us of runtime call each: 2.5 ms of a 256-image call)
  {
    const size_t kpitch = sizeof(BriskKeyPoint) * (size_t)(nmax > 0 ? nmax : 1);
    const size_t need = sizeof(int) * (size_t)nimages + 256 + kpitch * (size_t)nimages;
    if (need > ctx->kin_pin_bytes) {
      if (ctx->kin_pin_used) HIPCHK(ctx, hipEventSynchronize(ctx->kin_pin_ev));
      ctx->kin_pin_used = false;
      if (ctx->h_kin_pin) (void)hipHostFree(ctx->h_kin_pin);
      ctx->h_kin_pin = nullptr; ctx->kin_pin_bytes = 0;
      HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_kin_pin, need + need / 4, hipHostMallocDefault));
      ctx->kin_pin_bytes = need + need / 4;
      if (!ctx->kin_pin_ev) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->kin_pin_ev, hipEventDisableTiming));
    }
    if (ctx->kin_pin_used) HIPCHK(ctx, hipEventSynchronize(ctx->kin_pin_ev));  // (the previous call's copy out of the block)
    int* h_n = reinterpret_cast<int*>(ctx->h_kin_pin);
    uint8_t* h_k = ctx->h_kin_pin + ((sizeof(int) * (size_t)nimages + 255) & ~(size_t)255);
    for (int f = 0; f < nimages; ++f) {
      h_n[f] = nkps[f];
      if (nkps[f] > 0) memcpy(h_k + (size_t)f * kpitch, kps[f], sizeof(BriskKeyPoint) * (size_t)nkps[f]);
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_n_in, h_n, sizeof(int) * (size_t)nimages, hipMemcpyHostToDevice, s));
    if (nmax > 0)
      HIPCHK(ctx, hipMemcpy2DAsync(ctx->d_kp_in, sizeof(BriskKeyPoint) * (size_t)ctx->B.kp_cap, h_k, kpitch, kpitch, (size_t)nimages, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipEventRecord(ctx->kin_pin_ev, s));
    ctx->kin_pin_used = true;
  }
  rc = describe_batch_locked(ctx, pat, ctx->d_imgs, nimages, w, h, (long)dframe, dpitch, rotation_invariant, scale_invariant, nmax, s);
  if (rc) return rc;
  if (guard.release()) return fail(ctx, BRISK_HIP_ERR_HIP, "hipEventRecord failed");
  return download_all_locked(ctx, 1, dst, s, ticket);
}

// ---- host-buffer calls ---------------------------------------------------------------------------
static int detect_host(brisk_hip_ctx* ctx, const uint8_t* img, int w, int h, int stride, int threshold, int octaves,
                       int suppress_scale_nonmaxima, const uint8_t* mask, int mask_stride, double uni_radius, int uni_max,
                       brisk_hip_keypoint* out, int cap, int* n, const int* bucketing = nullptr) {
  if (!ctx || !img || !n || (cap > 0 && !out) || cap < 0) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  *n = 0;
  if (uni_radius >= 0.0 && ((uni_radius > 0.0 && uni_radius < 1.0) || uni_max < 1))
    return fail(ctx, BRISK_HIP_ERR_ARG, "uniformity: radius must be 0 (off) or >= 1, max_keypoints >= 1");
  // suppressScaleNonmaxima = false (brisk-scale-space.cc:131-170): with octaves == 0 the branch is the single-layer
  // 2-D refinement (:172-209) verbatim (fast path); with more layers it runs on the ordered path, `at(0)` indexing
  // included (k_ordered_keypoints)
  int rc = check_detect_args(ctx, w, h, threshold, octaves);
  if (rc) return rc;
  if (stride < w || (mask && mask_stride < w)) return fail(ctx, BRISK_HIP_ERR_ARG, "stride smaller than width");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t img_bytes = (size_t)brisk_align_up(w, 64) * h;
  rc = ensure_stage(ctx, img_bytes * 2);
  if (rc) return rc;
  const int pitch = brisk_align_up(w, 64);
  if (workspace_acquire(ctx, ctx->stream)) return fail(ctx, BRISK_HIP_ERR_HIP, "hipStreamWaitEvent failed");
  WorkspaceGuard guard(ctx, ctx->stream);  // (run_batch records the event again at its own end: harmless)
  HIPCHK(ctx, upload_rows(ctx->d_stage, pitch, img, stride, w, h, ctx->stream));
  const uint8_t* d_mask = nullptr;
  if (mask) {
    HIPCHK(ctx, upload_rows(ctx->d_stage + img_bytes, pitch, mask, mask_stride, w, h, ctx->stream));
    d_mask = ctx->d_stage + img_bytes;
  }
  rc = run_batch(ctx, nullptr, ctx->d_stage, 1, w, h, (long)img_bytes, pitch, threshold, octaves, d_mask, (long)img_bytes,
                 pitch, ctx->stream, true, false, uni_radius, uni_max, !suppress_scale_nonmaxima, BRISK_LOWER_THRESHOLD, bucketing);
  if (rc) return rc;
  ctx->img_cache.valid = true;
  ctx->img_cache.ptr = img; ctx->img_cache.w = w; ctx->img_cache.h = h; ctx->img_cache.stride = stride;
  ctx->img_cache.hash = image_hash_reuse_enabled() ? image_sample_hash(img, w, h, stride) : 0;  // (while the GPU works)
  ctx->img_cache.l0_ext = ctx->last_l0_ext;
  return download_single(ctx, 0, out, cap, n, nullptr, 0, 0, 0, ctx->spec_nkp < cap ? ctx->spec_nkp : cap);
}

int brisk_hip_detect(brisk_hip_ctx* ctx, const uint8_t* img, int w, int h, int stride, int threshold, int octaves,
                     int suppress_scale_nonmaxima, const uint8_t* mask, int mask_stride, brisk_hip_keypoint* out,
                     int cap, int* n) {
  return detect_host(ctx, img, w, h, stride, threshold, octaves, suppress_scale_nonmaxima, mask, mask_stride, -1.0, 0, out,
                     cap, n);
}

int brisk_hip_detect_uniform(brisk_hip_ctx* ctx, const uint8_t* img, int w, int h, int stride, int threshold, int octaves,
                             int suppress_scale_nonmaxima, const uint8_t* mask, int mask_stride, double uniformity_radius,
                             int max_keypoints, brisk_hip_keypoint* out, int cap, int* n) {
  if (uniformity_radius < 0.0) return ctx ? fail(ctx, BRISK_HIP_ERR_ARG, "uniformity: negative radius") : BRISK_HIP_ERR_ARG;
  return detect_host(ctx, img, w, h, stride, threshold, octaves, suppress_scale_nonmaxima, mask, mask_stride,
                     uniformity_radius, max_keypoints, out, cap, n);
}

int brisk_hip_detect_filtered(brisk_hip_ctx* ctx, const uint8_t* img, int w, int h, int stride, int threshold, int octaves,
                              int suppress_scale_nonmaxima, const uint8_t* mask, int mask_stride,
                              const brisk_hip_postfilter* pf, brisk_hip_keypoint* out, int cap, int* n) {
  if (!pf) return ctx ? fail(ctx, BRISK_HIP_ERR_ARG, "null post-filter description") : BRISK_HIP_ERR_ARG;
  if (pf->uniformity_radius < 0.0) return ctx ? fail(ctx, BRISK_HIP_ERR_ARG, "uniformity: negative radius") : BRISK_HIP_ERR_ARG;
  const bool bk_off = pf->num_buckets_u == 0 && pf->num_buckets_v == 0;
  if (!bk_off && (pf->num_buckets_u < 1 || pf->num_buckets_v < 1 || pf->bucket_max_keypoints < 1 ||
                  (long)pf->num_buckets_u * pf->num_buckets_v > (1 << 24)))
    return ctx ? fail(ctx, BRISK_HIP_ERR_ARG, "bucketing: buckets >= 1 each way (0, 0 = off), max_keypoints >= 1") : BRISK_HIP_ERR_ARG;
  const int bk[3] = {bk_off ? 0 : pf->num_buckets_u, bk_off ? 0 : pf->num_buckets_v, bk_off ? 0 : pf->bucket_max_keypoints};
  return detect_host(ctx, img, w, h, stride, threshold, octaves, suppress_scale_nonmaxima, mask, mask_stride,
                     pf->uniformity_radius, pf->uniformity_radius > 0.0 ? pf->uniformity_max_keypoints : 0x7FFFFFFF, out, cap, n, bk);
}

int brisk_hip_compute_scale(brisk_hip_ctx* ctx, const uint8_t* img, int w, int h, int stride, int threshold, int octaves,
                            int suppress_scale_nonmaxima, const brisk_hip_keypoint* in, int n_in, brisk_hip_keypoint* out,
                            int cap, int* n) {
  if (!ctx || !img || !n || n_in < 0 || (n_in > 0 && !in) || (cap > 0 && !out) || cap < 0) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  *n = 0;
  int rc = check_detect_args(ctx, w, h, threshold, octaves);
  if (rc) return rc;
  if (stride < w) return fail(ctx, BRISK_HIP_ERR_ARG, "stride smaller than width");
  if (n_in > ctx->kp_cap) return fail(ctx, BRISK_HIP_ERR_CAPACITY, "more keypoints than the configured capacity");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int pitch = brisk_align_up(w, 64);
  const size_t img_bytes = (size_t)pitch * h;
  rc = ensure_stage(ctx, img_bytes * 2);
  if (rc) return rc;
  if (workspace_acquire(ctx, ctx->stream)) return fail(ctx, BRISK_HIP_ERR_HIP, "hipStreamWaitEvent failed");
  WorkspaceGuard guard(ctx, ctx->stream);
  ctx->img_cache.valid = false;
  HIPCHK(ctx, upload_rows(ctx->d_stage, pitch, img, stride, w, h, ctx->stream));
  if (n_in == 0) {
    // an empty list makes GetKeypoints detect (brisk-scale-space.cc:104): plain detection on the pyramid ComputeScale
    // builds (lowerThreshold_ = 0, brisk-feature-detector.cc:90), without the mask filter
    rc = run_batch(ctx, nullptr, ctx->d_stage, 1, w, h, (long)img_bytes, pitch, threshold, octaves, nullptr, 0, 0, ctx->stream,
                   true, false, 0.0, 0x7FFFFFFF, !suppress_scale_nonmaxima, 0);
    if (rc) return rc;
    return download_single(ctx, 0, out, cap, n, nullptr, 0, 0, 0, ctx->spec_nkp < cap ? ctx->spec_nkp : cap);
  }
  make_geometry(w, h, threshold, octaves, &ctx->G, &ctx->T);
  ctx->G.debug_flags = dbg_flags(ctx);
  ctx->G.lower_threshold = 0;
  rc = ensure_buffers(ctx, 1, ctx->G);
  if (rc) return rc;
  if (ctx->dirty_frames > 0) brisk_launch_smap_clear(ctx->dirtyG, ctx->B, ctx->dirty_frames, ctx->stream);
  ctx->dirtyG = ctx->G;
  ctx->dirty_frames = 1;  // (the kernel marks the frame for a complete clear: the walk writes the cache anywhere)
  HIPCHK(ctx, hipMemcpyAsync(ctx->d_kp_in, in, sizeof(BriskKeyPoint) * (size_t)n_in, hipMemcpyHostToDevice, ctx->stream));
  ctx->last_l0_ext = nullptr;
  brisk_launch_compute_scale(ctx->G, ctx->B, ctx->d_stage, pitch, ctx->d_kp_in, n_in, suppress_scale_nonmaxima ? 1 : 0,
                             ctx->stream);
  HIPCHK(ctx, hipGetLastError());
  ctx->last_nframes = 1;
  ctx->last_has_desc = false;
  if (guard.release()) return fail(ctx, BRISK_HIP_ERR_HIP, "hipEventRecord failed");
  return download_single(ctx, 0, out, cap, n, nullptr, 0, 0, 0, ctx->spec_nkp < cap ? ctx->spec_nkp : cap);
}

static int describe_host(brisk_hip_ctx* ctx, const brisk_hip_pattern* pat, const uint8_t* img, int w, int h, int stride,
                         brisk_hip_keypoint* kps, int* n, uint8_t* desc, int desc_stride, int rotation_invariant,
                         int scale_invariant, bool same_image) {
  if (!ctx || !pat || !img || !n || *n < 0 || (*n > 0 && (!kps || !desc))) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (w <= 0 || h <= 0 || w > 8191 || h > 8191 || stride < w) return fail(ctx, BRISK_HIP_ERR_ARG, "bad image description");
  if (*n > 0 && desc_stride < pat->host.strings) return fail(ctx, BRISK_HIP_ERR_ARG, "descriptor stride too small");
  if (*n > ctx->kp_cap) return fail(ctx, BRISK_HIP_ERR_CAPACITY, "more keypoints than the configured capacity");
  if (int rcp = check_pattern_device(ctx, pat)) return rcp;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int pitch = brisk_align_up(w, 64);
  const size_t img_bytes = (size_t)pitch * h;
  int rc = ensure_stage(ctx, img_bytes * 2);
  if (rc) return rc;
  make_geometry(w, h, 20, 0, &ctx->G, &ctx->T);  // only layer 0 is needed
  if (pat->host.strings > ctx->desc_pitch) ctx->desc_pitch = brisk_align_up(pat->host.strings, 16);
  rc = ensure_buffers(ctx, 1, ctx->G);
  if (rc) return rc;
  if (workspace_acquire(ctx, ctx->stream)) return fail(ctx, BRISK_HIP_ERR_HIP, "hipStreamWaitEvent failed");
  WorkspaceGuard guard(ctx, ctx->stream);
  // the image of the last detect call, still on the device?  Only on the caller's word (same_image) - or, opted in through
  // the environment, when the sampled hash of the buffer has not changed
  const bool reuse = ctx->img_cache.valid && ctx->img_cache.ptr == img && ctx->img_cache.w == w && ctx->img_cache.h == h &&
                     ctx->img_cache.stride == stride &&
                     (same_image || (image_hash_reuse_enabled() && ctx->img_cache.hash == image_sample_hash(img, w, h, stride)));
  ctx->img_cache.valid = reuse;  // an uploaded image overwrites the staging buffer (and is not remembered itself)
  if (reuse) ctx->img_cache.hits++;
  if (!reuse) HIPCHK(ctx, upload_rows(ctx->d_stage, pitch, img, stride, w, h, ctx->stream));
  const int n_in = *n;
  // (the count where the kernels read it: a word of the context's pinned block - the call is synchronous, nothing of an earlier
  // call still reads it)
  if (int rcb = ensure_single_buffer(ctx)) return rcb;
  int* const h_n_in = reinterpret_cast<int*>(ctx->h_res + 16);
  __atomic_store_n(h_n_in, n_in, __ATOMIC_RELEASE);
  if (n_in) HIPCHK(ctx, hipMemcpyAsync(ctx->d_kp_in, kps, sizeof(BriskKeyPoint) * (size_t)n_in, hipMemcpyHostToDevice, ctx->stream));
  if (ctx->dirty_frames > 0) {  // the clear needs the last detect batch's counters, which are reset below
    brisk_launch_smap_clear(ctx->dirtyG, ctx->B, ctx->dirty_frames, ctx->stream);
    ctx->dirty_frames = 0;
  }
  HIPCHK(ctx, hipMemsetAsync(ctx->B.counters, 0, sizeof(BriskFrameCounters), ctx->stream));
  BriskDetectBuffers Bd = ctx->B;
  brisk_prof_begin_call(&ctx->prof);
  brisk_prof_mark(&ctx->prof, BRISK_STG_PYRAMID, ctx->stream);  // (the layer-0 pass, if any, shows as k_pyramid)
  if (reuse) {
    // layer 0 (in the pyramid buffer, or read in place from the staging buffer) and the pyramid kernel's 96-row band
    // sums are what the detect call left
    ctx->G.l0_ext = ctx->img_cache.l0_ext;
    ctx->G.l0_pitch = (long)img_bytes;
    ctx->last_l0_ext = ctx->img_cache.l0_ext;
    ctx->last_l0_pitch = (long)img_bytes;
  } else {
    ctx->last_l0_ext = nullptr;
    brisk_launch_layer0_only(ctx->G, ctx->B, 1, ctx->d_stage, (long)img_bytes, pitch, ctx->stream);
    Bd.band_h = 64;  // brisk_launch_layer0_only: k_pyramid_even's 64-row band sums
  }
  brisk_prof_mark(&ctx->prof, BRISK_STG_DETECT, ctx->stream);
  BriskPatternDev P = pat->dev;
  P.rotation_invariant = rotation_invariant ? 1 : 0;
  P.scale_invariant = scale_invariant ? 1 : 0;
  // descriptor rows of this call at the caller's pitch when the caller's rows are packed (a cv::Mat of K x strings bytes):
  // the download is then one linear copy
  BriskDescribeBuffers Dd = ctx->D;
  integral_format(ctx, pat, false, &Dd.ibits);
  if (desc_stride == pat->host.strings && pat->host.strings % 8 == 0 && pat->host.strings <= ctx->D.desc_pitch) Dd.desc_pitch = pat->host.strings;
  brisk_launch_describe(ctx->G, P, Bd, Dd, 1, ctx->d_kp_in, h_n_in, sizeof(int), ctx->stream, &ctx->prof, nullptr, n_in);
  if (ctx->prof.on) ctx->prof.calls++;
  HIPCHK(ctx, hipGetLastError());
  ctx->last_nframes = 1;
  ctx->last_has_desc = true;
  ctx->last_desc_pitch = Dd.desc_pitch;
  ctx->last_strings = pat->host.strings;
  if (guard.release()) return fail(ctx, BRISK_HIP_ERR_HIP, "hipEventRecord failed");
  return download_single(ctx, 1, kps, n_in, n, desc, desc_stride, pat->host.strings, Dd.desc_pitch, n_in);
}

int brisk_hip_describe(brisk_hip_ctx* ctx, const brisk_hip_pattern* pat, const uint8_t* img, int w, int h, int stride,
                       brisk_hip_keypoint* kps, int* n, uint8_t* desc, int desc_stride, int rotation_invariant,
                       int scale_invariant) {
  return describe_host(ctx, pat, img, w, h, stride, kps, n, desc, desc_stride, rotation_invariant, scale_invariant, false);
}

int brisk_hip_describe_same_image(brisk_hip_ctx* ctx, const brisk_hip_pattern* pat, const uint8_t* img, int w, int h, int stride,
                                  brisk_hip_keypoint* kps, int* n, uint8_t* desc, int desc_stride, int rotation_invariant,
                                  int scale_invariant) {
  return describe_host(ctx, pat, img, w, h, stride, kps, n, desc, desc_stride, rotation_invariant, scale_invariant, true);
}

// ---- per-stage timing (HIP events on the launch stream) -----------------------------------------------
// ---------------------------------------------------------------------------------------------------------------
// Hamming brute-force matcher
// ---------------------------------------------------------------------------------------------------------------
static_assert(sizeof(brisk_hip_dmatch) == 16 && sizeof(BriskDMatch) == 16, "cv::DMatch layout");

namespace {
struct DevBuf {  // frees its allocations when the call returns
  std::vector<void*> ptrs;
  ~DevBuf() { for (void* p : ptrs) (void)hipFree(p); }
  hipError_t alloc(void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e == hipSuccess) ptrs.push_back(*p);
    return e;
  }
};
}  // namespace

// mode 0: knn (param k), mode 1: radius (param cap)
static int match_host(brisk_hip_ctx* ctx, const uint8_t* query, int nq, int q_pitch, int dim, int nimg,
                      const uint8_t* const* train, const int* ntrain, const int* t_pitch, const uint8_t* const* masks,
                      const int* mask_pitch, int mode, int k_or_cap, float max_distance, brisk_hip_dmatch* out,
                      int* out_count) {
  if (!ctx) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lock(ctx->mu);
  if (nq < 0 || nimg < 0 || k_or_cap < 0 || !out_count || (nq > 0 && (!query || (k_or_cap > 0 && !out))) ||
      (nimg > 0 && (!train || !ntrain || !t_pitch)))
    return fail(ctx, BRISK_HIP_ERR_ARG, "match: bad argument");
  if (dim < 16 || dim > 224) return fail(ctx, BRISK_HIP_ERR_UNSUPPORTED, "match: descriptor size must be 16..224 bytes");
  if (q_pitch < dim) return fail(ctx, BRISK_HIP_ERR_ARG, "match: query pitch smaller than the descriptor");
  if (nq == 0) return BRISK_HIP_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int dim16 = (dim / 16) * 16;  // brisk::Hamming ignores bytes beyond the last full 128-bit word
  std::vector<int> img_start(nimg + 1, 0), has_mask(nimg > 0 ? nimg : 1, 0);
  bool any_mask = false;
  for (int i = 0; i < nimg; ++i) {
    if (ntrain[i] < 0 || (ntrain[i] > 0 && (!train[i] || t_pitch[i] < dim)))
      return fail(ctx, BRISK_HIP_ERR_ARG, "match: bad train set");
    img_start[i + 1] = img_start[i] + ntrain[i];
    if (masks && masks[i] && ntrain[i] > 0) {
      if (!mask_pitch || mask_pitch[i] < ntrain[i]) return fail(ctx, BRISK_HIP_ERR_ARG, "match: bad mask pitch");
      has_mask[i] = 1;
      any_mask = true;
    }
  }
  const int nt = img_start[nimg];
  const long tp = dim16;                                        // packed train rows
  const long dist_pitch = ((long)nt + 63) / 64 * 64 + 64;
  long qblock = (256L << 20) / (dist_pitch * 2);                // <= 256 MB of distances at a time
  if (qblock < 64) qblock = 64;
  if (qblock > nq) qblock = nq;
  DevBuf mem;
  uint8_t *d_q = nullptr, *d_t = nullptr, *d_mask = nullptr;
  uint16_t* d_dist = nullptr;
  int *d_start = nullptr, *d_has = nullptr, *d_masked = nullptr, *d_cnt = nullptr;
  BriskDMatch* d_out = nullptr;
  HIPCHK(ctx, mem.alloc((void**)&d_q, (size_t)nq * dim16));
  HIPCHK(ctx, mem.alloc((void**)&d_t, (size_t)(nt > 0 ? nt : 1) * tp));
  HIPCHK(ctx, mem.alloc((void**)&d_dist, (size_t)qblock * dist_pitch * 2));
  HIPCHK(ctx, mem.alloc((void**)&d_start, sizeof(int) * (nimg + 1)));
  HIPCHK(ctx, mem.alloc((void**)&d_has, sizeof(int) * (nimg > 0 ? nimg : 1)));
  HIPCHK(ctx, mem.alloc((void**)&d_masked, sizeof(int) * (size_t)qblock));
  HIPCHK(ctx, mem.alloc((void**)&d_cnt, sizeof(int) * (size_t)nq));
  HIPCHK(ctx, mem.alloc((void**)&d_out, sizeof(BriskDMatch) * (size_t)nq * (k_or_cap > 0 ? k_or_cap : 1)));
  hipStream_t s = ctx->stream;
  HIPCHK(ctx, hipMemcpy2DAsync(d_q, dim16, query, q_pitch, dim16, nq, hipMemcpyHostToDevice, s));
  for (int i = 0; i < nimg; ++i)
    if (ntrain[i] > 0)
      HIPCHK(ctx, hipMemcpy2DAsync(d_t + (long)img_start[i] * tp, tp, train[i], t_pitch[i], dim16, ntrain[i],
                                   hipMemcpyHostToDevice, s));
  HIPCHK(ctx, hipMemcpyAsync(d_start, img_start.data(), sizeof(int) * (nimg + 1), hipMemcpyHostToDevice, s));
  HIPCHK(ctx, hipMemcpyAsync(d_has, has_mask.data(), sizeof(int) * (nimg > 0 ? nimg : 1), hipMemcpyHostToDevice, s));
  long mpitch = 0;
  if (any_mask && nt > 0) {  // concatenated mask, nq x nt (255 where an image has no mask)
    mpitch = nt;
    HIPCHK(ctx, mem.alloc((void**)&d_mask, (size_t)nq * mpitch));
    HIPCHK(ctx, hipMemsetAsync(d_mask, 0xFF, (size_t)nq * mpitch, s));
    for (int i = 0; i < nimg; ++i)
      if (has_mask[i])
        HIPCHK(ctx, hipMemcpy2DAsync(d_mask + img_start[i], mpitch, masks[i], mask_pitch[i], ntrain[i], nq,
                                     hipMemcpyHostToDevice, s));
  }
  bool fused = false;
  if (mode == 0 && !d_mask && !(dbg_flags(ctx) & 0x20000)) {
    int nonempty = 0, first = -1;
    for (int i = 0; i < nimg; ++i)
      if (ntrain[i] > 0) { ++nonempty; if (first < 0) first = i; }
    // one non-empty train image that is also the last one (the imgIdx the kernel writes is patched below)
    if (nonempty == 1)
      fused = brisk_launch_match_knn_fused(d_q, dim16, nq, d_t, (int)tp, nt, dim16 / 4, k_or_cap, d_out, d_cnt, s);
    if (fused && first != 0) {  // imgIdx of the single non-empty image
      HIPCHK(ctx, hipStreamSynchronize(s));
      std::vector<BriskDMatch> tmp((size_t)nq * k_or_cap);
      HIPCHK(ctx, hipMemcpy(tmp.data(), d_out, sizeof(BriskDMatch) * tmp.size(), hipMemcpyDeviceToHost));
      for (BriskDMatch& m : tmp) m.imgIdx = first;
      HIPCHK(ctx, hipMemcpy(d_out, tmp.data(), sizeof(BriskDMatch) * tmp.size(), hipMemcpyHostToDevice));
    }
  }
  for (long q0 = 0; q0 < nq && !fused; q0 += qblock) {
    const int nqb = (int)((nq - q0 < qblock) ? nq - q0 : qblock);
    brisk_launch_match_dist(d_q, dim16, (int)q0, nqb, d_t, (int)tp, nt, dim16 / 8, d_mask, mpitch, d_dist, dist_pitch, s);
    const int* masked = nullptr;
    if (d_mask) {
      brisk_launch_match_masked_out(d_mask, mpitch, (int)q0, nqb, d_start, d_has, nimg, d_masked, s);
      masked = d_masked;
    }
    if (mode == 0)
      brisk_launch_match_knn(d_dist, dist_pitch, (int)q0, nqb, nt, d_start, nimg, masked, k_or_cap, d_out, d_cnt, s);
    else
      brisk_launch_match_radius(d_dist, dist_pitch, (int)q0, nqb, nt, d_start, nimg, masked, max_distance, k_or_cap, d_out,
                                d_cnt, dim16, s);
  }
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out_count, d_cnt, sizeof(int) * (size_t)nq, hipMemcpyDeviceToHost, s));
  if (k_or_cap > 0)
    HIPCHK(ctx, hipMemcpyAsync(out, d_out, sizeof(BriskDMatch) * (size_t)nq * k_or_cap, hipMemcpyDeviceToHost, s));
  HIPCHK(ctx, hipStreamSynchronize(s));
  return BRISK_HIP_OK;
}

int brisk_hip_match_knn(brisk_hip_ctx* ctx, const uint8_t* query, int nq, int q_pitch, int dim_bytes, int nimg,
                        const uint8_t* const* train, const int* ntrain, const int* t_pitch,
                        const uint8_t* const* masks, const int* mask_pitch, int k, brisk_hip_dmatch* out,
                        int* out_count) {
  return match_host(ctx, query, nq, q_pitch, dim_bytes, nimg, train, ntrain, t_pitch, masks, mask_pitch, 0, k, 0.f, out,
                    out_count);
}

int brisk_hip_match_radius(brisk_hip_ctx* ctx, const uint8_t* query, int nq, int q_pitch, int dim_bytes, int nimg,
                           const uint8_t* const* train, const int* ntrain, const int* t_pitch,
                           const uint8_t* const* masks, const int* mask_pitch, float max_distance,
                           int cap_per_query, brisk_hip_dmatch* out, int* out_count) {
  return match_host(ctx, query, nq, q_pitch, dim_bytes, nimg, train, ntrain, t_pitch, masks, mask_pitch, 1, cap_per_query,
                    max_distance, out, out_count);
}

int brisk_hip_match_knn_device(brisk_hip_ctx* ctx, const uint8_t* d_query, int nq, int q_pitch, const uint8_t* d_train,
                               int nt, int t_pitch, int dim_bytes, int k, brisk_hip_dmatch* d_out, int* d_out_count,
                               void* stream) {
  if (!ctx) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lock(ctx->mu);
  if (nq < 0 || nt < 0 || k < 0 || !d_out_count || (nq > 0 && (!d_query || (k > 0 && !d_out))) || (nt > 0 && !d_train))
    return fail(ctx, BRISK_HIP_ERR_ARG, "match: bad argument");
  if (dim_bytes < 16 || dim_bytes > 224) return fail(ctx, BRISK_HIP_ERR_UNSUPPORTED, "match: descriptor size must be 16..224 bytes");
  if (q_pitch < dim_bytes || (nt > 0 && t_pitch < dim_bytes)) return fail(ctx, BRISK_HIP_ERR_ARG, "match: pitch smaller than the descriptor");
  if (nq == 0) return BRISK_HIP_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int dim16 = (dim_bytes / 16) * 16;
  hipStream_t st = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
  if (!(dbg_flags(ctx) & 0x20000) &&
      brisk_launch_match_knn_fused(d_query, q_pitch, nq, d_train, t_pitch, nt, dim16 / 4, k,
                                   reinterpret_cast<BriskDMatch*>(d_out), d_out_count, st)) {
    HIPCHK(ctx, hipGetLastError());
    return BRISK_HIP_OK;
  }
  const long dist_pitch = ((long)nt + 63) / 64 * 64 + 64;
  const size_t need = (size_t)nq * dist_pitch * 2;
  if (workspace_acquire(ctx, st)) return fail(ctx, BRISK_HIP_ERR_HIP, "hipStreamWaitEvent failed");
  if (ctx->match_bytes < need) {  // workspace kept by the context (the call is asynchronous)
    HIPCHK(ctx, hipDeviceSynchronize());
    if (ctx->d_match) (void)hipFree(ctx->d_match);
    ctx->d_match = nullptr; ctx->match_bytes = 0;
    HIPCHK(ctx, hipMalloc(&ctx->d_match, need));
    ctx->match_bytes = need;
  }
  uint16_t* d_dist = static_cast<uint16_t*>(ctx->d_match);
  brisk_launch_match_dist(d_query, q_pitch, 0, nq, d_train, t_pitch, nt, dim16 / 8, nullptr, 0, d_dist, dist_pitch, st);
  brisk_launch_match_knn(d_dist, dist_pitch, 0, nq, nt, nullptr, 1, nullptr, k,
                         reinterpret_cast<BriskDMatch*>(d_out), d_out_count, st);
  HIPCHK(ctx, hipGetLastError());
  if (workspace_release(ctx, st)) return fail(ctx, BRISK_HIP_ERR_HIP, "hipEventRecord failed");
  return BRISK_HIP_OK;
}

int brisk_hip_set_uniformity(brisk_hip_ctx* ctx, double radius, int max_keypoints) {
  if (!ctx) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (radius < 0.0 || (radius > 0.0 && radius < 1.0) || max_keypoints < 1) return fail(ctx, BRISK_HIP_ERR_ARG, "uniformity: radius must be 0 (off) or >= 1, max_keypoints >= 1");
  ctx->uni_radius = radius;
  ctx->uni_max = max_keypoints;
  return BRISK_HIP_OK;
}

int brisk_hip_set_bucketing(brisk_hip_ctx* ctx, int num_buckets_u, int num_buckets_v, int max_keypoints) {
  if (!ctx) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (num_buckets_u == 0 && num_buckets_v == 0) { ctx->bk_u = ctx->bk_v = ctx->bk_max = 0; return BRISK_HIP_OK; }
  // key-point-bucketing-inl.h:78-86: at least one bucket each way, at least one keypoint
  if (num_buckets_u < 1 || num_buckets_v < 1 || max_keypoints < 1 || (long)num_buckets_u * num_buckets_v > (1 << 24))
    return fail(ctx, BRISK_HIP_ERR_ARG, "bucketing: buckets >= 1 each way (0, 0 = off), max_keypoints >= 1");
  ctx->bk_u = num_buckets_u; ctx->bk_v = num_buckets_v; ctx->bk_max = max_keypoints;
  return BRISK_HIP_OK;
}

// ---- 16-bit image functions (stand-alone: host buffers in, host buffers out) -------------------------------------
static int image16_call(brisk_hip_ctx* ctx, int which, const uint16_t* src, int w, int h, int src_stride, void* dst, int dst_stride) {
  if (!ctx || !src || !dst) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (w <= 0 || h <= 0 || w > 8191 || h > 8191 || src_stride < w) return fail(ctx, BRISK_HIP_ERR_ARG, "bad image description");
  int dw, dh;
  size_t delem;
  if (which == 0) {
    dw = w / 2; dh = h / 2; delem = 2;
    // a single row, or fewer than 16 usable columns: the reference's loops do not run and nothing is written
    // (image-down-sampling.cc:69-74) - the same here, for both degenerate shapes
    if (dh < 1 || dw * 2 < 16) return BRISK_HIP_OK;
  } else if (which == 1) {
    dw = w / 3 * 2; dh = h / 3 * 2; delem = 2;
    // fewer than three rows or fewer than 12 usable columns: nothing to write (image-down-sampling.cc:407-413)
    if (dh < 2 || w / 3 * 3 < 12) return BRISK_HIP_OK;
  } else {
    dw = w + 1; dh = h + 1; delem = 4;
  }
  if (dst_stride < dw) return fail(ctx, BRISK_HIP_ERR_ARG, "destination stride too small");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  // scratch kept by the context (the functions may be called per frame): grown when a call needs more, never shrunk
  const size_t need[3] = {(size_t)w * h * 2, (size_t)dw * dh * delem, which == 2 ? (size_t)w * h * 4 : 0};
  for (int i = 0; i < 3; ++i) {
    if (need[i] <= ctx->img16_bytes[i]) continue;
    HIPCHK(ctx, hipStreamSynchronize(s));
    if (ctx->d_img16[i]) (void)hipFree(ctx->d_img16[i]);
    ctx->d_img16[i] = nullptr; ctx->img16_bytes[i] = 0;
    HIPCHK(ctx, hipMalloc(&ctx->d_img16[i], need[i]));
    ctx->img16_bytes[i] = need[i];
  }
  uint16_t* d_src = static_cast<uint16_t*>(ctx->d_img16[0]);
  void* d_dst = ctx->d_img16[1];
  float* d_tmp = static_cast<float*>(ctx->d_img16[2]);
  HIPCHK(ctx, hipMemcpy2DAsync(d_src, (size_t)w * 2, src, (size_t)src_stride * 2, (size_t)w * 2, h, hipMemcpyHostToDevice, s));
  if (which == 0) brisk_launch_halfsample16(d_src, w, w, h, (uint16_t*)d_dst, dw, s);
  else if (which == 1) brisk_launch_twothirdsample16(d_src, w, w, h, (uint16_t*)d_dst, dw, s);
  else brisk_launch_integral16(d_src, w, w, h, d_tmp, (float*)d_dst, dw, s);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpy2DAsync(dst, (size_t)dst_stride * delem, d_dst, (size_t)dw * delem, (size_t)dw * delem, dh, hipMemcpyDeviceToHost, s));
  HIPCHK(ctx, hipStreamSynchronize(s));
  return BRISK_HIP_OK;
}
int brisk_hip_halfsample16(brisk_hip_ctx* ctx, const uint16_t* src, int w, int h, int src_stride, uint16_t* dst, int dst_stride) {
  return image16_call(ctx, 0, src, w, h, src_stride, dst, dst_stride);
}
int brisk_hip_twothirdsample16(brisk_hip_ctx* ctx, const uint16_t* src, int w, int h, int src_stride, uint16_t* dst, int dst_stride) {
  return image16_call(ctx, 1, src, w, h, src_stride, dst, dst_stride);
}
int brisk_hip_integral_image16(brisk_hip_ctx* ctx, const uint16_t* src, int w, int h, int src_stride, float* dst, int dst_stride) {
  return image16_call(ctx, 2, src, w, h, src_stride, dst, dst_stride);
}

int brisk_hip_profile_enable(brisk_hip_ctx* ctx, int enable) {
  if (!ctx) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->prof.on = enable != 0;
  ctx->prof.calls = 0;
  return BRISK_HIP_OK;
}

int brisk_hip_profile_stages(void) { return BRISK_PROF_STAGES; }
const char* brisk_hip_profile_stage_name(int i) { return brisk_stage_name(i); }

int brisk_hip_set_streams(brisk_hip_ctx* ctx, int n) {
  if (!ctx || n < 1 || n > 8) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->nsub = n;
  return BRISK_HIP_OK;
}

int brisk_hip_profile_frames_per_launch(brisk_hip_ctx* ctx) { return ctx ? ctx->last_frames_per_launch : 0; }

int brisk_hip_profile_read(brisk_hip_ctx* ctx, float* avg_ms, int* calls) {
  if (!ctx || !avg_ms) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, wait_own_work(ctx));
  BriskProfiler& P = ctx->prof;
  const int n = P.calls < BRISK_PROF_MAX_CALLS ? P.calls : BRISK_PROF_MAX_CALLS;
  for (int k = 0; k < BRISK_PROF_STAGES; ++k) {
    double sum = 0;
    int cnt = 0;
    for (int c = 0; c < n; ++c) {
      if (!P.used[c][k] || !P.used[c][k + 1]) continue;
      float ms = 0;
      if (hipEventElapsedTime(&ms, P.ev[c][k], P.ev[c][k + 1]) == hipSuccess) { sum += ms; cnt++; }
    }
    avg_ms[k] = cnt ? (float)(sum / cnt) : 0.f;
  }
  {  // the integral kernel of an overlapped batch ran on the side stream: its duration there replaces the (empty)
     // interval on the launch stream
    double sum = 0;
    int cnt = 0;
    for (int c = 0; c < n; ++c) {
      if (!P.side_used[c]) continue;
      float ms = 0;
      if (hipEventElapsedTime(&ms, P.side_ev[c][0], P.side_ev[c][1]) == hipSuccess) { sum += ms; cnt++; }
    }
    if (cnt) avg_ms[BRISK_STG_INTEGRAL] = (float)(sum / cnt);
  }
  if (calls) *calls = n;
  P.calls = 0;
  return BRISK_HIP_OK;
}

#ifndef BRISK_KERNEL_REV
#define BRISK_KERNEL_REV "unversioned"
#endif
const char* brisk_hip_kernel_revision(void) { return BRISK_KERNEL_REV; }

int brisk_hip_stream_ceiling(brisk_hip_ctx* ctx, size_t bytes, double* copy_GBps, double* read_GBps) {
  if (!ctx || bytes < (1u << 20)) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  void *a = nullptr, *b = nullptr;
  HIPCHK(ctx, hipMalloc(&a, bytes));
  if (hipMalloc(&b, bytes) != hipSuccess) { (void)hipFree(a); return fail(ctx, BRISK_HIP_ERR_HIP, "hipMalloc failed"); }
  (void)hipMemset(a, 1, bytes);
  (void)hipMemset(b, 2, bytes);
  (void)hipDeviceSynchronize();  // (null-stream memsets are not ordered against the context's non-blocking stream)
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  double best[2] = {0, 0};
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipEventRecord(e0, ctx->stream);
      brisk_launch_stream_probe(a, b, bytes, mode, ctx->stream);
      (void)hipEventRecord(e1, ctx->stream);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double gbps = (mode ? 1.0 : 2.0) * (double)bytes / (ms * 1e-3) / 1e9;
      if (rep && gbps > best[mode]) best[mode] = gbps;
    }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(a);
  (void)hipFree(b);
  if (copy_GBps) *copy_GBps = best[0];
  if (read_GBps) *read_GBps = best[1];
  HIPCHK(ctx, hipGetLastError());
  return BRISK_HIP_OK;
}

int brisk_hip_set_integral_format(brisk_hip_ctx* ctx, int format) {
  if (!ctx) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (format != BRISK_HIP_INTEGRAL_AUTO && format != BRISK_HIP_INTEGRAL_U24 && format != BRISK_HIP_INTEGRAL_U32)
    return fail(ctx, BRISK_HIP_ERR_ARG, "integral format must be BRISK_HIP_INTEGRAL_AUTO, _U24 or _U32");
  ctx->integral_fmt = format;
  return BRISK_HIP_OK;
}

// ---- test / tuning entry points (include/brisk_hip_debug.h): BRISK_HIP_TUNING builds only -------------------------------
#ifdef BRISK_HIP_TUNING
int brisk_hip_debug_forge_pattern_device(brisk_hip_pattern* p, int device) {
  if (!p) return BRISK_HIP_ERR_ARG;
  p->device = device < 0 ? p->true_device : device;
  return BRISK_HIP_OK;
}

int brisk_hip_debug_set_flags(brisk_hip_ctx* ctx, int flags) {
  if (!ctx) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->debug_flags = flags;
  return BRISK_HIP_OK;
}

// ---- debug / per-stage parity ------------------------------------------------------------------
int brisk_hip_debug_layer(brisk_hip_ctx* ctx, int frame, int layer, int which, uint8_t* out, int* w, int* h) {
  if (!ctx || !w || !h) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (frame < 0 || frame >= ctx->slots || layer < 0 || layer >= ctx->G.nlayers) return fail(ctx, BRISK_HIP_ERR_ARG, "bad index");
  const BriskLayerGeom& L = ctx->G.L[layer];
  *w = L.w; *h = L.h;
  if (!out) return BRISK_HIP_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, wait_own_work(ctx));
  const size_t base = (size_t)frame * ctx->G.pyr_elems + L.off;
  if (which == 0) {
    // (layer 0 of a batch that read the caller's frames in place: the caller's buffer must still be alive)
    const uint8_t* src = (layer == 0 && ctx->last_l0_ext) ? ctx->last_l0_ext + (size_t)frame * ctx->last_l0_pitch : ctx->B.pyr + base;
    HIPCHK(ctx, hipMemcpy2D(out, L.w, src, L.stride, L.w, L.h, hipMemcpyDeviceToHost));
  } else {
    std::vector<uint16_t> tmp((size_t)L.stride * L.h);
    HIPCHK(ctx, hipMemcpy(tmp.data(), ctx->B.smap + base, tmp.size() * 2, hipMemcpyDeviceToHost));
    for (int y = 0; y < L.h; ++y)
      for (int x = 0; x < L.w; ++x) {
        const uint16_t v = tmp[(size_t)y * L.stride + x];
        out[(size_t)y * L.w + x] = (which == 1) ? (uint8_t)(v & 0xFF) : (uint8_t)(v >> 8);
      }
  }
  return BRISK_HIP_OK;
}

int brisk_hip_debug_image_reuse(brisk_hip_ctx* ctx) { return ctx ? ctx->img_cache.hits + ctx->image_reuse_multi : 0; }

// the uniformity filter alone on a given keypoint list (parity tests of the filter kernels on lists no detector produces)
int brisk_hip_debug_filter_keypoints(brisk_hip_ctx* ctx, const brisk_hip_keypoint* in, int n_in, int rows, int cols, double radius,
                                     int max_keypoints, brisk_hip_keypoint* out, int* n) {
  if (!ctx || !in || !out || !n || n_in < 0 || rows <= 0 || cols <= 0 || rows > 8191 || cols > 8191 || radius < 1.0 || max_keypoints < 1)
    return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n_in > ctx->kp_cap) return fail(ctx, BRISK_HIP_ERR_CAPACITY, "more keypoints than the configured capacity");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  make_geometry(cols, rows, 20, 0, &ctx->G, &ctx->T);
  int rc = ensure_buffers(ctx, 1, ctx->G);
  if (rc) return rc;
  rc = ensure_filter_buffers(ctx, cols, rows, 1, radius);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  if (workspace_acquire(ctx, s)) return fail(ctx, BRISK_HIP_ERR_HIP, "hipStreamWaitEvent failed");
  WorkspaceGuard guard(ctx, s);
  ctx->img_cache.valid = false;
  BriskFrameCounters c0;
  memset(&c0, 0, sizeof(c0));
  c0.nkp = n_in;
  HIPCHK(ctx, hipMemcpyAsync(ctx->B.counters, &c0, sizeof(c0), hipMemcpyHostToDevice, s));
  if (n_in) HIPCHK(ctx, hipMemcpyAsync(ctx->B.kp_out, in, sizeof(BriskKeyPoint) * (size_t)n_in, hipMemcpyHostToDevice, s));
  HIPCHK(ctx, hipStreamSynchronize(s));  // (c0 lives on this stack frame)
  const float scaling = (float)(15.0 / (float)radius);
  const int oh = (int)(rows * ceil(scaling) + 32), ow = (int)(cols * ceil(scaling) + 32);
  const long occ_frame = ((long)oh * ow + 255) / 256 * 256;
  brisk_launch_uniformity(ctx->B.kp_out, ctx->B.counters, ctx->d_uni_order, ctx->d_uni_tmp, ctx->d_occ, occ_frame, ow, ctx->B.kp_cap, scaling,
                          max_keypoints, 1, s);
  HIPCHK(ctx, hipGetLastError());
  ctx->last_nframes = 1;
  ctx->last_has_desc = false;
  if (guard.release()) return fail(ctx, BRISK_HIP_ERR_HIP, "hipEventRecord failed");
  return download_locked(ctx, 0, 0, out, n_in, n, nullptr, 0, 0);
}

int brisk_hip_debug_integral(brisk_hip_ctx* ctx, int frame, uint32_t* out) {
  if (!ctx || !out) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (frame < 0 || frame >= ctx->slots) return fail(ctx, BRISK_HIP_ERR_ARG, "bad index");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, wait_own_work(ctx));
  const uint32_t* src = ctx->D.integral + (size_t)frame * ctx->D.iframe_elems;
  BriskFrameCounters fc;
  HIPCHK(ctx, hipMemcpy(&fc, ctx->B.counters + frame, sizeof(fc), hipMemcpyDeviceToHost));
  if (fc.i24) {  // 3-byte elements (values modulo 2^24), row pitch istride * 3 bytes: zero-extended
    const int wi = ctx->G.w + 1, hi = ctx->G.h + 1;
    std::vector<uint8_t> tmp((size_t)wi * 3 * hi);
    HIPCHK(ctx, hipMemcpy2D(tmp.data(), (size_t)wi * 3, src, (size_t)ctx->D.istride * 3, (size_t)wi * 3, hi, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < (size_t)wi * hi; ++i) out[i] = (uint32_t)tmp[3 * i] | ((uint32_t)tmp[3 * i + 1] << 8) | ((uint32_t)tmp[3 * i + 2] << 16);
    return BRISK_HIP_OK;
  }
  HIPCHK(ctx, hipMemcpy2D(out, (size_t)(ctx->G.w + 1) * 4, src, (size_t)ctx->D.istride * 4, (size_t)(ctx->G.w + 1) * 4, ctx->G.h + 1,
                          hipMemcpyDeviceToHost));
  return BRISK_HIP_OK;
}

int brisk_hip_debug_integral_bits(brisk_hip_ctx* ctx, int frame) {
  if (!ctx) return 0;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (frame < 0 || frame >= ctx->slots || !ctx->B.counters) return 0;
  if (hipSetDevice(ctx->device) != hipSuccess || wait_own_work(ctx) != hipSuccess) return 0;
  BriskFrameCounters fc;
  if (hipMemcpy(&fc, ctx->B.counters + frame, sizeof(fc), hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return fc.i24 ? 24 : 32;
}

// per-frame work counts of the last batch: out[0] = candidates, out[1] = keypoints, out[2] = described keypoints,
// out[3] = overflow flags, out[4 ..] = tie candidates per layer (nlayers entries; returns nlayers in *nlayers), out[20 .. 27] = experiment words
int brisk_hip_debug_counters(brisk_hip_ctx* ctx, int frame, int* out, int* nlayers) {
  if (!ctx || !out || !nlayers) return BRISK_HIP_ERR_ARG;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (frame < 0 || frame >= ctx->slots || !ctx->B.counters) return fail(ctx, BRISK_HIP_ERR_ARG, "bad index");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, wait_own_work(ctx));
  BriskFrameCounters c;
  HIPCHK(ctx, hipMemcpy(&c, ctx->B.counters + frame, sizeof(c), hipMemcpyDeviceToHost));
  out[0] = c.ncand; out[1] = c.nkp; out[2] = c.ndesc; out[3] = c.overflow;
  *nlayers = ctx->G.nlayers;
  for (int l = 0; l < ctx->G.nlayers; ++l) out[4 + l] = c.ntie[l];
  for (int i = 0; i < 2; ++i) out[20 + i] = c.pad[i];
  out[22] = c.i24;
  out[23] = c.low_score;
  out[25] = c.nestimate;
  out[26] = c.orient_ticket; out[27] = c.desc_ticket;
  return BRISK_HIP_OK;
}

// experiments: the raw counter record of a frame (build variants append fields); returns its size
int brisk_hip_debug_counters_raw(brisk_hip_ctx* ctx, int frame, void* out, int bytes) {
  if (!ctx || !out) return -1;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (frame < 0 || frame >= ctx->slots || !ctx->B.counters || bytes < (int)sizeof(BriskFrameCounters)) return -1;
  if (hipSetDevice(ctx->device) != hipSuccess || wait_own_work(ctx) != hipSuccess) return -1;
  if (hipMemcpy(out, ctx->B.counters + frame, sizeof(BriskFrameCounters), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int)sizeof(BriskFrameCounters);
}

#endif  // BRISK_HIP_TUNING

}  // extern "C"

#include "brisk_pool.inc"
