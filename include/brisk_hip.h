/*
 * brisk_hip.h - C ABI of the MI355X-native BRISK detect+describe engine (libbrisk_hip.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  The two host classes
 * brisk::BriskFeatureDetector / brisk::BriskDescriptorExtractor (include/brisk/, mirroring the
 * reference headers) are thin wrappers over these entry points, and a maintainer of the reference
 * would bind exactly these from brisk-feature-detector.cc / brisk-descriptor-extractor.cc
 * (see INTEGRATION.md).  Reference paths below are relative to the ethzasl_brisk tree.
 *
 * All functions return BRISK_HIP_OK (0) or an error code; brisk_hip_last_error() gives the text.
 * There is no CPU fallback: without a HIP device every compute entry point fails with
 * BRISK_HIP_ERR_NO_DEVICE.
 */
#ifndef BRISK_HIP_H_
#define BRISK_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  BRISK_HIP_OK = 0,
  BRISK_HIP_ERR_ARG = 1,        /* null pointer / non-positive size / bad enum */
  BRISK_HIP_ERR_NO_DEVICE = 2,  /* no usable HIP device */
  BRISK_HIP_ERR_HIP = 3,        /* a HIP runtime call failed */
  BRISK_HIP_ERR_CAPACITY = 4,   /* more candidates / keypoints than the configured capacity */
  BRISK_HIP_ERR_THRESHOLD = 5,  /* AGAST threshold outside [1, 255] (below 20 some frames run the ordered path) */
  BRISK_HIP_ERR_PATTERN = 6,    /* malformed pattern (reference: CHECK_EQ(noShortPairs_, 384), :286) */
  BRISK_HIP_ERR_UNSUPPORTED = 7 /* no defined result in the reference on this input (see brisk_hip_detect), or an unsupported size */
};

/* Binary-identical to cv::KeyPoint {Point2f pt; float size, angle, response; int octave, class_id;} */
typedef struct brisk_hip_keypoint {
  float x, y, size, angle, response;
  int octave, class_id;
} brisk_hip_keypoint;

typedef struct brisk_hip_ctx brisk_hip_ctx;          /* device workspace + stream          */
typedef struct brisk_hip_pattern brisk_hip_pattern;  /* sampling pattern tables (extractor) */

/* ---- context ------------------------------------------------------------------------------- */
/* device: HIP device ordinal.  Buffers are sized lazily for the largest (w, h, octaves, batch) seen.  The calls of one
 * context are serialised (mutex) and share one workspace: every call orders its stream behind the previous call's work,
 * whichever stream that ran on.  For concurrency use one context per host thread (what include/brisk/hip-context.h
 * does); pattern handles are plain device tables and may be shared by all contexts of a device. */
int brisk_hip_create(int device, brisk_hip_ctx** out);
void brisk_hip_destroy(brisk_hip_ctx* ctx);
const char* brisk_hip_last_error(const brisk_hip_ctx* ctx);
/* per-frame capacities: AGAST candidates (default 65536) and keypoints (default 16384; below 2^23) */
int brisk_hip_set_capacity(brisk_hip_ctx* ctx, int max_candidates, int max_keypoints);
int brisk_hip_device_count(void);
/* CPUs the process may use (affinity mask, cut by a cgroup quota).  A one-frame call POLLS for its results while fewer threads
 * than that are polling (a wake-up costs more than the call's tail) and sleeps on a blocking event otherwise; the classes of
 * include/brisk/ hand calls to the device's shared pool from more concurrent callers than CPUs on. */
int brisk_hip_usable_cpus(void);
/* raises the capacities to at least these values; never lowers them (no reallocation for smaller requests) */
int brisk_hip_reserve(brisk_hip_ctx* ctx, int min_candidates, int min_keypoints);

/* Page-locks a caller's host buffer (an image, a frame ring, result arrays) so that the device reads / writes it directly:
 * hipHostRegister / hipHostUnregister for callers that do not link the HIP runtime themselves.  A registered image is
 * uploaded by DMA straight from the buffer (a pageable one goes through the runtime's staging copies, on the calling
 * thread); registered result arrays make brisk_hip_batch_download_all write them without a bounce buffer.  The buffer must
 * be unregistered before it is freed.  Registration costs ~0.1 ms per MB: register buffers that are reused. */
int brisk_hip_host_register(void* ptr, size_t bytes);
int brisk_hip_host_unregister(void* ptr);

/* ---- pattern: replaces the BriskDescriptorExtractor constructors ---------------------------- */
/* brisk-descriptor-extractor.cc:293-343: version 2 = built-in 66-point pattern (InitFromStream :180-291),
 * version 1 = generated 60-point BRISK 1.0 kernel (generateKernel :65-178, 512 bits). */
int brisk_hip_pattern_create(brisk_hip_ctx* ctx, int version, float pattern_scale, brisk_hip_pattern** out);
/* :345-367: pattern file contents (.ptn syntax: N, N x {x y sigma}, S, S x {i j}, L, L x {i j}) */
int brisk_hip_pattern_create_from_text(brisk_hip_ctx* ctx, const char* ptn_text, float pattern_scale,
                                       brisk_hip_pattern** out);
void brisk_hip_pattern_destroy(brisk_hip_pattern* p);
int brisk_hip_pattern_descriptor_size(const brisk_hip_pattern* p); /* descriptorSize() :780-782 (48; 64 for briskV1; 16 ... 224 for briskV1 at other pattern scales) */
int brisk_hip_pattern_points(const brisk_hip_pattern* p);
/* host copies of the derived tables, for inspection / tests: 64 entries each */
int brisk_hip_pattern_tables(const brisk_hip_pattern* p, float* scale_list, int* size_list, float* size_thresh);

/* ---- host-buffer calls: what the two host classes forward to --------------------------------- */
/* BriskFeatureDetector::detectImpl (brisk-feature-detector.cc:77-85): clears/overwrites `out`.
 * img: h x w u8, row pitch `stride` bytes.  mask: optional h x w u8 (0 = drop keypoint), or NULL.
 * threshold: 1..255 (20..255 on the fast path; below 20 a frame in which a detection stores a score <= 2 - nearly every
 * frame below 10, few at 10..19 - takes the sequential ordered path, bit-exact but slow: ~19 us per AGAST candidate).
 * suppress_scale_nonmaxima = 0 (brisk-scale-space.cc:131-170): with octaves == 0 the single-layer 2-D refinement; with
 * more layers the reference takes every layer's point coordinates from layer 0's list (`agastPoints.at(0)[n]`, :137) -
 * reproduced on the ordered path; where that indexing leaves layer 0's list or a score matrix (the usual case for
 * ordinary images: undefined behaviour in the reference) the call fails with BRISK_HIP_ERR_UNSUPPORTED.
 * out: capacity `cap` keypoints; *n receives the count (BRISK_HIP_ERR_CAPACITY if cap is too small). */
int brisk_hip_detect(brisk_hip_ctx* ctx, const uint8_t* img, int w, int h, int stride, int threshold, int octaves,
                     int suppress_scale_nonmaxima, const uint8_t* mask, int mask_stride, brisk_hip_keypoint* out,
                     int cap, int* n);
/* The same call with the uniformity post-filter (see brisk_hip_set_uniformity) given per call instead of as context
 * state: uniformity_radius 0 = off, else >= 1; at most max_keypoints keypoints are kept.  Nothing of the context's
 * settings applies to this call: with radius 0 it returns the unfiltered detections even if brisk_hip_set_uniformity /
 * brisk_hip_set_bucketing are set on the context (those are for the batch path and brisk_hip_detect). */
int brisk_hip_detect_uniform(brisk_hip_ctx* ctx, const uint8_t* img, int w, int h, int stride, int threshold, int octaves,
                             int suppress_scale_nonmaxima, const uint8_t* mask, int mask_stride, double uniformity_radius,
                             int max_keypoints, brisk_hip_keypoint* out, int cap, int* n);
/* The same call with BOTH optional post-filters given per call: nothing is read from or written to the context's settings
 * (brisk_hip_set_uniformity / brisk_hip_set_bucketing), so detector objects with different settings can share a
 * context.  uniformity_radius 0 = off, else >= 1; bucketing takes effect while uniformity is off, (0, 0) buckets = off. */
typedef struct brisk_hip_postfilter {
  double uniformity_radius;
  int uniformity_max_keypoints;
  int num_buckets_u, num_buckets_v, bucket_max_keypoints;
} brisk_hip_postfilter;
int brisk_hip_detect_filtered(brisk_hip_ctx* ctx, const uint8_t* img, int w, int h, int stride, int threshold, int octaves,
                              int suppress_scale_nonmaxima, const uint8_t* mask, int mask_stride,
                              const brisk_hip_postfilter* pf, brisk_hip_keypoint* out, int cap, int* n);
/* BriskFeatureDetector::ComputeScale (brisk-feature-detector.cc:87-92; brisk-scale-space.cc:104-123 and the branches
 * behind it): scores / scales for PROVIDED keypoints on a pyramid with lower threshold 0.  Every provided keypoint yields
 * up to one output per layer that admits it (layer order, provided order inside a layer; class_id is kept).  Sequential
 * on the device - one lane per (layer, provided point): the walk's phases are order-free among themselves (3 000 points on a
 * 1080p frame: 0.6 ms); a call in which some layer admits no provided point, and therefore detects, runs the reference's
 * sequential walk on one lane -, parity unpinned (nothing in the reference exercises this entry).
 * BRISK_HIP_ERR_UNSUPPORTED where the reference has no defined result: a provided point in the last admitted rows of a
 * layer (within about 5 rows x the layer's scale of the bottom border) makes the reference read beyond the image.
 * in: n_in keypoints (only x, y and the copied-through fields matter); out: capacity cap; *n receives the count. */
int brisk_hip_compute_scale(brisk_hip_ctx* ctx, const uint8_t* img, int w, int h, int stride, int threshold, int octaves,
                            int suppress_scale_nonmaxima, const brisk_hip_keypoint* in, int n_in, brisk_hip_keypoint* out,
                            int cap, int* n);
/* BriskDescriptorExtractor::compute (brisk-descriptor-extractor.cc:612-778): filters `kps` in place
 * (border test), fills kps[i].angle, writes *n rows of descriptorSize() bytes at pitch desc_stride. */
int brisk_hip_describe(brisk_hip_ctx* ctx, const brisk_hip_pattern* pat, const uint8_t* img, int w, int h, int stride,
                       brisk_hip_keypoint* kps, int* n, uint8_t* desc, int desc_stride, int rotation_invariant,
                       int scale_invariant);
/* The same call for the usual detect() -> compute() pair on one image (test-binary-equal.cc:215,237): the caller STATES
 * that `img` is the buffer the context's last brisk_hip_detect / _detect_uniform call was given and that its pixels have
 * not changed since.  The upload and the layer-0 pass are then skipped (the device still holds the image).  If pointer,
 * size or stride differ from that call's, or another call used the context in between, the image is uploaded as in
 * brisk_hip_describe.  A caller whose pixels DID change gets the descriptors of the old pixels: brisk_hip_describe never
 * makes that assumption (it always uploads, as the reference's compute() always reads the current pixels), unless the
 * environment opts in with BRISK_HIP_IMAGE_CACHE=1 - the buffer is then recognised by pointer, size, stride and a 64-bit
 * hash over one 8-byte word per 128 bytes of every row, and a change that avoids every sampled word (a small overlay, rows
 * narrower than 8 + the row's phase) goes unnoticed. */
int brisk_hip_describe_same_image(brisk_hip_ctx* ctx, const brisk_hip_pattern* pat, const uint8_t* img, int w, int h, int stride,
                                  brisk_hip_keypoint* kps, int* n, uint8_t* desc, int desc_stride, int rotation_invariant,
                                  int scale_invariant);

/* ---- device-resident batch path (frames already in HBM; results stay in HBM) ----------------- */
/* d_frames: nframes images, frame f at d_frames + f*frame_pitch, row pitch row_pitch.  Runs
 * detect (threshold, octaves) then describe (pat) for every frame on `stream` (hipStream_t, may be
 * NULL = the context's stream, a non-blocking stream that is NOT ordered with the legacy default stream: pass
 * your own stream if other work has to be ordered with the batch).  Asynchronous: synchronise the stream before
 * reading results.  The frames must stay unchanged until then: when they already have the pyramid's layer-0 layout
 * (row_pitch = width, a multiple of 64; 16-byte aligned base and frame pitch) every kernel reads them in place
 * instead of from a private copy (the reference clones the image, brisk-scale-space.cc:74; the result is the same). */
int brisk_hip_detect_describe_batch(brisk_hip_ctx* ctx, const brisk_hip_pattern* pat, const uint8_t* d_frames,
                                    int nframes, int w, int h, long frame_pitch, int row_pitch, int threshold,
                                    int octaves, void* stream);
/* Element format of the integral image the descriptor kernels build and sample (IntegralImage8,
 * brisk/include/brisk/internal/integral-image.h:56-161).  The result never depends on it - both forms are bit-equal to
 * the reference wherever the 24-bit form is used at all - only the speed does:
 *   BRISK_HIP_INTEGRAL_AUTO (default)  3-byte elements (values modulo 2^24) in detect + describe batches while the
 *        context's PREVIOUS batch was sparse (at most 3 000 AGAST candidates per megapixel), u32 otherwise and for
 *        descriptor-only calls: a stream's batches look alike, and the sparse ones gain from the smaller image.  The
 *        choice of a call therefore depends on the call before it - a benchmark that wants one form says so:
 *   BRISK_HIP_INTEGRAL_U24 / _U32      that form for every call of the context (U24 only where it is exact: patterns
 *        whose boxes cover fewer than 2^24 / 255 pixels - every built-in one; other patterns use u32).
 * (Test / tuning builds - libbrisk_hip.so with BRISK_HIP_TUNING, include/brisk_hip_debug.h - also know BRISK_INTEGRAL_BITS=24 / 32
 * in the environment and debug bits 18 / 24, which take precedence, and report the format of the last call with
 * brisk_hip_debug_integral_bits; the release library has neither.) */
enum { BRISK_HIP_INTEGRAL_AUTO = 0, BRISK_HIP_INTEGRAL_U24 = 24, BRISK_HIP_INTEGRAL_U32 = 32 };
int brisk_hip_set_integral_format(brisk_hip_ctx* ctx, int format);
/* Host-fed form of the batch (the PCIe-fed stream): h_frames is HOST memory (pinned with hipHostMalloc /
 * hipHostRegister for full speed; pageable memory works but copies synchronously).  The frames are moved in slices of 64
 * into two device staging buffers on a copy stream while the previous slice is computed on the context's stream; the
 * results stay in HBM exactly as after brisk_hip_detect_describe_batch (brisk_hip_batch_results / _download /
 * _status).  Returns when everything is queued; brisk_hip_batch_status synchronises.
 * LIFETIME: the copy stream keeps reading h_frames after the call has returned.  The host frames must stay valid AND
 * unchanged until brisk_hip_batch_status / brisk_hip_batch_download has returned for this batch (or the next call on the
 * context has been synchronised): a caller that refills a pinned ring right after the call gets results from mixed
 * frames, silently. */
int brisk_hip_detect_describe_batch_host(brisk_hip_ctx* ctx, const brisk_hip_pattern* pat, const uint8_t* h_frames,
                                         int nframes, int w, int h, long frame_pitch, int row_pitch, int threshold,
                                         int octaves);
/* detect only / describe only variants of the batch path (roofline + stage timing) */
int brisk_hip_detect_batch(brisk_hip_ctx* ctx, const uint8_t* d_frames, int nframes, int w, int h, long frame_pitch,
                           int row_pitch, int threshold, int octaves, void* stream);
/* Device pointers of the last batch's results.  d_counts: per frame {detected, described} at
 * byte stride *count_stride (ints); keypoints [frame][kp_cap]; descriptors [frame][kp_cap][desc_pitch].  desc_pitch is the
 * pitch of the rows the LAST call wrote - read it after every call, not once: batches write 64-byte rows (more once a
 * pattern with longer descriptors was used on the context), a host-buffer brisk_hip_describe whose destination rows are
 * packed (desc_stride == descriptor size) writes slot 0's rows at that packed pitch (48 for the default pattern). */
int brisk_hip_batch_results(brisk_hip_ctx* ctx, const int** d_detected, const int** d_described, int* count_stride,
                            const brisk_hip_keypoint** d_detected_kps, const brisk_hip_keypoint** d_described_kps,
                            const uint8_t** d_desc, int* kp_cap, int* desc_pitch);
/* Copies one frame's results of the last batch to the host (synchronises). kps/desc may be NULL. */
int brisk_hip_batch_download(brisk_hip_ctx* ctx, int frame, int which /*0 detected, 1 described*/,
                             brisk_hip_keypoint* kps, int cap, int* n, uint8_t* desc, int desc_stride);
/* error / overflow flags of the last batch, OR-ed over frames (0 = clean); synchronises */
int brisk_hip_batch_status(brisk_hip_ctx* ctx, int nframes, int* overflow_flags);

/* ---- the batch path's exit to HOST memory: every frame's results in one asynchronous transfer ------------------------
 * The reference hands a call's results to the caller's std::vector<cv::KeyPoint> and descriptor cv::Mat
 * (brisk-feature-detector.cc:77-85, brisk-descriptor-extractor.cc:601-604); for a batch that is, per frame f, the rows
 * [offsets[f], offsets[f + 1]) of ONE keypoint array and ONE descriptor matrix in the caller's memory - exact prefix sums,
 * no padding rows.  The caller fills in the capacities and the five destination pointers:
 *   counts  [frames]      rows frame f HAS (detected or described keypoints)
 *   flags   [frames]      0 = clean, else the frame's rows are NOT stored: the engine's capacity flags of the frame (bit 0
 *                         candidates, bit 1 ties, bit 2 keypoints: brisk_hip_set_capacity), bit 3 / 4 as brisk_hip_batch_status,
 *                         BRISK_HIP_ROWS_CUT = the frame (and every frame behind it) did not fit rows_cap
 *   offsets [frames + 1]  first row of frame f; offsets[frames] = rows stored
 *   kps     [rows_cap]    desc [rows_cap][desc_stride] (NULL: no descriptors; bytes of a stored row behind the descriptor: 0)
 * Destinations in pinned / registered host memory (hipHostMalloc, hipHostRegister, torch pin_memory) are written by the
 * device directly, the stored rows only; pageable destinations go through a pinned buffer of the context and a host copy
 * inside brisk_hip_batch_download_wait.  4-byte aligned pointers, desc_stride a multiple of 4 and >= the descriptor size. */
#define BRISK_HIP_ROWS_CUT 0x100
typedef struct brisk_hip_batch_host_results {
  int frames_cap;          /* frames the arrays hold: >= the frames of the batch */
  int desc_stride;         /* bytes between descriptor rows */
  long long rows_cap;      /* rows kps / desc hold */
  int* counts;
  int* flags;
  long long* offsets;
  brisk_hip_keypoint* kps;
  uint8_t* desc;
} brisk_hip_batch_host_results;
/* Queues the transfer of the context's LAST batch (which: 0 detected keypoints, 1 described keypoints + descriptors) behind
 * that batch and returns: two small kernels on `stream` (hipStream_t the batch ran on; NULL = the context's stream) pack the
 * rows into a device slab, the transfer itself runs on the context's second stream (where the integral kernel runs beside the detector's
 * tail; a process has four hardware queues: a stream of its own for the transfer would share one) beside whatever the context does next -
 * the next batch may be issued at once.  *ticket names the transfer.  At most two transfers are in flight per context: a
 * third call first completes the oldest (as brisk_hip_batch_download_wait would).  `dst` (the struct) is copied; the arrays
 * it points to must stay valid until the ticket has been waited for. */
int brisk_hip_batch_download_all(brisk_hip_ctx* ctx, int which, const brisk_hip_batch_host_results* dst, void* stream,
                                 unsigned* ticket);
/* Blocks until transfer `ticket` (and every earlier one) is complete; the context's lock is not held while waiting.
 * *frames_flagged = frames whose flags[] entry is non-zero.  BRISK_HIP_OK, or - with the rows of all clean frames in place -
 * BRISK_HIP_ERR_CAPACITY (some frame hit an engine capacity or was cut) / the code brisk_hip_batch_status would give. */
int brisk_hip_batch_download_wait(brisk_hip_ctx* ctx, unsigned ticket, int* frames_flagged);
/* brisk_hip_detect_describe_batch_host followed by brisk_hip_batch_download_all(which = 1) in one call: frames from host
 * memory in, keypoints + descriptors back in host memory, everything queued when the call returns.  The transfer of
 * batch n overlaps the H2D copies and kernels of batch n + 1 (alternate two destination sets). */
int brisk_hip_detect_describe_batch_host_results(brisk_hip_ctx* ctx, const brisk_hip_pattern* pat, const uint8_t* h_frames,
                                                 int nframes, int w, int h, long frame_pitch, int row_pitch, int threshold,
                                                 int octaves, const brisk_hip_batch_host_results* dst, unsigned* ticket);

/* ---- the multi-image overloads of the reference's base classes, as ONE batch -------------------------------------------------
 * cv::FeatureDetector::detect(const vector<Mat>& images, vector<vector<KeyPoint>>& keypoints, ...) and
 * cv::DescriptorExtractor::compute(const vector<Mat>& images, vector<vector<KeyPoint>>& keypoints, vector<Mat>& descriptors), which the
 * reference's classes inherit (brisk-feature-detector.h:51, brisk-descriptor-extractor.h:54), loop over detectImpl / computeImpl;
 * here the images of a call - separate host buffers of ONE size, given as an array of pointers, row pitch `stride` - go through the
 * batch path and come back through brisk_hip_batch_download_all's destination (`dst`, `ticket`: brisk_hip_batch_download_wait).
 * _detect_images: plain detection (suppressScaleNonmaxima = true, no masks); frame f's keypoints = rows [offsets[f], offsets[f + 1]).
 * _describe_images: kps[f] / nkps[f] = the provided keypoints of image f (not modified); the rows of frame f are its border-filtered
 * keypoints with their angles and the descriptors; same_images != 0 is the caller's word (as brisk_hip_describe_same_image) that images[]
 * are the very buffers of the context's last multi-image call, unchanged: the frames are then taken from their device copies (no second
 * upload); a list that is not that list is uploaded.  images must stay valid until the ticket has been waited for (kps / nkps are
 * copied before the call returns).
 * The drop-in classes' vector overloads forward here (include/brisk/). */
int brisk_hip_detect_images(brisk_hip_ctx* ctx, const uint8_t* const* images, int nimages, int w, int h, int stride, int threshold,
                            int octaves, const brisk_hip_batch_host_results* dst, unsigned* ticket);
int brisk_hip_describe_images(brisk_hip_ctx* ctx, const brisk_hip_pattern* pat, const uint8_t* const* images, int nimages, int w, int h,
                              int stride, const brisk_hip_keypoint* const* kps, const int* nkps, int rotation_invariant,
                              int scale_invariant, int same_images, const brisk_hip_batch_host_results* dst, unsigned* ticket);

/* ---- call combining: the one-frame host calls of MANY threads as batches ------------------------------------------------
 * The reference's classes are re-entrant (detectImpl is const and builds its state per call, brisk-feature-detector.cc:77-85);
 * with a context per thread every call is ~15 kernel launches and 2 - 4 copies of its own, and the HIP runtime serialises the
 * API calls of a process - 16 threads get 3.6 x one thread.  A pool runs the calls that are inside it at the same
 * time as ONE batch: every caller copies its frame into a pinned staging slot (a CPU copy on its own thread), the first
 * caller of a group submits the group - one H2D copy, the batch kernels, one result transfer - and every caller takes its
 * own rows.  Groups close when the device can take them (two are in flight at most): batches grow with the load, a lone
 * caller is a batch of one (slower than brisk_hip_detect by the staging copy - use the pool from about four threads on, as
 * include/brisk/hip-context.h does).  All calls of a pool are thread-safe and blocking; results are bit-identical to
 * brisk_hip_detect / brisk_hip_describe.  max_batch <= 64 frames per group, max_keypoints per frame as brisk_hip_set_capacity. */
typedef struct brisk_hip_pool brisk_hip_pool;
int brisk_hip_pool_create(int device, int max_batch, int max_keypoints, brisk_hip_pool** out);
void brisk_hip_pool_destroy(brisk_hip_pool* pool);
const char* brisk_hip_pool_last_error(const brisk_hip_pool* pool); /* of the calling thread's last failed call */
/* groups run so far and the calls they carried (calls / groups = mean batch size) */
int brisk_hip_pool_stats(brisk_hip_pool* pool, unsigned long long* groups, unsigned long long* calls);
/* brisk_hip_detect (suppressScaleNonmaxima = true, no mask, no post-filter) through the pool.  *image_token (may be NULL)
 * names the device copy of this frame for a following brisk_hip_pool_describe. */
int brisk_hip_pool_detect(brisk_hip_pool* pool, const uint8_t* img, int w, int h, int stride, int threshold, int octaves,
                          brisk_hip_keypoint* out, int cap, int* n, unsigned long long* image_token);
/* brisk_hip_describe through the pool.  image_token: 0, or the token of the brisk_hip_pool_detect call that was given the
 * SAME, unchanged pixels (the caller's word, as brisk_hip_describe_same_image): the frame is then taken from its device
 * copy while the pool still holds it (a stale token is harmless: the frame is uploaded). */
int brisk_hip_pool_describe(brisk_hip_pool* pool, const brisk_hip_pattern* pat, const uint8_t* img, int w, int h, int stride,
                            brisk_hip_keypoint* kps, int* n, uint8_t* desc, int desc_stride, int rotation_invariant,
                            int scale_invariant, unsigned long long image_token);

/* Number of internal streams a batch is sliced over (1..8, default 1: measured no gain from slicing).  The slices fork from / join into the
 * caller's stream with events, so the call stays asynchronous and ordered on that stream. */
int brisk_hip_set_streams(brisk_hip_ctx* ctx, int n);

/* ---- multi-GPU: result gather of the batch path (BASELINE config 3) --------------------------------------
 * Frames are independent units (brisk-feature-detector.cc:77-85 builds all state per call): one process or host thread
 * per GPU, each with its own context, runs brisk_hip_detect_describe_batch on its shard of the frame stream - there is no
 * collective inside detect + describe.  The only exchange is this gather of the results to one rank, on RCCL directly
 * (grouped ncclSend / ncclRecv over xGMI; librccl is opened at run time, BRISK_HIP_ERR_UNSUPPORTED where it is missing).
 *   rank 0:      brisk_hip_comm_unique_id(id); pass the 128 bytes to the other ranks (file, socket, MPI, ...)
 *   every rank:  brisk_hip_comm_create(ctx, rank, world, id, &comm)         (collective: all ranks call it)
 *   per batch:   brisk_hip_detect_describe_batch(ctx, ...); brisk_hip_comm_gather_results(ctx, comm, root, ...)
 *   root:        brisk_hip_comm_wait(comm, stream or NULL) before reading the destination buffers
 * The gather is asynchronous and double-buffered: the rank's rows are packed into a slab on the batch's stream, the
 * transfer runs on the communicator's own stream beside the next batch's kernels. */
#define BRISK_HIP_COMM_ID_BYTES 128
typedef struct brisk_hip_comm brisk_hip_comm;
int brisk_hip_comm_unique_id(uint8_t* id /* BRISK_HIP_COMM_ID_BYTES */);
int brisk_hip_comm_create(brisk_hip_ctx* ctx, int rank, int world, const uint8_t* id, brisk_hip_comm** out);
void brisk_hip_comm_destroy(brisk_hip_comm* comm);
int brisk_hip_comm_rank(const brisk_hip_comm* comm);
int brisk_hip_comm_world(const brisk_hip_comm* comm);
/* Collective over the communicator: every rank contributes the DESCRIBED results of its context's last batch as fixed
 * slabs - per-frame counts [frames_max] (0 for the frames beyond its own batch), keypoints [frames_max][kpad] and
 * descriptors [frames_max][kpad][strings] (the first kpad rows of every frame; strings = descriptor bytes) - and `root`
 * receives them, rank after rank, in d_counts [world][frames_max], d_kps [world][frames_max][kpad],
 * d_desc [world][frames_max][kpad][strings] (device memory of the root's GPU; ignored on the other ranks).  frames_max,
 * kpad and strings must be the same on every rank; a frame with more than kpad keypoints is cut (its count is not: the
 * root can tell).  stream: the stream the batch ran on (NULL = the context's stream). */
int brisk_hip_comm_gather_results(brisk_hip_ctx* ctx, brisk_hip_comm* comm, int root, int frames_max, int kpad, int strings,
                                  int* d_counts, brisk_hip_keypoint* d_kps, uint8_t* d_desc, void* stream);
/* orders `stream` behind every gather issued so far on this rank (NULL: blocks the host until they are done) */
int brisk_hip_comm_wait(brisk_hip_comm* comm, void* stream);

/* ---- optional post-filter: keypoint uniformity enforcement (BASELINE config 4) ---- */
/* EnforceKeyPointUniformity (brisk/include/brisk/internal/uniformity-enforcement-inl.h:44-194, mask LUT
 * scale-space-layer-inl.h:88-97) applied to the detector's keypoints (x, y, response) in every following detect call
 * of this context (host-buffer and batch): keypoints in descending response order are accepted greedily against an
 * occupancy image at scale 15 / radius, at most max_keypoints are kept, the output is in acceptance order.
 * radius = 0 switches it off (default).  In the reference the filter is only wired into the Harris
 * ScaleSpaceFeatureDetector (scale-space-layer-inl.h:372-375), so this is an engine option, not reference behaviour of
 * BriskFeatureDetector; equal responses keep their (layer, y, x) order (the reference's std::sort is unstable). */
int brisk_hip_set_uniformity(brisk_hip_ctx* ctx, double radius, int max_keypoints);
/* KeyPointBucketing (brisk/include/brisk/internal/key-point-bucketing-inl.h:40-112; what the reference's
 * ScaleSpaceLayer uses when uniformity enforcement is off, scale-space-layer-inl.h:372-378), applied like the uniformity
 * filter to the detector's keypoints of every following detect call while uniformity is off: keypoints in descending
 * response order; with one bucket in either direction the best max_keypoints are kept, otherwise a keypoint is kept while
 * its bucket of (1 + (cols - 1) / num_buckets_u) x (1 + (rows - 1) / num_buckets_v) pixels holds fewer than
 * max_keypoints / (num_buckets_u * num_buckets_v).  Output in descending response order - except with one bucket in either
 * direction and no more than max_keypoints keypoints: the reference then leaves the vector untouched (:87-88), and so
 * does the engine (detector order).  (0, 0, *) switches it off.
 * The reference requires num_buckets_u < cols and num_buckets_v < rows (CHECK_LT, :82-83): checked per call. */
int brisk_hip_set_bucketing(brisk_hip_ctx* ctx, int num_buckets_u, int num_buckets_v, int max_keypoints);

/* ---- 16-bit image functions (not on the 8-bit detect + describe path) --------------- */
/* Halfsample16 (brisk/src/image-down-sampling.cc:56-139): dst is (w / 2) x (h / 2); Twothirdsample16 (:394-548): dst is
 * (w / 3 * 2) x (h / 3 * 2); IntegralImage16 (brisk/include/brisk/internal/integral-image.h:163-218): dst is
 * (h + 1) x (w + 1) floats.  Host buffers, strides in ELEMENTS; the reference's arithmetic bit for bit (saturating adds,
 * signed pack to 32767, float sums in the reference's order incl. its unscaled last 0..3 columns).  Images with fewer
 * than 16 (half) / 12 (two thirds) usable columns or without an output row, where the reference's loops write nothing:
 * BRISK_HIP_OK, dst untouched. */
int brisk_hip_halfsample16(brisk_hip_ctx* ctx, const uint16_t* src, int w, int h, int src_stride, uint16_t* dst, int dst_stride);
int brisk_hip_twothirdsample16(brisk_hip_ctx* ctx, const uint16_t* src, int w, int h, int src_stride, uint16_t* dst, int dst_stride);
int brisk_hip_integral_image16(brisk_hip_ctx* ctx, const uint16_t* src, int w, int h, int src_stride, float* dst, int dst_stride);

/* ---- Hamming brute-force matcher (the step after the path) ----------------------- */
/* binary-identical to cv::DMatch */
typedef struct brisk_hip_dmatch {
  int queryIdx, trainIdx, imgIdx;
  float distance;
} brisk_hip_dmatch;
/* BruteForceMatcher::knnMatchImpl -> commonKnnMatchImpl (brisk/src/brute-force-matcher.cc:54-64, 80-162) with
 * brisk::Hamming (brisk/include/brisk/internal/hamming.h:98-112: popcount of a ^ b over dim_bytes / 16 128-bit
 * words; dim_bytes 16 ... 224).  query: nq rows of dim_bytes at pitch q_pitch; train[i]: ntrain[i] rows at pitch t_pitch[i] (the
 * trainDescCollection, nimg images).  masks: NULL, or nimg pointers (each NULL or an nq x ntrain[i] u8 matrix at
 * pitch mask_pitch[i]; 0 = pair not allowed).  out: nq * k matches, row q at out + q * k, sorted by
 * (distance, imgIdx, trainIdx); out_count[q] = entries of row q (0 for a masked-out query; rows with fewer
 * than k possible matches are topped up exactly as the reference does).
 * All pointers are host pointers. */
int brisk_hip_match_knn(brisk_hip_ctx* ctx, const uint8_t* query, int nq, int q_pitch, int dim_bytes, int nimg,
                        const uint8_t* const* train, const int* ntrain, const int* t_pitch,
                        const uint8_t* const* masks, const int* mask_pitch, int k, brisk_hip_dmatch* out,
                        int* out_count);
/* BruteForceMatcher::radiusMatchImpl -> commonRadiusMatchImpl (:66-78, 164-213): every pair with
 * distance < max_distance.  out: nq rows of cap_per_query entries; out_count[q] = matches FOUND for query q
 * (only the first cap_per_query of them, in (distance, imgIdx, trainIdx) order, are stored). */
int brisk_hip_match_radius(brisk_hip_ctx* ctx, const uint8_t* query, int nq, int q_pitch, int dim_bytes, int nimg,
                           const uint8_t* const* train, const int* ntrain, const int* t_pitch,
                           const uint8_t* const* masks, const int* mask_pitch, float max_distance,
                           int cap_per_query, brisk_hip_dmatch* out, int* out_count);
/* Device-resident form used by pipelines that keep descriptors in HBM (e.g. the rows brisk_hip_batch_results
 * returns): one train set, no masks, all pointers are device pointers; asynchronous on `stream` (hipStream_t,
 * NULL = the context's stream). */
int brisk_hip_match_knn_device(brisk_hip_ctx* ctx, const uint8_t* d_query, int nq, int q_pitch, const uint8_t* d_train,
                               int nt, int t_pitch, int dim_bytes, int k, brisk_hip_dmatch* d_out, int* d_out_count,
                               void* stream);

/* ---- per-stage timing: HIP events recorded on the launch stream around every kernel of the batch path ---- */
int brisk_hip_profile_enable(brisk_hip_ctx* ctx, int enable);       /* resets the accumulated calls */
int brisk_hip_profile_stages(void);                                 /* number of stages */
const char* brisk_hip_profile_stage_name(int stage);
/* average milliseconds per stage over the calls since enable/read (at most the last 64); synchronises */
int brisk_hip_profile_read(brisk_hip_ctx* ctx, float* avg_ms, int* calls);
/* frames handled by each timed kernel launch (= frames of the first stream slice of the last batch) */
int brisk_hip_profile_frames_per_launch(brisk_hip_ctx* ctx);

/* The box's own streaming ceiling, measured with the engine's float4 kernels over two buffers of `bytes` each:
 * a device-to-device copy (read + write bytes per second) and a read-only pass.  Reported by bench.py next to the
 * roofline numbers (the spec peak is never reached by any kernel). */
int brisk_hip_stream_ceiling(brisk_hip_ctx* ctx, size_t bytes, double* copy_GBps, double* read_GBps);
/* identifies the kernel sources this library was built from (hash); committed PMC traffic data names the revision
 * it was measured on */
const char* brisk_hip_kernel_revision(void);

#ifdef __cplusplus
}
#endif
#endif /* BRISK_HIP_H_ */
