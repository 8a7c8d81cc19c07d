// brisk/hip-context.h - per-thread device context used by the BRISK host classes.
#ifndef BRISK_HIP_CONTEXT_H_
#define BRISK_HIP_CONTEXT_H_

#include <brisk_hip.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>

namespace brisk {
namespace hip {

// Converts a C-ABI status into the reference's error behaviour: the reference aborts through glog
// CHECKs or throws std::runtime_error (brisk-descriptor-extractor.cc:341,678); here every failure is a
// std::runtime_error carrying the engine's message.
inline void Check(brisk_hip_ctx* ctx, int rc, const char* what) {
  if (rc == BRISK_HIP_OK) return;
  std::string msg = std::string(what) + " failed (code " + std::to_string(rc) + "): " +
                    (ctx ? brisk_hip_last_error(ctx) : "no HIP device / context");
  throw std::runtime_error(msg);
}

// One workspace per THREAD: the reference's classes are re-entrant (detectImpl is const and builds its state per call,
// brisk-feature-detector.cc:77-85), so concurrent calls from several threads must not serialise on one workspace or race
// on its settings.  Every thread that uses the classes gets its own context (stream + lazily sized buffers, ~30 MB per
// 1080p frame slot), destroyed when the thread ends.  Pattern tables (BriskDescriptorExtractor) are plain device memory
// and are shared by all contexts of the device.
// Device of a thread's context: SetThreadDevice(d) (one host thread per GPU of a node: call it first thing in the
// thread), else the BRISK_HIP_DEVICE environment variable, else device 0.
inline int& ThreadDeviceRequest() {
  static thread_local int dev = -1;
  return dev;
}
struct ThreadContext {
  brisk_hip_ctx* ctx = nullptr;
  int rc = BRISK_HIP_OK;
  int device = 0;
  ThreadContext() {
    const char* dev = std::getenv("BRISK_HIP_DEVICE");
    device = ThreadDeviceRequest() >= 0 ? ThreadDeviceRequest() : (dev ? std::atoi(dev) : 0);
    rc = brisk_hip_create(device, &ctx);
  }
  ~ThreadContext() {
    if (ctx) brisk_hip_destroy(ctx);
  }
  void Recreate(int dev) {
    if (ctx) brisk_hip_destroy(ctx);
    ctx = nullptr;
    device = dev;
    rc = brisk_hip_create(device, &ctx);
  }
  ThreadContext(const ThreadContext&) = delete;
  ThreadContext& operator=(const ThreadContext&) = delete;
};

inline ThreadContext& ThisThread() {
  static thread_local ThreadContext tc;
  return tc;
}

inline brisk_hip_ctx* DefaultContext() {
  ThreadContext& tc = ThisThread();
  if (tc.rc != BRISK_HIP_OK || !tc.ctx) Check(nullptr, tc.rc ? tc.rc : BRISK_HIP_ERR_NO_DEVICE, "brisk_hip_create");
  return tc.ctx;
}

// Selects the GPU of the calling thread's context.  Call it before the thread constructs a BriskDescriptorExtractor or
// uses a detector: extractor objects keep pattern tables on the device of the context they were built with.  A context the
// thread already has on another device is destroyed (its workspace with it) and re-created.
inline void SetThreadDevice(int device) {
  ThreadDeviceRequest() = device;
  ThreadContext& tc = ThisThread();
  if (tc.device != device || !tc.ctx) tc.Recreate(device);
  if (tc.rc != BRISK_HIP_OK || !tc.ctx) Check(nullptr, tc.rc ? tc.rc : BRISK_HIP_ERR_NO_DEVICE, "brisk_hip_create");
}

// Engine option (per thread, off by default): while set, BriskDescriptorExtractor::compute() tells the engine that its
// image is the very cv::Mat buffer the thread's last detect() call was given and that the pixels have not changed in
// between - the usual detect() -> compute() pair (test-binary-equal.cc:215,237) - so the image is not uploaded a second
// time (brisk_hip_describe_same_image).  Without it compute() always uploads, as the reference always reads the current
// pixels.
inline bool& SameImageHint() {
  static thread_local bool on = false;
  return on;
}
struct ScopedSameImage {
  bool prev;
  ScopedSameImage() : prev(SameImageHint()) { SameImageHint() = true; }
  ~ScopedSameImage() { SameImageHint() = prev; }
};

// ---- many threads at once: call combining (brisk_hip_pool, include/brisk_hip.h) -----------------------------------------
// A thread's own context is the fastest way to serve one caller, or a few; with many, every call's ~15 launches and 2 - 4 copies
// queue behind each other in the HIP runtime (it serialises the API calls of a process: 16 threads with their own contexts reach
// 3.6 x one thread) and the calls that are in the engine at the same time are better off as one batch.  The classes count the threads that are inside detect() / compute() right
// now and hand a call to the device's shared pool when that count has reached PoolThreshold().  Default: one more than the
// CPUs the process may use - measured: up to 16 threads on 16 CPUs a context per thread is as fast or faster
// (9 - 10 k against 8 k 1080p frames/s), with more threads than CPUs the pool keeps its rate (10 - 12 k at 32 threads) where the
// per-thread contexts lose theirs.  0 = never; BRISK_HIP_POOL_THREADS in the environment or SetPoolThreshold() change it.
// Results are bit-identical either way.
inline std::atomic<int>& ActiveCalls() {
  static std::atomic<int> n{0};
  return n;
}
struct CallScope {
  int n;
  CallScope() : n(ActiveCalls().fetch_add(1, std::memory_order_relaxed) + 1) {}
  ~CallScope() { ActiveCalls().fetch_sub(1, std::memory_order_relaxed); }
  CallScope(const CallScope&) = delete;
  CallScope& operator=(const CallScope&) = delete;
};
inline std::atomic<int>& PoolThresholdRef() {
  static std::atomic<int> t{[] { const char* e = std::getenv("BRISK_HIP_POOL_THREADS"); return e ? std::atoi(e) : brisk_hip_usable_cpus() + 1; }()};
  return t;
}
inline int PoolThreshold() { return PoolThresholdRef().load(std::memory_order_relaxed); }
inline void SetPoolThreshold(int concurrent_callers) { PoolThresholdRef().store(concurrent_callers, std::memory_order_relaxed); }
// the shared pool of a device: created by the first call that needs it, lives as long as the process (null: creation failed,
// the caller uses its own context)
inline brisk_hip_pool* SharedPool(int device) {
  static std::mutex mu;
  static brisk_hip_pool* pools[16] = {};
  static bool tried[16] = {};
  if (device < 0 || device >= 16) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!tried[device]) {
    tried[device] = true;
    if (brisk_hip_pool_create(device, 32, 16384, &pools[device]) != BRISK_HIP_OK) pools[device] = nullptr;
  }
  return pools[device];
}
// the thread's last pooled detect(): the buffer it was given and the token of its device copy (compute() presents the token
// only under ScopedSameImage - the caller's word that the pixels are unchanged - and for that very buffer)
struct PooledImage {
  const void* data = nullptr;
  int rows = 0, cols = 0;
  unsigned long long token = 0;
};
inline PooledImage& LastPooledImage() {
  static thread_local PooledImage p;
  return p;
}

// Destination of a multi-image call's results (brisk_hip_batch_host_results), one per thread, grown when a call needs more and
// page-locked (brisk_hip_host_register): the device writes the rows straight into it, nothing is zero-filled or faulted in per
// call, and the classes copy every image's rows out of it into the caller's vectors.
struct ResultScratch {
  void* mem = nullptr;
  size_t bytes = 0;
  bool locked = false;
  brisk_hip_batch_host_results dst;
  ResultScratch() { std::memset(&dst, 0, sizeof dst); }
  ~ResultScratch() { Release(); }
  ResultScratch(const ResultScratch&) = delete;
  ResultScratch& operator=(const ResultScratch&) = delete;
  void Release() {
    if (mem && locked) (void)brisk_hip_host_unregister(mem);
    std::free(mem);
    mem = nullptr; bytes = 0; locked = false;
  }
  // arrays for `frames` frames and `rows` rows with descriptor rows of desc_stride bytes (0: keypoints only)
  brisk_hip_batch_host_results* Prepare(int frames, long long rows, int desc_stride) {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_flags = up(sizeof(int) * (size_t)frames), o_offs = up(o_flags + sizeof(int) * (size_t)frames);
    const size_t o_kps = up(o_offs + sizeof(long long) * ((size_t)frames + 1));
    const size_t o_desc = up(o_kps + sizeof(brisk_hip_keypoint) * (size_t)rows);
    const size_t need = up(o_desc + (size_t)rows * (size_t)desc_stride) + 256;
    if (need > bytes) {
      Release();
      const size_t want = need + need / 4;  // (room to grow: a registration costs ~0.1 ms per MB)
      if (posix_memalign(&mem, 4096, want) != 0) { mem = nullptr; throw std::bad_alloc(); }
      bytes = want;
      locked = brisk_hip_host_register(mem, bytes) == BRISK_HIP_OK;  // (pageable still works: the engine goes through its bounce buffer)
    }
    unsigned char* b = static_cast<unsigned char*>(mem);
    dst.frames_cap = frames; dst.desc_stride = desc_stride ? desc_stride : 4; dst.rows_cap = rows;
    dst.counts = reinterpret_cast<int*>(b); dst.flags = reinterpret_cast<int*>(b + o_flags);
    dst.offsets = reinterpret_cast<long long*>(b + o_offs);
    dst.kps = reinterpret_cast<brisk_hip_keypoint*>(b + o_kps);
    dst.desc = desc_stride ? b + o_desc : nullptr;
    return &dst;
  }
};
inline ResultScratch& ThreadResultScratch() {
  static thread_local ResultScratch s;
  return s;
}

}  // namespace hip
}  // namespace brisk
#endif  // BRISK_HIP_CONTEXT_H_
