// brisk/hip-context.h - per-thread device context used by the BRISK host classes.
#ifndef BRISK_HIP_CONTEXT_H_
#define BRISK_HIP_CONTEXT_H_

#include <brisk_hip.h>

#include <cstdlib>
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

// One workspace per THREAD (device taken from BRISK_HIP_DEVICE, default 0): the reference's classes are re-entrant
// (detectImpl is const and builds its state per call, brisk-feature-detector.cc:77-85), so concurrent calls from
// several threads must not serialise on one workspace or race on its settings.  Every thread that uses the classes
// gets its own context (stream + lazily sized buffers, ~30 MB per 1080p frame slot), destroyed when the thread ends.
// Pattern tables (BriskDescriptorExtractor) are plain device memory and are shared by all contexts of the device.
struct ThreadContext {
  brisk_hip_ctx* ctx = nullptr;
  int rc = BRISK_HIP_OK;
  ThreadContext() {
    const char* dev = std::getenv("BRISK_HIP_DEVICE");
    rc = brisk_hip_create(dev ? std::atoi(dev) : 0, &ctx);
  }
  ~ThreadContext() {
    if (ctx) brisk_hip_destroy(ctx);
  }
  ThreadContext(const ThreadContext&) = delete;
  ThreadContext& operator=(const ThreadContext&) = delete;
};

inline brisk_hip_ctx* DefaultContext() {
  static thread_local ThreadContext tc;
  if (tc.rc != BRISK_HIP_OK || !tc.ctx) Check(nullptr, tc.rc ? tc.rc : BRISK_HIP_ERR_NO_DEVICE, "brisk_hip_create");
  return tc.ctx;
}

}  // namespace hip
}  // namespace brisk
#endif  // BRISK_HIP_CONTEXT_H_
