// brisk/hip-context.h - process-wide device context shared by the BRISK host classes.
#ifndef BRISK_HIP_CONTEXT_H_
#define BRISK_HIP_CONTEXT_H_

#include <brisk_hip.h>

#include <cstdlib>
#include <mutex>
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

// One workspace per process (device taken from BRISK_HIP_DEVICE, default 0).  Calls are serialised
// inside the C ABI, so the classes stay re-entrant like the reference's (which is stateless per call).
inline brisk_hip_ctx* DefaultContext() {
  static std::once_flag once;
  static brisk_hip_ctx* ctx = nullptr;
  static int rc = BRISK_HIP_OK;
  std::call_once(once, [] {
    const char* dev = std::getenv("BRISK_HIP_DEVICE");
    rc = brisk_hip_create(dev ? std::atoi(dev) : 0, &ctx);
  });
  if (rc != BRISK_HIP_OK || !ctx) Check(nullptr, rc ? rc : BRISK_HIP_ERR_NO_DEVICE, "brisk_hip_create");
  return ctx;
}

}  // namespace hip
}  // namespace brisk
#endif  // BRISK_HIP_CONTEXT_H_
