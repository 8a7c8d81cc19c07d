// test_gather.cc - the multi-GPU result gather of the batch path through the C ABI alone (no torch, no Python):
// brisk_hip_comm_* on RCCL.  What a C++ host with one thread (or process) per GPU runs per rank; here world = 1 (one GPU
// per box in this pool; RCCL refuses two ranks on one device), two batches in a row so that both send slabs are used.
// Compares every gathered row with brisk_hip_batch_download.  usage: test_gather   (exit code 2: no GPU, 0: ok)
#include <brisk/hip-context.h>
#include <brisk_hip.h>
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK_RC(call)                                                                      \
  do {                                                                                      \
    const int rc_ = (call);                                                                 \
    if (rc_ != BRISK_HIP_OK) {                                                              \
      std::printf("%s failed (%d): %s\n", #call, rc_, ctx ? brisk_hip_last_error(ctx) : ""); \
      return 1;                                                                             \
    }                                                                                       \
  } while (0)

static void make_frame(std::vector<uint8_t>& f, int w, int h, unsigned seed) {
  // blocks of random grey levels + a little noise: a few hundred AGAST corners
  f.assign((size_t)w * h, 0);
  unsigned s = seed * 2654435761u + 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
  const int bw = 24, bh = 20;
  std::vector<uint8_t> lv((size_t)(w / bw + 1) * (h / bh + 1));
  for (auto& v : lv) v = (uint8_t)(40 + rnd() % 170);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) f[(size_t)y * w + x] = (uint8_t)(lv[(size_t)(y / bh) * (w / bw + 1) + x / bw] + rnd() % 5);
}

int main() {
  brisk_hip_ctx* ctx = nullptr;
  if (brisk_hip_device_count() <= 0) {
    std::printf("no HIP device: brisk_hip_create failed\n");
    return 2;
  }
  try {
    brisk::hip::SetThreadDevice(0);  // (one host thread per GPU: the thread's context lives on that GPU)
    ctx = brisk::hip::DefaultContext();
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 2;
  }
  const int w = 640, h = 480, nframes = 5, frames_max = 8, kpad = 2048, strings = 48;
  brisk_hip_pattern* pat = nullptr;
  CHECK_RC(brisk_hip_pattern_create(ctx, 2, 1.0f, &pat));
  uint8_t id[BRISK_HIP_COMM_ID_BYTES];
  CHECK_RC(brisk_hip_comm_unique_id(id));
  brisk_hip_comm* comm = nullptr;
  CHECK_RC(brisk_hip_comm_create(ctx, 0, 1, id, &comm));
  if (brisk_hip_comm_rank(comm) != 0 || brisk_hip_comm_world(comm) != 1) { std::printf("rank / world wrong\n"); return 1; }
  int* d_counts = nullptr;
  brisk_hip_keypoint* d_kps = nullptr;
  uint8_t* d_desc = nullptr;
  if (hipMalloc((void**)&d_counts, sizeof(int) * frames_max) != hipSuccess ||
      hipMalloc((void**)&d_kps, sizeof(brisk_hip_keypoint) * (size_t)frames_max * kpad) != hipSuccess ||
      hipMalloc((void**)&d_desc, (size_t)frames_max * kpad * strings) != hipSuccess) {
    std::printf("hipMalloc failed\n");
    return 1;
  }
  std::vector<uint8_t> frames((size_t)w * h * nframes), one;
  long total = 0;
  for (int batch = 0; batch < 3; ++batch) {
    for (int f = 0; f < nframes; ++f) {
      make_frame(one, w, h, 100u * batch + f);
      std::memcpy(&frames[(size_t)f * w * h], one.data(), one.size());
    }
    CHECK_RC(brisk_hip_detect_describe_batch_host(ctx, pat, frames.data(), nframes, w, h, (long)w * h, w, 60, 4));
    // (a wrong argument is refused before anything is queued)
    if (brisk_hip_comm_gather_results(ctx, comm, 0, nframes - 1, kpad, strings, d_counts, d_kps, d_desc, nullptr) != BRISK_HIP_ERR_ARG) {
      std::printf("frames_max below the batch size was accepted\n");
      return 1;
    }
    CHECK_RC(brisk_hip_comm_gather_results(ctx, comm, 0, frames_max, kpad, strings, d_counts, d_kps, d_desc, nullptr));
    CHECK_RC(brisk_hip_comm_wait(comm, nullptr));
    std::vector<int> counts(frames_max);
    std::vector<brisk_hip_keypoint> kps((size_t)frames_max * kpad);
    std::vector<uint8_t> desc((size_t)frames_max * kpad * strings);
    if (hipMemcpy(counts.data(), d_counts, sizeof(int) * frames_max, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(kps.data(), d_kps, sizeof(brisk_hip_keypoint) * kps.size(), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(desc.data(), d_desc, desc.size(), hipMemcpyDeviceToHost) != hipSuccess) {
      std::printf("hipMemcpy failed\n");
      return 1;
    }
    for (int f = 0; f < frames_max; ++f) {
      if (f >= nframes) {
        if (counts[f] != 0) { std::printf("batch %d: frame %d beyond the batch has count %d\n", batch, f, counts[f]); return 1; }
        continue;
      }
      std::vector<brisk_hip_keypoint> k(kpad);
      std::vector<uint8_t> d((size_t)kpad * strings);
      int n = 0;
      CHECK_RC(brisk_hip_batch_download(ctx, f, 1, k.data(), kpad, &n, d.data(), strings));
      if (n != counts[f] || n < 50 || std::memcmp(k.data(), &kps[(size_t)f * kpad], sizeof(brisk_hip_keypoint) * n) != 0 ||
          std::memcmp(d.data(), &desc[(size_t)f * kpad * strings], (size_t)n * strings) != 0) {
        std::printf("batch %d frame %d: gathered rows differ from the download (%d vs %d keypoints)\n", batch, f, counts[f], n);
        return 1;
      }
      total += n;
    }
  }
  brisk_hip_comm_destroy(comm);
  brisk_hip_pattern_destroy(pat);
  (void)hipFree(d_counts); (void)hipFree(d_kps); (void)hipFree(d_desc);
  std::printf("gather OK: 3 batches x %d frames, %ld keypoint rows equal\n", nframes, total);
  return 0;
}
