// test_host_results.cc - the batch path from host memory to host memory through the C ABI alone (no torch, no Python, no HIP
// header): what a C++ caller with a stream of frames in host memory writes.  A ring of frames page-locked with
// brisk_hip_host_register, brisk_hip_detect_describe_batch_host_results with two destination sets alternating (the transfer of
// batch n runs beside batch n + 1), results as exact prefix-summed rows (who receives the results in the reference: the caller's
// std::vector<cv::KeyPoint> and descriptor cv::Mat, brisk-feature-detector.cc:77-85, brisk-descriptor-extractor.cc:601-604).
// Every frame of every batch is compared with brisk_hip_batch_download of that frame; a destination that is one row short must
// report the cut frame.  usage: test_host_results   (exit code 2: no GPU, 0: ok)
#include <brisk_hip.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
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

static void make_frame(uint8_t* f, int w, int h, unsigned seed) {
  unsigned s = seed * 2654435761u + 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
  const int bw = 24, bh = 20;
  std::vector<uint8_t> lv((size_t)(w / bw + 1) * (h / bh + 1));
  for (auto& v : lv) v = (uint8_t)(40 + rnd() % 170);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) f[(size_t)y * w + x] = (uint8_t)(lv[(size_t)(y / bh) * (w / bw + 1) + x / bw] + rnd() % 5);
}

struct Dest {  // one destination set: ONE page-aligned block, page-locked once, the five arrays carved out of it
  void* mem = nullptr;
  int* counts; int* flags; long long* offsets; brisk_hip_keypoint* kps; uint8_t* desc;
  brisk_hip_batch_host_results r;
  Dest(int frames, long long rows, int strings) {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_flags = up(sizeof(int) * (size_t)frames), o_offs = up(o_flags + sizeof(int) * (size_t)frames);
    const size_t o_kps = up(o_offs + sizeof(long long) * ((size_t)frames + 1)), o_desc = up(o_kps + sizeof(brisk_hip_keypoint) * (size_t)rows);
    const size_t bytes = up(o_desc + (size_t)rows * strings) + 4096;
    if (posix_memalign(&mem, 4096, bytes) != 0) { std::printf("out of memory\n"); std::exit(1); }
    std::memset(mem, 0xEE, bytes);
    uint8_t* b = static_cast<uint8_t*>(mem);
    counts = reinterpret_cast<int*>(b); flags = reinterpret_cast<int*>(b + o_flags); offsets = reinterpret_cast<long long*>(b + o_offs);
    kps = reinterpret_cast<brisk_hip_keypoint*>(b + o_kps); desc = b + o_desc;
    r.frames_cap = frames; r.desc_stride = strings; r.rows_cap = rows;
    r.counts = counts; r.flags = flags; r.offsets = offsets; r.kps = kps; r.desc = desc;
    // (page-locked: the device writes the rows itself; without this the engine goes through its bounce buffer - same results)
    (void)brisk_hip_host_register(mem, bytes);
  }
  ~Dest() {
    (void)brisk_hip_host_unregister(mem);
    std::free(mem);
  }
  Dest(const Dest&) = delete;
  Dest& operator=(const Dest&) = delete;
};

int main() {
  brisk_hip_ctx* ctx = nullptr;
  if (brisk_hip_device_count() <= 0 || brisk_hip_create(0, &ctx) != BRISK_HIP_OK) {
    std::printf("no HIP device: brisk_hip_create failed\n");
    return 2;
  }
  brisk_hip_pattern* pat = nullptr;
  CHECK_RC(brisk_hip_pattern_create(ctx, 2, 1.0f, &pat));
  const int W = 640, H = 480, N = 24, STRINGS = brisk_hip_pattern_descriptor_size(pat);
  // a ring of three batches of frames in page-locked host memory
  const size_t ring_bytes = (size_t)3 * N * W * H;
  void* ring_mem = nullptr;
  if (posix_memalign(&ring_mem, 4096, ring_bytes) != 0) return 1;
  uint8_t* ring = static_cast<uint8_t*>(ring_mem);
  for (int i = 0; i < 3 * N; ++i) make_frame(ring + (size_t)i * W * H, W, H, 77u + (unsigned)i);
  CHECK_RC(brisk_hip_host_register(ring, ring_bytes));
  {
    Dest d0(N, (long long)N * 2048, STRINGS), d1(N, (long long)N * 2048, STRINGS);
    Dest* dst[2] = {&d0, &d1};
    unsigned ticket[2] = {0, 0};
    int bad = 0;
    long long rows_seen = 0;
    // the last batch's frames against the per-frame download (the batch is still the context's last one)
    auto check_last = [&](const Dest& d) {
      std::vector<brisk_hip_keypoint> k(4096);
      std::vector<uint8_t> dd((size_t)4096 * STRINGS);
      for (int f = 0; f < N; ++f) {
        int n = 0;
        if (brisk_hip_batch_download(ctx, f, 1, k.data(), (int)k.size(), &n, dd.data(), STRINGS) != BRISK_HIP_OK) { ++bad; continue; }
        const long long a = d.offsets[f], cnt = d.offsets[f + 1] - a;
        if (d.flags[f] || cnt != n || d.counts[f] != n || std::memcmp(d.kps + a, k.data(), sizeof(brisk_hip_keypoint) * (size_t)n) != 0 ||
            std::memcmp(d.desc + (size_t)a * STRINGS, dd.data(), (size_t)n * STRINGS) != 0) {
          std::printf("frame %d differs from its per-frame download (%lld vs %d rows, flags %d)\n", f, cnt, n, d.flags[f]);
          ++bad;
        }
        rows_seen += n;
      }
    };
    for (int b = 0; b < 7; ++b) {  // seven batches over the ring, two transfers in flight
      const int s = b & 1;
      int flagged = 0;
      if (ticket[s]) CHECK_RC(brisk_hip_batch_download_wait(ctx, ticket[s], &flagged));  // the set is free again (its rows were consumed below)
      CHECK_RC(brisk_hip_detect_describe_batch_host_results(ctx, pat, ring + (size_t)(b % 3) * N * W * H, N, W, H, (long)W * H, W, 70, 4, &dst[s]->r,
                                                            &ticket[s]));
      if (b == 6 || b == 3) {  // consume at once and compare while the batch is the context's last one
        CHECK_RC(brisk_hip_batch_download_wait(ctx, ticket[s], &flagged));
        check_last(*dst[s]);
        if (flagged) { std::printf("batch %d: %d frame(s) flagged\n", b, flagged); ++bad; }
      }
    }
    for (int s = 0; s < 2; ++s) {
      int flagged = 0;
      CHECK_RC(brisk_hip_batch_download_wait(ctx, ticket[s], &flagged));
    }
    // batches 5 and 6 used the same frames as 2 and 0 of the ring: set 1 (batch 5) must equal what batch 2 produced - compare
    // its totals with a fresh run of those frames
    {
      Dest again(N, (long long)N * 2048, STRINGS);
      unsigned t = 0;
      int flagged = 0;
      CHECK_RC(brisk_hip_detect_describe_batch_host_results(ctx, pat, ring + (size_t)(5 % 3) * N * W * H, N, W, H, (long)W * H, W, 70, 4, &again.r, &t));
      CHECK_RC(brisk_hip_batch_download_wait(ctx, t, &flagged));
      if (again.offsets[N] != dst[1]->offsets[N] ||
          std::memcmp(again.kps, dst[1]->kps, sizeof(brisk_hip_keypoint) * (size_t)again.offsets[N]) != 0 ||
          std::memcmp(again.desc, dst[1]->desc, (size_t)again.offsets[N] * STRINGS) != 0) {
        std::printf("batch 5's rows differ from a fresh run of the same frames\n");
        ++bad;
      }
      // one row short: the last frame with rows is cut, its count still reported
      Dest small(N, again.offsets[N] - 1, STRINGS);
      CHECK_RC(brisk_hip_detect_describe_batch_host_results(ctx, pat, ring + (size_t)(5 % 3) * N * W * H, N, W, H, (long)W * H, W, 70, 4, &small.r, &t));
      const int rc = brisk_hip_batch_download_wait(ctx, t, &flagged);
      int cut = 0;
      for (int f = 0; f < N; ++f) cut += (small.flags[f] & BRISK_HIP_ROWS_CUT) ? 1 : 0;
      if (rc != BRISK_HIP_ERR_CAPACITY || flagged < 1 || cut != flagged || small.offsets[N] >= again.offsets[N]) {
        std::printf("short destination: rc %d, %d flagged, %d cut\n", rc, flagged, cut);
        ++bad;
      }
    }
    if (bad) { std::printf("FAILED: %d difference(s)\n", bad); return 1; }
    std::printf("host results OK: 7 batches of %d frames, %lld rows compared with the per-frame download\n", N, rows_seen);
  }
  (void)brisk_hip_host_unregister(ring);
  std::free(ring_mem);
  brisk_hip_pattern_destroy(pat);
  brisk_hip_destroy(ctx);
  return 0;
}
