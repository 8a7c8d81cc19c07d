// test_gather.cc - the multi-GPU result gather of the batch path through the C ABI alone (no torch, no Python):
// brisk_hip_comm_* on RCCL.  What a C++ host with one thread (or process) per GPU runs per rank.  Without arguments:
// world = 1 on the real RCCL (one GPU per box in this pool; RCCL refuses two ranks on one device).  With --rank / --world /
// --id-file: one process per rank, several ranks on one GPU over the test suite's socket double of RCCL
// (tests/cpp/fake_rccl.cc, BRISK_HIP_RCCL_LIB) - the peer branches of the gather.  Three batches in a row so that both send
// slabs are reused.  usage: test_gather [--rank R --world W --id-file F]   (exit code 2: no GPU, 0: ok)
#include <brisk/hip-context.h>
#include <brisk_hip.h>
#include <hip/hip_runtime_api.h>

#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

// One rank of the gather.  world 1: the original single-process test.  world > 1: `--rank R --world W --id-file F`, one
// process per rank (all on device 0 when BRISK_HIP_RCCL_LIB points to the test suite's socket double - RCCL itself refuses
// two ranks on one device); rank 0 writes the unique id to F, the others wait for it.  11 frames per batch are dealt in
// contiguous shards of UNEQUAL size (6 + 5, or 4 + 4 + 3), three batches so that both send slabs are reused; the root
// checks its own slab against brisk_hip_batch_download and every peer's slab - at d_counts + r * frames_max, d_kps + r *
// frames_max * kpad, d_desc + ... - against its own recomputation of that peer's frames.
int main(int argc, char** argv) {
  int rank = 0, world = 1;
  std::string id_file;
  for (int i = 1; i + 1 < argc; i += 2) {
    if (!std::strcmp(argv[i], "--rank")) rank = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--world")) world = std::atoi(argv[i + 1]);
    else if (!std::strcmp(argv[i], "--id-file")) id_file = argv[i + 1];
  }
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
  const int w = 640, h = 480, kpad = 2048, strings = 48;
  const int total_frames = world == 1 ? 5 : 11;
  auto shard_of = [&](int r, int* first) {
    const int base = total_frames / world, extra = total_frames % world;
    *first = r * base + (r < extra ? r : extra);
    return base + (r < extra ? 1 : 0);
  };
  int first = 0;
  const int nframes = shard_of(rank, &first);
  const int frames_max = world == 1 ? 8 : total_frames / world + (total_frames % world ? 1 : 0);
  brisk_hip_pattern* pat = nullptr;
  CHECK_RC(brisk_hip_pattern_create(ctx, 2, 1.0f, &pat));
  uint8_t id[BRISK_HIP_COMM_ID_BYTES];
  if (rank == 0) {
    CHECK_RC(brisk_hip_comm_unique_id(id));
    if (!id_file.empty()) {  // (the 128 bytes travel to the other ranks by whatever the host has: here a file)
      FILE* f = std::fopen((id_file + ".tmp").c_str(), "wb");
      if (!f || std::fwrite(id, 1, sizeof id, f) != sizeof id) { std::printf("cannot write %s\n", id_file.c_str()); return 1; }
      std::fclose(f);
      std::rename((id_file + ".tmp").c_str(), id_file.c_str());
    }
  } else {
    bool got = false;
    for (int tries = 0; tries < 3000 && !got; ++tries) {
      FILE* f = std::fopen(id_file.c_str(), "rb");
      if (f) { got = std::fread(id, 1, sizeof id, f) == sizeof id; std::fclose(f); }
      if (!got) usleep(20000);
    }
    if (!got) { std::printf("rank %d: no unique id in %s\n", rank, id_file.c_str()); return 1; }
  }
  brisk_hip_comm* comm = nullptr;
  CHECK_RC(brisk_hip_comm_create(ctx, rank, world, id, &comm));
  if (brisk_hip_comm_rank(comm) != rank || brisk_hip_comm_world(comm) != world) { std::printf("rank / world wrong\n"); return 1; }
  int* d_counts = nullptr;
  brisk_hip_keypoint* d_kps = nullptr;
  uint8_t* d_desc = nullptr;
  const size_t slab_c = (size_t)frames_max, slab_k = (size_t)frames_max * kpad, slab_d = (size_t)frames_max * kpad * strings;
  if (rank == 0 && (hipMalloc((void**)&d_counts, sizeof(int) * slab_c * world) != hipSuccess ||
                    hipMalloc((void**)&d_kps, sizeof(brisk_hip_keypoint) * slab_k * world) != hipSuccess ||
                    hipMalloc((void**)&d_desc, slab_d * world) != hipSuccess)) {
    std::printf("hipMalloc failed\n");
    return 1;
  }
  std::vector<uint8_t> frames, one;
  auto build = [&](int batch, int r) {  // the frames of rank r's shard of this batch
    int f0 = 0;
    const int n = shard_of(r, &f0);
    frames.resize((size_t)w * h * n);
    for (int f = 0; f < n; ++f) {
      make_frame(one, w, h, 100u * batch + (unsigned)(f0 + f));
      std::memcpy(&frames[(size_t)f * w * h], one.data(), one.size());
    }
    return n;
  };
  long total = 0;
  for (int batch = 0; batch < 3; ++batch) {
    build(batch, rank);
    CHECK_RC(brisk_hip_detect_describe_batch_host(ctx, pat, frames.data(), nframes, w, h, (long)w * h, w, 60, 4));
    // (a wrong argument is refused before anything is queued - on every rank alike, so nobody is left waiting)
    if (brisk_hip_comm_gather_results(ctx, comm, 0, nframes - 1, kpad, strings, d_counts, d_kps, d_desc, nullptr) != BRISK_HIP_ERR_ARG) {
      std::printf("frames_max below the batch size was accepted\n");
      return 1;
    }
    CHECK_RC(brisk_hip_comm_gather_results(ctx, comm, 0, frames_max, kpad, strings, d_counts, d_kps, d_desc, nullptr));
    CHECK_RC(brisk_hip_comm_wait(comm, nullptr));
    if (rank != 0) continue;
    std::vector<int> counts(slab_c * world);
    std::vector<brisk_hip_keypoint> kps(slab_k * world);
    std::vector<uint8_t> desc(slab_d * world);
    if (hipMemcpy(counts.data(), d_counts, sizeof(int) * counts.size(), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(kps.data(), d_kps, sizeof(brisk_hip_keypoint) * kps.size(), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(desc.data(), d_desc, desc.size(), hipMemcpyDeviceToHost) != hipSuccess) {
      std::printf("hipMemcpy failed\n");
      return 1;
    }
    for (int r = 0; r < world; ++r) {
      // rank r's rows as THIS context computes them: its own last batch for r = 0, a recomputation of r's frames otherwise
      int f0 = 0;
      const int nr = shard_of(r, &f0);
      if (r > 0) {
        build(batch, r);
        CHECK_RC(brisk_hip_detect_describe_batch_host(ctx, pat, frames.data(), nr, w, h, (long)w * h, w, 60, 4));
      }
      for (int f = 0; f < frames_max; ++f) {
        const int cnt = counts[r * slab_c + f];
        if (f >= nr) {
          if (cnt != 0) { std::printf("batch %d rank %d: frame %d beyond the shard has count %d\n", batch, r, f, cnt); return 1; }
          continue;
        }
        std::vector<brisk_hip_keypoint> k(kpad);
        std::vector<uint8_t> d((size_t)kpad * strings);
        int n = 0;
        CHECK_RC(brisk_hip_batch_download(ctx, f, 1, k.data(), kpad, &n, d.data(), strings));
        if (n != cnt || n < 50 || std::memcmp(k.data(), &kps[r * slab_k + (size_t)f * kpad], sizeof(brisk_hip_keypoint) * n) != 0 ||
            std::memcmp(d.data(), &desc[r * slab_d + (size_t)f * kpad * strings], (size_t)n * strings) != 0) {
          std::printf("batch %d rank %d frame %d: gathered rows differ from the download (%d vs %d keypoints)\n", batch, r, f, cnt, n);
          return 1;
        }
        total += n;
      }
    }
  }
  brisk_hip_comm_destroy(comm);
  brisk_hip_pattern_destroy(pat);
  (void)hipFree(d_counts); (void)hipFree(d_kps); (void)hipFree(d_desc);
  if (rank == 0) std::printf("gather OK: world %d, 3 batches x %d frames, %ld keypoint rows equal\n", world, total_frames, total);
  else std::printf("rank %d of %d done: 3 batches x %d frames sent\n", rank, world, nframes);
  return 0;
}
