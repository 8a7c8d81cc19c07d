// microbench_gather.hip - what the vector memory path of one CU sustains for GATHERS (one distinct 128-byte line per
// lane and instruction), the access shape of k_describe.  Standalone:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mbg tools/microbench_gather.hip && /tmp/mbg
//
// Every wave runs ITERS rounds of K independent `global_load_dwordx2` gathers (addresses from a per-lane LCG, never
// from loaded data) and then consumes them, so K x 64 line requests per wave are in flight at the wait.  Swept:
//   table   : 16 KB per CU (L1 hits), 2 MB (every XCD's L2 holds it), 8 MB per XCD group (the integral image of one
//             1080p frame, blocks of an XCD share it), 1 GiB (HBM / Infinity Cache misses)
//   waves   : 1..8 per SIMD (one or two workgroups per CU, the rest of the CU blocked by dynamic LDS)
//   K       : gathers in flight per wave
//   share   : lanes per line (1 = 64 distinct lines per instruction, 2 = 32, 4 = 16)
// Output: one JSON line per configuration with line requests per clock and CU (s_memtime clock and wall clock).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);  \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef uint32_t __attribute__((ext_vector_type(2), aligned(4))) u32x2;

// mode 0: table shared by all blocks.  mode 1: one table slice per CU (block index).  mode 2: one slice per XCD group
// (blockIdx.x & 7).
template <int K>
__global__ void __launch_bounds__(1024) k_gather(const uint32_t* __restrict__ tbl, unsigned line_mask, long slice_dwords, int mode,
                                                 int share_shift, int iters, unsigned long long* cyc, unsigned* sink) {
  extern __shared__ int dyn[];
  const int lane = threadIdx.x & 63;
  const long slice = mode == 1 ? (long)(blockIdx.x % 256) : mode == 2 ? (long)(blockIdx.x & 7) : 0;
  const uint32_t* base = tbl + slice * slice_dwords;
  unsigned s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
  // lanes of a share group use the same stream of lines
  unsigned sg = ((blockIdx.x * 1024u + (threadIdx.x >> share_shift)) * 2654435761u) ^ 0x9E3779B9u;
  unsigned acc = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int it = 0; it < iters; ++it) {
    u32x2 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      sg = sg * 1664525u + 1013904223u;
      s = s * 22695477u + 1u;
      const unsigned line = (sg >> 7) & line_mask;
      const unsigned dw = (s >> 20) & 30u;  // even dword inside the line
      v[k] = *reinterpret_cast<const u32x2*>(base + (long)line * 32 + dw);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) acc += v[k].x ^ v[k].y;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) cyc[(long)blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc;
}


// k_box: the access shape of k_describe itself.  A wave = one keypoint pass: 64 pattern points (lane = point) inside a
// disc of radius R around a random centre of a 1921 x 1081 u32 integral image (one image per XCD group, as the engine
// places a frame), box half side 3..12 px per point; per point the 4 x 4 integral samples as 8 dwordx2 gathers
// (rows yt, yt+1, yb, yb+1 x column pairs xl, xr), optionally two byte gathers from the u8 image, then `valu`
// dependent multiply-adds per lane before the next pass (the duty cycle of the real kernel).
//   order 0: row-major (r0xl r0xr r1xl r1xr ...), order 1: column-major (r0xl r1xl r2xl r3xl r0xr ...)
//   align 1: xl, xr forced even (8-byte aligned pairs)
template <int ORDER>
__global__ void __launch_bounds__(1024) k_box(const uint32_t* __restrict__ integ, const uint8_t* __restrict__ img, int iw, int ih,
                                              long frame_dwords, int radius, int align, int bytes, int valu, int band, int iters,
                                              unsigned long long* cyc, unsigned* sink) {
  extern __shared__ int dyn[];
  const int lane = threadIdx.x & 63;
  const uint32_t* I = integ + (long)(blockIdx.x & 7) * frame_dwords;
  const uint8_t* P = img + (long)(blockIdx.x & 7) * (long)(iw - 1) * (ih - 1);
  unsigned s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 777u;
  unsigned sw = (blockIdx.x * 16u + (threadIdx.x >> 6)) * 2246822519u + 99u;  // wave-uniform stream (keypoint centres)
  // pattern point of this lane: polar offset inside the disc, box half side
  s = s * 1664525u + 1013904223u;
  const float ang = (float)(s >> 8) * (6.2831853f / 16777216.0f);
  s = s * 1664525u + 1013904223u;
  const float rad = sqrtf((float)(s >> 8) * (1.0f / 16777216.0f)) * (float)radius;
  s = s * 1664525u + 1013904223u;
  const int hs = 3 + (int)((s >> 8) % 10u) * radius / 50;
  float ox = rad * cosf(ang), oy = rad * sinf(ang);
  unsigned acc = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int margin = radius + 3 + 9 * radius / 50 + 3;
  for (int it = 0; it < iters; ++it) {
    sw = sw * 1664525u + 1013904223u;
    const int cx = margin + (int)((sw >> 8) % (unsigned)(iw - 2 * margin));
    sw = sw * 1664525u + 1013904223u;
    int cy = margin + (int)((sw >> 8) % (unsigned)(ih - 2 * margin));
    // band > 0: all waves pick their centres inside a window of `band` rows that moves down the frame (the concurrency
    // window of a sorted keypoint list); 0: anywhere in the frame
    if (band) cy = margin + (it * 5) % (ih - 2 * margin - band) + (int)((sw >> 8) % (unsigned)band);
    // rotate the pattern a little every pass (as theta does)
    const float c = 0.9553365f, sn = 0.2955202f;
    const float nx = ox * c - oy * sn, ny = ox * sn + oy * c;
    ox = nx; oy = ny;
    int xl = cx + (int)ox - hs, xr = cx + (int)ox + hs, yt = cy + (int)oy - hs, yb = cy + (int)oy + hs;
    if (align) { xl &= ~1; xr &= ~1; }
    const uint32_t* r0 = I + (long)yt * iw;
    const uint32_t* r1 = r0 + iw;
    const uint32_t* r2 = I + (long)yb * iw;
    const uint32_t* r3 = r2 + iw;
    u32x2 v0, v1, v2, v3, v4, v5, v6, v7;
#define LD(p) (*reinterpret_cast<const u32x2*>(p))
    if (ORDER == 2) {
      // rows 2k and 2k + 1 interleaved dword by dword (element (y, x) at (y >> 1) * 2 iw + 2 x + (y & 1)): the 2 x 2 cluster
      // of (y, x) is ONE 16-byte group when y is even and two (pair rows y >> 1 and (y + 1) >> 1) when it is odd; the
      // code always loads both groups (the second repeats the first for an even y)
      const long pr = 2L * iw;
      unsigned t = 0;
      const int ys[2] = {yt, yb}, xs[2] = {xl, xr};
      uint4 A[4], B[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int y = ys[q >> 1], x = xs[q & 1];
        A[q] = *reinterpret_cast<const uint4*>(I + (long)(y >> 1) * pr + 2 * x);
        B[q] = *reinterpret_cast<const uint4*>(I + (long)((y + 1) >> 1) * pr + 2 * x);
      }
      unsigned b0 = 0, b1 = 0;
      if (bytes) {
        b0 = P[(long)(yb - 1) * (iw - 1) + xr + 1];
        b1 = P[(long)(yb - 1) * (iw - 1) + xl + 1];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool odd = ys[q >> 1] & 1;
        t += (odd ? A[q].y : A[q].x) ^ (odd ? B[q].x : A[q].y);
        t += (odd ? A[q].w : A[q].z) ^ (odd ? B[q].z : A[q].w);
      }
      t += b0 + b1;
      for (int k = 0; k < valu; ++k) t = t * 1664525u + 1013904223u;
      acc += t;
      continue;
    }
    if (ORDER == 0) {
      v0 = LD(r0 + xl); v1 = LD(r0 + xr); v2 = LD(r1 + xl); v3 = LD(r1 + xr);
      v4 = LD(r2 + xl); v5 = LD(r2 + xr); v6 = LD(r3 + xl); v7 = LD(r3 + xr);
    } else {
      v0 = LD(r0 + xl); v2 = LD(r1 + xl); v4 = LD(r2 + xl); v6 = LD(r3 + xl);
      v1 = LD(r0 + xr); v3 = LD(r1 + xr); v5 = LD(r2 + xr); v7 = LD(r3 + xr);
    }
    unsigned b0 = 0, b1 = 0;
    if (bytes) {
      b0 = P[(long)(yb - 1) * (iw - 1) + xr + 1];
      b1 = P[(long)(yb - 1) * (iw - 1) + xl + 1];
    }
    unsigned t = (v0.x ^ v0.y) + (v1.x ^ v1.y) + (v2.x ^ v2.y) + (v3.x ^ v3.y) + (v4.x ^ v4.y) + (v5.x ^ v5.y) + (v6.x ^ v6.y) +
                 (v7.x ^ v7.y) + b0 + b1;
    for (int k = 0; k < valu; ++k) t = t * 1664525u + 1013904223u;  // 2 dependent VALU per step
    acc += t;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) cyc[(long)blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc;
}

// dependent chain of one lane: load latency
__global__ void k_chase(const uint32_t* __restrict__ tbl, int n, unsigned start, unsigned long long* cyc, unsigned* sink) {
  unsigned i = start;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int k = 0; k < n; ++k) i = tbl[(long)i * 32];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  cyc[0] = t1 - t0;
  sink[1] = i;
}

typedef void (*gk_t)(const uint32_t*, unsigned, long, int, int, int, unsigned long long*, unsigned*);
static gk_t pick(int K) {
  switch (K) {
    case 1: return k_gather<1>;
    case 2: return k_gather<2>;
    case 4: return k_gather<4>;
    case 8: return k_gather<8>;
    case 10: return k_gather<10>;
    case 16: return k_gather<16>;
    default: return k_gather<20>;
  }
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const long big = 1L << 30;
  uint32_t* tbl;
  CHECK(hipMalloc(&tbl, big));
  {
    std::vector<uint32_t> h(big / 4);
    uint32_t x = 1;
    for (size_t i = 0; i < h.size(); ++i) { x = x * 1664525u + 1013904223u; h[i] = x; }
    // pointer-chase ring over the first 1 MiB / 64 MiB / 1 GiB is set up below per run
    CHECK(hipMemcpy(tbl, h.data(), big, hipMemcpyHostToDevice));
  }
  unsigned long long* d_cyc;
  unsigned* d_sink;
  CHECK(hipMalloc(&d_cyc, 8 * 16 * 4096));
  CHECK(hipMalloc(&d_sink, 64));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));

  struct Tab { const char* name; long lines; long slice_dwords; int mode; };
  const Tab tabs[] = {
      {"L1_8KB_per_workgroup", 64, 64 * 32, 1},
      {"L2_2MB_shared", 16384, 0, 0},
      {"frame_8MB_per_XCD", 65536, 65536L * 32, 2},
      {"HBM_1GiB", big / 128, 0, 0},
  };
  const int Ws[] = {1, 2, 3, 4, 5, 6, 8};
  const int Ks[] = {1, 2, 4, 8, 10, 16, 20};
  const int shares[] = {0, 1, 2};
  printf("{\"device\": \"%s\", \"cus\": %d, \"unit\": \"128-byte line requests per clock and CU\", \"rows\": [\n", prop.gcnArchName, ncu);
  bool first = true;
  const bool box_only = argc > 1;  // any argument: only the describe-shaped sweep
  for (const Tab& T : tabs) {
    if (box_only) break;
    for (int sh : shares) {
      if (sh && T.mode != 2) continue;
      for (int W : Ws) {
        for (int K : Ks) {
          if (T.mode != 2 && !(K == 1 || K == 4 || K == 10 || K == 20)) continue;
          const int bpc = W > 4 ? 2 : 1;            // workgroups per CU
          const int threads = 64 * 4 * W / bpc;     // W waves per SIMD in total
          const size_t lds = bpc == 1 ? 100 * 1024 : 70 * 1024;
          gk_t fn = pick(K);
          CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          const int grid = ncu * bpc;
          const int iters = std::max(8, 4096 / (K * W));
          double best_clk = 1e30, best_ms = 1e30;
          std::vector<unsigned long long> h((size_t)grid * 16);
          for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(d_cyc, 0, (size_t)grid * 16 * 8));
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(fn, dim3(grid), dim3(threads), lds, 0, tbl, (unsigned)(T.lines - 1), T.slice_dwords, T.mode, sh, iters,
                               d_cyc, d_sink);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<unsigned long long> nz;
            for (auto c : h) if (c) nz.push_back(c);
            std::sort(nz.begin(), nz.end());
            if (rep) {
              best_clk = std::min(best_clk, (double)nz[nz.size() / 2]);
              best_ms = std::min(best_ms, (double)ms);
            }
          }
          const double lines_per_cu = (double)iters * K * (64 >> sh) * 4 * W;
          printf("%s{\"table\": \"%s\", \"lanes_per_line\": %d, \"waves_per_simd\": %d, \"K\": %d, \"lines_per_clk_cu\": %.4f, "
                 "\"lines_per_ns_chip_wall\": %.2f, \"instr_per_kclk_cu\": %.2f, \"clk_per_round\": %.0f}",
                 first ? "" : ",\n", T.name, 1 << sh, W, K, lines_per_cu / best_clk, lines_per_cu * ncu / (best_ms * 1e6),
                 1000.0 * iters * K * 4 * W / best_clk, best_clk / iters);
          first = false;
          fflush(stdout);
        }
      }
    }
  }

  printf("\n], \"box_rows\": [\n");
  {
    // describe-shaped passes: 8 integral images of 1921 x 1081 u32 (one per XCD group) + their u8 images
    const int iw = 1921, ih = 1081;
    const long fd = (long)iw * ih;
    const long fd2 = (long)iw * (ih + 1);  // the row-pair interleaved layout: (ih + 1) / 2 pair rows
    uint8_t* img = (uint8_t*)(tbl + 8 * fd2);
    struct Cfg { int radius, order, align, bytes, valu, band; };
    const Cfg cfgs[] = {{50, 0, 0, 1, 0, 0},   {50, 0, 0, 0, 0, 0},    {50, 1, 0, 0, 0, 0},   {50, 0, 1, 0, 0, 0},   {50, 0, 0, 1, 100, 0},
                        {50, 0, 0, 1, 400, 0}, {50, 0, 0, 1, 1000, 0}, {20, 0, 0, 1, 0, 0},   {120, 0, 0, 1, 0, 0},  {300, 0, 0, 1, 0, 0},
                        {50, 0, 0, 1, 0, 32},  {50, 0, 0, 1, 0, 64},   {50, 0, 0, 1, 0, 128}, {50, 0, 0, 1, 0, 256}, {50, 0, 0, 1, 0, 512},
                        {50, 0, 0, 0, 0, 64},  {50, 0, 0, 0, 0, 256},  {20, 0, 0, 1, 0, 64},  {120, 0, 0, 1, 0, 64}, {120, 0, 0, 1, 0, 256},
                        // order 2: the row-pair interleaved layout (two 16-byte groups per 2 x 2 cluster) beside order 0
                        {50, 2, 0, 1, 0, 0},   {50, 2, 0, 1, 0, 64},   {50, 2, 0, 1, 0, 256}, {120, 2, 0, 1, 0, 64}, {120, 2, 0, 1, 0, 256},
                        {120, 2, 0, 1, 0, 0},  {20, 2, 0, 1, 0, 64},   {50, 2, 0, 1, 400, 64}, {50, 0, 0, 1, 400, 64}};
    const bool layout_only = argc > 1 && !strcmp(argv[1], "layout");  // only the rows needed for the layout comparison
    const int Wb[] = {1, 2, 3, 4, 5, 6, 8};
    bool firstb = true;
    for (const Cfg& c : cfgs) {
      if (layout_only && !(c.order == 2 || (c.order == 0 && c.bytes == 1 && c.align == 0 && (c.valu == 0 || c.valu == 400) && (c.band == 0 || c.band == 64 || c.band == 256)))) continue;
      for (int W : Wb) {
        if (layout_only && W != 2 && W != 3) continue;
        const int bpc = W > 4 ? 2 : 1;
        const int threads = 64 * 4 * W / bpc;
        const size_t lds = bpc == 1 ? 100 * 1024 : 70 * 1024;
        auto fn = c.order == 2 ? k_box<2> : c.order ? k_box<1> : k_box<0>;
        CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int grid = ncu * bpc;
        const int iters = std::max(8, 2048 / W);
        double best_ms = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
          CHECK(hipEventRecord(e0, 0));
          hipLaunchKernelGGL(fn, dim3(grid), dim3(threads), lds, 0, tbl, img, iw, ih, c.order == 2 ? fd2 : fd, c.radius, c.align, c.bytes, c.valu, c.band, iters, d_cyc,
                             d_sink);
          CHECK(hipEventRecord(e1, 0));
          CHECK(hipEventSynchronize(e1));
          float ms;
          CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (rep) best_ms = std::min(best_ms, (double)ms);
        }
        const double samples = (double)iters * 64 * 4 * W * ncu;
        printf("%s{\"radius\": %d, \"order\": %d, \"aligned\": %d, \"byte_gathers\": %d, \"valu_steps\": %d, \"band_rows\": %d, \"waves_per_simd\": %d, "
               "\"samples_per_ns_chip\": %.2f, \"us_per_pass_cu\": %.3f}",
               firstb ? "" : ",\n", c.radius, c.order, c.align, c.bytes, c.valu, c.band, W, samples / (best_ms * 1e6),
               best_ms * 1e3 / ((double)iters * 4 * W));
        firstb = false;
        fflush(stdout);
      }
    }
  }
  printf("\n], \"latency_clk\": {");
  {
    // chase rings: i -> (i * 40503 + 1) mod n lines, n a power of two (full period: odd multiplier == 1 mod 4? use LCG rule)
    const long ns[] = {64, 8192, 262144, big / 128};
    const char* names[] = {"L1_8KB", "L2_1MB", "MALL_32MB", "HBM_1GiB"};
    for (int t = 0; t < 4; ++t) {
      const long n = ns[t];
      std::vector<uint32_t> ring(n * 32);
      for (long i = 0; i < n; ++i) ring[i * 32] = (uint32_t)((i * 1664525L + 1013904223L) & (n - 1));
      CHECK(hipMemcpy(tbl, ring.data(), n * 128, hipMemcpyHostToDevice));
      unsigned long long c = 0;
      const int hops = 2000;
      for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, tbl, hops, (unsigned)((rep * 7919L + 13) & (n - 1)), d_cyc, d_sink);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost));
      }
      printf("%s\"%s\": %.0f", t ? ", " : "", names[t], (double)c / hops);
    }
  }
  printf("}}\n");
  return 0;
}
