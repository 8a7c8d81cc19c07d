// tools/microbench_unaligned.hip - do 8-byte buffer gathers at byte offsets that are no multiple of 4 cost more?  (Would a
// 24-bit integral image - 3-byte elements, two adjacent columns = 6 bytes at offset 3 x - be read as fast as the 32-bit
// one?)  Random lines of an L2-resident table, one 8-byte gather per lane and iteration, offsets 4-byte aligned vs 3 x.
// build: hipcc -O3 --offload-arch=gfx950 -o build/microbench_unaligned tools/microbench_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t __attribute__((ext_vector_type(2))) u32x2;
template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const uint8_t* tab, int bytes, uint32_t* out, int iters) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(tab), 0, bytes, 0x00020000);
  uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u, acc = 0;
  for (int i = 0; i < iters; ++i) {
    u32x2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s = s * 1664525u + 1013904223u;
      const uint32_t col = (s >> 8) % (uint32_t)(bytes / 4 - 4);
      const int off = MODE == 0 ? (int)(col * 4) : MODE == 1 ? (int)((col * 3) & ~0u) : (int)(col * 3) | 0;  // 1: 3 x (any alignment)
      v[k] = __builtin_amdgcn_raw_buffer_load_b64(rs, MODE == 2 ? (int)((col * 3) & ~3u) : off, 0, 0);          // 2: 3 x rounded down to 4
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k].x ^ v[k].y;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
  const int bytes = 2 << 20;  // L2-resident
  uint8_t* tab; uint32_t* out;
  hipMalloc(&tab, bytes); hipMalloc(&out, 1024 * 256 * 4);
  hipMemset(tab, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"4-byte aligned offsets (col * 4)", "byte offsets col * 3 (unaligned 3 of 4 times)", "col * 3 rounded down to a multiple of 4"};
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3(1024), dim3(256), 0, 0, tab, bytes, out, 200);
      else if (mode == 1) hipLaunchKernelGGL(k_gather<1>, dim3(1024), dim3(256), 0, 0, tab, bytes, out, 200);
      else hipLaunchKernelGGL(k_gather<2>, dim3(1024), dim3(256), 0, 0, tab, bytes, out, 200);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    const double gathers = 1024.0 * 256 * 200 * 8;
    printf("%-50s %.3f ms  %.1f G gathers/s\n", names[mode], best, gathers / best / 1e6);
  }
  return 0;
}
