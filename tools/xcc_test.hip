#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (3 << 11));
}
__global__ void k2(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
}
int main() {
  int* d; hipMalloc(&d, 4096 * 4);
  int h[4096];
  hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, d); hipMemcpy(h, d, 64 * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; ++i) printf("%d ", h[i]); printf("\n");
  hipLaunchKernelGGL(k2, dim3(16), dim3(64), 0, 0, d); hipMemcpy(h, d, 16 * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < 16; ++i) printf("%08x ", h[i]); printf("\n");
  hipLaunchKernelGGL(k, dim3(1536), dim3(256), 100*1024, 0, d); hipMemcpy(h, d, 1536 * 4, hipMemcpyDeviceToHost);
  int cnt[16] = {0}; for (int i = 0; i < 1536; ++i) cnt[h[i] & 15]++;
  for (int i = 0; i < 16; ++i) printf("%d ", cnt[i]); printf("\n");
  return 0;
}
