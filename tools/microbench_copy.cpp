// tools/microbench_copy.cpp - what the host-buffer calls pay for their transfers on this box (config 5 sizes):
// pageable vs pinned hipMemcpy in both directions, a pitched D2H copy, plain CPU memcpy, hipHostRegister.
// build: hipcc -O2 -o build/microbench_copy tools/microbench_copy.cpp ; run on the GPU box
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static double timeit(F f, int reps = 20) {
  f();
  double best = 1e9;
  for (int i = 0; i < reps; ++i) { const double t0 = now(); f(); const double dt = now() - t0; if (dt < best) best = dt; }
  return best * 1e6;
}
int main() {
  const size_t sizes[] = {64 << 10, 2073600, 2800000, 3627456, 8 << 20};
  void* d = nullptr;
  hipMalloc(&d, 64 << 20);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  void* pin = nullptr;
  hipHostMalloc(&pin, 64 << 20, hipHostMallocDefault);
  std::vector<char> page(64 << 20, 1), page2(64 << 20, 2);
  printf("{\n");
  for (size_t n : sizes) {
    const double h2d_page = timeit([&] { hipMemcpy(d, page.data(), n, hipMemcpyHostToDevice); });
    const double h2d_page_async = timeit([&] { hipMemcpyAsync(d, page.data(), n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); });
    const double h2d_pin = timeit([&] { hipMemcpyAsync(d, pin, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); });
    const double d2h_page = timeit([&] { hipMemcpy(page.data(), d, n, hipMemcpyDeviceToHost); });
    const double d2h_pin = timeit([&] { hipMemcpyAsync(pin, d, n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); });
    const double cpu = timeit([&] { memcpy(page2.data(), page.data(), n); });
    const double cpu_to_pin = timeit([&] { memcpy(pin, page.data(), n); });
    const double cpu_from_pin = timeit([&] { memcpy(page.data(), pin, n); });
    const double reg = timeit([&] { hipHostRegister(page.data(), n, hipHostRegisterDefault); hipHostUnregister(page.data()); }, 5);
    printf(" \"%zu\": {\"h2d_pageable_us\": %.1f, \"h2d_pageable_async_us\": %.1f, \"h2d_pinned_us\": %.1f, \"d2h_pageable_us\": %.1f, \"d2h_pinned_us\": %.1f, "
           "\"cpu_memcpy_us\": %.1f, \"cpu_to_pinned_us\": %.1f, \"cpu_from_pinned_us\": %.1f, \"register_unregister_us\": %.1f},\n",
           n, h2d_page, h2d_page_async, h2d_pin, d2h_page, d2h_pin, cpu, cpu_to_pin, cpu_from_pin, reg);
  }
  {  // pitched D2H: 75572 rows of 48 bytes out of a 64-byte pitch (descriptors)
    const int rows = 75572;
    const double p2d_page = timeit([&] { hipMemcpy2D(page.data(), 48, d, 64, 48, rows, hipMemcpyDeviceToHost); });
    const double p2d_pin = timeit([&] { hipMemcpy2DAsync(pin, 48, d, 64, 48, rows, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); });
    const double sync_only = timeit([&] { hipStreamSynchronize(s); });
    int v = 0;
    const double tiny_d2h = timeit([&] { hipMemcpy(&v, d, 4, hipMemcpyDeviceToHost); });
    const double tiny_h2d_async = timeit([&] { hipMemcpyAsync(d, &v, 4, hipMemcpyHostToDevice, s); });
    printf(" \"pitched_d2h_75572x48_of_64\": {\"pageable_us\": %.1f, \"pinned_us\": %.1f}, \"stream_sync_idle_us\": %.2f, \"d2h_4_bytes_us\": %.1f, \"h2d_4_bytes_async_us\": %.1f\n",
           p2d_page, p2d_pin, sync_only, tiny_d2h, tiny_h2d_async);
  }
  printf("}\n");
  return 0;
}
