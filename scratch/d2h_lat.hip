// microbench: latency of "tiny kernel + 4-byte D2H + stream sync" with a pageable vs a pinned destination
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void bump(unsigned *p) { if (threadIdx.x == 0) *p += 1; }
int main() {
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  unsigned *d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
  unsigned pageable = 0; unsigned *pinned; hipHostMalloc((void **)&pinned, 64, hipHostMallocDefault);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      auto t0 = std::chrono::steady_clock::now();
      const int N = 2000;
      for (int i = 0; i < N; ++i) {
        bump<<<1, 64, 0, st>>>(d);
        if (mode == 0) hipMemcpyAsync(&pageable, d, 4, hipMemcpyDeviceToHost, st);
        else if (mode == 1) hipMemcpyAsync(pinned, d, 4, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
      }
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
      if (rep) printf("%s: %.1f us per kernel+copy+sync\n", mode == 0 ? "pageable" : mode == 1 ? "pinned" : "no copy", us);
    }
  }
  return 0;
}
