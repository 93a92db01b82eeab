// rocPRIM onesweep configurations on gfx950 for the index build's big sort (64-bit keys, bits [20, 58)): time per sort of n keys.   hipcc --offload-arch=gfx950 -O3 scratch/r5_sort_cfg.hip -o /tmp/sortcfg && /tmp/sortcfg [n]
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
__global__ void fill(u64 *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; const size_t st = (size_t)gridDim.x * blockDim.x; for (; i < n; i += st) { u64 x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; p[i] = x >> 6; } }
template <class Config> static void run(const char *name, u64 *in, u64 *out, size_t n) {
  size_t bytes = 0; void *tmp = nullptr;
  rocprim::radix_sort_keys<Config>(nullptr, bytes, in, out, n, 20u, 58u, 0);
  hipMalloc(&tmp, bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(a, 0);
    rocprim::radix_sort_keys<Config>(tmp, bytes, in, out, n, 20u, 58u, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (it && ms < best) best = ms;
  }
  printf("%-28s %8.2f ms  %.2f TB/s (5 passes x 16 B + histogram 8 B per key)\n", name, best, (double)n * 88 / (best * 1e-3) / 1e12); fflush(stdout);
  hipFree(tmp);
}
using namespace rocprim;
#define CFG(T, I, R, ALG) radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<T, I>, kernel_config<T, I>, R, block_radix_rank_algorithm::ALG>, 1024 * 1024>
int main(int argc, char **argv) {
  const size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 730000000ull;
  u64 *in, *out; hipMalloc(&in, n * 8); hipMalloc(&out, n * 8);
  fill<<<4096, 256>>>(in, n); hipDeviceSynchronize();
  run<default_config>("default", in, out, n);
  run<CFG(1024, 8, 8, match)>("1024x8 r8 match", in, out, n);
  run<CFG(1024, 10, 8, match)>("1024x10 r8 match", in, out, n);
  run<CFG(768, 10, 8, match)>("768x10 r8 match", in, out, n);
  // round 6: 38 key bits are five passes of 8 bits — four of 10?
  run<CFG(1024, 8, 10, match)>("1024x8 r10 match (4 passes)", in, out, n);
  run<CFG(512, 12, 10, match)>("512x12 r10 match (4 passes)", in, out, n);
  run<CFG(1024, 6, 10, match)>("1024x6 r10 match (4 passes)", in, out, n);
  return 0;
}
