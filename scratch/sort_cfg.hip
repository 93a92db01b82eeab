// Device radix sort of 19 M u64 keys on bits [14, 52) (the index build's packed entries) under rocPRIM onesweep configurations
// of 8 / 9 / 10 / 11 bits per pass: fewer passes against more bins. Prints ms per sort.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstring>
#include <vector>
typedef unsigned long long u64;
template <class Config> static float run(const u64 *in, u64 *out, size_t n, int b0, int b1) {
  size_t bytes = 0; void *tmp = nullptr;
  rocprim::radix_sort_keys<Config>(nullptr, bytes, in, out, n, (unsigned)b0, (unsigned)b1, 0);
  hipMalloc(&tmp, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    rocprim::radix_sort_keys<Config>(tmp, bytes, in, out, n, (unsigned)b0, (unsigned)b1, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  hipFree(tmp);
  return best;
}
template <unsigned BITS, unsigned BS, unsigned IPT> using Cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<BS, IPT>, rocprim::kernel_config<BS, IPT>, BITS, rocprim::block_radix_rank_algorithm::match>>;
int main() {
  const size_t n = 19000000;
  std::vector<u64> h(n); u64 x = 88172645463325252ULL;
  for (size_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = ((x >> 26) << 14) | (i & 0x3FFF); }
  u64 *in, *out; hipMalloc(&in, n * 8); hipMalloc(&out, n * 8); hipMemcpy(in, h.data(), n * 8, hipMemcpyHostToDevice);
  printf("default            : %.3f ms\n", run<rocprim::default_config>(in, out, n, 14, 52));
  printf("8 bits  512 x 12   : %.3f ms\n", run<Cfg<8, 512, 12>>(in, out, n, 14, 52));
  printf("8 bits 1024 x 6    : %.3f ms\n", run<Cfg<8, 1024, 6>>(in, out, n, 14, 52));
  printf("9 bits  512 x 12   : %.3f ms\n", run<Cfg<9, 512, 12>>(in, out, n, 14, 52));
  printf("10 bits 512 x 12   : %.3f ms\n", run<Cfg<10, 512, 12>>(in, out, n, 14, 52));
  printf("10 bits 1024 x 6   : %.3f ms\n", run<Cfg<10, 1024, 6>>(in, out, n, 14, 52));
  printf("10 bits 1024 x 8   : %.3f ms\n", run<Cfg<10, 1024, 8>>(in, out, n, 14, 52));
  return 0;
}
