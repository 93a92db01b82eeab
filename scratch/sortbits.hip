// microbench: rocPRIM radix_sort_pairs u64->u32, 18.25M entries, key bits 42 / 40 / 38 / 37
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <cstdint>
int main() {
  const size_t n = 18250191;
  std::vector<uint64_t> h(n); uint64_t x = 88172645463325252ull;
  for (size_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = x >> 22; }
  uint64_t *kin, *kout; uint32_t *vin, *vout; void *tmp = nullptr; size_t tb = 0;
  hipMalloc(&kin, n * 8); hipMalloc(&kout, n * 8); hipMalloc(&vin, n * 4); hipMalloc(&vout, n * 4);
  hipMemcpy(kin, h.data(), n * 8, hipMemcpyHostToDevice); hipMemset(vin, 0, n * 4);
  rocprim::radix_sort_pairs(nullptr, tb, kin, kout, vin, vout, n, 0, 42, 0);
  hipMalloc(&tmp, tb);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int bitsList[] = {42, 40, 38, 37, 32, 24};
  for (int bits : bitsList) {
    for (int w = 0; w < 2; ++w) rocprim::radix_sort_pairs(tmp, tb, kin, kout, vin, vout, n, 0, bits, 0);
    hipEventRecord(a, 0);
    for (int it = 0; it < 5; ++it) rocprim::radix_sort_pairs(tmp, tb, kin, kout, vin, vout, n, 0, bits, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("bits %d: %.3f ms\n", bits, ms / 5);
  }
  return 0;
}
