// Does rocprim::radix_sort_keys honour a bit range that ends at bit 64 on small inputs (merge-sort / single-block paths)?
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned long long u64;
int main() {
  for (size_t n : {300u, 3000u, 40000u, 2000000u})
    for (int b0 : {4, 14})
      for (int b1 : {52, 63, 64}) {
        std::vector<u64> h(n); u64 x = 88172645463325252ULL + n;
        for (size_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; u64 q = (x >> (64 - (b1 - b0))) / 3 * 3 % ((u64)1 << (b1 - b0 - 1)); if (i % 3) q = h[i - 1] >> b0; h[i] = (q << b0) | (i & (((u64)1 << b0) - 1)); }
        u64 *in, *out; hipMalloc(&in, n * 8); hipMalloc(&out, n * 8); hipMemcpy(in, h.data(), n * 8, hipMemcpyHostToDevice);
        size_t bytes = 0; void *tmp = nullptr;
        rocprim::radix_sort_keys(nullptr, bytes, in, out, n, (unsigned)b0, (unsigned)b1, 0); hipMalloc(&tmp, bytes);
        rocprim::radix_sort_keys(tmp, bytes, in, out, n, (unsigned)b0, (unsigned)b1, 0); hipDeviceSynchronize();
        std::vector<u64> g(n); hipMemcpy(g.data(), out, n * 8, hipMemcpyDeviceToHost);
        std::vector<u64> e = h; std::stable_sort(e.begin(), e.end(), [b0](u64 a, u64 b) { return (a >> b0) < (b >> b0); });
        size_t bad = 0; for (size_t i = 0; i < n; ++i) if (g[i] != e[i]) ++bad;
        printf("n %8zu bits [%2d, %2d): %s (%zu differ)\n", n, b0, b1, bad ? "WRONG" : "ok", bad);
        hipFree(in); hipFree(out); hipFree(tmp);
      }
  return 0;
}
