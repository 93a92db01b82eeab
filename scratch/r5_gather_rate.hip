// How fast does this chip do independent random 8-byte reads from a table of a given size? (the index build's entry look-ups: one per (barcode, hash) entry into a
// 1-4 GB table; DESIGN 3 "HBM random-sector rate").   hipcc --offload-arch=gfx950 -O3 scratch/r5_gather_rate.hip -o scratch/bin/r5_gather_rate && scratch/bin/r5_gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef unsigned long long u64;
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 27; x *= 0x94d049bb133111ebULL; return x ^ (x >> 31); }
template <int PER, typename T>
__global__ __launch_bounds__(256) void gather(const T *__restrict__ tab, u64 mask, u64 n, u64 *__restrict__ out) {
  u64 acc = 0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x * PER) {
    T v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] = tab[mix(i + (u64)k * gridDim.x * blockDim.x) & mask];
#pragma unroll
    for (int k = 0; k < PER; ++k) acc += (u64)v[k];
  }
  if (acc == 0x1234567) out[0] = acc;
}
template <int PER, typename T> void run(const char *name, size_t tableBytes, u64 n, int gridMul) {
  T *tab; u64 *out; hipMalloc(&tab, tableBytes); hipMalloc(&out, 8); hipMemset(tab, 1, tableBytes);
  const u64 mask = tableBytes / sizeof(T) - 1;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int grid = 256 * gridMul;
  gather<PER, T><<<grid, 256>>>(tab, mask, n, out); hipDeviceSynchronize();
  hipEventRecord(a); gather<PER, T><<<grid, 256>>>(tab, mask, n, out); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-8s table %6.0f MB  %d loads in flight per thread, grid %5d x 256: %.1f G look-ups/s (%.1f ms for %.1f G)\n", name, tableBytes / 1e6, PER, grid, n / ms / 1e6, ms, n / 1e9);
  hipFree(tab); hipFree(out);
}
int main() {
  const u64 n = 2ull << 30;
  for (size_t mb : {64, 128, 256, 1024, 4096}) {
    run<1, u64>("u64", mb << 20, n, 32); run<4, u64>("u64", mb << 20, n, 32); run<8, u64>("u64", mb << 20, n, 16);
  }
  run<8, uint8_t>("u8", (size_t)226 << 20 > 0 ? (size_t)256 << 20 : 0, n, 16);
  run<8, uint8_t>("u8", (size_t)128 << 20, n, 16);
  return 0;
}
