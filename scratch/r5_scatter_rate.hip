// The companion of r5_gather_rate.hip: how fast does this chip do independent random WRITES (4 and 8 bytes) and random 8-byte compare-and-swaps into a table of a given size?
// (the sharded index build's reply_kernel scatters 4-byte indices into arrival order; priv_insert_kernel / owner_table_insert_kernel claim 8-byte slots by CAS; DESIGN 3 and 5 quote the rates.)
//   hipcc --offload-arch=gfx950 -O3 scratch/r5_scatter_rate.hip -o scratch/bin/r5_scatter_rate && scratch/bin/r5_scatter_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef unsigned long long u64;
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 27; x *= 0x94d049bb133111ebULL; return x ^ (x >> 31); }
template <typename T>
__global__ __launch_bounds__(256) void scatter(T *__restrict__ tab, u64 mask, u64 n) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) tab[mix(i) & mask] = (T)i;
}
__global__ __launch_bounds__(256) void cas8(u64 *__restrict__ tab, u64 mask, u64 n) {          // every key claims the first empty slot from its home on (linear probing), as the insert kernels do
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    u64 slot = mix(i) & mask;
    for (int d = 0; d < 256; ++d, slot = (slot + 1) & mask)
      if (atomicCAS((unsigned long long *)&tab[slot], ~0ULL, (unsigned long long)i) == ~0ULL) break;
  }
}
template <typename T> void runScatter(const char *name, size_t tableBytes, u64 n) {
  T *tab; hipMalloc(&tab, tableBytes); hipMemset(tab, 0, tableBytes);
  const u64 mask = tableBytes / sizeof(T) - 1;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  scatter<T><<<256 * 32, 256>>>(tab, mask, n); hipDeviceSynchronize();
  hipEventRecord(a); scatter<T><<<256 * 32, 256>>>(tab, mask, n); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("scatter %-4s table %6.0f MB: %.1f G writes/s (%.1f ms for %.2f G)\n", name, tableBytes / 1e6, n / ms / 1e6, ms, n / 1e9);
  hipFree(tab);
}
void runCas(size_t tableBytes, double load) {
  u64 *tab; hipMalloc(&tab, tableBytes);
  const u64 slots = tableBytes / 8, mask = slots - 1, n = (u64)(slots * load);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipMemset(tab, 0xFF, tableBytes); cas8<<<256 * 32, 256>>>(tab, mask, n); hipDeviceSynchronize();
  hipMemset(tab, 0xFF, tableBytes); hipDeviceSynchronize();
  hipEventRecord(a); cas8<<<256 * 32, 256>>>(tab, mask, n); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("cas 8 B      table %6.0f MB filled to %2.0f %%: %.1f G inserts/s (%.1f ms for %.2f G keys)\n", tableBytes / 1e6, 100 * load, n / ms / 1e6, ms, n / 1e9);
  hipFree(tab);
}
int main() {
  const u64 n = 1ull << 30;
  for (size_t mb : {64, 256, 1024, 4096}) { runScatter<uint32_t>("u32", mb << 20, n); runScatter<u64>("u64", mb << 20, n); }
  for (size_t mb : {64, 512, 4096}) { runCas(mb << 20, 0.25); runCas(mb << 20, 0.42); }
  return 0;
}
