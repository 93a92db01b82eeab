// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access widths of the cluster kernels (MI355X_MICROARCH.md: "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern"): each kernel moves exactly 1 GiB —
//   store_u16 : 64 lanes x 2 bytes per instruction, contiguous (the handle stores of the translated placement)
//   load_u16  : the same as loads (its pass B)
//   load_u32  : 64 lanes x 4 bytes, 256-byte pieces at random 4-byte-aligned places of a 4 GiB array (the list loads)
//   store_u64 : one 8-byte store per 64 lanes (the result words)
// build: hipcc --offload-arch=gfx950 -O3 -o scratch/bin/cal_widths scratch/cal_widths.hip ; run under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void store_u16(uint16_t *p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint16_t)i; }
__global__ void load_u16(const uint16_t *p, size_t n, unsigned *out) { unsigned s = 0; for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i]; if (s == 0x12345678u) *out = s; }
__global__ void load_u32(const uint32_t *p, size_t pieces, size_t words, unsigned *out) {
  unsigned s = 0; const unsigned lane = threadIdx.x & 63; const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t k = wave; k < pieces; k += nw) { const size_t at = ((k * 0x9E3779B97F4A7C15ull) >> 20) % (words - 64); s += p[at + lane]; }
  if (s == 0x12345678u) *out = s;
}
__global__ void store_u64(unsigned long long *p, size_t n) { const unsigned lane = threadIdx.x & 63; const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t k = wave; k < n; k += nw) if (lane == 0) p[k] = k; }
int main() {
  const size_t GiB = (size_t)1 << 30; void *a = nullptr, *b = nullptr; unsigned *out = nullptr;
  hipMalloc(&a, 4 * GiB); hipMalloc(&b, GiB); hipMalloc(&out, 4); hipMemset(a, 1, 4 * GiB);
  store_u16<<<4096, 256>>>((uint16_t *)b, GiB / 2);
  load_u16<<<4096, 256>>>((const uint16_t *)b, GiB / 2, out);
  load_u32<<<4096, 256>>>((const uint32_t *)a, GiB / 256, 4 * GiB / 4, out);
  store_u64<<<4096, 256>>>((unsigned long long *)b, GiB / 8);
  hipDeviceSynchronize(); printf("done\n"); return 0;
}
