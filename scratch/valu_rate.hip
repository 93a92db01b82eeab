// VALU issue rate on gfx950: N waves per SIMD each running a chain of independent 32-bit integer adds / ands / compares.
// prints wave-instructions per SIMD-cycle. usage: valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void k(unsigned *out, int iters) {
  unsigned a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (KIND == 0) { a0 += a1; a1 += a2; a2 += a3; a3 += a4; a4 += a5; a5 += a6; a6 += a7; a7 += a0; }
      else if (KIND == 1) { a0 = (a0 & a1) ^ i; a1 = (a1 | a2) ^ i; a2 = (a2 & a3) + 1; a3 = (a3 ^ a4) + 1; a4 = a4 + (a5 >> 1); a5 = a5 + (a6 << 1); a6 = a6 ^ a7; a7 = a7 + a0; }
      else { a0 = a0 * a1; a1 = a1 * a2; a2 = a2 * a3; a3 = a3 * a4; a4 = a4 * a5; a5 = a5 * a6; a6 = a6 * a7; a7 = a7 * a0; }   // v_mul_lo_u32
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
  unsigned *d; hipMalloc(&d, 256 * 8 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int kind = 0; kind < 3; ++kind)
    for (int wavesPerSimd : {1, 2, 4, 8}) {
      const int threads = 256 * wavesPerSimd > 1024 ? 1024 : 256 * wavesPerSimd, blocksPerCU = 256 * wavesPerSimd / threads;
      const int grid = 256 * blocksPerCU;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) k<0><<<grid, threads>>>(d, iters); else if (kind == 1) k<1><<<grid, threads>>>(d, iters); else k<2><<<grid, threads>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double insts = (double)iters * 16 * 8 * (grid * (threads / 64));          // wave-instructions (the adds alone)
      printf("kind %d waves/SIMD %d: %.3f ms, %.3f wave-inst per SIMD-cycle at 2.4 GHz (%.2f G wave-inst/s)\n", kind, wavesPerSimd, ms, insts / (ms * 1e-3) / (1024 * 2.4e9), insts / (ms * 1e-3) / 1e9);
    }
  return 0;
}
