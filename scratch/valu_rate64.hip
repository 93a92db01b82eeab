// Issue rate of the 64-bit integer instructions the k-mer hash needs (gfx950): v_mad_u64_u32, v_lshlrev_b64 / v_lshrrev_b64,
// v_cmp_lt_u64 + v_cndmask pair, next to v_mul_lo_u32 and v_add_u32. Prints wave-instructions per SIMD-cycle (2.4 GHz assumed).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
template <int KIND>
__global__ void k(u64 *out, int iters, u64 F, int sh) {
  u64 a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
  unsigned b0 = threadIdx.x, b1 = b0 + 3, b2 = b0 + 5, b3 = b0 + 9;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (KIND == 0) {        // v_mad_u64_u32: 32x32 -> 64 plus 64
        a0 = (u64)(unsigned)a0 * (unsigned)F + a1; a1 = (u64)(unsigned)a1 * (unsigned)F + a2; a2 = (u64)(unsigned)a2 * (unsigned)F + a3; a3 = (u64)(unsigned)a3 * (unsigned)F + a0;
      } else if (KIND == 1) { // 64-bit shifts by a scalar amount
        a0 = (a0 << sh) ^ a1; a1 = (a1 >> sh) ^ a2; a2 = (a2 << sh) ^ a3; a3 = (a3 >> sh) ^ a0;
      } else if (KIND == 2) { // 64-bit compare + select
        a0 = a0 < a1 ? a0 + 1 : a1; a1 = a1 < a2 ? a1 + 1 : a2; a2 = a2 < a3 ? a2 + 1 : a3; a3 = a3 < a0 ? a3 + 1 : a0;
      } else if (KIND == 3) { // v_mul_lo_u32
        b0 = b0 * b1; b1 = b1 * b2; b2 = b2 * b3; b3 = b3 * b0;
      } else if (KIND == 4) { // v_mul_hi_u32
        b0 = __umulhi(b0, b1); b1 = __umulhi(b1, b2) + 3; b2 = __umulhi(b2, b3) + 5; b3 = __umulhi(b3, b0) + 7;
      } else if (KIND == 5) { // full 64 x 64 -> low 64 (what (f * factor1) compiles to)
        a0 = a0 * F + 1; a1 = a1 * F + 1; a2 = a2 * F + 1; a3 = a3 * F + 1;
      } else {                // v_mul_u32_u24
        b0 = __umul24(b0, b1) + 1; b1 = __umul24(b1, b2) + 1; b2 = __umul24(b2, b3) + 1; b3 = __umul24(b3, b0) + 1;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3;
}
int main() {
  u64 *d; hipMalloc(&d, 256 * 8 * 1024 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 5000;
  const char *names[] = {"v_mad_u64_u32", "64-bit shift + xor", "u64 compare + select + add", "v_mul_lo_u32", "v_mul_hi_u32 (+add)", "u64 * u64 low (+add)", "v_mul_u32_u24 (+add)"};
  for (int kind = 0; kind < 7; ++kind)
    for (int wavesPerSimd : {2, 8}) {
      const int threads = 256 * wavesPerSimd > 1024 ? 1024 : 256 * wavesPerSimd, blocksPerCU = 256 * wavesPerSimd / threads;
      const int grid = 256 * blocksPerCU;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        switch (kind) {
          case 0: k<0><<<grid, threads>>>(d, iters, 0x9E3779B97F4A7C15ULL, 3); break; case 1: k<1><<<grid, threads>>>(d, iters, 1, 3); break;
          case 2: k<2><<<grid, threads>>>(d, iters, 1, 3); break; case 3: k<3><<<grid, threads>>>(d, iters, 1, 3); break;
          case 4: k<4><<<grid, threads>>>(d, iters, 1, 3); break; case 5: k<5><<<grid, threads>>>(d, iters, 0x9E3779B97F4A7C15ULL, 3); break;
          default: k<6><<<grid, threads>>>(d, iters, 1, 3); break;
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double groups = (double)iters * 16 * 4 * (grid * (threads / 64));          // source-level operations per wave
      printf("%-28s waves/SIMD %d: %.3f ms, %.1f SIMD-cycles per source operation per wave\n", names[kind], wavesPerSimd, ms, (ms * 1e-3) * 2.4e9 * 1024 / groups);
    }
  return 0;
}
