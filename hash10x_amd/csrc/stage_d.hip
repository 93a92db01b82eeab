// stage_d.hip — the crib: where two truth genomes hold each hash (accuracy check of SURVEY §8f-2).
//
// Replaces cribAddGenome + the classification loop of cribBuild (hash10x.c:426-521). The reference walks every
// sequence with the mosh iterator and updates crib[index] in order of appearance: the first occurrence sets
// (chr, pos >> 10), the second turns chr into -1, every further one decrements it. Order-independent form:
//   count[index] = occurrences,   first[index] = min (sequence number, position)   =>   chr = count == 1 ? seq : -(count - 1)
// so one lane can own any run of k-mer positions: it rolls the forward and reverse-complement words over its run
// (the same arithmetic as mosh_lds_kernel: seqhash.c:58-80,154-195), looks the moshes up in the probe table
// (find only: hashIndexFind(hash, FALSE)) and posts count / first with atomics.
#include "common.hpp"
#include "prim.hpp"

namespace h10x {

constexpr int CRIB_RUN = 64;                               // k-mer start positions per lane
enum { CRIB_ERR = 0, CRIB_HTA = 1, CRIB_HTB = 2, CRIB_HOM = 3, CRIB_MUL = 4 };   // hash10x.c:411-415

// runStart[s] = number of runs in sequences before s; a sequence shorter than k has none (seqhash.c:162)
__global__ __launch_bounds__(256)
void crib_scan_kernel(const u8 *__restrict__ codes, const u64 *__restrict__ seqStart, const u64 *__restrict__ runStart, u32 nSeq,
                      int k, int w, u64 factor1, const u32 *__restrict__ table, const u64 *__restrict__ hashValue, int B,
                      u32 *__restrict__ count, u64 *__restrict__ first, u64 *__restrict__ tallies /* present, absent */) {
  const u64 nRuns = runStart[nSeq];
  const int shift1 = 64 - 2 * k, k2 = 2 * k;
  const u64 mask = k2 == 64 ? ~0ULL : ((1ULL << k2) - 1);
  u64 present = 0, absent = 0;
  for (u64 run = (u64)blockIdx.x * blockDim.x + threadIdx.x; run < nRuns; run += (u64)gridDim.x * blockDim.x) {
    u32 lo = 0, hi = nSeq;                                   // largest s with runStart[s] <= run
    while (hi - lo > 1) { const u32 mid = (lo + hi) / 2; if (runStart[mid] <= run) lo = mid; else hi = mid; }
    const u32 s = lo;
    const u64 len = seqStart[s + 1] - seqStart[s];
    const u64 p0 = (run - runStart[s]) * CRIB_RUN, nK = len - (u64)k + 1;
    const int cnt = (int)(nK - p0 < (u64)CRIB_RUN ? nK - p0 : (u64)CRIB_RUN);
    const u8 *b = codes + seqStart[s] + p0;
    u64 f = 0, rc = 0;
    for (int j = 0; j < k - 1; ++j) { const u64 x = b[j] & 3; f = (f << 2) | x; rc = (rc >> 2) | ((3 - x) << (k2 - 2)); }
    for (int j = 0; j < cnt; ++j) {
      const u64 x = b[k - 1 + j] & 3;
      f = ((f << 2) | x) & mask; rc = (rc >> 2) | ((3 - x) << (k2 - 2));          // seqhash.c:72-76
      const u64 hf = (f * factor1) >> shift1, hr = (rc * factor1) >> shift1;      // seqhash.c:58-59
      const u64 h = hf < hr ? hf : hr;
      if (h % (u64)w) continue;
      const u32 ix = probe_find(table, hashValue, B, h);
      if (ix) {
        atomicAdd(&count[ix], 1u);
        atomicMin(&first[ix], ((u64)(s + 1) << 32) | (u64)(u32)(p0 + j));
        ++present;
      } else ++absent;
    }
  }
  for (int o = 32; o; o >>= 1) { present += __shfl_down(present, o); absent += __shfl_down(absent, o); }
  if ((threadIdx.x & (WAVE - 1)) == 0) { if (present) atomicAdd(&tallies[0], present); if (absent) atomicAdd(&tallies[1], absent); }
}

// crib[] of one genome as the reference leaves it: chr = sequence number (I16) for one occurrence, -(count - 1) for more
__device__ __forceinline__ void crib_of(u32 count, u64 first, int &chr, u32 &pos) {
  chr = 0; pos = 0;
  if (!count) return;
  chr = (int)(u32)(first >> 32);                             // c->chr = chr (hash10x.c:439); sequence numbers are 1..32767 (checked by the driver)
  pos = ((u32)first >> 10) & 0xFFFFu;                        // c->pos = pos >> 10 into U16
  if (count >= 2) {                                          // hash10x.c:440-441: -1, -2, ..., -32768, then the I16 wraps to 32767 and the next hit restarts at -1
    const u32 t = (count - 2) % 32769u;
    chr = t < 32768u ? -(int)(t + 1) : 32767;
  }
}

__global__ void crib_classify_kernel(const u32 *__restrict__ c1, const u64 *__restrict__ f1, const u32 *__restrict__ c2, const u64 *__restrict__ f2,
                                     const u32 *__restrict__ depth, u32 hashNumber, u32 histDim,
                                     int16_t *__restrict__ chr, u16 *__restrict__ pos, u8 *__restrict__ type, u32 *__restrict__ hist) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hashNumber) return;
  if (i == 0) { chr[0] = 0; pos[0] = 0; type[0] = 0; return; }
  int a, b; u32 pa, pb;
  crib_of(c1[i], f1[i], a, pa); crib_of(c2[i], f2[i], b, pb);
  int t, slot;                                               // hash10x.c:479-494; histogram slots: 0 err, 1 het, 2 hom, 3 mul
  if (a == 0 && b == 0) { t = CRIB_ERR; slot = 0; }
  else if (a > 0 && b > 0) { t = CRIB_HOM; slot = 2; }
  else if (a < 0 || b < 0) { t = CRIB_MUL; slot = 3; if (b < a) a = b; }
  else { t = a ? CRIB_HTA : CRIB_HTB; slot = 1; if (!a) { a = b; pa = pb; } }
  chr[i] = (int16_t)a; pos[i] = (u16)pa; type[i] = (u8)t;
  const u32 d = depth[i];
  if (d < histDim) atomicAdd(&hist[(size_t)slot * histDim + d], 1u);
  atomicMax(&hist[(size_t)4 * histDim + slot], d + 1);       // arrayMax of the reference's Array after array(a, d, int)
}

int stageD_cribGenome(Ctx *c, const u8 *hostCodes, const u64 *seqStart, u32 nSeq, int which, u64 *nPresent, u64 *nAbsent) {
  hipStream_t st = c->stream;
  if (!c->haveState) return c->fail("no hash state loaded: use readFQB or readHash first");
  if (c->sharded) H10X_TRY(shard_materializeTables(c));     // collective: every rank looks the genomes up in the whole table (replicated crib)
  if (which < 0 || which > 1) return c->fail("crib genome %d: must be 0 or 1", which);
  if (nSeq > 32767) return c->fail("crib genome with %u sequences: CribInfo.chr is a 16-bit integer (hash10x.c:407), at most 32767 are supported", nSeq);
  const u32 U1 = c->hashNumber; const int k = c->prm.k;
  H10X_HIP(c, c->cribCount[which].alloc(U1)); H10X_HIP(c, c->cribFirst[which].alloc(U1));
  H10X_HIP(c, hipMemsetAsync(c->cribCount[which].p, 0, (size_t)U1 * 4, st));
  H10X_HIP(c, hipMemsetAsync(c->cribFirst[which].p, 0xFF, (size_t)U1 * 8, st));
  c->haveCrib = false;
  std::vector<u64> runStart((size_t)nSeq + 1, 0);
  for (u32 s = 0; s < nSeq; ++s) {
    const u64 len = seqStart[s + 1] - seqStart[s];
    runStart[s + 1] = runStart[s] + (len >= (u64)k ? (len - (u64)k + 1 + CRIB_RUN - 1) / CRIB_RUN : 0);
  }
  const u64 total = seqStart[nSeq], nRuns = runStart[nSeq];
  DevBuf<u8> dCodes; DevBuf<u64> dSeq, dRun, tallies;
  H10X_HIP(c, dCodes.alloc(total + 1)); H10X_HIP(c, dSeq.alloc((size_t)nSeq + 1)); H10X_HIP(c, dRun.alloc((size_t)nSeq + 1)); H10X_HIP(c, tallies.alloc(2));
  if (total) H10X_HIP(c, hipMemcpyAsync(dCodes.p, hostCodes, total, hipMemcpyHostToDevice, st));
  H10X_HIP(c, hipMemcpyAsync(dSeq.p, seqStart, ((size_t)nSeq + 1) * 8, hipMemcpyHostToDevice, st));
  H10X_HIP(c, hipMemcpyAsync(dRun.p, runStart.data(), ((size_t)nSeq + 1) * 8, hipMemcpyHostToDevice, st));
  H10X_HIP(c, hipMemsetAsync(tallies.p, 0, 16, st));
  if (nRuns) {
    const unsigned grid = (unsigned)hmin<u64>(divUp(nRuns, 256), (u64)c->numCU * 64);
    crib_scan_kernel<<<grid, 256, 0, st>>>(dCodes.p, dSeq.p, dRun.p, nSeq, k, c->prm.w, c->prm.factor1, c->hashIndex.p, c->hashValue.p, c->prm.B,
                                          c->cribCount[which].p, c->cribFirst[which].p, tallies.p);
    H10X_HIP(c, hipGetLastError());
  }
  u64 ht[2];
  H10X_HIP(c, hipMemcpyAsync(ht, tallies.p, 16, hipMemcpyDeviceToHost, st));
  H10X_HIP(c, hipStreamSynchronize(st));                     // also: the host buffers may be released by the caller now
  if (nPresent) *nPresent = ht[0];
  if (nAbsent) *nAbsent = ht[1];
  c->cribGenomes |= 1 << which;
  return 0;
}

int stageD_cribFinish(Ctx *c) {
  hipStream_t st = c->stream; PrimTemp pt;
  if (c->cribGenomes != 3) return c->fail("cribBuild needs both genomes before the crib can be classified");
  const u32 U1 = c->hashNumber;
  DevBuf<u32> red; H10X_HIP(c, red.alloc(1));
  H10X_TRY(prim_reduce_max_u32(c, pt, c->hashDepth.p, red.p, U1));
  u32 maxDepth = 0;
  H10X_HIP(c, hipMemcpyAsync(&maxDepth, red.p, 4, hipMemcpyDeviceToHost, st));
  H10X_HIP(c, hipStreamSynchronize(st));
  c->cribHistDim = maxDepth + 1;
  H10X_HIP(c, c->cribChr.alloc(U1)); H10X_HIP(c, c->cribPos.alloc(U1)); H10X_HIP(c, c->cribType.alloc(U1));
  H10X_HIP(c, c->cribHist.alloc((size_t)4 * c->cribHistDim + 4));
  H10X_HIP(c, hipMemsetAsync(c->cribHist.p, 0, ((size_t)4 * c->cribHistDim + 4) * 4, st));
  crib_classify_kernel<<<divUp(U1, 256), 256, 0, st>>>(c->cribCount[0].p, c->cribFirst[0].p, c->cribCount[1].p, c->cribFirst[1].p, c->hashDepth.p, U1,
                                                      c->cribHistDim, c->cribChr.p, c->cribPos.p, c->cribType.p, c->cribHist.p);
  H10X_HIP(c, hipGetLastError());
  H10X_HIP(c, hipStreamSynchronize(st));
  for (int g = 0; g < 2; ++g) { c->cribCount[g].release(); c->cribFirst[g].release(); }
  c->cribGenomes = 0; c->haveCrib = true;
  return 0;
}

// h10x_warm: the first launch of a kernel loads the code object of its translation unit (HIP loads them on first use); this one is launched ahead of time
__global__ void warm_stageD_kernel() {}
void warm_stageD(hipStream_t st) { warm_stageD_kernel<<<1, 1, 0, st>>>(); }

}  // namespace h10x
