// stage_c.hip — depth filter, good-hash lists and per-barcode clustering.
//
// Replaces hashWithinRangeBuild + goodHashesBuild (hash10x.c:528-539, 738-766), codeClusterFind
// (hash10x.c:770-835), codeClusterReadMerge (hash10x.c:837-868) and the OMP --cluster loop
// (hash10x.c:1241-1261).
//
// The reference's codeClusterFind is dense: per good hash i it clears and rescans an n-entry count
// array (O(n^2) per barcode) and callocs an int per barcode of the data set. Here it is sparse
// (verified bit-exact there): one workgroup owns one barcode and keeps everything
// in LDS —
//   (a) first[cj]   = lowest good-hash rank i >= 1 whose barcode list contains cj   (LDS u16 table,
//                     filled by all waves with CAS-min while streaming the lists from HBM)
//   (b) per rank i  : one wavefront gathers first[] for the <= 256 entries of list i into registers and
//                     finds the mode among values < i (ties -> lowest rank) and their count with
//                     ballot/popcount/readlane — the reference's msBest / msMax / msTot
//   (c) one lane replays the order-dependent part (cluster creation, > 255 abort, labels)
//   (d) per rank i  : the count for the cluster's founding rank (a second gather only when it is not
//                     msBest) and the IEEE double quotient
//   (e) point_sum_kernel: one LANE per barcode adds the quotients in rank order => bit-identical pointToMin
//       (a serial fp64 chain per barcode: 64 barcodes per wave instruction instead of one)
//   (f) read_merge_kernel: connected components over <= 255 labels linked by shared reads, renumbered by
//       ascending minimum label — small workgroups of its own, not 1024 lanes waiting on 255 labels.
// Barcodes whose working set exceeds the LDS budget run the same code on a per-workgroup HBM scratch.
#include "common.hpp"
#include <chrono>
#include "prim.hpp"
#include <type_traits>
#include <rocprim/block/block_radix_sort.hpp>

namespace h10x {


constexpr int CL_THREADS_SMALL = 1024;                     // <= 79 KB working sets, two workgroups per CU
constexpr int CL_THREADS_HUGE = 1024;                      // the whole LDS of a CU, one workgroup per CU
constexpr u16 NONE16 = 0xFFFF;
constexpr u32 NOHANDLE = 0xFFFFFFFFu;                        // (argument of row_mode_hist for a chunk a list does not have)
constexpr int ROWS_IN_FLIGHT = 4;                                 // barcode lists a wavefront keeps in flight
constexpr int RCHUNK = 4;                                   // register chunks: lists up to 256 entries

// ------------------------------------------------------------------------------------------ depth range
__global__ void within_kernel(const u32 *__restrict__ depth, u32 hashNumber, int lo, int hi, u8 *__restrict__ within) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hashNumber) return;
  const int n = (int)depth[i];
  if (n >= lo && n < hi) within[i] = 1;                     // only ever set (hash10x.c:535)
}
__global__ void within_depth_kernel(const u32 *__restrict__ depth, const u8 *__restrict__ within, u32 hashNumber, u32 *__restrict__ out,
                                    const u64 *__restrict__ rowStart, u32 rowShift, u64 *__restrict__ rowInfo /* null, or per hash: where its barcode list starts (>> rowShift) | (depth + 1, 0 outside the range) << 32 */) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < hashNumber) {
    const u32 wd = within[i] ? depth[i] + 1 : 0;             // 0 = outside the range(s); depth + 1 otherwise (one word tells both)
    out[i] = wd; if (rowInfo) rowInfo[i] = (u64)(u32)(rowStart[i] >> rowShift) | ((u64)wd << 32);
  }
}

// per block: keys (depth << 16 | position) of its in-range hashes, appended in any order; KT = u32 while the largest
// in-range depth fits 16 bits (the segmented sort then moves half the bytes)
template <typename KT>
__global__ __launch_bounds__(256)
void good_keys_kernel(const h10x_clushash *__restrict__ ch, const u64 *__restrict__ blockOff, const h10x_block *__restrict__ blocks,
                      u32 nBlocks, const u32 *__restrict__ wdepth /* 0 = not in range, else depth + 1 */,
                      KT *__restrict__ key, u32 *__restrict__ nGood, u32 *__restrict__ segEnd, u32 *__restrict__ entries /* sum of depths, saturating */) {
  __shared__ u32 sCount; __shared__ unsigned long long sDepth;
  for (u32 c = blockIdx.x; c < nBlocks; c += gridDim.x) {
    const u64 o = blockOff[c]; const u32 nHash = blocks[c].nHash;
    __syncthreads();
    if (threadIdx.x == 0) { sCount = 0; sDepth = 0; }
    __syncthreads();
    unsigned long long myDepth = 0;
    if (nHash <= 65535) {                                    // hash10x.c:748-753: bigger blocks are ignored
      for (u32 base = 0; base < nHash; base += blockDim.x) {
        const u32 p = base + threadIdx.x;
        u32 wd = 0; bool good = false;
        if (p < nHash) { wd = wdepth[ch[o + p].hash]; good = wd != 0; if (good) myDepth += wd - 1; }
        const u64 bal = __ballot(good);
        const int lane = threadIdx.x & (WAVE - 1);
        u32 wb = 0;
        if (lane == 0 && bal) wb = atomicAdd(&sCount, (u32)__popcll(bal));
        wb = __shfl(wb, 0);
        if (good) key[o + wb + (u32)__popcll(bal & ((1ULL << lane) - 1))] = (KT)(((KT)(wd - 1) << 16) | (KT)p);
      }
    }
    for (int sft = 32; sft; sft >>= 1) myDepth += __shfl_down(myDepth, sft);
    if ((threadIdx.x & (WAVE - 1)) == 0 && myDepth) atomicAdd(&sDepth, myDepth);
    __syncthreads();
    if (threadIdx.x == 0) { nGood[c] = sCount; segEnd[c] = (u32)o + sCount; entries[c] = sDepth > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)sDepth; }
  }
}
template <typename KT>
__global__ void good_pos_kernel(const KT *__restrict__ key, u64 n, u16 *__restrict__ pos) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) pos[i] = (u16)(key[i] & 0xFFFF);
}
__global__ void offsets32c_kernel(const u64 *__restrict__ off, u32 n, u32 *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (u32)off[i];
}

// The good list of a block in one workgroup: the in-range hashes' depths as keys, their positions as values, sorted in LDS
// (rocPRIM block radix sort, stable), positions written out — instead of key kernel + device-wide segmented sort + position
// kernel. Blocks up to BLOCK_SORT_MAX entries, depths below 2^16; three launch classes by block size like clushash_block_kernel.
// Round 5: ONE random read per entry. The chip does some 55 G independent random reads a second whatever the table's size (scratch/r5_gather_rate.hip: 64 MB .. 4 GB
// tables, 1 .. 8 byte entries, all within 15 %), and this kernel used to make two — the depth byte of every entry, then the list offset of every good one: 4.1 G look-ups
// in 72 ms on the 3 Gb set, i.e. at that wall. rowInfo[] (within_depth_kernel) holds both in a word; the offset rides through the sort as part of the value.
template <int THREADS, int IPT>
__global__ __launch_bounds__(THREADS)
void good_block_kernel(const h10x_clushash *__restrict__ ch, const u64 *__restrict__ blockOff, const h10x_block *__restrict__ blocks,
                       const u32 *__restrict__ list, const u32 *__restrict__ count /* this class's blocks: stageB_blockClassLists */,
                       const u64 *__restrict__ rowInfo /* per hash: list offset | (0 = not in range, else depth + 1) << 32 */, int sortBits,
                       u16 *__restrict__ goodPos, u32 *__restrict__ nGood, u32 *__restrict__ entries /* sum of depths, saturating */,
                       u64 *__restrict__ goodRow /* list descriptor per rank: see good_rows_kernel */) {
  using Sort = rocprim::block_radix_sort<u32, THREADS, IPT, u64>;
  __shared__ typename Sort::storage_type storage;
  __shared__ u32 sCount; __shared__ unsigned long long sDepth;
  const u32 nList = *count;
  for (u32 wi = blockIdx.x; wi < nList; wi += gridDim.x) {
    const u32 c = list[wi];
    const u32 nHash = blocks[c].nHash;                       // (the smallest class also takes the empty blocks: their counts are written too)
    const u64 o = blockOff[c];
    __syncthreads();
    if (threadIdx.x == 0) { sCount = 0; sDepth = 0; }
    __syncthreads();
    // blocked arrangement: a thread holds IPT consecutive positions, so the positions are ascending in the order the sort
    // takes as given, and a STABLE sort on the depth bits alone (one 8-bit pass for depths below 128, where depth and
    // position in one key needed three) leaves equal depths in ascending position
    u32 k[IPT]; u64 v[IPT]; u32 mine = 0; unsigned long long myDepth = 0;
#pragma unroll
    for (int j = 0; j < IPT; ++j) { k[j] = 0xFFFFFFFFu; v[j] = threadIdx.x * IPT + (u32)j; }
    if (nHash) {
      // (positions past the block's end read its last record: loads without a branch around them — written as `if (p < nHash) load` every entry was a branch, a load and a
      // wait of its own, twelve memory latencies in a row per lane — the IPT records and then the IPT look-ups are in flight together)
      u32 hs[IPT]; u64 ri[IPT];
#pragma unroll
      for (int j = 0; j < IPT; ++j) { const u32 p = threadIdx.x * IPT + (u32)j; hs[j] = ch[o + (p < nHash ? p : nHash - 1)].hash; }
#pragma unroll
      for (int j = 0; j < IPT; ++j) ri[j] = rowInfo[hs[j]];
#pragma unroll
      for (int j = 0; j < IPT; ++j) {
        const u32 p = threadIdx.x * IPT + (u32)j; const u32 wd = (u32)(ri[j] >> 32);
        if (p < nHash && wd) { k[j] = wd - 1; ++mine; myDepth += wd - 1; v[j] = (u64)p | ((ri[j] & 0xFFFFFFFFull) << 16); }   // value: position (16 bits: blocks of at most BLOCK_SORT_MAX entries) | list offset << 16
      }
    }
    for (int sft = 32; sft; sft >>= 1) { mine += (u32)__shfl_down((int)mine, sft); myDepth += __shfl_down(myDepth, sft); }
    if ((threadIdx.x & (WAVE - 1)) == 0 && mine) { atomicAdd(&sCount, mine); atomicAdd(&sDepth, myDepth); }
    Sort().sort_to_striped(k, v, storage, 0, sortBits);      // ascending (depth, position): hash10x.c:726-730,758; padding keys last
    __syncthreads();
    const u32 nG = sCount;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {                          // striped: coalesced stores
      const u32 e = (u32)j * THREADS + threadIdx.x;
      if (e < nG) {
        goodPos[o + e] = (u16)v[j];
        // where the rank's barcode list lies and how long it is (the sort key IS the depth): the cluster kernel streams these
        // instead of gathering position -> hash index -> offset, depth per barcode
        goodRow[o + e] = (u64)(u32)(v[j] >> 16) | ((u64)k[j] << 32);
      }
    }
    if (threadIdx.x == 0) { nGood[c] = nG; entries[c] = sDepth > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)sDepth; }
  }
}

__global__ void good_rows_kernel(const h10x_clushash *__restrict__ ch, const u64 *__restrict__ blockOff, const u32 *__restrict__ nGood, const u16 *__restrict__ goodPos, u32 nBlocks,
                                 const u32 *__restrict__ hashDepth, const u64 *__restrict__ rowStart, u32 rowShift, u64 *__restrict__ goodRow);
static int bitsForC(u64 maxValue) { int b = 1; while (b < 64 && (maxValue >> b)) ++b; return b; }

int stageC_depthRange(Ctx *c, int lo, int hi) {
  hipStream_t st = c->stream; PrimTemp pt;
  if (!c->haveState) return c->fail("no hash state loaded: use readFQB or readHash first");
  c->haveGood = false;                                       // (a call that fails half-way must not leave the lists of an earlier one standing: rows[] and the offsets are replaced below)
  c->tstart(T_GOOD);
  const u32 U1 = c->hashNumber; const u32 nBlocks = c->nBlocks; const u64 H = c->nEntries;
  if (!(c->haveRange && lo == c->rangeMin && hi == c->rangeMax)) {       // hash10x.c:530
    if (!c->haveRange) { H10X_HIP(c, c->within.alloc(U1)); H10X_HIP(c, hipMemsetAsync(c->within.p, 0, U1, st)); }
    within_kernel<<<divUp(U1, 256), 256, 0, st>>>(c->hashDepth.p, U1, lo, hi, c->within.p);
    c->haveRange = true; c->rangeMin = lo; c->rangeMax = hi;
    if (hi > 0 && (u32)hi > c->rangeHiMax) c->rangeHiMax = (u32)hi;
  }
  // sharded: the barcode lists of the in-range hashes come from their owners first (they depend on the ranges alone), so that
  // the good lists below can point into them
  XGuard xGuard(c);                                          // (the lists may still be arriving on the exchange stream when this function leaves early)
  if (c->sharded) { c->tstop(T_GOOD); H10X_TRY(shard_exchangeRows(c)); c->tstart(T_GOOD); }
  // goodHashesBuild (hash10x.c:738-766)
  DevBuf<u64> key, keyS; DevBuf<u32> key32, keyS32, off32, segEnd, wdepth, red;
  H10X_HIP(c, off32.alloc((size_t)nBlocks + 1));
  H10X_HIP(c, segEnd.alloc(nBlocks)); H10X_HIP(c, wdepth.alloc(U1)); H10X_HIP(c, red.alloc(2));
  H10X_HIP(c, c->nGood.alloc(nBlocks)); H10X_HIP(c, c->goodPos.alloc(H)); H10X_HIP(c, c->goodEntries.alloc(nBlocks)); H10X_HIP(c, c->goodRow.alloc(H));
  // no in-range depth exceeds this, known without a round trip: the data set's barcode count (Ctx::depthBound) or the ranges' limit
  const u32 goodDepthBound = hmin<u32>(c->depthBound, c->rangeHiMax ? c->rangeHiMax - 1 : 0);
  const bool narrow = goodDepthBound <= 65535u;
  const bool byBlocks = narrow && c->maxBlockHashes <= BLOCK_SORT_MAX;       // every block's list is built and sorted by one workgroup
  DevBuf<u64> rowInfo; if (byBlocks) H10X_HIP(c, rowInfo.alloc(U1));
  within_depth_kernel<<<divUp(U1, 256), 256, 0, st>>>(c->hashDepth.p, c->within.p, U1, wdepth.p, c->rowStart.p, (u32)c->rowShift, rowInfo.p);
  H10X_TRY(prim_reduce_max_u32(c, pt, wdepth.p, red.p, U1));
  if (byBlocks) {
    int db = 1; while (db < 16 && (goodDepthBound >> db)) ++db;
    const int sortBits = db + 1;                                              // one bit more than the largest depth: padding keys sort last
    DevBuf<u32> lists, counts;
    H10X_TRY(stageB_blockClassLists(c, lists, counts));
    const unsigned gridBig = hmin<u32>(nBlocks, (u32)c->numCU * 8), gridSmall = hmin<u32>(nBlocks, 65535u * 4);
    const int side = c->maxBlockHashes > BLOCK_SORT_CAP1 ? 2 : (c->maxBlockHashes > BLOCK_SORT_CAP0 ? 1 : 0);
    ForkGuard forkGuard(c);
    if (side) H10X_TRY(c->forkStreams(side));
#define H10X_GOOD_LAUNCH(T, I, CLS, STREAM)                                                                                         \
    { const u32 *const L = lists.p + (size_t)CLS * nBlocks, *const N = counts.p + CLS; const unsigned grid = CLS == 0 ? gridSmall : gridBig; \
      good_block_kernel<T, I><<<grid, T, 0, STREAM>>>(c->clusHash.p, c->blockOff.p, c->blocks.p, L, N, rowInfo.p, sortBits, c->goodPos.p, c->nGood.p, c->goodEntries.p, c->goodRow.p); }
    H10X_GOOD_LAUNCH(H10X_BS_T0, H10X_BS_I0, 0, st)
    if (side >= 1) H10X_GOOD_LAUNCH(H10X_BS_T1, H10X_BS_I1, 1, c->aux[0])
    if (side >= 2) H10X_GOOD_LAUNCH(1024, 8, 2, c->aux[1])
#undef H10X_GOOD_LAUNCH
    H10X_TRY(c->faultAt(3));
    if (side) H10X_TRY(c->joinStreams(side));
    forkGuard.done();
  }
  else if (narrow) { H10X_HIP(c, key32.alloc(H)); H10X_HIP(c, keyS32.alloc(H)); } else { H10X_HIP(c, key.alloc(H)); H10X_HIP(c, keyS.alloc(H)); }
  if (byBlocks) {}
  else if (narrow) good_keys_kernel<u32><<<hmin<u32>(nBlocks, 16384), 256, 0, st>>>(c->clusHash.p, c->blockOff.p, c->blocks.p, nBlocks, wdepth.p,
                                                                             key32.p, c->nGood.p, segEnd.p, c->goodEntries.p);
  else good_keys_kernel<u64><<<hmin<u32>(nBlocks, 16384), 256, 0, st>>>(c->clusHash.p, c->blockOff.p, c->blocks.p, nBlocks, wdepth.p,
                                                                      key.p, c->nGood.p, segEnd.p, c->goodEntries.p);
  H10X_TRY(prim_reduce_max_u32(c, pt, c->nGood.p, red.p + 1, nBlocks));
  DevBuf<u64> redSum; H10X_HIP(c, redSum.alloc(1));
  H10X_TRY(prim_reduce_sum_u32_u64(c, pt, c->nGood.p, redSum.p, nBlocks));
  u32 hr[2]; u64 sumGood = 0;
  H10X_TRY(c->readback(hr, red.p, 8));
  H10X_TRY(c->readback(&sumGood, redSum.p, 8));
  H10X_TRY(c->syncReadbacks());
  hr[0] = hr[0] ? hr[0] - 1 : 0;                              // wdepth holds depth + 1
  c->maxGoodDepth = hr[0]; c->maxGood = hr[1]; c->meanGood = nBlocks > 1 ? (u32)(sumGood / (nBlocks - 1)) : 0;
  offsets32c_kernel<<<divUp((u64)nBlocks + 1, 256), 256, 0, st>>>(c->blockOff.p, nBlocks + 1, off32.p);
  // ascending (depth, position): qsort by depth, stable => ties by position (hash10x.c:726-730,758)
  if (H && !byBlocks) {
    if (narrow) {
      H10X_TRY(prim_seg_sort_keys_u32(c, pt, key32.p, keyS32.p, (u32)H, nBlocks, off32.p, segEnd.p, 0, 16 + bitsForC(hr[0])));
      good_pos_kernel<u32><<<(unsigned)hmin<u64>(divUp(H, 256), 65535u * 2), 256, 0, st>>>(keyS32.p, H, c->goodPos.p);
    } else {
      H10X_TRY(prim_seg_sort_keys_u64(c, pt, key.p, keyS.p, (u32)H, nBlocks, off32.p, segEnd.p, 0, 16 + bitsForC(hr[0])));
      good_pos_kernel<u64><<<(unsigned)hmin<u64>(divUp(H, 256), 65535u * 2), 256, 0, st>>>(keyS.p, H, c->goodPos.p);
    }
  }
  // list descriptors per good hash, in rank order: written by good_block_kernel; the device-wide sort path adds them here
  if (!byBlocks && nBlocks) good_rows_kernel<<<hmin<u32>(nBlocks, 65535u * 4), 256, 0, st>>>(c->clusHash.p, c->blockOff.p, c->nGood.p, c->goodPos.p, nBlocks, c->hashDepth.p, c->rowStart.p,
                                                                                           (u32)c->rowShift, c->goodRow.p);
  H10X_HIP(c, hipGetLastError());                            // (no round trip here: --cluster, the next command, starts with one)
  H10X_TRY(c->xJoin());                                      // sharded: the in-range lists have arrived before anything queued from here on runs (and before the next collective)
  c->haveGood = true;
  c->tstop(T_GOOD);
  return 0;
}

// ------------------------------------------------------------------------------------------ cluster kernel
struct ClusterArgs {
  h10x_block *blocks; const u64 *blockOff; h10x_clushash *clusHash;
  const u16 *goodPos; const u32 *nGood;
  const u64 *goodRow;                                       // per rank (slice of block c at blockOff[c]): low word = list offset in rows[] (>> rowShift), high word = list length = hashDepth
  const u32 *rows;
  const u32 *list; u32 nList; u32 *workCounter;
  u32 *started;                 // host-visible word per workgroup, set when the workgroup starts (whole-CU class only)
  const u32 *front; u32 nFront;                             // handed out before list[]: the largest barcodes of the launch
  u32 nBlocks; int threshold;
  SegMap segs;                                              // local block number -> global barcode number (what the lists hold); identity when unsharded
  u32 rowShift;                                             // rows[]: a list starts at entry (rs << rowShift) — sharded runs with more than 2^32 list entries
  u32 nBlocksFirst;                                         // size of first[]: barcodes of the whole data set + 1
  u32 firstCap;                                             // ranked placement, test knob: cap on the first[] entries of a block (0 = what the budget leaves)
  u32 hashMinSlots;                                         // translated placement: table slots below which the 10-bit tag is too narrow for the data set's barcode count
  u32 hashBits;                                             // b
  u16 *handles; size_t handleStride; u32 hStride;           // translated placement: HBM slot of the workgroup (handleStride u16 each), u16 per list (a power of two >= 64 that holds the longest list)
  const u32 *entries;                                       // per block: entries of its barcode lists (sum of depths of its good hashes)
  u32 *overflow, *overflowCount;                            // ranked / hashed placement: blocks whose table was too small (re-run in the next larger placement)
  unsigned char *scratch; size_t scratchStride;             // global-mode working set per workgroup
  u32 maxGood;
  u32 ldsBudget;                                            // LDS bytes of the launch class (list loop: as many waves as have room for a histogram)
  u64 *res;                                                 // out, per rank (slice of block c at blockOff[c]): RES_PACK(msBest or NONE16 if msMax < threshold,
                                                            // minShareCount[clusterMin], msTot) — replay_kernel turns it into the rank's pointToMin term in place
  u64 *stats;                                               // [0] sum good, [1] sum good depth, [2] sum nHash, [3] codes
  u64 *phase;                                               // diagnostic per-phase ticks (null = off)
  u32 narrowFirst;                                          // test / A-B knob: keep first[] at 2 bytes per entry in every block
  u32 tpClassT;                                             // packed translated placement: class T (lists of 65 .. 96 entries, two to a unit of three chunks) in use
};

// Working set of the list loop inside a region (LDS or HBM scratch). Since round 3 the per-rank arrays of the reference's
// bookkeeping (list offset and length, msMax, msTot, msBest, label: 15 bytes per rank) are gone from it: the list
// descriptors are read from goodRow[] (built once per --hashDepthRange), what a rank contributes to pointToMin is read off
// the list's histogram in the list loop itself and written to HBM in one word per rank, and the order-dependent labelling
// runs in a kernel of its own (replay_kernel) from those words. What a block keeps on chip is first[], 2 bytes per rank
// and one byte-histogram per wave.
// one word per rank: bits 0-15 msBest of an active rank (NONE16: inactive), 16-39 minShareCount[clusterMin[label]], 40-63 msTot (hash10x.c:803-821)
#define RES_PACK(best, q, tot) ((u64)(best) | ((u64)(q) << 16) | ((u64)(tot) << 40))
constexpr u32 RES_COUNT_MAX = (1u << 24) - 1;               // list lengths beyond this do not fit the word: refused by stageC_cluster
struct Work {
  u16 *first;        // NONE16 = unseen; indexed by barcode (dense) or by the barcode's rank among those present (ranked)
  u32 *bm; u16 *pre; // ranked placement: presence bitmap over all barcodes and its exclusive popcount prefix per word (< 65536 present)
  u16 *root;         // founding rank of the cluster the rank joins (clusterMin[label], hash10x.c:819-821): the rank itself while it is
                     // inactive; NONE16 = active but not known yet (its msBest belongs to the same round: settled after the loop)
  u32 *hist;         // CL_WAVES private byte-histograms over ranks (4 counters per word), all zero between lists
  u32 histWords;     // words per wave
};
__host__ __device__ inline size_t pad16(size_t b) { return (b + 15) & ~(size_t)15; }
__host__ __device__ inline size_t workBytes(u32 nFirst, u32 n, u32 nWaves, u32 bmWords) {
  size_t b = pad16((size_t)nFirst * 2) + pad16((size_t)bmWords * 4) + pad16((size_t)bmWords * 2);
  b += pad16((size_t)n * 2);                                // root
  b += (size_t)nWaves * (((size_t)n + 3) / 4) * 4;          // hist (the list of unsettled ranks, 2 n bytes at most, overlays it after the loop: nWaves >= 5)
  return b + 16;
}
// waves of the workgroup that take part in the list loop of a barcode with n ranks: as many as have room for a private
// histogram in what the fixed part leaves of `budget` (0 = does not fit)
constexpr u32 MIN_HIST_WAVES = 5;
constexpr u32 RANKED_FIRST_PER_RANK = 6;                    // ranked placement: first[] entries budgeted per rank when a block is classified
                                                           // (3-7 barcodes present per rank on the synthetic sets); the kernel then takes
                                                           // whatever the budget leaves, and a block that still overflows is re-run with
                                                           // first[] on an HBM slot
__host__ __device__ inline u32 rankedFirstEstimate(u32 nBarcodes, u32 n) { const u64 e = (u64)RANKED_FIRST_PER_RANK * n; return e < nBarcodes ? (u32)e : nBarcodes; }
// … and, where the entries of the block's lists are known (classification): a seventh of them (6-8 % on the yeast-like
// sets, more on deeper ones), whichever is larger. Erring low is cheap: the ranked kernel knows the true number right
// after its bitmap pass and hands the block on before the list loop.
__host__ __device__ inline u32 rankedFirstEstimateE(u32 nBarcodes, u32 n, u32 entries, u32 div = 7) { const u32 a = rankedFirstEstimate(nBarcodes, n), b = entries / div; const u32 m = a > b ? a : b; return m < nBarcodes ? m : nBarcodes; }
__host__ __device__ inline u32 histWaves(u32 nFirst, u32 n, u32 maxWaves, u32 bmWords, size_t budget) {
  const size_t fixed = workBytes(nFirst, n, 0, bmWords), per = (((size_t)n + 3) / 4) * 4;
  if (fixed + MIN_HIST_WAVES * per > budget) return 0;
  const size_t wv = (budget - fixed) / (per ? per : 1);
  return wv < maxWaves ? (u32)wv : maxWaves;
}
__device__ inline Work carve(unsigned char *base, u32 nFirst, u32 n, u32 bmWords = 0) {
  Work w; size_t o = 0;
  w.first = (u16 *)(base + o); o += pad16((size_t)nFirst * 2);
  w.bm = (u32 *)(base + o); o += pad16((size_t)bmWords * 4);
  w.pre = (u16 *)(base + o); o += pad16((size_t)bmWords * 2);
  w.root = (u16 *)(base + o); o += pad16((size_t)n * 2);
  w.hist = (u32 *)(base + o); w.histWords = (n + 3) / 4;
  return w;
}
// read-merge working set: readRep[nRep] bytes, adj[256][8] u32, comp[256] u32, newLab[512] u32
__host__ __device__ inline size_t mergeBytes(u32 nRep) {
  return (((size_t)nRep + 15) & ~(size_t)15) + 256 * 8 * 4 + 256 * 4 + 512 * 4 + 16;
}

// Loads of words that other waves of the workgroup modify with ATOMICS. Atomics on global memory execute in
// L2 and leave this CU's vector L1 untouched, so in the HBM-scratch instantiation a plain load may return a
// stale L1 line (left there by an earlier barcode of the same workgroup); agent-scope loads bypass L1.
// In the LDS instantiation they are ordinary ds_reads.
template <bool IN_LDS, typename T> __device__ __forceinline__ T ld_shared(const T *p) {
  if (IN_LDS) return *p;
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// CAS-min on a u16 living in a u32 word (LDS or global)
template <bool IN_LDS>
__device__ __forceinline__ void min_u16(u16 *arr, u32 idx, u32 val) {
  u32 *w = (u32 *)arr + (idx >> 1); const int sh = (idx & 1) * 16;
  u32 old = ld_shared<IN_LDS>(w);
  for (;;) {
    if (((old >> sh) & 0xFFFFu) <= val) return;
    const u32 nw = (old & ~(0xFFFFu << sh)) | (val << sh);
    const u32 prev = atomicCAS(w, old, nw);
    if (prev == old) return;
    old = prev;
  }
}

// first[cj] = lowest rank >= 1 of this barcode's good hashes whose list holds barcode cj (the reference's minShare), three
// placements behind one interface (update = minimum with rank i, lookup = current value, NONE16 if unseen):
//  * FirstDense : a u16 per barcode of the data set — in LDS while that is small, else on a per-workgroup HBM slot;
//  * FirstRanked: a u16 per barcode PRESENT in this block's lists, found through a presence bitmap over all barcodes and a
//    per-word popcount prefix (3 LDS reads instead of 1; 1 bit + 1/16 B per barcode of the data set);
//  * (data sets whose bitmap no longer fits: the translated placement, SlotTable / FirstSlots below — a barcode's lists only ever
//    meet some thousand others, whatever the data set holds)
// update() returns a handle under which peek() finds the entry again without repeating the search: the list loop reads
// every entry twice (before and after the round's barrier).
// Dense and ranked placements in LDS come in two widths, chosen per block: a u32 per entry where the block's working set
// leaves the room (minimum = one fire-and-forget ds_min_u32, no loop), the u16 of the CAS-min otherwise. The list loop is
// bound by scalar and branch instructions (0.8 per CU cycle), and the CAS loop is mostly those.
template <bool FIRST_LDS> struct FirstDense {
  static constexpr bool MASK_TAIL = false;
  static constexpr bool SELF = true;                          // a lane without an entry (and the barcode's own number) keeps the barcode's own slot as its handle: first[] of the barcode
                                                             // itself is never written and reads "unseen", so the read-back needs no test and no exec mask
  u32 none;                                                  // (the barcode's own number: set per barcode)
  u16 *first; u32 wide;                                      // wide: 1 = 4-byte entries (their low half, at the same address, is the value), else 0
  __device__ __forceinline__ u32 update(u32 cj, u32 i) const {
    if (FIRST_LDS && wide) atomicMin(&((u32 *)first)[cj], i); else min_u16<FIRST_LDS>(first, cj, i);
    return cj;
  }
  __device__ __forceinline__ u32 peek(u32 h) const {         // one read for both widths, no branch
    if (FIRST_LDS) return *(const u16 *)((const unsigned char *)first + (u32)(h << (1 + wide)));
    return ld_shared<FIRST_LDS>(&first[h]);
  }
  __device__ __forceinline__ u32 lookup(u32 cj) const { return peek(cj); }
  __device__ __forceinline__ u32 entry(const void *row, u32 j, u32 code) const { const u32 cj = ((const u32 *)row)[j]; return cj != code ? lookup(cj) : (u32)NONE16; }   // first[] of entry j of a list
};
struct FirstRanked {
  static constexpr bool SELF = false;
  static constexpr bool MASK_TAIL = false;
  u16 *first; const u32 *bm; const u16 *pre; u32 wide;
  u32 none;                                                  // handle of a lane without an entry: one slot behind the last barcode present, always unseen
  __device__ __forceinline__ u32 at(u32 cj) const { const u32 w = cj >> 5; return pre[w] + (u32)__popc(bm[w] & ((1u << (cj & 31)) - 1u)); }
  __device__ __forceinline__ u32 update(u32 cj, u32 i) const { const u32 h = at(cj); if (wide) atomicMin(&((u32 *)first)[h], i); else min_u16<true>(first, h, i); return h; }
  __device__ __forceinline__ u32 peek(u32 h) const { return *(const u16 *)((const unsigned char *)first + (u32)(h << (1 + wide))); }   // (a 32-bit offset: as size_t the shift is a 64-bit one)
  __device__ __forceinline__ u32 lookup(u32 cj) const { return peek(at(cj)); }
  __device__ __forceinline__ u32 entry(const void *row, u32 j, u32 code) const { const u32 cj = ((const u32 *)row)[j]; return cj != code ? lookup(cj) : (u32)NONE16; }
};
// ---- translated placement (round 4), for data sets whose barcodes no longer fit a dense or ranked first[]: the lists of a
// block are read ONCE, by a pass of their own (pass A) that looks every entry's barcode up in an open-addressing table in LDS
// (SlotTable: the entry of a barcode holds the lowest rank that met it — the final first[] value, hash10x.c:794-799 — and is
// inserted the first time the barcode turns up) and parks the entry's SLOT NUMBER, 16 bits, on an HBM slot of the workgroup.
// The list loop proper (pass B) then runs the dense form on those handles: one ds_read_u16 per entry (FirstSlots), no probing,
// no update and no barrier per round — first[] is final when it starts. Between the two the table gives its LDS back: pass B
// needs 2 of its 4 bytes per slot (the ranks, compacted in place), and that is what lets nearly every block of a million-
// barcode set run two workgroups per CU where the one-pass hashed form needed a whole CU for table + histograms.
// Table layout: buckets of 4 words, entry = rank << 16 | bucket displacement << 10 | tag, 0xFFFFFFFF = empty;
// (home bucket, tag) of barcode cj: x = the b scrambled bits of cj at the top of a word (b = bits of the data set's barcode
// count), home = floor(x NB / 2^32) (one v_mul_hi), tag = the top bits of the fraction (x NB mod 2^32) — barcodes of one home
// bucket are consecutive x, their fractions NB apart, so b - floor(log2 NB) <= 10 bits of it tell them apart.
constexpr u32 TR_QUEUE = 16384;                              // entries of a wave's queue in pass A (8 bytes each, behind the handles on the workgroup's HBM slot)
constexpr u32 TR_MIN_SLOTS = 1024;                             // tables are not made smaller than this
constexpr u32 TR_MAX_SLOTS = 65532;                          // handles are 16 bits; slot S (<= 65532) is the handle of "no entry" and reads unseen
struct SlotTable {
  u32 *tab; u32 NB /* buckets */, xsh /* 32 - b */, tsh; u32 *ovf; u32 hbase /* handle of the table's slot 0 */;
  u32 base3;                                                 // the table's LDS byte address (probeHome; 0 = the table is not in LDS), kept in a scalar register
  typedef __attribute__((address_space(3))) u32 lds_u32;
  typedef u32 u32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
  static constexpr u32 SCRAMBLE = 0x9E3779u;                 // odd: cj -> cj * SCRAMBLE mod 2^b is a bijection; 24 bits: full-rate multiply (cj < 2^22)
  static constexpr u32 MAXD = 63;
  static constexpr u32 NOTFOUND = 0xFFFFFFFFu;
  __device__ __forceinline__ void shape(u32 *t, u32 slots, u32 hashBits, u32 *o, u32 hb) {
    tab = t; NB = slots / 4; xsh = 32 - hashBits; ovf = o; hbase = hb; base3 = 0;
    const u32 lg = 31 - (u32)__clz((int)NB), x = 32 - hashBits + lg; tsh = x < 31 ? x : 31;
  }
  // (probeHome addresses the table as LDS: the claim / minimum address is the bucket's plus the word. Kept as an opaque SCALAR the base went straight into the shift-adds —
  //  2 vector instructions per chunk instead of 6 — and the kernel was 0.7 % SLOWER on full configs[2]: one more scalar register to spill. HISTORY §9)
  __device__ __forceinline__ void inLds() { base3 = (u32)(size_t)(lds_u32 *)tab; }
  // minimum of the barcode's entry with rank i: returns its handle. The barcode is inserted if it is not there yet and `insert` says
  // so (ins = true then); otherwise NOTFOUND. A table without a free word within MAXD buckets of the barcode's home: *ovf = 1 (the
  // block is abandoned and re-run in the next larger class) when inserting, NOTFOUND when only looking.
  __device__ __forceinline__ u32 search(u32 cj, u32 i, bool insert, bool &ins) const {
    const u32 x = __umul24(cj, SCRAMBLE) << xsh;
    u32 b = __umulhi(x, NB);
    const u32 tag = (x * NB) >> tsh;
    const u32 step = 2 * (tag & 7u) + 1;                     // the probe sequence strides by the tag's low bits: no long runs of full buckets at 7/8 load
    ins = false;                                             // (displacement d and tag still say which home bucket an entry belongs to: home = b - d * step)
    for (u32 key = tag; key < (MAXD << 10); key += 1u << 10) {
      const u32 mine = (i << 16) | key;
      const uint4 e4 = *(const uint4 *)&tab[4 * b];
      const u32 e[4] = {e4.x, e4.y, e4.z, e4.w};
      u32 hit = 4, eh = 0;                                   // already here? (an empty word's low half, displacement 63 with tag 1023, is no key)
#pragma unroll
      for (int w = 3; w >= 0; --w) { const bool m = (e[w] & 0xFFFFu) == key; hit = m ? (u32)w : hit; eh = m ? e[w] : eh; }
      if (hit < 4) { if (eh > (mine | 0xFFFFu)) atomicMin(&tab[4 * b + hit], mine); return hbase + 4 * b + hit; }
#pragma unroll
      for (int w = 0; w < 4; ++w)                            // first empty word, in order: every inserter of a barcode walks the same words, and a word never empties
        if (e[w] == 0xFFFFFFFFu) {
          if (!insert) return NOTFOUND;                      // (a barcode is never put behind an empty word of its probe sequence)
          const u32 old = atomicCAS(&tab[4 * b + w], 0xFFFFFFFFu, mine);
          if (old == 0xFFFFFFFFu) { ins = true; return hbase + 4 * b + w; }
          if ((old & 0xFFFFu) == key) { if (old > (mine | 0xFFFFu)) atomicMin(&tab[4 * b + w], mine); return hbase + 4 * b + w; }
        }
      b += step; if (b >= NB) b -= NB;                       // (NB >= 16 > step: shape())
    }
    if (insert) *ovf = 1;
    return NOTFOUND;
  }
  // The same for the home bucket alone, straight-line: 98 in 100 entries meet their barcode there or an empty word. false = the search
  // must go on (home bucket full of others, the empty word went to another barcode meanwhile, or the barcode is new and the table
  // closed): the caller parks such entries and runs search() on them 64 at a time — inside the list pass a wave would otherwise loop as
  // long as the unluckiest of its 64 lanes, for every chunk.
  __device__ __forceinline__ bool probeHome(u32 cj, u32 i, bool valid /* the lane has an entry */, bool insert, u32 &slot, bool &ins) const {
    const u32 x = __umul24(cj, SCRAMBLE) << xsh;
    const u32 b4 = __umulhi(x, NB) << 2;
    const u32 key = (x * NB) >> tsh;                         // displacement 0
    const u32 mine = (i << 16) | key;
    const u32 aBucket = base3 + (b4 << 2);
    const u32x4 e4 = *(const lds_u32x4 *)(uintptr_t)aBucket;   // (every lane reads: a lane without an entry holds some barcode of the list all the same)
    // which word holds the barcode, else which is the first empty one (selects, no branch per word)
    const bool m0 = (e4.x & 0xFFFFu) == key, m1 = (e4.y & 0xFFFFu) == key, m2 = (e4.z & 0xFFFFu) == key, m3 = (e4.w & 0xFFFFu) == key;
    const bool z0 = e4.x == 0xFFFFFFFFu, z1 = e4.y == 0xFFFFFFFFu, z2 = e4.z == 0xFFFFFFFFu, z3 = e4.w == 0xFFFFFFFFu;
    const bool hit = m0 | m1 | m2 | m3;
    const u32 wh = m0 ? 0u : (m1 ? 1u : (m2 ? 2u : 3u)), wz = z0 ? 0u : (z1 ? 1u : (z2 ? 2u : 3u));
    const u32 word = hit ? wh : wz;
    slot = b4 | word;
    lds_u32 *const pSlot = (lds_u32 *)(uintptr_t)(aBucket + (word << 2));
    // the masks the wave's scalar unit has to form are what this code costs (the pass issues as many scalar as vector instructions): ONE flag leaves
    // the insertion branch — settled: the word was empty, or the same barcode got there first — and the minimum is queued where the flag is known
    // (round 5) two flat exec regions and no flag that lives across a branch: a claim for the lanes whose barcode is new here and that see an empty word, then ONE minimum for
    // the lanes that found their barcode — before or through the claim. (Flags set inside nested branches came back as 0 / 1 registers that were compared again to form the masks.)
    const bool claim = valid && !hit && insert && (z0 | z1 | z2 | z3);
    u32 seen = 0xFFFFFFFEu;                                  // (neither empty nor anybody's key: displacement 63 is never stored; -2 is an inline constant, 0xFFFF was a v_mov of a literal)
    if (claim) { u32 expect = 0xFFFFFFFFu; __hip_atomic_compare_exchange_strong(pSlot, &expect, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); seen = expect; }   // claim the first empty word
    ins = seen == 0xFFFFFFFFu;
    const bool same = (seen & 0xFFFFu) == key;               // the same barcode got there first: its rank takes part in the minimum
    if ((valid && hit) || same) (void)__hip_atomic_fetch_min(pSlot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // the entry's rank: minimum with this one, looked at or not — a fire-and-forget
                                                             // LDS atomic per chunk costs less than finding out whether a later rank got here first
    return !valid || hit || ins || same;
  }
};
struct FirstSlots {                                          // pass B: first[] by handle
  static constexpr bool SELF = false;
  static constexpr bool MASK_TAIL = false;                   // lanes past a list's end hold the handle of `none` (pass A writes whole chunks)
  const unsigned char *base; u32 sh;                         // value of slot h: the u16 at base + (h << sh) — sh = 1 after the compaction, 2 (base at the words' high halves) without
  u32 none;
  __device__ __forceinline__ u32 peek(u32 h) const { return *(const u16 *)(base + (u32)(h << sh)); }
  __device__ __forceinline__ u32 lookup(u32 h) const { return peek(h); }
  __device__ __forceinline__ u32 entry(const void *row, u32 j, u32) const { return peek(((const u16 *)row)[j]); }   // row: the list's handles
};
// slots of a block's table and what pass B then looks like: S = 0 if the block does not fit `budget`. The table gets every byte
// pass A can give it (the emptier, the shorter the probes) but no more than twice the entries of the block's lists — a table at
// most half full cannot overflow — and no more than pass B can keep at 2 bytes per slot beside root[] and five histograms.
// S2: slots of a SECOND table, should the first fill up (only then: it is "closed" at 7/8 and the barcodes that turn up afterwards
// are parked): it lives in the half of the first table's LDS that the compaction of its ranks frees, as far as pass B then still
// has room for 2 bytes per slot of both. 0 = none (the first table holds every barcode the lists can bring, or nothing is left).
__host__ __device__ inline void translatedShape(u32 n, u32 maxWaves, size_t budget, u32 minSlots, u32 entries, u32 nBarcodes, u32 cap /* test knob */,
                                                u32 &S, u32 &S2, u32 &nW, bool &compact) {
  const size_t per = (((size_t)n + 3) / 4) * 4, fixedB = pad16((size_t)n * 2) + 64;
  S = 0; S2 = 0; nW = 0; compact = false;
  if (fixedB + MIN_HIST_WAVES * per + 2 * 256 + 64 >= budget) return;
  const size_t sA = (budget - 64) / 4, sB = (budget - fixedB - MIN_HIST_WAVES * per) / 2 - 16;
  size_t s = sA < sB ? sA : sB;
  if (s > TR_MAX_SLOTS) s = TR_MAX_SLOTS;
  size_t useful = 2 * (size_t)(entries < nBarcodes ? entries : nBarcodes) + 64;
  if (useful < TR_MIN_SLOTS) useful = TR_MIN_SLOTS;
  if (useful < minSlots) useful = minSlots;                  // (the tag's width asks for this many: a block with few entries still gets them)
  const bool all = s >= useful;                              // holds whatever the lists bring
  if (all) s = useful;
  if (cap && s > cap) s = cap;
  s &= ~(size_t)3;
  if (s < 64 || s < minSlots) return;                        // (16 buckets at least: the probe stride is below that)
  S = (u32)s;
  if (!all || cap) {
    const long a2 = ((long)budget - (long)pad16(2 * (s + 8)) - 64) / 4, b2 = ((long)budget - (long)fixedB - (long)(MIN_HIST_WAVES * per) - 2 * (long)(s + 8) - 64) / 2;
    long s2 = a2 < b2 ? a2 : b2;
    if (s2 > (long)TR_MAX_SLOTS - (long)s - 4) s2 = (long)TR_MAX_SLOTS - (long)s - 4;
    if (cap && s2 > (long)cap) s2 = (long)cap;
    s2 &= ~3L;
    if (s2 >= 64 && s2 >= (long)minSlots) S2 = (u32)s2;
  }
  if (pad16(4 * (s + 4)) + fixedB + (size_t)maxWaves * per <= budget) { nW = maxWaves; return; }   // the table stays as it is: every wave has its histogram anyway
  compact = true;
  const size_t wv = (budget - fixedB - pad16(2 * (s + 8))) / (per ? per : 1);
  nW = wv < maxWaves ? (u32)wv : maxWaves;
}
__host__ __device__ inline u32 translatedCloseAt(u32 S, u32 S2) { return S2 ? S - S / 8 : 0xFFFFFFFFu; }   // inserts into the first table stop here when there is a second one
// list-loop waves of pass B when both tables are in use
__host__ __device__ inline u32 translatedWaves2(u32 n, u32 maxWaves, size_t budget, u32 S, u32 S2) {
  const size_t per = (((size_t)n + 3) / 4) * 4, fixedB = pad16((size_t)n * 2) + 64, wv = (budget - fixedB - pad16(2 * ((size_t)S + S2 + 16))) / (per ? per : 1);
  return wv < maxWaves ? (u32)wv : maxWaves;
}
__host__ __device__ inline bool translatedFits(u32 S, u32 S2, u32 est, u32 entries, u32 nBarcodes) {      // classification: do the tables hold the barcodes expected in the block's lists?
  if (!S) return false;
  if ((size_t)S >= 2 * (size_t)(entries < nBarcodes ? entries : nBarcodes)) return true;
  const size_t capacity = S2 ? (size_t)(S - S / 8) + (S2 - S2 / 8) : (size_t)S - S / 8;
  return capacity >= (size_t)est + est / 8;
}

// wave64 max in registers: DPP row shifts (1,2,4,8) then row broadcasts 15/31 (gfx9 DPP), result in lane 63.
// Identity 0 (keys are unsigned); no LDS round trips unlike ds_bpermute-based shuffles.
__device__ __forceinline__ u32 wave_max_u32(u32 v) {
#define H10X_DPP_MAX(ctrl, rowmask) { const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rowmask, 0xf, false); v = o > v ? o : v; }
  H10X_DPP_MAX(0x111, 0xf)      // row_shr:1
  H10X_DPP_MAX(0x112, 0xf)      // row_shr:2
  H10X_DPP_MAX(0x114, 0xf)      // row_shr:4
  H10X_DPP_MAX(0x118, 0xf)      // row_shr:8
  H10X_DPP_MAX(0x142, 0xa)      // row_bcast:15 into rows 1 and 3
  H10X_DPP_MAX(0x143, 0xc)      // row_bcast:31 into rows 2 and 3
#undef H10X_DPP_MAX
  return (u32)__builtin_amdgcn_readlane((int)v, 63);
}

// msBest / msMax / msTot of one barcode list (< 256 entries) for rank i, one wavefront. The gathered first[]
// values are counted in this wave's private byte histogram with RETURNING ds atomics: the lane that arrives
// last at a value sees its full count, so a DPP wave max over (arrival count, lowest rank) is the mode —
// one LDS round trip for the gather, one for the atomics. Lists with fewer usable entries than the
// threshold are skipped (only "msMax < threshold" matters to the caller then).
// Round 3: while the histogram of the list still stands, an active rank (msMax >= threshold) also reads from it what the
// reference adds to pointToMin for it (hash10x.c:821): minShareCount[clusterMin[label]] = the count of the founding rank of
// the cluster it joins, rb = root[msBest] — known as soon as msBest's own round is over, i.e. for all but the few ranks
// whose msBest lies in the round being processed (rb = NONE16: settled after the loop). The second gather of round 2's
// phase (d) is gone for the rest, and with it the need to keep msMax / msTot per rank on chip.
template <bool IN_LDS, int RCHUNK, typename FT>
__device__ __forceinline__ void row_mode_hist(const void *__restrict__ row /* entries beyond the two chunks held in registers: barcodes, or handles (FirstSlots) */, u32 f0, u32 f1 /* first[] of entries lane, 64 + lane, read for all lists of the round together; NONE16 = no entry */, u32 d, u32 code, u32 i, const FT &ft, u32 *hist,
                                              const u16 *root, u32 thr, u32 &best, u32 &bcnt, u32 &tot, u32 &rb, u32 &q) {
  const int lane = threadIdx.x & (WAVE - 1);
  u32 f[RCHUNK]; bool ok[RCHUNK];
  tot = 0;
#pragma unroll
  for (int r = 0; r < RCHUNK; ++r) {
    f[r] = NONE16; ok[r] = false;
    if ((u32)(r * WAVE) < d) {
      const u32 j = r * WAVE + lane;
      if (r < 2) {
        f[r] = ft.peek(r == 0 ? f0 : f1);                     // (f0 / f1 are handles: a lane without an entry holds one that reads "unseen" — or, FirstSlots, the list's
        if (FT::MASK_TAIL && j >= d) f[r] = NONE16;           //  last handle again: told apart here)
        ok[r] = f[r] < i;
      }
      else if (j < d) { f[r] = ft.entry(row, j, code); ok[r] = f[r] < i; }
      tot += (u32)__popcll(__ballot(ok[r]));
    }
  }
  best = NONE16; bcnt = 0; rb = NONE16; q = 0;
  // lists of up to two chunks (nearly all) run straight through: a list with fewer usable entries than the threshold then
  // counts and clears its few values for nothing, which is cheaper than two branches per list (2.41 -> 2.36 ms)
  if (RCHUNK > 2 && tot < thr) return;
  u32 key = 0;
#pragma unroll
  for (int r = 0; r < RCHUNK; ++r)
    if ((u32)(r * WAVE) < d && ok[r]) {
      const int sh8 = (f[r] & 3) * 8;
      const u32 c = ((atomicAdd(&hist[f[r] >> 2], 1u << sh8) >> sh8) & 0xFFu) + 1;
      const u32 k = (c << 16) | (0xFFFFu - f[r]);
      key = k > key ? k : key;
      if (RCHUNK == 1) {                                     // a one-chunk list is cleared in the exec region of its atomics (ds ops of a wave stay in order)
        if (IN_LDS) hist[f[r] >> 2] = 0;
        else atomicAnd(&hist[f[r] >> 2], 0u);                // HBM scratch: a plain store could be overtaken by the next list's atomics
      }
    }
  if (RCHUNK > 1) {                                          // longer lists: once every chunk has been counted (a value's entries may lie in several chunks)
#pragma unroll
    for (int r = 0; r < RCHUNK; ++r)
      if ((u32)(r * WAVE) < d && ok[r]) {
        if (IN_LDS) hist[f[r] >> 2] = 0;
        else atomicAnd(&hist[f[r] >> 2], 0u);
      }
  }
  // no value reached the threshold (most long lists): the caller only tests msMax >= threshold (hash10x.c:807), so the
  // wave reduction is skipped and the rank is reported inactive
  if (!(RCHUNK > 2) || __builtin_amdgcn_ballot_w64(key >= (thr << 16))) {
    key = wave_max_u32(key);
    bcnt = key >> 16;
    best = key ? 0xFFFFu - (key & 0xFFFFu) : NONE16;
    if (bcnt >= thr) {                                       // wave-uniform (readlane result)
      rb = ld_shared<IN_LDS>(&root[best]);
      if (rb != NONE16) {
#pragma unroll
        for (int r = 0; r < RCHUNK; ++r) if ((u32)(r * WAVE) < d) q += (u32)__popcll(__ballot(f[r] == rb));   // (rb <= msBest < i: an entry that equals it is a usable one; the bare compare is a ballot as it stands, the conjunction cost two more instructions)
      }
    }
  }
}
// entries of a list whose first[] value equals v (and, in tot, those below i): one wavefront, any length
template <typename FT>
__device__ __forceinline__ u32 row_count_value(const void *__restrict__ row, u32 d, u32 code, u32 i, const FT &ft, u32 v, u32 &tot) {
  const int lane = threadIdx.x & (WAVE - 1);
  u32 q = 0; tot = 0;
  for (u32 b0 = 0; b0 < d; b0 += WAVE) {
    u32 f = NONE16;
    if (b0 + lane < d) f = ft.entry(row, b0 + lane, code);
    q += (u32)__popcll(__ballot(f == v && v != NONE16)); tot += (u32)__popcll(__ballot(f < i));
  }
  return q;
}
// lists of 256 entries and more (exotic depth ranges): re-gather per candidate
template <typename FT>
__device__ __forceinline__ void row_mode_long(const void *__restrict__ row, u32 d, u32 code, u32 i, const FT &ft, u32 &best, u32 &bcnt, u32 &tot) {
  const int lane = threadIdx.x & (WAVE - 1);
  best = NONE16; bcnt = 0; tot = 0;
  for (u32 a0 = 0; a0 < d; a0 += WAVE) {
    u32 fa = NONE16; bool va = false;
    if (a0 + lane < d) { fa = ft.entry(row, a0 + lane, code); va = fa < i; }
    u64 rem = __ballot(va); tot += (u32)__popcll(rem);
    while (rem) {
      const int src = __ffsll((long long)rem) - 1;
      const u32 v = (u32)__builtin_amdgcn_readlane((int)fa, src);
      rem &= ~__ballot(fa == v);
      u32 cnt = 0; bool seenBefore = false;
      for (u32 b0 = 0; b0 < d; b0 += WAVE) {
        u32 fb = NONE16;
        if (b0 + lane < d) fb = ft.entry(row, b0 + lane, code);
        const u32 m = (u32)__popcll(__ballot(fb == v));
        if (b0 < a0 && m) { seenBefore = true; break; }
        cnt += m;
      }
      if (!seenBefore && (cnt > bcnt || (cnt == bcnt && v < best))) { bcnt = cnt; best = v; }
    }
  }
}
#define STAMP(k) do { if (a.phase && threadIdx.x == 0) { const u64 t__ = wall_clock64(); atomicAdd((u64 *)&a.phase[k], t__ - (u64)acc.s[4]); acc.s[4] = t__; } } while (0)   /* (the previous stamp lives in LDS: diagnostic code must not cost the kernels a register pair) */

// Workgroup barrier. In the HBM-scratch instantiation the working set lives in global memory and is re-used
// by successive phases: drop this CU's vector-L1 copies after every barrier (buffer_inv sc1) so that no phase
// reads a line cached before another wave rewrote it.
#define SYNC() do { __syncthreads(); if (!IN_LDS) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); } while (0)
// LDS-only rendezvous for the list loop: waits for this wave's LDS traffic but leaves its prefetched global loads in
// flight (a full __syncthreads() drains vmcnt too). The HBM-scratch instantiation keeps the full barrier + L1 drop.
#define SYNC_LDS() do { if (IN_LDS) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } else SYNC(); } while (0)

// The list descriptors of a block (goodRow[]) travel in ONE vector register per round, one list per LANE: lane t of a wave
// loads the offset and lane 32 + t the length of the wave's t-th list of a round (one load instruction for all of them,
// three rounds ahead), and the scalars a list's loads and loops need are taken from those lanes with v_readlane when they
// are needed — no LDS array and no init pass, and no scalar register holds a descriptor across rounds (scalar registers
// are what this kernel spills).
struct Desc { u32 x /* list offset in rows[] >> rowShift */, y /* list length */; };
__device__ __forceinline__ Desc descLane(u32 v, int t) { Desc d; d.x = (u32)__builtin_amdgcn_readlane((int)v, t); d.y = (u32)__builtin_amdgcn_readlane((int)v, 32 + t); return d; }
// lane t < nLists: low word of the descriptor of rank i0 + t, lane 32 + t: its high word; 0 for rank 0 and past the last rank
__device__ __forceinline__ u32 descLoad(const u64 *gr, u32 i0, u32 nLists, u32 n) {
  const u32 lane = threadIdx.x & (WAVE - 1), t = lane & 31, i = i0 + t;
  return (t < nLists && i >= 1 && i < n) ? ((const u32 *)gr)[2 * (size_t)i + (lane >> 5)] : 0u;
}

// The same with a load no lane skips (the index is clamped): a load behind a branch may not have been issued as far as the compiler's
// wait-count bookkeeping can tell, and every later wait for an OLDER load then becomes s_waitcnt vmcnt(0) — a prefetch pipeline that pays a
// full memory round trip per round. The offset lanes keep what was loaded (a real list: addresses formed from it are valid), the length
// lanes read 0 for rank 0 and past the last rank.
__device__ __forceinline__ u32 descLoadU(const u64 *gr, u32 i0, u32 n) {   // raw: to be passed through descFix() when the round comes up (not before: that would wait for the load)
  const u32 lane = threadIdx.x & (WAVE - 1), i = i0 + (lane & 31);
  return ((const u32 *)gr)[2 * (size_t)(i < n ? i : n - 1) + (lane >> 5)];
}
__device__ __forceinline__ u32 descFix(u32 v, u32 i0, u32 nLists, u32 n) {
  const u32 lane = threadIdx.x & (WAVE - 1), t = lane & 31, i = i0 + t;
  return (lane < 32 || (t < nLists && i >= 1 && i < n)) ? v : 0u;
}

// work counters of a workgroup: the list entries gathered are summed per lane (32 bits: a wave's share of a launch) and posted when the kernel ends; what thread 0
// alone counts — good hashes, nHash, barcodes, the lengths of the rank-0 lists — lives in LDS (the four 64-bit registers per lane these took were a tenth of the
// 64 a wave has where two 1024-lane workgroups share a CU)
// `bytes` (a multiple of 16, the region 16-byte aligned: every part of the working set is padded so) set to the word `v`, 16 bytes per store. The init phases of a block
// are workgroup-synchronous — every wave does the same thing between two barriers, nobody but the CU's other workgroup hides their latencies — and were written a word
// per store: 20 store instructions per lane for the 79 KB table of the packed form, 4 per lane now (round 6: with the rank compaction 285 -> 275 ms on the 3 Gb set).
template <int CL_THREADS> __device__ __forceinline__ void fill16(void *p, size_t bytes, u32 v) {
  const uint4 v4 = make_uint4(v, v, v, v);
  for (u32 i = threadIdx.x; i < (u32)(bytes / 16); i += CL_THREADS) ((uint4 *)p)[i] = v4;
}
struct WorkAcc { u32 depth; unsigned long long *s; };
template <bool IN_LDS, int FIRST_MODE /* 0 dense in LDS, 1 ranked in LDS, 2 dense on an HBM slot, 3 hashed in LDS */, int CL_THREADS, int KLASS>
__device__ __forceinline__ void cluster_one_block(const ClusterArgs &a, u32 code, unsigned char *region, u16 *firstGlobal, u32 *sh /* small shared ints */, WorkAcc &acc) {   // code: local block number
  constexpr int CL_WAVES = CL_THREADS / WAVE;
  // lists a wave keeps in flight: 4, but 2 where the kernel must stay within 64 VGPRs (two workgroups per CU) AND carries
  // the ranked / hashed lookup: fewer registers spilled is worth more there than the deeper prefetch (8x set: 52.8 -> 41.4 ms);
  constexpr int RIF = (KLASS == 0 && CL_THREADS == 1024 && FIRST_MODE == 1) ? 2 : ROWS_IN_FLIGHT;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
  code = (u32)__builtin_amdgcn_readfirstlane((int)code);     // (the block number reaches every lane through LDS: say that it is uniform)
  const u32 n = (u32)__builtin_amdgcn_readfirstlane((int)a.nGood[code]);
  if (n == 0) return;                                        // hash10x.c:780: block left untouched
  const u64 o = a.blockOff[code];                            // (made scalar with readfirstlane as in the packed form: dense unchanged, ranked 0.8 % slower — left as it is)
  constexpr bool FIRST_LDS = IN_LDS && FIRST_MODE != 2;
  constexpr bool RANKED = IN_LDS && FIRST_MODE == 1;
  const u32 bmWords = RANKED ? (a.nBlocksFirst + 31) / 32 : 0;
  // list-loop waves: all of them where the histograms fit, fewer for a barcode with many ranks (one kernel and one
  // work queue then serve nearly every barcode; the phases behind the loop always use the whole workgroup)
  u32 nW;
  if constexpr (RANKED) nW = 0;                       // ranked placement: decided once the bitmap pass has counted the barcodes present
  else nW = IN_LDS ? histWaves(FIRST_LDS ? a.nBlocksFirst : 0, n, CL_WAVES, bmWords, a.ldsBudget) : (u32)CL_WAVES;
  if (!RANKED && !nW) return;                                // cannot happen: the classification sends such a barcode to the HBM-scratch class
  // dense placement in LDS: 4-byte entries where that costs the list loop no wave (see FirstDense)
  bool wideFirst = false;
  if (FIRST_LDS && !RANKED && a.narrowFirst != 1) {
    const u32 nW4 = histWaves(2 * a.nBlocksFirst, n, CL_WAVES, bmWords, a.ldsBudget);
    wideFirst = nW4 >= nW || (a.narrowFirst >= 2 && nW4 >= a.narrowFirst);   // (A/B: accept down to narrowFirst waves)
    if (wideFirst) nW = nW4;
  }
  // ranked placement: first[] lies BEHIND the histograms and both are laid out after the bitmap pass, when the number of
  // barcodes present is known: the list loop then runs on as many waves as what is left of the budget has room for
  Work w = carve(region, RANKED ? 0 : (FIRST_LDS || !IN_LDS ? (wideFirst ? 2 * a.nBlocksFirst : a.nBlocksFirst) : 0), n, bmWords);
  if (IN_LDS && FIRST_MODE == 2) w.first = firstGlobal;      // hybrid: first[] on this workgroup's HBM slot, the rest in LDS
  typename std::conditional<RANKED, FirstRanked, FirstDense<FIRST_LDS>>::type ft{};
  if constexpr (RANKED) { ft.first = w.first; ft.bm = w.bm; ft.pre = w.pre; ft.wide = 0; }
  else { ft.first = w.first; ft.wide = wideFirst ? 1u : 0u; }
  const u32 lcode = code; code = a.segs.globalOf(lcode);
  if constexpr (!RANKED) ft.none = code;     // from here on `code` is the global barcode number (what the lists hold)
  const u32 rsh = a.rowShift;
#define ROWP(rs) (a.rows + ((size_t)(rs) << rsh))
  const u64 *const gr = a.goodRow + o;
  u64 *const res = a.res + o;
  const u32 thr = (u32)a.threshold;

  if (a.phase && threadIdx.x == 0) acc.s[4] = wall_clock64();
  // ---- init: clear the tables; nothing is fetched per rank any more
  if (FIRST_LDS && !RANKED) {                              // (16 bytes per store: fill16; the parts of the working set are padded to 16 bytes, the histograms have 16 bytes of slack behind them: workBytes)
    if (wideFirst) fill16<CL_THREADS>(w.first, pad16((size_t)a.nBlocksFirst * 4), (u32)NONE16);
    else fill16<CL_THREADS>(w.first, pad16((size_t)a.nBlocksFirst * 2), 0xFFFFFFFFu);
  }
  if (RANKED) fill16<CL_THREADS>(w.bm, pad16((size_t)bmWords * 4), 0u);
  if (!RANKED) fill16<CL_THREADS>(w.hist, pad16((size_t)nW * w.histWords * 4), 0u);
  fill16<CL_THREADS>(w.root, pad16((size_t)n * 2), 0xFFFFFFFFu);     // NONE16 everywhere ...
  if (tid == 0) { w.root[0] = 0; res[0] = RES_PACK(NONE16, 0, 0); }  // ... but rank 0, which is never processed (hash10x.c:789): inactive, its own root (lane 0 wrote its word just above)
  SYNC();
  if constexpr (RANKED) {
    // ---- (0) which barcodes occur in this block's lists: presence bitmap, per-word popcount prefix, and a first[] sized by
    // the number present. More present than the table holds => the block is handed to the HBM-slot variant.
    // eight lists per wave in flight (their chunks are requested together). Round 2 looked at the word before the atomic (most entries
    // meet a bit that is already set, and LDS atomics cost per active lane); since the kernel is bound by its vector instructions
    // and not by LDS time, the unconditional OR is the cheaper one (1/10 config-3 set: cluster 40.9 -> 40.0 ms)
    // (the bitmap is addressed as LDS explicitly: through the generic pointer this build of the compiler emits an invalid compare
    // against the shared aperture base for the volatile look — "Illegal instruction detected" — in the RIF = 2 instantiation)
    typedef __attribute__((address_space(3))) u32 lds_u32;
    lds_u32 *const bm3 = (lds_u32 *)w.bm;
    auto mark = [&](u32 cj) { if (cj != code) __hip_atomic_fetch_or(&bm3[cj >> 5], 1u << (cj & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    // (both chunks of the eight lists are requested together: met one list at a time, the second chunk — most lists have one where the
    // depth range reaches 100 and the lists are long — was a load waited for per list: a third of the launch on the 1/10 config-3 set)
    constexpr int BIF = 8;                                   // lists in flight per wave in this pass (few live registers here)
    const u32 uw = (u32)__builtin_amdgcn_readfirstlane(wave);
    u32 dvN = descLoad(gr, 1 + uw * BIF, BIF, n);
    for (u32 i0 = 1 + uw * BIF; i0 < n; i0 += CL_WAVES * BIF) {
      u32 c0[BIF], c1[BIF], d4[BIF], r4[BIF];
      const u32 dv = dvN;
      dvN = descLoad(gr, i0 + CL_WAVES * BIF, BIF, n);
#pragma unroll
      for (int t = 0; t < BIF; ++t) {
        const Desc g2 = descLane(dv, t);
        d4[t] = g2.y; r4[t] = g2.x;
        const u32 *row = ROWP(r4[t]);
        c0[t] = (u32)lane < d4[t] ? row[lane] : code; c1[t] = (u32)(WAVE + lane) < d4[t] ? row[WAVE + lane] : code;
      }
#pragma unroll
      for (int t = 0; t < BIF; ++t) {
        mark(c0[t]);
        if (d4[t] > WAVE) {
          mark(c1[t]);
          if (d4[t] > 2 * WAVE) { const u32 *row = ROWP(r4[t]); for (u32 j = 2 * WAVE + lane; j < d4[t]; j += WAVE) mark(row[j]); }
        }
      }
    }
    SYNC();
    const u32 ipt = (bmWords + CL_THREADS - 1) / CL_THREADS, s0 = tid * ipt < bmWords ? tid * ipt : bmWords, s1 = s0 + ipt < bmWords ? s0 + ipt : bmWords;
    u32 mine = 0;
    for (u32 q = s0; q < s1; ++q) mine += (u32)__popc(w.bm[q]);
    u32 inc = mine;
#pragma unroll
    for (int dd = 1; dd < WAVE; dd <<= 1) { const u32 o2 = (u32)__shfl_up((int)inc, dd); if (lane >= dd) inc += o2; }
    if (lane == WAVE - 1) sh[4 + wave] = inc;
    SYNC();
    u32 run = inc - mine, total = 0;
    for (int q = 0; q < CL_WAVES; ++q) { if (q < wave) run += sh[4 + q]; total += sh[4 + q]; }
    for (u32 q = s0; q < s1; ++q) { w.pre[q] = (u16)run; run += (u32)__popc(w.bm[q]); }
    {                                                         // uniform: every thread sees the same total
      const size_t fixed = workBytes(0, n, 0, bmWords), per = (size_t)w.histWords * 4, need = pad16((size_t)total * 2) + 32;
      const bool fits = total <= 65535u && fixed + need + MIN_HIST_WAVES * per <= a.ldsBudget && !(a.firstCap && total > a.firstCap);   // (firstCap: test knob)
      if (!fits) {                                             // more barcodes present than this class has room for: the next larger class takes the block
        if (tid == 0) a.overflow[atomicAdd(a.overflowCount, 1u)] = lcode;
        SYNC();
        return;
      }
      const size_t wv = (a.ldsBudget - fixed - need) / (per ? per : 1);
      nW = wv < (size_t)CL_WAVES ? (u32)wv : (u32)CL_WAVES;
      const size_t need4 = pad16((size_t)total * 4) + 32;      // 4-byte entries where that costs the list loop no wave
      ft.wide = (a.narrowFirst != 1 && fixed + need4 + (size_t)nW * per <= a.ldsBudget) ? 1u : 0u;
      w.first = (u16 *)((unsigned char *)w.hist + pad16((size_t)nW * per));
      ft.first = w.first;
    }
    for (u32 i = tid; i < nW * w.histWords; i += CL_THREADS) w.hist[i] = 0;
    ft.none = total;                                          // (the 32 spare bytes of `need` hold this slot)
    if (ft.wide) for (u32 q = tid; q <= total; q += CL_THREADS) ((u32 *)w.first)[q] = NONE16;
    else for (u32 q = tid; q < (total + 2) / 2; q += CL_THREADS) ((u32 *)w.first)[q] = 0xFFFFFFFFu;
    SYNC();
  }
  STAMP(0);

  // ---- (a)+(b) in one pass over the lists, in rank order, CL_WAVES * RIF ranks per round:
  //   (a) first[cj] = lowest rank >= 1 sharing barcode cj (hash10x.c:794-799, minShare) by CAS-min,
  //   (b) msBest / msMax / msTot of the same ranks (hash10x.c:801-806) from the list entries still in registers, and from
  //       them, while the list's histogram stands, the rank's pointToMin term (hash10x.c:819-821).
  // first[cj] for a barcode of list i is final once every list <= i has been merged (later lists can only
  // offer larger ranks), so one barrier per round is enough.
  // A wave keeps the first two chunks (128 entries) of each of its lists in registers; longer lists re-read the rest.
  // Pipeline per wave, in rounds: descriptors (scalar loads) three rounds ahead, the first chunk of a list two rounds ahead
  // (the loop is bound by the latency of these short random reads), the second chunk — one list in seven has one at
  // yeast scale, most at config-3 scale — one round ahead.
  // The rounds start at rank 0 (never processed, hash10x.c:789: its slot gets length 0).
#define H10X_LOAD_D(I0, DV) { DV = descLoad(gr, (I0), RIF, n); }
#define H10X_LOAD_A(CJ, DV)                                                                                   \
  { _Pragma("unroll") for (int t = 0; t < RIF; ++t) { const Desc g2 = descLane(DV, t); CJ[t] = (u32)lane < g2.y ? ROWP(g2.x)[lane] : code; } }
#define H10X_LOAD_B(CJ2, DV)                                                                                  \
  { _Pragma("unroll") for (int t = 0; t < RIF; ++t) { const Desc g2 = descLane(DV, t); CJ2[t] = (u32)(WAVE + lane) < g2.y ? ROWP(g2.x)[WAVE + lane] : code; } }
  nW = (u32)__builtin_amdgcn_readfirstlane((int)nW);
  const u32 uwave = (u32)__builtin_amdgcn_readfirstlane(wave);
  const bool listWave = uwave < nW;                          // the other waves only keep the barriers company
  const u32 stepR = nW * RIF;
  u32 cjN[RIF], cj2N[RIF], cjNN[RIF];
  u32 dvCur, dvN, dvNN, dvD;                                 // descriptors of this round's lists and of the next three rounds' (see descLoad)
  u32 sDepth = 0;                                            // entries of the lists this wave has worked through (work counter)
  H10X_LOAD_D(listWave ? uwave * RIF : n, dvN)
  H10X_LOAD_D(listWave ? uwave * RIF + stepR : n, dvNN)
  H10X_LOAD_D(listWave ? uwave * RIF + 2 * stepR : n, dvD)
  H10X_LOAD_A(cjN, dvN)
  H10X_LOAD_B(cj2N, dvN)
  H10X_LOAD_A(cjNN, dvNN)
  for (u32 r0 = 0; r0 < n; r0 += stepR) {
    const u32 i0 = listWave ? r0 + uwave * RIF : n;
    u32 cj[RIF], cj2[RIF], dl[RIF];
    dvCur = dvN; dvN = dvNN; dvNN = dvD;
#pragma unroll
    for (int t = 0; t < RIF; ++t) { cj[t] = cjN[t]; cj2[t] = cj2N[t]; cjN[t] = cjNN[t]; dl[t] = (u32)__builtin_amdgcn_readlane((int)dvCur, 32 + t); }
    if constexpr (FIRST_MODE != 0) H10X_LOAD_B(cj2N, dvN) else
    // (dense placement = small data sets, where most lists fit one chunk; lanes 32 .. 32 + RIF - 1 of the descriptor register hold the lengths: one compare and a ballot say whether any list of the next round is longer than
    // a chunk — at yeast scale one list in seven is, so more than half of the rounds skip the RIF address set-ups; the registers then keep stale values that
    // nobody reads: every use of a second chunk is behind `length > 64`)
    if (__builtin_amdgcn_ballot_w64(lane >= 32 && dvN > (u32)WAVE)) H10X_LOAD_B(cj2N, dvN)
    H10X_LOAD_A(cjNN, dvNN)
    H10X_LOAD_D(listWave ? i0 + 3 * stepR : n, dvD)
#define RS_OF(t) ((u32)__builtin_amdgcn_readlane((int)dvCur, t))   /* list offset of this round's list t: only lists of more than two chunks ask */
#pragma unroll
    for (int t = 0; t < RIF; ++t) {
      const u32 i = i0 + t;
      using FTT = decltype(ft);
      if constexpr (FTT::SELF) { if (cj[t] != code) ft.update(cj[t], i); }          // dense: the barcode number IS the handle (the barcode's own: reads "unseen")
      else cj[t] = cj[t] != code ? ft.update(cj[t], i) : ft.none;    // from here on cj / cj2 hold handles
      if (dl[t] > WAVE) {
        if constexpr (FTT::SELF) { if (cj2[t] != code) ft.update(cj2[t], i); }
        else cj2[t] = cj2[t] != code ? ft.update(cj2[t], i) : ft.none;
        if (dl[t] > 2 * WAVE) { const u32 *row = ROWP(RS_OF(t)); for (u32 j = 2 * WAVE + lane; j < dl[t]; j += WAVE) { const u32 c2 = row[j]; if (c2 != code) ft.update(c2, i); } }
      }
    }
    SYNC_LDS();
    // the first[] values of all the round's lists are read back in one go (one wait for up to 2 RIF LDS reads instead of one per list):
    // from here on cj / cj2 hold first[] values, NONE16 where a lane has no entry
    constexpr u32 NOENTRY = NOHANDLE;
    u32 rootV = 0; u64 resV = 0; bool mine = false;          // results of the round's lists, list t in lane t: one LDS and one HBM store per round (H10X_BATCH)
#pragma unroll
    for (int t = 0; t < RIF; ++t) {
      const u32 i = i0 + t;
      if (dl[t]) {                                           // 0 for rank 0 and past the last rank; a list of the depth range holds at least one barcode
        const u32 d = dl[t];
        u32 best, bcnt, tot, rb, q;
        // (the list's address is only formed where entries beyond the two chunks held in registers are read)
        if (d <= WAVE) row_mode_hist<IN_LDS, 1>(nullptr, cj[t], NOENTRY, d, code, i, ft, w.hist + wave * w.histWords, w.root, thr, best, bcnt, tot, rb, q);
        else if (d <= 2 * WAVE) row_mode_hist<IN_LDS, 2>(nullptr, cj[t], cj2[t], d, code, i, ft, w.hist + wave * w.histWords, w.root, thr, best, bcnt, tot, rb, q);
        else if (d < RCHUNK * WAVE) row_mode_hist<IN_LDS, RCHUNK>(ROWP(RS_OF(t)), cj[t], cj2[t], d, code, i, ft, w.hist + wave * w.histWords, w.root, thr, best, bcnt, tot, rb, q);
        else {
          row_mode_long(ROWP(RS_OF(t)), d, code, i, ft, best, bcnt, tot); rb = NONE16; q = 0;
          if (bcnt >= thr) { rb = ld_shared<IN_LDS>(&w.root[best]); if (rb != NONE16) { u32 t2; q = row_count_value(ROWP(RS_OF(t)), d, code, i, ft, rb, t2); } }
        }
        const bool act = bcnt >= thr;                        // hash10x.c:807
        // rb = NONE16: msBest belongs to this round (or to an earlier list of this wave's round: its root is stored with the round's),
        // settled behind the loop. The quotient is formed by replay_kernel: an fp64 divide per list is 40 instructions in this loop.
        if (lane == 0) { w.root[i] = act ? (u16)rb : (u16)i; res[i] = RES_PACK(act ? best : NONE16, q, tot); }
        sDepth += d;
      }
    }
    (void)rootV; (void)resV; (void)mine;
  }
#undef RS_OF
#undef H10X_LOAD_D
#undef H10X_LOAD_A
#undef H10X_LOAD_B
  STAMP(1);
  SYNC();
  STAMP(2);

  // ---- the ranks left open by the loop: active ranks whose msBest was processed in the same round (its root was not on
  // record yet). Collected, their roots settled by walking down the msBest chain to the first rank with a root on record
  // (an inactive rank is its own root; chains only run downwards), then their lists are gathered once more — RIF lists per
  // wave with the loads in flight together — for minShareCount[root] and msTot. A few per cent of the ranks.
  {
    u16 *todo = (u16 *)w.hist;                               // the histograms are idle from here on
    if (tid == 0) sh[2] = 0;
    SYNC();
      for (u32 i0 = 0; i0 < n; i0 += CL_THREADS) {
        const u32 i = i0 + tid;
        bool need = false;
        if (i >= 1 && i < n) need = ld_shared<IN_LDS>(&w.root[i]) == NONE16;
        const u64 bal = __ballot(need);
        if (bal) {
          u32 base = 0;
          if (lane == 0) base = atomicAdd(&sh[2], (u32)__popcll(bal));
          base = (u32)__shfl((int)base, 0);
          if (need) todo[base + (u32)__popcll(bal & ((1ULL << lane) - 1))] = (u16)i;
        }
      }
    SYNC();
    const u32 nTodo = sh[2];
    for (u32 k = tid; k < nTodo; k += CL_THREADS) {
      const u32 i = ld_shared<IN_LDS>(&todo[k]);
      // msBest of an unsettled rank comes from its result word in HBM (written before the barriers above; read past this CU's L1)
      u32 r = (u32)(__hip_atomic_load(&res[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFFFu);
      for (u32 hop = 0; hop < n && r < n; ++hop) {           // (r < n always on consistent data)
        const u32 rr = ld_shared<IN_LDS>(&w.root[r]);
        if (rr != NONE16) { r = rr; break; }
        r = (u32)(__hip_atomic_load(&res[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFFFu);
      }
      w.root[i] = (u16)r;                                    // (a walker passing through i meanwhile reads NONE16 or r: the same answer either way)
    }
    SYNC();
    STAMP(3);
    const u32 uw = (u32)__builtin_amdgcn_readfirstlane(wave);
    for (u32 k0 = uw * RIF; k0 < nTodo; k0 += CL_WAVES * RIF) {
      u32 ii[RIF], cj[RIF], cjB[RIF], dl[RIF], rsv[RIF], qv[RIF];
#pragma unroll
      for (int t = 0; t < RIF; ++t) {
        const bool on = k0 + t < nTodo;
        ii[t] = (u32)__builtin_amdgcn_readfirstlane((int)(on ? (u32)ld_shared<IN_LDS>(&todo[k0 + t]) : 0u));
        qv[t] = on ? (u32)ld_shared<IN_LDS>(&w.root[ii[t]]) : NONE16;
        const u64 g2 = gr[ii[t]];
        dl[t] = on ? (u32)__builtin_amdgcn_readfirstlane((int)(u32)(g2 >> 32)) : 0u; rsv[t] = (u32)__builtin_amdgcn_readfirstlane((int)(u32)g2);
        const u32 *row = ROWP(rsv[t]);
        cj[t] = (u32)lane < dl[t] ? row[lane] : code; cjB[t] = (u32)(WAVE + lane) < dl[t] ? row[WAVE + lane] : code;   // both chunks up front
      }
#pragma unroll
      for (int t = 0; t < RIF; ++t) {
        if (dl[t] == 0) continue;
        const u32 i = ii[t];
        u32 f = cj[t] != code ? ft.lookup(cj[t]) : NONE16;
        u32 q = (u32)__popcll(__ballot(f == qv[t])), tt = (u32)__popcll(__ballot(f < i));
        if (dl[t] > WAVE) { f = cjB[t] != code ? ft.lookup(cjB[t]) : NONE16; q += (u32)__popcll(__ballot(f == qv[t])); tt += (u32)__popcll(__ballot(f < i)); }
        if (dl[t] > 2 * WAVE) { const u32 *row = ROWP(rsv[t]); for (u32 b0 = 2 * WAVE; b0 < dl[t]; b0 += WAVE) { f = NONE16; if (b0 + lane < dl[t]) { const u32 c2 = row[b0 + lane]; if (c2 != code) f = ft.lookup(c2); } q += (u32)__popcll(__ballot(f == qv[t])); tt += (u32)__popcll(__ballot(f < i)); } }
        if (lane == 0) res[i] = RES_PACK(__hip_atomic_load(&res[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFFFu, q, tt);
      }
    }
  }
  STAMP(4);

  // ---- out: nothing is left to write — every rank's result word is in place; replay_kernel (labels, cluster count, the > 255
  // clusters cut, the quotients), point_sum_kernel and read_merge_kernel finish the block.
  // work counters: kept per lane over the barcodes of the workgroup, posted once when the kernel ends (19 atomics per
  // barcode on four shared words would queue up in L2 behind those of every other workgroup)
  if (lane == 0) acc.depth += sDepth;
  if (tid == 0) { acc.s[3] += (u32)(gr[0] >> 32); acc.s[0] += n; acc.s[1] += a.blocks[lcode].nHash; acc.s[2] += 1; }
  if (!FIRST_LDS) {                                          // leave first[] clean for the next barcode of this workgroup
    SYNC();
    const u32 uw = (u32)__builtin_amdgcn_readfirstlane(wave);
    for (u32 i = 1 + uw; i < n; i += CL_WAVES) {
      const u64 g2 = gr[i]; const u32 d = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(g2 >> 32)); const u32 *row = ROWP((u32)__builtin_amdgcn_readfirstlane((int)(u32)g2));
      for (u32 j = lane; j < d; j += WAVE) w.first[row[j]] = NONE16;
    }
  }
  SYNC();
  STAMP(5);
}


// One block in the translated placement (SlotTable / FirstSlots above): pass A = every list entry -> 16-bit handle on the
// workgroup's HBM slot, first[] final in the table; pass B = the list loop on handles (msBest / msMax / msTot and the
// pointToMin counts per rank, hash10x.c:801-821, exactly as cluster_one_block's), then the ranks the loop left open.
template <int CL_THREADS, int KLASS>
__device__ __forceinline__ void cluster_one_block_tr(const ClusterArgs &a, u32 code, unsigned char *region, u32 *sh, WorkAcc &acc) {
  constexpr bool IN_LDS = true;
  constexpr int CL_WAVES = CL_THREADS / WAVE;
  constexpr int RIF = KLASS == 2 ? 8 : ROWS_IN_FLIGHT;         // (the whole-CU class has 128 registers per lane and only four waves per SIMD to hide latency behind: twice the lists in flight)
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
  code = (u32)__builtin_amdgcn_readfirstlane((int)code);
  const u32 n = (u32)__builtin_amdgcn_readfirstlane((int)a.nGood[code]);
  if (n == 0) return;                                        // hash10x.c:780: block left untouched
  const u64 o = a.blockOff[code];
  const u32 lcode = code; code = a.segs.globalOf(lcode);     // from here on `code` is the global barcode number (what the lists hold)
  u32 S, S2, nW; bool compact;
  translatedShape(n, CL_WAVES, a.ldsBudget, a.hashMinSlots, a.entries[lcode], a.nBlocksFirst, a.firstCap, S, S2, nW, compact);
  if (!S || nW < MIN_HIST_WAVES) {                           // cannot hold this barcode at all: hand it on
    if (tid == 0) a.overflow[atomicAdd(a.overflowCount, 1u)] = lcode;
    return;
  }
  const u32 closeAt = translatedCloseAt(S, S2);
  // the table's fill count, read as LDS: through the generic pointer the volatile read is a FLAT load followed by s_waitcnt vmcnt(0) — every round of pass A then waited
  // for the list loads it had just issued for the NEXT round (seen in the ISA, round 5)
  auto shFill = [&]() { typedef __attribute__((address_space(3))) u32 lds_u32; return *(volatile lds_u32 *)(lds_u32 *)&sh[1]; };
  const u32 rsh = a.rowShift;
#define ROWP(rs) (a.rows + ((size_t)(rs) << rsh))
  const u64 *const gr = a.goodRow + o;
  u64 *const res = a.res + o;
  u32 thr = (u32)a.threshold;
  asm volatile("" : "+s"(thr));                              // (a value of its own from here on: left as a kernel argument, the compiler reloads it for every list — two v_readlane for the
                                                             // spilled argument pointer, an s_load and a wait that also covers the list's LDS traffic — rather than keep a register for it)
  u16 *const hs = a.handles + (size_t)blockIdx.x * a.handleStride;   // this workgroup's handles: list i at hs + i * hst
  const u32 hst = a.hStride, lgH = 31 - (u32)__clz((int)hst);
  const u32 uwave = (u32)__builtin_amdgcn_readfirstlane(wave);
  if (a.phase && threadIdx.x == 0) acc.s[4] = wall_clock64();

  // ---- pass A
  u32 *const tab = (u32 *)region;
  for (u32 i = tid; i <= S; i += CL_THREADS) tab[i] = 0xFFFFFFFFu;   // (+ the word of handle `none`)
  if (tid == 0) { sh[0] = 0; sh[1] = 0; sh[2] = 0; res[0] = RES_PACK(NONE16, 0, 0); }   // [0] entries left for the second table, [1] barcodes in the first, [2] a table overflowed
  SYNC();
  STAMP(0);
  // entries whose search goes beyond the home bucket (1.5 % on the config-3 sets) are parked — barcode | position of the handle << 22;
  // hst is a power of two: the rank is the position's top bits — in a queue of the wave's own behind the handles on the HBM slot,
  // and searched for with search() 64 lanes at a time when the pass is over (or the queue full). What search() cannot settle —
  // a barcode that is new when the table is closed — stays in the queue, for the second table.
  unsigned long long *const queue = (unsigned long long *)(hs + a.handleStride - (size_t)CL_WAVES * TR_QUEUE * 4) + (size_t)uwave * TR_QUEUE;
  u32 qn = 0;                                                // (uniform)
  SlotTable st; st.shape(tab, S, a.hashBits, &sh[2], 0); st.inLds();
  // the queue against table `t`: entries [qk, qn) are searched for, 64 at a time (whole batches only unless `all`); settled entries get
  // their handle, the others — barcodes that are new when the table no longer takes any — are kept at the queue's front, [0, qk)
  u32 qk = 0;
  auto settle = [&](const SlotTable &t, bool mayInsert, u32 fillLimit, bool all) {
    u32 k0 = qk;
    for (; all ? k0 < qn : k0 + WAVE <= qn; k0 += WAVE) {
      const bool on = k0 + lane < qn;
      const unsigned long long e = on ? queue[k0 + lane] : 0ull;
      const bool insert = mayInsert && shFill() < fillLimit;   // (uniform: one LDS word)
      u32 h = 0; bool ins = false;
      if (on) { const u32 dest = (u32)(e >> 22); h = t.search((u32)e & 0x3FFFFFu, dest >> lgH, insert, ins); if (h != SlotTable::NOTFOUND) hs[dest] = (u16)h; }
      const u64 balI = __ballot(ins);
      if (balI && lane == 0) atomicAdd(&sh[1], (u32)__popcll(balI));
      const u64 balU = __ballot(on && h == SlotTable::NOTFOUND);
      if (balU) {
        if (on && h == SlotTable::NOTFOUND) queue[qk + (u32)__popcll(balU & ((1ULL << lane) - 1))] = e;   // (qk <= k0: behind what has been read)
        qk += (u32)__popcll(balU);
      }
    }
    // what is left of [k0, qn) (less than a batch) moves down behind the kept ones
    const u32 rest = k0 < qn ? qn - k0 : 0u;
    if (rest && k0 != qk) { const unsigned long long e = (u32)lane < rest ? queue[k0 + lane] : 0ull; if ((u32)lane < rest) queue[qk + lane] = e; }
    qn = qk + rest;
  };
  {
    bool insert = true;                                      // (uniform) the table takes new barcodes: looked up once per round
    u32 myIns = 0;                                           // barcodes this wave put into the table since it last said so
    const u32 laneU = (u32)lane;
    // one chunk: up to 64 entries of list i -> their handles at hl[0..] (position dest0 on the slot): `none` for the barcode itself, a guess where the
    // search is not over (the queue brings the real one)
    auto place = [&](u32 cj, u32 i, u32 drem /* entries of the list from this chunk on */, u16 *hl, u32 dest0) {
      const bool valid = laneU < drem && cj != code;
      u32 slot; bool ins;
      const bool done = st.probeHome(cj, i, valid, insert, slot, ins);
      hl[laneU] = (u16)(valid ? slot : S);                    // all 64 lanes (a list's row is a power of two >= 64 wide): past the list's end the handle of `none`, which reads unseen — pass B
                                                             // then loads and counts a chunk without looking at the list's length, and no exec mask is formed here
      myIns += (u32)__popcll(__ballot(ins));
      const u64 bal = __ballot(!done);
      if (bal) {
        if (!done) queue[qn + (u32)__popcll(bal & ((1ULL << lane) - 1))] = (unsigned long long)cj | ((unsigned long long)(dest0 + laneU) << 22);
        qn += (u32)__popcll(bal);
      }
    };
    // both chunks of a wave's RIF lists are requested a round ahead of their probes; descriptors two rounds ahead. Rounds start at
    // rank 0, which is never processed (hash10x.c:789): its slot gets length 0.
    constexpr u32 stepA = CL_WAVES * RIF;
    u32 dv, dvN = descFix(descLoadU(gr, uwave * RIF, n), uwave * RIF, RIF, n), dvNN = descLoadU(gr, uwave * RIF + stepA, n);   // (dvNN: raw)
    u32 c0N[RIF], c1N[RIF];
    // (no lane skips a load and none looks at the list's length: rows[] has ROWS_PAD entries of slack, what lies behind a list's end is told
    // apart when the chunk is used — see descLoadU)
    // (a list without a second chunk asks for its first one again: a load that is always issued, yet no bytes from HBM that nobody looks at)
#define TR_LOAD_C(DV) { _Pragma("unroll") for (int t = 0; t < RIF; ++t) { const u32 *row = ROWP((u32)__builtin_amdgcn_readlane((int)DV, t)); c0N[t] = row[laneU]; \
      c1N[t] = row[((u32)__builtin_amdgcn_readlane((int)DV, 32 + t) > (u32)WAVE ? (u32)WAVE : 0u) + laneU]; } }
    TR_LOAD_C(dvN)
    for (u32 i0 = uwave * RIF;; i0 += stepA) {
      const bool last = i0 >= n;                             // one more turn behind the last round: the queue's remainder (ONE copy of the search loop in the code)
      u32 lt = RIF;                                          // first list of the round with more than two chunks (RIF: none)
      dv = dvN;
      if (!last) {
        u32 c0[RIF], c1[RIF];
        dvN = descFix(dvNN, i0 + stepA, RIF, n);
#pragma unroll
        for (int t = 0; t < RIF; ++t) { c0[t] = c0N[t]; c1[t] = c1N[t]; }
        dvNN = descLoadU(gr, i0 + 2 * stepA, n);             // (in front of the list loads: the next round's first wait is for this one)
        TR_LOAD_C(dvN)
        if (S2) {                                            // a table that may fill up: say what this wave has put in, see whether it still takes barcodes
          if (myIns) { if (lane == 0) atomicAdd(&sh[1], myIns); myIns = 0; }
          insert = (u32)__builtin_amdgcn_readfirstlane((int)shFill()) < closeAt;
        }
#pragma unroll
        for (int t = RIF - 1; t >= 0; --t) if ((u32)__builtin_amdgcn_readlane((int)dv, 32 + t) > 2u * WAVE) lt = (u32)t;
#pragma unroll
        for (int t = 0; t < RIF; ++t) {
          const u32 d = (u32)__builtin_amdgcn_readlane((int)dv, 32 + t);
          if (d) {
            const u32 i = i0 + t, dest0 = i << lgH; u16 *const hl = hs + dest0;
            place(c0[t], i, d, hl, dest0);
            if (d > WAVE) place(c1[t], i, d - WAVE, hl + WAVE, dest0 + WAVE);
          }
        }
      }
      for (u32 lj = 2 * WAVE;;) {                            // further chunks (depth ranges beyond 128), list after list; the queue is emptied in between when it fills up
        while (lt < RIF) {
          const u32 d = (u32)__builtin_amdgcn_readlane((int)dv, 32 + (int)lt);
          if (d <= 2u * WAVE || lj >= d) { ++lt; lj = 2 * WAVE; continue; }
          const u32 i = i0 + lt; const u32 *row = ROWP((u32)__builtin_amdgcn_readlane((int)dv, (int)lt));
          place(row[lj + laneU], i, d - lj, hs + (i << lgH) + lj, (i << lgH) + lj);
          lj += WAVE;
          if (qn - qk >= WAVE) break;
        }
        if (last || qn - qk >= WAVE) {                       // (the next round's loads are under way meanwhile)
          if (myIns) { if (lane == 0) atomicAdd(&sh[1], myIns); myIns = 0; }
          settle(st, true, closeAt, last);
          if (qn > TR_QUEUE - 2 * RIF * WAVE) { sh[2] = 1; qn = qk = 0; }   // more entries wait for the second table than the queue holds: the block is handed on
        }
        if (lt >= RIF) break;
      }
      if (last) break;
    }
#undef TR_LOAD_C
    STAMP(1);
    if (qn && lane == 0) sh[0] = 1;
  }
  __syncthreads();                                           // (the handles other waves wrote are plain stores of this CU, read back through its own L1: workgroup scope is
                                                             // enough — an agent-scope acquire here drops the XCD's L2 for everybody, twice per block)
  STAMP(2);
  const bool spill = sh[0] != 0 && !sh[2];                   // (uniform: read after the barrier)
  if (spill) {                                               // the first table is final now: what is still queued is looked up once more (another wave may have put the barcode in
    qk = 0; settle(st, false, 0, true);                      // while this one found the table closed); what is still not there goes into the second table
    SYNC();
  }
  if (sh[2]) {
    if (tid == 0) a.overflow[atomicAdd(a.overflowCount, 1u)] = lcode;
    SYNC();
    return;
  }
  // ---- the table gives its LDS back: the ranks, 2 bytes per slot, compacted in place (a batch of words is read, then — behind a
  // barrier — written as halves: the halves land below every word still to be read)
  FirstSlots ft; ft.none = S;
  size_t firstBytes;
  auto compactRanks = [&](const u32 *src, u16 *dst, u32 count) {   // dst below src (or equal)
    for (u32 base = 0; base < count; base += 4 * CL_THREADS) {
      u32 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const u32 idx = base + (u32)k * CL_THREADS + tid; v[k] = idx < count ? src[idx] : 0u; }
      SYNC();
#pragma unroll
      for (int k = 0; k < 4; ++k) { const u32 idx = base + (u32)k * CL_THREADS + tid; if (idx < count) dst[idx] = (u16)(v[k] >> 16); }   // (an empty word and the word of `none` read 0xFFFF = unseen)
    }
  };
  if (compact || spill) {
    compactRanks(tab, (u16 *)region, S + 1);
    ft.base = region; ft.sh = 1; firstBytes = pad16(2 * ((size_t)S + 8));
  } else { ft.base = region + 2; ft.sh = 2; firstBytes = pad16(4 * ((size_t)S + 4)); }
  if (spill) {
    // ---- the second table, in what the first has freed: the barcodes that turned up after the first was closed
    u32 *const tab2 = (u32 *)(region + firstBytes);
    SYNC();
    for (u32 i = tid; i < S2; i += CL_THREADS) tab2[i] = 0xFFFFFFFFu;
    if (tid == 0) sh[1] = 0;
    SYNC();
    SlotTable st2; st2.shape(tab2, S2, a.hashBits, &sh[2], S + 1);
    qk = 0; settle(st2, true, 0xFFFFFFFFu, true);
    if (qn) sh[2] = 1;                                       // (cannot happen: search() inserts or reports the overflow)
    SYNC();
    if (sh[2]) {
      if (tid == 0) a.overflow[atomicAdd(a.overflowCount, 1u)] = lcode;
      SYNC();
      return;
    }
    compactRanks(tab2, (u16 *)region + S + 1, S2);
    firstBytes = pad16(2 * ((size_t)S + S2 + 16));
    nW = translatedWaves2(n, CL_WAVES, a.ldsBudget, S, S2);
  }
  u16 *const root = (u16 *)(region + firstBytes);
  u32 *const hist = (u32 *)(region + firstBytes + pad16((size_t)n * 2));
  const u32 histWords = (n + 3) / 4;
  SYNC();
  for (u32 i = tid; i < nW * histWords; i += CL_THREADS) hist[i] = 0;
  for (u32 i = tid; i < n; i += CL_THREADS) root[i] = i ? NONE16 : (u16)0;      // rank 0 is never processed (hash10x.c:789): inactive, its own root
  SYNC();
  STAMP(3);

  // ---- pass B: the list loop on handles. No barrier per round: first[] is final; root[] of an earlier rank is either on record or
  // reads "open" (then the rank is settled behind the loop), whichever wave gets there first.
  nW = (u32)__builtin_amdgcn_readfirstlane((int)nW);
  const bool listWave = uwave < nW;
  const u32 stepR = nW * RIF;
  u32 sDepth = 0;
  {
    const u32 laneU = (u32)lane;
    u32 hN[RIF], h2N[RIF], hNN[RIF];
    u32 dvCur, dvN, dvNN, dvD;
    // the handles of a list's two chunks: all 64 lanes, no test and no clamp (pass A wrote every lane of a chunk the list has: `none` past the list's end)
#define TR_LOAD_A(H, I0, DV) { _Pragma("unroll") for (int t = 0; t < RIF; ++t) { const u32 ic = (I0) + t < n ? (I0) + t : n - 1; H[t] = (hs + ((size_t)ic << lgH))[laneU]; } }
#define TR_LOAD_B(H2, I0, DV) { _Pragma("unroll") for (int t = 0; t < RIF; ++t) { const u32 ic = (I0) + t < n ? (I0) + t : n - 1; \
      H2[t] = (hs + ((size_t)ic << lgH))[((u32)__builtin_amdgcn_readlane((int)DV, 32 + t) > (u32)WAVE ? (u32)WAVE : 0u) + laneU]; } }   /* (no second chunk: the first again, from the caches) */
    const u32 iw = listWave ? uwave * RIF : n;
    dvN = descFix(descLoadU(gr, iw, n), iw, RIF, n); dvNN = descLoadU(gr, listWave ? iw + stepR : n, n); dvD = descLoadU(gr, listWave ? iw + 2 * stepR : n, n);   // (dvNN, dvD: raw)
    TR_LOAD_A(hN, iw, dvN)
    TR_LOAD_B(h2N, iw, dvN)
    TR_LOAD_A(hNN, listWave ? iw + stepR : n, dvNN)
    for (u32 r0 = 0; r0 < n; r0 += stepR) {
      const u32 i0 = listWave ? r0 + uwave * RIF : n;
      u32 h[RIF], h2[RIF], dl[RIF];
      dvCur = dvN; dvN = descFix(dvNN, listWave ? i0 + stepR : n, RIF, n); dvNN = dvD;
#pragma unroll
      for (int t = 0; t < RIF; ++t) { h[t] = hN[t]; h2[t] = h2N[t]; hN[t] = hNN[t]; dl[t] = (u32)__builtin_amdgcn_readlane((int)dvCur, 32 + t); }
      dvD = descLoadU(gr, listWave ? i0 + 3 * stepR : n, n);
      TR_LOAD_B(h2N, listWave ? i0 + stepR : n, dvN)
      TR_LOAD_A(hNN, listWave ? i0 + 2 * stepR : n, dvNN)
      u32 rootV = 0; u64 resV = 0;                           // the results of the round's lists, list t in lane t: one LDS and one HBM store per round
#pragma unroll
      for (int t = 0; t < RIF; ++t) {
        const u32 i = i0 + t;
        if (dl[t]) {                                         // 0 for rank 0 and past the last rank
          const u32 d = dl[t];
          u32 best, bcnt, tot, rb, q;
          const u16 *const hrow = hs + ((size_t)i << lgH);
          if (d <= WAVE) row_mode_hist<IN_LDS, 1>(nullptr, h[t], NOHANDLE, d, code, i, ft, hist + wave * histWords, root, thr, best, bcnt, tot, rb, q);
          else if (d <= 2 * WAVE) row_mode_hist<IN_LDS, 2>(nullptr, h[t], h2[t], d, code, i, ft, hist + wave * histWords, root, thr, best, bcnt, tot, rb, q);
          else if (d < RCHUNK * WAVE) row_mode_hist<IN_LDS, RCHUNK>(hrow, h[t], h2[t], d, code, i, ft, hist + wave * histWords, root, thr, best, bcnt, tot, rb, q);
          else {
            row_mode_long(hrow, d, code, i, ft, best, bcnt, tot); rb = NONE16; q = 0;
            if (bcnt >= thr) { rb = root[best]; if (rb != NONE16) { u32 t2; q = row_count_value(hrow, d, code, i, ft, rb, t2); } }
          }
          const bool act = bcnt >= thr;                      // hash10x.c:807
          if (laneU == (u32)t) { rootV = act ? rb : i; resV = RES_PACK(act ? best : NONE16, q, tot); }
          sDepth += d;
        }
      }
      { const u32 i = i0 + laneU; if (laneU < (u32)RIF && i >= 1 && i < n) { root[i] = (u16)rootV; res[i] = resV; } }
    }
#undef TR_LOAD_A
#undef TR_LOAD_B
  }
  STAMP(4);
  SYNC();
  // ---- the ranks left open by the loop (their msBest's root was not on record yet): roots by walking down the msBest chain, then
  // their lists' handles once more for minShareCount[root] and msTot — as in cluster_one_block
  {
    u16 *todo = (u16 *)hist;                                 // the histograms are idle from here on
    if (tid == 0) sh[2] = 0;
    SYNC();
    for (u32 i0 = 0; i0 < n; i0 += CL_THREADS) {
      const u32 i = i0 + tid;
      const bool need = i >= 1 && i < n && root[i] == NONE16;
      const u64 bal = __ballot(need);
      if (bal) {
        u32 base = 0;
        if (lane == 0) base = atomicAdd(&sh[2], (u32)__popcll(bal));
        base = (u32)__shfl((int)base, 0);
        if (need) todo[base + (u32)__popcll(bal & ((1ULL << lane) - 1))] = (u16)i;
      }
    }
    SYNC();
    const u32 nTodo = sh[2];
    if (a.phase && tid == 0) { atomicAdd((u64 *)&a.phase[6], (u64)nTodo); atomicAdd((u64 *)&a.phase[7], (u64)n); }   // diagnostic: ranks settled behind the loop / ranks
    for (u32 k = tid; k < nTodo; k += CL_THREADS) {
      const u32 i = todo[k];
      u32 r = (u32)(__hip_atomic_load(&res[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFFFu);   // msBest: from the rank's result word (read past this CU's L1)
      for (u32 hop = 0; hop < n && r < n; ++hop) {
        const u32 rr = *(volatile u16 *)&root[r];
        if (rr != NONE16) { r = rr; break; }
        r = (u32)(__hip_atomic_load(&res[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFFFu);
      }
      root[i] = (u16)r;                                      // (a walker passing through i meanwhile reads NONE16 or r: the same answer either way)
    }
    SYNC();
    for (u32 k0 = uwave * RIF; k0 < nTodo; k0 += CL_WAVES * RIF) {
      u32 ii[RIF], hA[RIF], hB[RIF], dl[RIF], qv[RIF];
#pragma unroll
      for (int t = 0; t < RIF; ++t) {
        const bool on = k0 + t < nTodo;
        ii[t] = (u32)__builtin_amdgcn_readfirstlane((int)(on ? (u32)todo[k0 + t] : 0u));
        qv[t] = on ? (u32)root[ii[t]] : NONE16;
        dl[t] = on ? (u32)__builtin_amdgcn_readfirstlane((int)(u32)(gr[ii[t]] >> 32)) : 0u;
        const u16 *const hrow = hs + (size_t)ii[t] * hst;
        hA[t] = (u32)lane < dl[t] ? (u32)hrow[lane] : S; hB[t] = (u32)(WAVE + lane) < dl[t] ? (u32)hrow[WAVE + lane] : S;   // (only the list's own entries have a handle)
      }
#pragma unroll
      for (int t = 0; t < RIF; ++t) {
        if (dl[t] == 0) continue;
        const u32 i = ii[t];
        u32 f = ft.peek(hA[t]);
        u32 q = (u32)__popcll(__ballot(f == qv[t])), tt = (u32)__popcll(__ballot(f < i));
        if (dl[t] > WAVE) { f = ft.peek(hB[t]); q += (u32)__popcll(__ballot(f == qv[t])); tt += (u32)__popcll(__ballot(f < i)); }
        if (dl[t] > 2 * WAVE) { const u16 *const hrow = hs + (size_t)i * hst; for (u32 b0 = 2 * WAVE; b0 < dl[t]; b0 += WAVE) { f = b0 + lane < dl[t] ? ft.peek(hrow[b0 + lane]) : (u32)NONE16; q += (u32)__popcll(__ballot(f == qv[t])); tt += (u32)__popcll(__ballot(f < i)); } }
        if (lane == 0) res[i] = RES_PACK(__hip_atomic_load(&res[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFFFu, q, tt);
      }
    }
  }
  if (lane == 0) acc.depth += sDepth;
  if (tid == 0) { acc.s[3] += (u32)(gr[0] >> 32); acc.s[0] += n; acc.s[1] += a.blocks[lcode].nHash; acc.s[2] += 1; }
  SYNC();
  STAMP(5);
}


// ---- translated placement, PACKED (round 5) -------------------------------------------------------------------------------------
// The ranks of a block ascend in list length (the good list is ordered by depth: hash10x.c:726-730,758), so its lists fall into
// contiguous rank classes: Q up to 16 entries, H up to 32, F up to 64, D up to 128, X beyond. A wave instruction serves FOUR lists of
// class Q (one per 16-lane DPP row), TWO of class H, one of class F / D; every per-list quantity — rank, length, msBest / msMax / msTot
// (hash10x.c:801-806), the founding rank and the two counts of the pointToMin term (hash10x.c:819-821) — lives in the lanes of the
// list's segment: segmented DPP maximum, per-segment popcounts of ballots, per-segment histogram regions. Nothing per list goes through
// the scalar unit, which is what bounded the per-list form (one list per wave instruction: 130 scalar + 126 vector instructions per list,
// a third of the lanes in use where the depth range is 6-45), and the units of a round carry no branches, so their LDS round trips overlap.
// Handles lie packed in the same order on the workgroup's HBM slot: class Q at 16 u16 per rank, H 32, F 64, D 128, X the slot's stride;
// pass B loads a unit's 64 handles with one coalesced load and needs no list descriptor at all (pass A writes `none` where a lane has no entry).
// Class T (round 5, last): lists of 65 .. 96 entries — four in five of all lists where the depth range is 30 - 100 and the homozygous peak sits in the eighties
// (BASELINE configs[2]: scratch/r5_c3_lengths.py) — run TWO to a unit of THREE chunks: a whole chunk for the first 64 entries of each and one chunk shared by their tails
// (lanes 0-31 the first list's entries 64 .. 95, lanes 32-63 the second's). As class D they took two chunks each, the second one a quarter full: 22 % fewer chunks
// on that set, lanes in use 66 % -> 85 %. Handles: 192 u16 per unit (first list | second list | tails); both lists count in the wave's one histogram region, one after the other.
struct TpShape { u32 m16, m32, m64, m96, m128, posH, posF, posT, posD, posX; };
__device__ __forceinline__ u32 tpPos(const TpShape &s, u32 i, u32 hst) {        // where the handles of rank i start
  if (i < s.m16) return 16u * i;
  if (i < s.m32) return s.posH + 32u * (i - s.m16);
  if (i < s.m64) return s.posF + 64u * (i - s.m32);
  if (i < s.m96) { const u32 k = i - s.m64; return s.posT + 192u * (k >> 1) + 64u * (k & 1u); }
  if (i < s.m128) return s.posD + 128u * (i - s.m96);
  return s.posX + hst * (i - s.m128);
}
__device__ __forceinline__ u32 tpPos64(const TpShape &s, u32 i, u32 hst) {      // where the handles of rank i's entries 64 .. start (class T: in the unit's third chunk)
  if (i >= s.m64 && i < s.m96) { const u32 k = i - s.m64; return s.posT + 192u * (k >> 1) + 128u + 32u * (k & 1u); }
  return tpPos(s, i, hst) + 64u;
}
// maximum over the lanes of a segment (16, 32 or 64 lanes), handed to every lane of the segment: DPP prefix maxima inside the rows, row broadcasts
// across them (as far as the segment reaches), then one LDS-crossbar permute from the segment's last lane
template <int SEGW> __device__ __forceinline__ u32 seg_max_u32(u32 v, u32 laneU) {
#define H10X_DPP_MAX(ctrl, rowmask) { const u32 o = (u32)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rowmask, 0xf, false); v = o > v ? o : v; }
  H10X_DPP_MAX(0x111, 0xf) H10X_DPP_MAX(0x112, 0xf) H10X_DPP_MAX(0x114, 0xf) H10X_DPP_MAX(0x118, 0xf)
  if (SEGW >= 32) H10X_DPP_MAX(0x142, 0xa)
  if (SEGW >= 64) H10X_DPP_MAX(0x143, 0xc)
#undef H10X_DPP_MAX
  if (SEGW == 64) return (u32)__builtin_amdgcn_readlane((int)v, 63);
  return (u32)__builtin_amdgcn_ds_bpermute((int)((laneU | (u32)(SEGW - 1)) << 2), (int)v);
}
// bits of a ballot that belong to the lane's segment, counted
template <int SEGW> __device__ __forceinline__ u32 seg_popc(u64 bal, u32 laneU) {
  if (SEGW == 64) return (u32)__popcll(bal);
  if (SEGW == 32) { const u32 lo = (u32)__popc((u32)bal), hi = (u32)__popc((u32)(bal >> 32)); return laneU < 32 ? lo : hi; }
  return (u32)__popc((u32)(bal >> (laneU & 48u)) & 0xFFFFu);
}

template <int CL_THREADS, int KLASS>
__device__ __forceinline__ void cluster_one_block_tp(const ClusterArgs &a, u32 code, unsigned char *region, u32 *sh, WorkAcc &acc) {
  constexpr bool IN_LDS = true;
  constexpr int CL_WAVES = CL_THREADS / WAVE;
#ifndef H10X_TP_RIF0
#define H10X_TP_RIF0 3                                        /* (2 until the end of round 5: tuned on the 1/10 sets, where 3 lost 1 %; on the full 3 Gb set 3 is 5.6 % ahead: cluster 301.4 -> 284.5 ms, 4 the same, 1 351) */
#define H10X_TP_RIF0F 3
#define H10X_TP_RIF2 4
#endif
  constexpr int RIF_Q = KLASS == 2 ? H10X_TP_RIF2 : H10X_TP_RIF0, RIF_F = KLASS == 2 ? H10X_TP_RIF2 : H10X_TP_RIF0F;   // units a wave keeps in flight (classes Q, H / F, D)
  constexpr int RIF = RIF_F > RIF_Q ? RIF_F : RIF_Q;          // (the queue's margin)
  constexpr int RIF_T = KLASS == 2 ? 3 : 2;                    // class T: units of three chunks (six / nine chunks of a wave in flight, as in class D)
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
  const u32 laneU = (u32)lane;
  code = (u32)__builtin_amdgcn_readfirstlane((int)code);
  const u32 n = (u32)__builtin_amdgcn_readfirstlane((int)a.nGood[code]);
  if (n == 0) return;                                        // hash10x.c:780: block left untouched
  u64 o = a.blockOff[code];
  o = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(o >> 32)) << 32) | (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)o);   // (uniform: say so — the pointers formed from it are then scalar)
  const u32 lcode = code; code = a.segs.globalOf(lcode);     // from here on `code` is the global barcode number (what the lists hold)
  u32 S, S2, nW; bool compactUnused;
  translatedShape(n, CL_WAVES, a.ldsBudget, a.hashMinSlots, a.entries[lcode], a.nBlocksFirst, a.firstCap, S, S2, nW, compactUnused);
  if (!S || nW < MIN_HIST_WAVES) {                           // cannot hold this barcode at all: hand it on
    if (tid == 0) a.overflow[atomicAdd(a.overflowCount, 1u)] = lcode;
    return;
  }
  const u32 closeAt = translatedCloseAt(S, S2);
  // the table's fill count, read as LDS: through the generic pointer the volatile read is a FLAT load followed by s_waitcnt vmcnt(0) — every round of pass A then waited
  // for the list loads it had just issued for the NEXT round (seen in the ISA, round 5)
  auto shFill = [&]() { typedef __attribute__((address_space(3))) u32 lds_u32; return *(volatile lds_u32 *)(lds_u32 *)&sh[1]; };
  const u32 rsh = a.rowShift;
#define ROWP(rs) (a.rows + ((size_t)(rs) << rsh))
  const u64 *const gr = a.goodRow + o;
  u32 thr = (u32)a.threshold;
  asm volatile("" : "+s"(thr));
  // (the result words go to a.res + o, formed where a result is written: as a kernel-long value of a NON-scalar o the pointer once sat in a spilled VGPR pair, and its reload in
  //  pass B's writer branch brought an s_waitcnt vmcnt(0) with it — a wait for the handle loads just issued for the next round)
  if (tid == 0) a.res[o] = RES_PACK(NONE16, 0, 0);
  u16 *const hs = a.handles + (size_t)blockIdx.x * a.handleStride;
  const u32 hst = a.hStride;
  const u32 uwave = (u32)__builtin_amdgcn_readfirstlane(wave);
  if (a.phase && threadIdx.x == 0) acc.s[4] = wall_clock64();

  // ---- the table of pass A is cleared, and meanwhile the classes are counted: ranks ascend in list length, so `lists of at most L entries` is a rank
  u32 *const tab = (u32 *)region;
  {                                                          // (16 bytes per store: S is a multiple of 4, the region 16-byte aligned; + the word of handle `none`)
    const uint4 ones = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    for (u32 i = tid; i < S / 4; i += CL_THREADS) ((uint4 *)tab)[i] = ones;
    if (tid == 0) tab[S] = 0xFFFFFFFFu;
  }
  if (tid < 8) sh[tid] = 0;                                  // [0] entries left for the second table, [1] barcodes in the first, [2] a table overflowed, [3..7] class counts (96, 16, 32, 64, 128)
  SYNC();
  {
    u32 c16 = 0, c32 = 0, c64 = 0, c96 = 0, c128 = 0;        // (uniform)
    for (u32 i0 = uwave * WAVE; i0 < n; i0 += CL_THREADS) {
      const u32 i = i0 + laneU;
      const u32 d = i < n ? (u32)(gr[i] >> 32) : 0xFFFFFFFFu;
      c16 += (u32)__popcll(__ballot(d <= 16u)); c32 += (u32)__popcll(__ballot(d <= 32u)); c64 += (u32)__popcll(__ballot(d <= 64u)); c96 += (u32)__popcll(__ballot(d <= 96u)); c128 += (u32)__popcll(__ballot(d <= 128u));
    }
    if (lane == 0 && c128) { atomicAdd(&sh[4], c16); atomicAdd(&sh[5], c32); atomicAdd(&sh[6], c64); atomicAdd(&sh[3], c96); atomicAdd(&sh[7], c128); }
  }
  SYNC();
  STAMP(0);
  // ---- how the lists are packed: four / two to a wave instruction where the histogram regions of that many segments fit beside first[] and root[]
  // (pass B: first[] at 2 bytes per slot, root[] 2 bytes per rank, one region per segment and wave). Decided here, before pass A: the handles are laid out by it.
  TpShape tp;
  u32 histBytes;                                             // per wave in pass B
  {
    const u32 c16 = sh[4], c32 = sh[5];
    tp.m64 = sh[6]; tp.m128 = sh[7]; tp.m96 = a.tpClassT ? sh[3] : tp.m64;
    if (tp.m96 < tp.m64) tp.m96 = tp.m64;                    // (cannot happen: counts of <= 64 and <= 96)
    if (tp.m96 > tp.m128) tp.m96 = tp.m128;
    auto p4 = [](u32 x) { return (x + 3u) & ~3u; };
    const long fixedB = (long)pad16((size_t)n * 2) + 64;
    const long space = (long)a.ldsBudget - (long)pad16(2 * ((size_t)S + 8)) - fixedB;
    const long spaceSpill = S2 ? (long)a.ldsBudget - (long)pad16(2 * ((size_t)S + S2 + 16)) - fixedB : space;
    const u32 nD = tp.m128 - tp.m96, nT = tp.m96 - tp.m64;
    u32 bestW = 0, bestU = 1; tp.m16 = 0; tp.m32 = 0; histBytes = p4(n);
    for (int opt = 0; opt < 3; ++opt) {
      const u32 q = opt == 0 ? c16 : 0u, h = opt <= 1 ? c32 : 0u;
      u32 R = p4(n); if (4u * p4(q) > R) R = 4u * p4(q); if (2u * p4(h) > R) R = 2u * p4(h);
      if ((long)R > spaceSpill) continue;
      u32 w = (u32)(space / (long)R); if (w > (u32)CL_WAVES) w = CL_WAVES;
      if (w < MIN_HIST_WAVES) continue;
      const u32 units = (q + 3) / 4 + (h - q + 1) / 2 + (n - h) + nD + (nT + 1) / 2;   // (chunks: F one, T three to two lists, D two)
      if (!bestW || (u64)w * bestU > (u64)bestW * units) { bestW = w; bestU = units ? units : 1; tp.m16 = q; tp.m32 = h; histBytes = R; }
    }
    if (tp.m32 > tp.m64) tp.m32 = tp.m64;                    // (cannot happen: counts of <= 32 and <= 64)
    nW = bestW ? bestW : (u32)MIN_HIST_WAVES;                // (option 2 always qualifies: translatedShape left room for MIN_HIST_WAVES regions of n bytes)
    tp.posH = (16u * tp.m16 + 63u) & ~63u;
    tp.posF = tp.posH + ((32u * (tp.m32 - tp.m16) + 63u) & ~63u);
    tp.posT = tp.posF + 64u * (tp.m64 - tp.m32);
    tp.posD = tp.posT + 192u * ((tp.m96 - tp.m64 + 1u) / 2u);
    tp.posX = tp.posD + 128u * (tp.m128 - tp.m96);
  }
  // entries whose search goes beyond the home bucket are parked — barcode | rank << 22 | position of the handle << 38 — in a queue of the wave's own
  // behind the handles on the HBM slot, and searched for with search() 64 lanes at a time when the pass is over (or the queue full). What search()
  // cannot settle — a barcode that is new when the table is closed — stays in the queue, for the second table.
  unsigned long long *const queue = (unsigned long long *)(hs + a.handleStride - (size_t)CL_WAVES * TR_QUEUE * 4) + (size_t)uwave * TR_QUEUE;
  u32 qn = 0, qk = 0;                                        // (uniform)
  SlotTable st; st.shape(tab, S, a.hashBits, &sh[2], 0); st.inLds();
  auto settle = [&](const SlotTable &t, bool mayInsert, u32 fillLimit, bool all) {
    u32 k0 = qk;
    for (; all ? k0 < qn : k0 + WAVE <= qn; k0 += WAVE) {
      const bool on = k0 + laneU < qn;
      const unsigned long long e = on ? queue[k0 + laneU] : 0ull;
      const bool insert = mayInsert && shFill() < fillLimit;   // (uniform: one LDS word)
      u32 h = 0; bool ins = false;
      if (on) { h = t.search((u32)e & 0x3FFFFFu, (u32)(e >> 22) & 0xFFFFu, insert, ins); if (h != SlotTable::NOTFOUND) hs[(u32)(e >> 38)] = (u16)h; }
      const u64 balI = __ballot(ins);
      if (balI && lane == 0) atomicAdd(&sh[1], (u32)__popcll(balI));
      const u64 balU = __ballot(on && h == SlotTable::NOTFOUND);
      if (balU) {
        if (on && h == SlotTable::NOTFOUND) queue[qk + (u32)__popcll(balU & ((1ULL << lane) - 1))] = e;   // (qk <= k0: behind what has been read)
        qk += (u32)__popcll(balU);
      }
    }
    const u32 rest = k0 < qn ? qn - k0 : 0u;
    if (rest && k0 != qk) { const unsigned long long e = laneU < rest ? queue[k0 + laneU] : 0ull; if (laneU < rest) queue[qk + laneU] = e; }
    qn = qk + rest;
  };
  // ---- pass A
  {
    bool insert = true;                                      // (uniform) the table takes new barcodes: looked up once per round
    u32 myIns = 0;
    // one chunk: 64 entries (of up to four lists) -> their handles at position pos + lane; i, drem: the lane's rank and what its list holds from this chunk's first lane of the segment on
    auto place = [&](u32 cj, u32 i, bool valid, u32 pos) {
      u32 slot; bool ins;
      const bool done = st.probeHome(cj, i, valid, insert, slot, ins);
      *(u16 *)((char *)hs + ((pos + laneU) << 1)) = (u16)(valid ? slot : S);   // every lane: `none` where the lane has no entry — pass B loads and counts a chunk without looking at a length
                                                             // (a 32-bit byte offset from the uniform base: no 64-bit address arithmetic per lane; a slot holds at most 2^24 handles)
      myIns += (u32)__popcll(__ballot(ins));
      const u64 bal = __ballot(!done);
      if (bal) {
        if (!done) queue[qn + (u32)__popcll(bal & ((1ULL << lane) - 1))] = (unsigned long long)cj | ((unsigned long long)i << 22) | ((unsigned long long)(pos + laneU) << 38);
        qn += (u32)__popcll(bal);
      }
    };
    auto roundEnd = [&](bool last) {
      if (last || qn - qk >= WAVE) {
        if (myIns) { if (lane == 0) atomicAdd(&sh[1], myIns); myIns = 0; }
        settle(st, true, closeAt, last);
        if (qn > TR_QUEUE - 4 * RIF * WAVE) { sh[2] = 1; qn = qk = 0; }   // more entries wait for the second table than the queue holds: the block is handed on
      }
    };
    // the units of one class: SEGW lanes per list, NCH chunks per unit (2: the lists of 65 .. 128 entries, one per unit)
    auto classA = [&](auto segw_, auto nch_, const u32 firstRank, const u32 endRank, const u32 posBase) {
      constexpr u32 SEGW = decltype(segw_)::value, NCH = decltype(nch_)::value, LPU = WAVE / SEGW;
      constexpr int RIF = SEGW == 64 ? RIF_F : RIF_Q;     // units in flight: three in every class of the half-CU kernel (class T: two units of three chunks) — see H10X_TP_RIF0
      const u32 nUnits = (endRank - firstRank + LPU - 1) / LPU;
      if (uwave * RIF >= nUnits) return;
      constexpr u32 stepA = CL_WAVES * RIF;
      // (seg / jl — the lane's segment and place in it — are formed from an opaque copy of the lane number wherever they are used: as function-long values the compiler kept a set of
      //  them per class in registers it did not have, and reloaded them from scratch inside the loops with a wait for every outstanding load)
      u32 lu = laneU, seg = 0, jl = 0;
      auto lanes = [&]() { asm volatile("" : "+v"(lu)); seg = lu / SEGW; jl = lu & (SEGW - 1); };
      // descriptors of the RIF * LPU lists of a round in ONE register: lane t the offset, lane 32 + t the length of the round's t-th list (clamped index: no lane skips the load)
      auto descBatch = [&](u32 u0) { const u32 i = firstRank + u0 * LPU + (lu & 31u); return ((const u32 *)gr)[2 * (size_t)(i < n ? i : n - 1) + (lu >> 5)]; };
      u32 cN[RIF][NCH], lenN[RIF];
      auto issue = [&](u32 dv, u32 u0) {                     // the entry loads of a round (no lane skips one: rows[] has ROWS_PAD entries of slack)
#pragma unroll
        for (int t = 0; t < RIF; ++t) {
          u32 lo, hi;
          if (SEGW == 64) { lo = (u32)__builtin_amdgcn_readlane((int)dv, t); hi = (u32)__builtin_amdgcn_readlane((int)dv, 32 + t); }
          else { const u32 src = (u32)t * LPU + seg; lo = (u32)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)dv); hi = (u32)__builtin_amdgcn_ds_bpermute((int)((32u + src) << 2), (int)dv); }
          const u32 r = firstRank + (u0 + (u32)t) * LPU + seg;
          lenN[t] = (r >= 1 && r < endRank) ? hi : 0u;        // rank 0 is never processed (hash10x.c:789); ranks of the next class and past the last rank: no entries
          const u32 *row = ROWP(lo) + jl;
#pragma unroll
          for (u32 c = 0; c < NCH; ++c) cN[t][c] = row[c * WAVE];
        }
      };
      u32 u0 = uwave * RIF;
      u32 dvN = descBatch(u0), dvNN = descBatch(u0 + stepA);
      lanes();
      issue(dvN, u0);
      for (; u0 < nUnits; u0 += stepA) {
        lanes();
        u32 c0[RIF][NCH], len[RIF];
#pragma unroll
        for (int t = 0; t < RIF; ++t) { len[t] = lenN[t];
#pragma unroll
          for (u32 c = 0; c < NCH; ++c) c0[t][c] = cN[t][c]; }
        dvN = dvNN; dvNN = descBatch(u0 + 2 * stepA);
        issue(dvN, u0 + stepA);
        if (S2) {                                            // a table that may fill up: say what this wave has put in, see whether it still takes barcodes
          if (myIns) { if (lane == 0) atomicAdd(&sh[1], myIns); myIns = 0; }
          insert = (u32)__builtin_amdgcn_readfirstlane((int)shFill()) < closeAt;
        }
#pragma unroll
        for (int t = 0; t < RIF; ++t) {
          if (u0 + (u32)t >= nUnits) break;                  // (uniform) the class region ends here: what lies behind belongs to the next class
          const u32 r = firstRank + (u0 + (u32)t) * LPU + seg;
#pragma unroll
          for (u32 c = 0; c < NCH; ++c) place(c0[t][c], r, jl + c * WAVE < len[t] && c0[t][c] != code, posBase + (u0 + (u32)t) * (WAVE * NCH) + c * WAVE);
        }
        roundEnd(false);
      }
    };
    // class T: two lists of 65 .. 96 entries to a unit — chunk 0 the first list's entries 0 .. 63, chunk 1 the second's, chunk 2 their tails (lanes 0-31 / 32-63)
    auto classAT = [&](const u32 firstRank, const u32 endRank, const u32 posBase) {
      constexpr int RIF = RIF_T;
      const u32 nUnits = (endRank - firstRank + 1u) / 2u;
      if (uwave * RIF >= nUnits) return;
      constexpr u32 stepA = CL_WAVES * RIF;
      u32 lu = laneU;
      auto descBatch = [&](u32 u0) { const u32 i = firstRank + u0 * 2u + (lu & 31u); return ((const u32 *)gr)[2 * (size_t)(i < n ? i : n - 1) + (lu >> 5)]; };
      u32 cN[RIF][3], lenAN[RIF], lenBN[RIF];                // (the lengths: uniform)
      auto issue = [&](u32 dv, u32 u0) {
#pragma unroll
        for (int t = 0; t < RIF; ++t) {
          const u32 loA = (u32)__builtin_amdgcn_readlane((int)dv, 2 * t), hiA = (u32)__builtin_amdgcn_readlane((int)dv, 32 + 2 * t);
          const u32 loB = (u32)__builtin_amdgcn_readlane((int)dv, 2 * t + 1), hiB = (u32)__builtin_amdgcn_readlane((int)dv, 32 + 2 * t + 1);
          const u32 rA = firstRank + (u0 + (u32)t) * 2u, rB = rA + 1u;
          lenAN[t] = (rA >= 1 && rA < endRank) ? hiA : 0u;    // rank 0 is never processed (hash10x.c:789); past the class: no entries
          lenBN[t] = rB < endRank ? hiB : 0u;
          const u32 *rowA = ROWP(loA), *rowB = ROWP(loB);
          cN[t][0] = rowA[lu]; cN[t][1] = rowB[lu];
          cN[t][2] = (lu >= 32u ? rowB : rowA)[64u + (lu & 31u)];   // (rows[] has ROWS_PAD entries of slack: no lane skips a load)
        }
      };
      u32 u0 = uwave * RIF;
      u32 dvN = descBatch(u0), dvNN = descBatch(u0 + stepA);
      issue(dvN, u0);
      for (; u0 < nUnits; u0 += stepA) {
        asm volatile("" : "+v"(lu));
        u32 c0[RIF][3], lenA[RIF], lenB[RIF];
#pragma unroll
        for (int t = 0; t < RIF; ++t) { lenA[t] = lenAN[t]; lenB[t] = lenBN[t];
#pragma unroll
          for (u32 c = 0; c < 3; ++c) c0[t][c] = cN[t][c]; }
        dvN = dvNN; dvNN = descBatch(u0 + 2 * stepA);
        issue(dvN, u0 + stepA);
        if (S2) {                                            // a table that may fill up: say what this wave has put in, see whether it still takes barcodes
          if (myIns) { if (lane == 0) atomicAdd(&sh[1], myIns); myIns = 0; }
          insert = (u32)__builtin_amdgcn_readfirstlane((int)shFill()) < closeAt;
        }
#pragma unroll
        for (int t = 0; t < RIF; ++t) {
          if (u0 + (u32)t >= nUnits) break;                  // (uniform)
          const u32 rA = firstRank + (u0 + (u32)t) * 2u, rB = rA + 1u;
          const bool up = lu >= 32u; const u32 jl = lu & 31u;
          const u32 posU = posBase + (u0 + (u32)t) * 192u;
          place(c0[t][0], rA, lu < lenA[t] && c0[t][0] != code, posU);
          place(c0[t][1], rB, lu < lenB[t] && c0[t][1] != code, posU + 64u);
          place(c0[t][2], up ? rB : rA, 64u + jl < (up ? lenB[t] : lenA[t]) && c0[t][2] != code, posU + 128u);
        }
        roundEnd(false);
      }
    };
    classA(std::integral_constant<u32, 16>{}, std::integral_constant<u32, 1>{}, 0u, tp.m16, 0u);
    classA(std::integral_constant<u32, 32>{}, std::integral_constant<u32, 1>{}, tp.m16, tp.m32, tp.posH);
    classA(std::integral_constant<u32, 64>{}, std::integral_constant<u32, 1>{}, tp.m32, tp.m64, tp.posF);
    classAT(tp.m64, tp.m96, tp.posT);
    classA(std::integral_constant<u32, 64>{}, std::integral_constant<u32, 2>{}, tp.m96, tp.m128, tp.posD);
    for (u32 i = tp.m128 + uwave; i < n; i += CL_WAVES) {     // class X (depth ranges beyond 128): list after list, chunk after chunk
      if (i == 0) continue;
      const u64 g2 = gr[i]; const u32 d = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(g2 >> 32)); const u32 *row = ROWP((u32)__builtin_amdgcn_readfirstlane((int)(u32)g2));
      const u32 pos = tp.posX + hst * (i - tp.m128);
      for (u32 lj = 0; lj < d; lj += WAVE) {
        const u32 cj = row[lj + laneU];
        place(cj, i, lj + laneU < d && cj != code, pos + lj);
        roundEnd(false);
      }
    }
    roundEnd(true);
    STAMP(1);
    if (qn && lane == 0) sh[0] = 1;
  }
  __syncthreads();                                           // (the handles other waves wrote are plain stores of this CU, read back through its own L1: workgroup scope is enough)
  STAMP(2);
  const bool spill = sh[0] != 0 && !sh[2];                   // (uniform: read after the barrier)
  if (spill) {                                               // the first table is final now: what is still queued is looked up once more; what is still not there goes into the second table
    qk = 0; settle(st, false, 0, true);
    SYNC();
  }
  if (sh[2]) {
    if (tid == 0) a.overflow[atomicAdd(a.overflowCount, 1u)] = lcode;
    SYNC();
    return;
  }
  // ---- the table gives its LDS back: the ranks, 2 bytes per slot, compacted in place
  FirstSlots ft; ft.none = S;
  size_t firstBytes;
  auto compactRanks = [&](const u32 *src, u16 *dst, u32 count) {   // dst below src (or equal)
    for (u32 base = 0; base < count; base += 4 * CL_THREADS) {
      u32 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const u32 idx = base + (u32)k * CL_THREADS + tid; v[k] = src[idx < count ? idx : count - 1]; }   // (reads without a branch around them)
      SYNC();
#pragma unroll
      for (int k = 0; k < 4; ++k) { const u32 idx = base + (u32)k * CL_THREADS + tid; if (idx < count) dst[idx] = (u16)(v[k] >> 16); }
    }
  };
  {
    // the first table, four entries per lane and step (round 6): 16-byte reads, 8-byte writes, rounds of 3 x 1024 x 4 entries with ONE barrier each — two rounds for the
    // largest table — where a word at a time took a barrier per 4096 entries and five times the LDS instructions; the phase is workgroup-synchronous (nobody to hide its
    // latencies behind but the CU's other workgroup): this and the 16-byte clears took the 3 Gb set's cluster time from 285 to ... ms. Round r writes bytes
    // [r * 24 K, (r + 1) * 24 K), which lie inside what rounds <= r have read (every round's reads are behind a barrier before the next round writes).
    const u32 noneWord = tab[S];                             // (the word of handle `none`, index S: read before anything is overwritten — S >= 64, so it is not in round 0's writes... kept by every lane, written by one)
    const uint4 *const src4 = (const uint4 *)tab; uint2 *const dst2 = (uint2 *)region; const u32 count4 = S / 4;
    for (u32 base = 0; base < count4; base += 3 * CL_THREADS) {
      uint4 v[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) { const u32 idx = base + (u32)k * CL_THREADS + tid; v[k] = src4[idx < count4 ? idx : count4 - 1]; }
      SYNC();
#pragma unroll
      for (int k = 0; k < 3; ++k) { const u32 idx = base + (u32)k * CL_THREADS + tid; if (idx < count4) dst2[idx] = make_uint2((v[k].x >> 16) | (v[k].y & 0xFFFF0000u), (v[k].z >> 16) | (v[k].w & 0xFFFF0000u)); }
    }
    SYNC();
    if (tid == 0) ((u16 *)region)[S] = (u16)(noneWord >> 16);
  }
  ft.base = region; ft.sh = 1; firstBytes = pad16(2 * ((size_t)S + 8));
  if (spill) {
    u32 *const tab2 = (u32 *)(region + firstBytes);
    SYNC();
    for (u32 i = tid; i < S2; i += CL_THREADS) tab2[i] = 0xFFFFFFFFu;
    if (tid == 0) sh[1] = 0;
    SYNC();
    SlotTable st2; st2.shape(tab2, S2, a.hashBits, &sh[2], S + 1);
    qk = 0; settle(st2, true, 0xFFFFFFFFu, true);
    if (qn) sh[2] = 1;                                       // (cannot happen: search() inserts or reports the overflow)
    SYNC();
    if (sh[2]) {
      if (tid == 0) a.overflow[atomicAdd(a.overflowCount, 1u)] = lcode;
      SYNC();
      return;
    }
    compactRanks(tab2, (u16 *)region + S + 1, S2);
    firstBytes = pad16(2 * ((size_t)S + S2 + 16));
    const long space = (long)a.ldsBudget - (long)firstBytes - (long)pad16((size_t)n * 2) - 64;
    nW = (u32)(space / (long)histBytes); if (nW > (u32)CL_WAVES) nW = CL_WAVES; if (nW < 1) nW = 1;   // (histBytes <= spaceSpill: at least one region set)
  }
  u16 *const root = (u16 *)(region + firstBytes);
  u32 *const hist = (u32 *)(region + firstBytes + pad16((size_t)n * 2));
  const u32 histWords = histBytes / 4;
  SYNC();
  {                                                          // (16 bytes per store; the region starts 16-byte aligned)
    const u32 words = nW * histWords;
    for (u32 i = tid; i < words / 4; i += CL_THREADS) ((uint4 *)hist)[i] = make_uint4(0, 0, 0, 0);
    if (tid < (words & 3u)) hist[(words & ~3u) + tid] = 0;
  }
  fill16<CL_THREADS>(root, pad16((size_t)n * 2), 0xFFFFFFFFu);
  if (tid == 0) root[0] = 0;                                 // rank 0 is never processed (hash10x.c:789): inactive, its own root
  SYNC();
  STAMP(3);

  // ---- pass B: the list loop on handles, a unit (64 handles of up to four lists) per step. No barrier: first[] is final; root[] of an earlier rank is
  // either on record or reads "open" (then the rank is settled behind the loop), whichever wave gets there first.
  nW = (u32)__builtin_amdgcn_readfirstlane((int)nW);
  const bool listWave = uwave < nW;
  if (listWave) {
    u32 *const myHist = hist + uwave * histWords;
    auto classB = [&](auto segw_, auto nch_, const u32 firstRank, const u32 endRank, const u32 posBase) {
      constexpr u32 SEGW = decltype(segw_)::value, NCH = decltype(nch_)::value, LPU = WAVE / SEGW;
      constexpr int RIF = SEGW == 64 ? RIF_F : RIF_Q;
      const u32 nUnits = (endRank - firstRank + LPU - 1) / LPU;
      if (uwave * RIF >= nUnits) return;
      const u32 stepB = nW * RIF;
      u32 lu = laneU, seg = 0, jl = 0; u32 *segHist = myHist;    // (formed per round from an opaque copy of the lane number: see pass A)
      auto lanes = [&]() { asm volatile("" : "+v"(lu)); seg = lu / SEGW; jl = lu & (SEGW - 1);
                           segHist = myHist + seg * ((endRank + 3u) / 4u); };   // a first[] value counted here is below the list's rank, i.e. below the class's last rank
      u32 hN[RIF][NCH];
      auto issue = [&](u32 u0) {                             // (uniform base + 32-bit position: no 64-bit pointer per lane to keep across the loop)
#pragma unroll
        for (int t = 0; t < RIF; ++t) { const u32 u = u0 + (u32)t < nUnits ? u0 + (u32)t : nUnits - 1;   // (past the class: its last unit again, from the caches)
#pragma unroll
          for (u32 c = 0; c < NCH; ++c) hN[t][c] = hs[posBase + u * (WAVE * NCH) + c * WAVE + lu]; }
      };
      u32 u0 = uwave * RIF;
      issue(u0);
      for (; u0 < nUnits; u0 += stepB) {
        lanes();
        u32 h[RIF][NCH];
#pragma unroll
        for (int t = 0; t < RIF; ++t) {
#pragma unroll
          for (u32 c = 0; c < NCH; ++c) h[t][c] = hN[t][c]; }
        issue(u0 + stepB);
#pragma unroll
        for (int t = 0; t < RIF; ++t) {
          if (u0 + (u32)t >= nUnits) break;                  // (uniform)
          const u32 r = firstRank + (u0 + (u32)t) * LPU + seg;   // the lane's rank
          const bool live = r >= 1 && r < endRank;
          u32 f[NCH]; bool ok[NCH]; u32 tot = 0, key = 0;
#pragma unroll
          for (u32 c = 0; c < NCH; ++c) { f[c] = ft.peek(h[t][c]); ok[c] = f[c] < r; tot += seg_popc<SEGW>(__ballot(ok[c]), lu); }   // (`none` reads 0xFFFF: never below a rank)
#pragma unroll
          for (u32 c = 0; c < NCH; ++c)
            if (ok[c]) {
              const int sh8 = (f[c] & 3) * 8;
              u32 *const cell = &segHist[f[c] >> 2];
              const u32 cnt = ((atomicAdd(cell, 1u << sh8) >> sh8) & 0xFFu) + 1;   // the lane that arrives last at a value sees its full count
              if (NCH == 1) *cell = 0;                       // cleared in the exec region of its count (ds ops of a wave stay in order); two chunks: behind both counts, below
              const u32 k = (cnt << 16) | (0xFFFFu - f[c]);
              key = k > key ? k : key;
            }
          if (NCH > 1) {
#pragma unroll
            for (u32 c = 0; c < NCH; ++c) if (ok[c]) segHist[f[c] >> 2] = 0;
          }
          key = seg_max_u32<SEGW>(key, lu);               // msMax << 16 | ~msBest: the mode, ties to the lowest rank (hash10x.c:801-806)
          const u32 bcnt = key >> 16, best = 0xFFFFu - (key & 0xFFFFu);       // (no usable entry: key 0 -> best NONE16, bcnt 0)
          const bool act = bcnt >= thr;                      // hash10x.c:807
          const u32 rbv = root[act ? best : 0u];             // founding rank of the cluster the rank joins (NONE16: not on record yet — settled behind the loop)
          const u32 rb = act ? rbv : (u32)NONE16;
          u32 q = 0;
#pragma unroll
          for (u32 c = 0; c < NCH; ++c) q += seg_popc<SEGW>(__ballot(f[c] == rb), lu);
          if (rb == NONE16) q = 0;
          if (live && jl == 0) { root[r] = (u16)(act ? rb : r); (a.res + o)[r] = RES_PACK(act ? best : (u32)NONE16, q, tot); }
        }
      }
    };
    // class T: the unit's two lists count in the wave's one region one after the other (ds operations of a wave stay in order: the second list's counts find the
    // first one's cells cleared); per-list quantities are uniform — the first list's in chunk 0 and lanes 0-31 of chunk 2, the second's in chunk 1 and lanes 32-63
    auto classBT = [&](const u32 firstRank, const u32 endRank, const u32 posBase) {
      constexpr int RIF = RIF_T;
      const u32 nUnits = (endRank - firstRank + 1u) / 2u;
      if (uwave * RIF >= nUnits) return;
      const u32 stepB = nW * RIF;
      u32 lu = laneU;
      u32 hN[RIF][3];
      auto issue = [&](u32 u0) {
#pragma unroll
        for (int t = 0; t < RIF; ++t) { const u32 u = u0 + (u32)t < nUnits ? u0 + (u32)t : nUnits - 1;
#pragma unroll
          for (u32 c = 0; c < 3; ++c) hN[t][c] = hs[posBase + u * 192u + c * WAVE + lu]; }
      };
      auto count = [&](u32 f, bool on, u32 &key) {             // one more of value f in the region: the lane that arrives last at a value sees its full count
        if (on) {
          const int sh8 = (f & 3) * 8;
          const u32 cnt = ((atomicAdd(&myHist[f >> 2], 1u << sh8) >> sh8) & 0xFFu) + 1;
          const u32 k = (cnt << 16) | (0xFFFFu - f);
          key = k > key ? k : key;
        }
      };
      u32 u0 = uwave * RIF;
      issue(u0);
      for (; u0 < nUnits; u0 += stepB) {
        asm volatile("" : "+v"(lu));
        u32 h[RIF][3];
#pragma unroll
        for (int t = 0; t < RIF; ++t) {
#pragma unroll
          for (u32 c = 0; c < 3; ++c) h[t][c] = hN[t][c]; }
        issue(u0 + stepB);
#pragma unroll
        for (int t = 0; t < RIF; ++t) {
          if (u0 + (u32)t >= nUnits) break;                  // (uniform)
          const u32 rA = firstRank + (u0 + (u32)t) * 2u, rB = rA + 1u;       // (uniform)
          const bool liveA = rA >= 1, liveB = rB < endRank, up = lu >= 32u;
          const u32 f0 = ft.peek(h[t][0]), f1 = ft.peek(h[t][1]), f2 = ft.peek(h[t][2]);   // (`none` reads 0xFFFF: never below a rank; a list that is not live has nothing but `none`)
          const bool ok0 = f0 < rA, ok1 = f1 < rB, ok2 = f2 < (up ? rB : rA);
          const u64 bal2 = __ballot(ok2);
          const u32 totA = (u32)__popcll(__ballot(ok0)) + (u32)__popc((u32)bal2), totB = (u32)__popcll(__ballot(ok1)) + (u32)__popc((u32)(bal2 >> 32));
          u32 keyA = 0, keyB = 0;
          count(f0, ok0, keyA); count(f2, ok2 && !up, keyA);
          if (ok0) myHist[f0 >> 2] = 0;
          if (ok2 && !up) myHist[f2 >> 2] = 0;
          count(f1, ok1, keyB); count(f2, ok2 && up, keyB);
          if (ok1) myHist[f1 >> 2] = 0;
          if (ok2 && up) myHist[f2 >> 2] = 0;
          keyA = wave_max_u32(keyA); keyB = wave_max_u32(keyB);   // msMax << 16 | ~msBest: the mode, ties to the lowest rank (hash10x.c:801-806)
          const u32 bcntA = keyA >> 16, bestA = 0xFFFFu - (keyA & 0xFFFFu), bcntB = keyB >> 16, bestB = 0xFFFFu - (keyB & 0xFFFFu);
          const bool actA = bcntA >= thr, actB = bcntB >= thr;   // hash10x.c:807
          const u32 rbvA = root[actA ? bestA : 0u], rbvB = root[actB ? bestB : 0u];
          const u32 rbA = actA ? rbvA : (u32)NONE16;
          const u32 rootA = actA ? rbA : rA;                 // what goes on record for the first list
          u32 rbB = actB ? rbvB : (u32)NONE16;
          if (actB && bestB == rA && liveA) rbB = rootA;     // the second list joins the first: its root is known here, not yet on record
          const u64 balq = __ballot(f2 == (up ? rbB : rbA));
          u32 qA = (u32)__popcll(__ballot(f0 == rbA)) + (u32)__popc((u32)balq), qB = (u32)__popcll(__ballot(f1 == rbB)) + (u32)__popc((u32)(balq >> 32));
          if (rbA == NONE16) qA = 0;
          if (rbB == NONE16) qB = 0;
          if (liveA && lu == 0) { root[rA] = (u16)rootA; (a.res + o)[rA] = RES_PACK(actA ? bestA : (u32)NONE16, qA, totA); }
          if (liveB && lu == 32u) { root[rB] = (u16)(actB ? rbB : rB); (a.res + o)[rB] = RES_PACK(actB ? bestB : (u32)NONE16, qB, totB); }
        }
      }
    };
    classB(std::integral_constant<u32, 16>{}, std::integral_constant<u32, 1>{}, 0u, tp.m16, 0u);
    classB(std::integral_constant<u32, 32>{}, std::integral_constant<u32, 1>{}, tp.m16, tp.m32, tp.posH);
    classB(std::integral_constant<u32, 64>{}, std::integral_constant<u32, 1>{}, tp.m32, tp.m64, tp.posF);
    classBT(tp.m64, tp.m96, tp.posT);
    classB(std::integral_constant<u32, 64>{}, std::integral_constant<u32, 2>{}, tp.m96, tp.m128, tp.posD);
    for (u32 i = tp.m128 + uwave; i < n; i += nW) {          // class X
      if (i == 0) continue;
      const u32 d = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(gr[i] >> 32));
      const u16 *const hrow = hs + tp.posX + (size_t)hst * (i - tp.m128);
      u32 best, bcnt, tot, rb, q;
      if (d < RCHUNK * WAVE) row_mode_hist<IN_LDS, RCHUNK>(hrow, (u32)hrow[laneU], (u32)hrow[WAVE + laneU], d, code, i, ft, myHist, root, thr, best, bcnt, tot, rb, q);
      else {
        row_mode_long(hrow, d, code, i, ft, best, bcnt, tot); rb = NONE16; q = 0;
        if (bcnt >= thr) { rb = root[best]; if (rb != NONE16) { u32 t2; q = row_count_value(hrow, d, code, i, ft, rb, t2); } }
      }
      const bool act = bcnt >= thr;
      if (lane == 0) { root[i] = (u16)(act ? rb : i); (a.res + o)[i] = RES_PACK(act ? best : (u32)NONE16, q, tot); }
    }
  }
  STAMP(4);
  SYNC();
  // ---- the ranks left open by the loop (their msBest's root was not on record yet): roots by walking down the msBest chain, then
  // their lists' handles once more for minShareCount[root] and msTot — as in cluster_one_block
  {
    u16 *todo = (u16 *)hist;                                 // the histograms are idle from here on
    if (tid == 0) sh[2] = 0;
    SYNC();
    for (u32 i0 = 0; i0 < n; i0 += CL_THREADS) {
      const u32 i = i0 + tid;
      const bool need = i >= 1 && i < n && root[i] == NONE16;
      const u64 bal = __ballot(need);
      if (bal) {
        u32 base = 0;
        if (lane == 0) base = atomicAdd(&sh[2], (u32)__popcll(bal));
        base = (u32)__shfl((int)base, 0);
        if (need) todo[base + (u32)__popcll(bal & ((1ULL << lane) - 1))] = (u16)i;
      }
    }
    SYNC();
    const u32 nTodo = sh[2];
    u64 *const res = a.res + o;
    if (a.phase && tid == 0) { atomicAdd((u64 *)&a.phase[6], (u64)nTodo); atomicAdd((u64 *)&a.phase[7], (u64)n); }   // diagnostic: ranks settled behind the loop / ranks
    u16 *const todoBest = todo + nTodo;                      // msBest of the open ranks, kept here for the recount below (read from the result word once, not twice; 4 nTodo bytes < the five histogram regions of >= n bytes)
    for (u32 k = tid; k < nTodo; k += CL_THREADS) {
      const u32 i = todo[k];
      u32 r = (u32)(__hip_atomic_load(&res[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFFFu);   // msBest: from the rank's result word (read past this CU's L1)
      todoBest[k] = (u16)r;
      for (u32 hop = 0; hop < n && r < n; ++hop) {
        const u32 rr = *(volatile u16 *)&root[r];
        if (rr != NONE16) { r = rr; break; }
        r = (u32)(__hip_atomic_load(&res[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFFFu);
      }
      root[i] = (u16)r;                                      // (a walker passing through i meanwhile reads NONE16 or r: the same answer either way)
    }
    SYNC();
    // the open ranks' lists once more, TIF of them in flight per wave. Round 6: every load of a round is issued without a branch around it — written as
    // `lane < length ? load : none` each of the 2 x TIF handle loads was a branch, a load and a wait of its own BEHIND the descriptor load its condition came from
    // (twelve memory latencies in a row for four ranks; the phase was 14-16 % of a block's time for 4-9 % of its ranks: HISTORY 9) — a rank past the end of the
    // list re-reads rank 0's places (valid memory: the slot's queue area lies behind the handles), lanes past a list's end are masked AFTER the loads.
    constexpr int TIF = ROWS_IN_FLIGHT;
    for (u32 k0 = uwave * TIF; k0 < nTodo; k0 += CL_WAVES * TIF) {
      u32 ii[TIF], hA[TIF], hB[TIF], dl[TIF], qv[TIF], bst[TIF]; u64 g[TIF];
#pragma unroll
      for (int t = 0; t < TIF; ++t) {
        const u32 kk = k0 + (u32)t < nTodo ? k0 + (u32)t : nTodo - 1;      // (past the end: the last open rank again, its result dropped below)
        ii[t] = (u32)__builtin_amdgcn_readfirstlane((int)(u32)todo[kk]);
        bst[t] = (u32)todoBest[kk];
        qv[t] = (u32)root[ii[t]];
      }
#pragma unroll
      for (int t = 0; t < TIF; ++t) g[t] = gr[ii[t]];
#pragma unroll
      for (int t = 0; t < TIF; ++t) {
        const u16 *const hrow = hs + tpPos(tp, ii[t], hst), *const hrow64 = hs + tpPos64(tp, ii[t], hst);   // (class T keeps entries 64 .. in the unit's third chunk)
        hA[t] = (u32)hrow[laneU]; hB[t] = (u32)hrow64[laneU];
      }
#pragma unroll
      for (int t = 0; t < TIF; ++t) {
        dl[t] = k0 + (u32)t < nTodo ? (u32)__builtin_amdgcn_readfirstlane((int)(u32)(g[t] >> 32)) : 0u;
        hA[t] = laneU < dl[t] ? hA[t] : S; hB[t] = WAVE + laneU < dl[t] ? hB[t] : S;   // (only the list's own entries: the neighbours in the chunk are other lists')
      }
#pragma unroll
      for (int t = 0; t < TIF; ++t) {
        if (dl[t] == 0) continue;
        const u32 i = ii[t];
        u32 f = ft.peek(hA[t]);
        u32 q = (u32)__popcll(__ballot(f == qv[t])), tt = (u32)__popcll(__ballot(f < i));
        if (dl[t] > WAVE) { f = ft.peek(hB[t]); q += (u32)__popcll(__ballot(f == qv[t])); tt += (u32)__popcll(__ballot(f < i)); }
        if (dl[t] > 2 * WAVE) { const u16 *const hrow = hs + tpPos(tp, i, hst); for (u32 b0 = 2 * WAVE; b0 < dl[t]; b0 += WAVE) { f = b0 + laneU < dl[t] ? ft.peek(hrow[b0 + laneU]) : (u32)NONE16; q += (u32)__popcll(__ballot(f == qv[t])); tt += (u32)__popcll(__ballot(f < i)); } }
        if (lane == 0) res[i] = RES_PACK(bst[t], q, tt);
      }
    }
  }
  if (tid == 0) { acc.s[3] += a.entries[lcode]; acc.s[0] += n; acc.s[1] += a.blocks[lcode].nHash; acc.s[2] += 1; }   // (the work counter: the block's list entries as --hashDepthRange summed them — per chunk it was two vector instructions)
  SYNC();
  STAMP(5);
}

#undef SYNC
#undef SYNC_LDS
#undef ROWP

// Two 1024-lane workgroups share a CU only if a wave stays within 64 VGPRs (8 waves per SIMD): the kernel is latency
// bound (chains of LDS round trips per list), so the second workgroup is worth far more than the few loop-invariant
// values the compiler then keeps in scratch (measured: 4.3 -> 3.4 ms on the yeast-scale set).
template <bool IN_LDS, int FIRST_MODE, int CL_THREADS, int KLASS = 0 /* distinct functions per launch class */>
__global__ __launch_bounds__(CL_THREADS) __attribute__((amdgpu_waves_per_eu(KLASS == 0 ? (CL_THREADS == 1024 ? 8 : (CL_THREADS == 768 ? 6 : 4)) : 4)))   // classes 2, 3: one workgroup per CU anyway
void cluster_kernel(ClusterArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ u32 sh[4 + 128];                               // [0..3] scalars, then per-wave scan totals
  unsigned char *region = IN_LDS ? smem : a.scratch + (size_t)blockIdx.x * a.scratchStride;
  u16 *firstGlobal = (IN_LDS && FIRST_MODE == 2) ? (u16 *)(a.scratch + (size_t)blockIdx.x * a.scratchStride) : nullptr;
  __shared__ unsigned long long accS[5];
  WorkAcc acc{0u, accS};
  if (threadIdx.x < 4) accS[threadIdx.x] = 0;               // (the work loop starts with a barrier)
  if (a.started && threadIdx.x == 0) { __hip_atomic_store(&a.started[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) sh[3] = atomicAdd(a.workCounter, 1u);
    __syncthreads();
    const u32 wi = sh[3];
    if (wi >= a.nFront + a.nList) break;                     // every wave of the workgroup leaves together
    // (making the block number scalar with readfirstlane — and with it the rank count and every loop bound — removes a third
    // of the spills and 170 of 6900 instructions, and is 3-4 % SLOWER: 2.49 against 2.41 ms, measured round 2)
    if constexpr (FIRST_MODE == 5) cluster_one_block_tp<CL_THREADS, KLASS>(a, wi < a.nFront ? a.front[wi] : a.list[wi - a.nFront], region, sh, acc);
    else if constexpr (FIRST_MODE == 4) cluster_one_block_tr<CL_THREADS, KLASS>(a, wi < a.nFront ? a.front[wi] : a.list[wi - a.nFront], region, sh, acc);
    else cluster_one_block<IN_LDS, FIRST_MODE, CL_THREADS, KLASS>(a, wi < a.nFront ? a.front[wi] : a.list[wi - a.nFront], region, firstGlobal, sh, acc);
  }
  u64 depth = acc.depth;
  for (int s = 32; s; s >>= 1) depth += __shfl_down(depth, s);
  if ((threadIdx.x & (WAVE - 1)) == 0 && depth) atomicAdd((u64 *)&a.stats[1], depth);
  if (threadIdx.x == 0 && accS[2]) { atomicAdd((u64 *)&a.stats[0], (u64)accS[0]); atomicAdd((u64 *)&a.stats[1], (u64)accS[3]); atomicAdd((u64 *)&a.stats[2], (u64)accS[1]); atomicAdd((u64 *)&a.stats[3], (u64)accS[2]); }
}

// ---- list descriptors: per good hash of a block, in rank order, where its barcode list starts in rows[] and how long it is.
// Built once per --hashDepthRange (after the list exchange of a sharded run), read by every --cluster that follows: the
// cluster kernel used to fetch them per barcode through three dependent gathers (position -> hash index -> offset, depth)
// into LDS arrays of 6 bytes per rank.
__global__ __launch_bounds__(256)
void good_rows_kernel(const h10x_clushash *__restrict__ ch, const u64 *__restrict__ blockOff, const u32 *__restrict__ nGood, const u16 *__restrict__ goodPos, u32 nBlocks,
                      const u32 *__restrict__ hashDepth, const u64 *__restrict__ rowStart, u32 rowShift, u64 *__restrict__ goodRow) {
  for (u32 c = blockIdx.x; c < nBlocks; c += gridDim.x) {
    const u32 n = nGood[c]; const u64 o = blockOff[c];
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) {
      const u32 x = ch[o + goodPos[o + i]].hash;
      goodRow[o + i] = (u64)(u32)(rowStart[x] >> rowShift) | ((u64)hashDepth[x] << 32);
    }
  }
}

// ---- (c) the order-dependent part of hash10x.c:807-822, restated without a serial walk, from the msBest column the list
// loop left behind (res[], RES_PACK: msBest of an active rank — msMax >= threshold — NONE16 otherwise, and the two counts
// of its pointToMin term, which this kernel divides and leaves in the same 8 bytes as a double: 0.0 for an inactive rank
// and for every rank from the abandoning turn on, so that point_sum_kernel simply adds the column in order).
// An active rank always ends up labelled at its own turn, and msBest < i, so: the cluster of an active rank is the one of
// the first INACTIVE rank on its msBest chain (its root = clusterMin of that cluster); an inactive rank founds a cluster at
// the first turn i' of an active rank pointing at it directly; clusters are numbered in founding order; the 256th founding
// turn is where the reference gives up (hash10x.c:810-816: labels wiped, pointToMin keeps the terms added so far).
// => roots by pointer jumping, founding turns by CAS-min, numbers by a scan. One workgroup per barcode, 8 bytes of LDS per
// rank; launched in classes by rank count (each launch skips the blocks of the others), the largest on HBM scratch.
struct ReplayArgs {
  h10x_block *blocks; const u64 *blockOff; h10x_clushash *clusHash; const u16 *goodPos; const u32 *nGood;
  u64 *res;                                                 // in: result words of the list loop; out: the ranks' pointToMin terms (double)
  u32 codeMin, span, nLo, nHi;                               // blocks [codeMin, codeMin + span) with nLo < nGood <= nHi ...
  const u32 *list;                                           // ... or, if not null, the `span` blocks of this list (the few large ones: no workgroup per barcode of the range)
  unsigned char *scratch; size_t scratchStride;              // IN_LDS = false: working set per workgroup
  u32 *raw;                                                  // out, two words per block: clusters before the read merge (bit 31: given up at the 256th), good hashes with a label —
                                                             // what the reference's --verbose line of codeClusterFind says (hash10x.c:827-834)
};
#define SYNC() do { __syncthreads(); if (!IN_LDS) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); } while (0)
template <bool IN_LDS, int THREADS>
__global__ __launch_bounds__(THREADS)
void replay_kernel(ReplayArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ u32 sh[4 + THREADS / WAVE];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
  unsigned char *region = IN_LDS ? smem : a.scratch + (size_t)blockIdx.x * a.scratchStride;
  for (u32 wi = blockIdx.x; wi < a.span; wi += gridDim.x) {
    const u32 c = a.list ? a.list[wi] : a.codeMin + wi;
    const u32 n = a.nGood[c];
    if (n == 0 || n <= a.nLo || n > a.nHi) continue;         // uniform; n == 0: block left untouched (hash10x.c:780)
    const u64 o = a.blockOff[c];
    const size_t col = ((size_t)n * 2 + 15) & ~(size_t)15;
    u16 *bst = (u16 *)region, *ptr = (u16 *)(region + col), *ft = (u16 *)(region + 2 * col), *fl = (u16 *)(region + 3 * col);
    SYNC();                                                  // the previous block of this workgroup is done with the region
    if (tid == 0) sh[2] = 0;
    // (global loads four rounds at a time, the index clamped instead of a branch around the load: a loop of `if (i < n) load` waits for every load alone)
    for (u32 i0 = tid; i0 < n; i0 += 4 * THREADS) {
      u64 rw[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const u32 i = i0 + (u32)k * THREADS; rw[k] = a.res[o + (i < n ? i : n - 1)]; }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32 i = i0 + (u32)k * THREADS;
        if (i < n) { const u16 m = (u16)(rw[k] & 0xFFFFu); bst[i] = m; ptr[i] = m != NONE16 ? m : (u16)i; ft[i] = NONE16; fl[i] = 0; }
      }
    }
    SYNC();
    for (u32 i = tid; i < n; i += THREADS) {
      const u32 b = ld_shared<IN_LDS>(&bst[i]);
      if (b != NONE16 && b < n && ld_shared<IN_LDS>(&bst[b]) == NONE16) min_u16<IN_LDS>(ft, b, i);   // b inactive: i's turn may be its founding turn
    }
    u32 rounds = 1; while ((1u << rounds) < n) ++rounds;
    SYNC();
    for (u32 r = 0; r <= rounds; ++r) {                      // chains only run downwards: in-place jumping converges; they are short, so
      int moved = 0;                                         // the rounds end with the first one in which no pointer moved
      for (u32 i = tid; i < n; i += THREADS) {
        const u32 p = ld_shared<IN_LDS>(&ptr[i]);
        if (p < n) { const u32 pp = IN_LDS ? (u32)*(volatile u16 *)&ptr[p] : (u32)ld_shared<false>(&ptr[p]); if (pp != p) { ptr[i] = (u16)pp; moved = 1; } }
      }
      const int any = __syncthreads_or(moved);
      if (!IN_LDS) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (!any) break;
    }
    for (u32 i = tid; i < n; i += THREADS) { const u32 t = ld_shared<IN_LDS>(&ft[i]); if (t != NONE16) fl[t] = 1; }   // founding turns are distinct
    SYNC();
    // block-wide inclusive scan of fl[0..n)
    const u32 ipt = (n + THREADS - 1) / THREADS, s0 = tid * ipt < n ? tid * ipt : n, s1 = s0 + ipt < n ? s0 + ipt : n;
    u32 mine = 0;
    for (u32 i = s0; i < s1; ++i) mine += ld_shared<IN_LDS>(&fl[i]);
    u32 inc = mine;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) { const u32 o2 = (u32)__shfl_up((int)inc, d); if (lane >= d) inc += o2; }
    if (lane == WAVE - 1) sh[4 + wave] = inc;
    __syncthreads();
    u32 run = inc - mine;
    for (int q = 0; q < wave; ++q) run += sh[4 + q];
    for (u32 i = s0; i < s1; ++i) { run += ld_shared<IN_LDS>(&fl[i]); fl[i] = (u16)run; }
    SYNC();
    const u32 nRoots = ld_shared<IN_LDS>(&fl[n - 1]);
    u32 stop = n;
    h10x_clushash *ch = a.clusHash + o; const u16 *g = a.goodPos + o;
    if (nRoots > 255) {                                      // hash10x.c:810-816: abandon at the 256th founding turn
      for (u32 i = tid; i < n; i += THREADS)
        if (ld_shared<IN_LDS>(&fl[i]) == 256 && (i == 0 || ld_shared<IN_LDS>(&fl[i - 1]) == 255)) sh[1] = i;
      __syncthreads();
      stop = sh[1];                                            // the terms of the turns before it stay in pointToMin
      for (u32 i0 = tid; i0 < n; i0 += 4 * THREADS) {
        u32 gp[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const u32 i = i0 + (u32)k * THREADS; gp[k] = g[i < n ? i : n - 1]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i0 + (u32)k * THREADS < n) ch[gp[k]].subCluster = 0;
      }
      if (tid == 0) { a.blocks[c].nSubCluster = 0; a.raw[2 * (size_t)c] = 0x80000000u; a.raw[2 * (size_t)c + 1] = 0; }   // (nSubClustered is reset with the labels: hash10x.c:811)
    } else {
      u32 labelled = 0;
      for (u32 i0 = tid; i0 < n; i0 += 4 * THREADS) {
        u32 gp[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const u32 i = i0 + (u32)k * THREADS; gp[k] = g[i < n ? i : n - 1]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const u32 i = i0 + (u32)k * THREADS;
          if (i < n) {
            u32 L = 0;
            if (ld_shared<IN_LDS>(&bst[i]) != NONE16) { const u32 rt = ld_shared<IN_LDS>(&ptr[i]); const u32 t = rt < n ? (u32)ld_shared<IN_LDS>(&ft[rt]) : NONE16; L = t != NONE16 ? (u32)ld_shared<IN_LDS>(&fl[t]) : 0u; }
            else { const u32 t = ld_shared<IN_LDS>(&ft[i]); if (t != NONE16) L = ld_shared<IN_LDS>(&fl[t]); }
            ch[gp[k]].subCluster = (u8)L;                    // includes the wipe of hash10x.c:783
            labelled += L != 0;
          }
        }
      }
      for (int sft = 32; sft; sft >>= 1) labelled += (u32)__shfl_down((int)labelled, sft);
      if (lane == 0 && labelled) atomicAdd(&sh[2], labelled);
      __syncthreads();
      if (tid == 0) { a.blocks[c].nSubCluster = nRoots; a.raw[2 * (size_t)c] = nRoots; a.raw[2 * (size_t)c + 1] = sh[2]; }
    }
    // the ranks' pointToMin terms (hash10x.c:821): one IEEE double divide per active rank, written over its result word
    double *term = (double *)(a.res + o);
    for (u32 i0 = tid; i0 < n; i0 += 4 * THREADS) {
      u64 rw[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const u32 i = i0 + (u32)k * THREADS; rw[k] = a.res[o + (i < n ? i : n - 1)]; }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32 i = i0 + (u32)k * THREADS; const u64 r = rw[k];
        const bool has = i >= 1 && i < stop && (u32)(r & 0xFFFFu) != NONE16;
        if (i < n) term[i] = has ? (double)(int)(u32)((r >> 16) & 0xFFFFFFu) / (double)(int)(u32)(r >> 40) : 0.0;
      }
    }
  }
}
#undef SYNC
__host__ __device__ inline size_t replayBytes(u32 n) { return 4 * (((size_t)n * 2 + 15) & ~(size_t)15) + 16; }
constexpr u32 REPLAY_SMALL = 2040, REPLAY_MID = 16376;       // rank counts up to which a block's replay runs in 16 KB / 128 KB of LDS

// ---- (e) pointToMin = the terms added in rank order (hash10x.c:821): a serial fp64 chain per barcode. One WAVE per
// barcode: 64 terms per coalesced load, added in order through readlane (the chain costs one v_add_f64 latency per
// term; the chip holds 8192 waves, so the chains of thousands of barcodes run side by side). Blocks without good
// hashes are left untouched.
constexpr int SUM_THREADS = 256;
__global__ __launch_bounds__(SUM_THREADS)
void point_sum_kernel(h10x_block *__restrict__ blocks, const u64 *__restrict__ blockOff, const u32 *__restrict__ nGood,
                      const double *__restrict__ term, u32 codeMin, u32 codeMax) {
  const int lane = threadIdx.x & (WAVE - 1);
  const u32 c = codeMin + blockIdx.x * (SUM_THREADS / WAVE) + threadIdx.x / WAVE;
  if (c >= codeMax) return;
  const u32 n = nGood[c];
  if (!n) return;
  const double *t = term + blockOff[c];
  double p = 0.0;
  double nxt = (u32)lane < n ? t[lane] : 0.0;                // term[0] is 0.0 (rank 0 is never processed)
  for (u32 base = 0; base < n; base += WAVE) {
    const double cur = nxt;
    nxt = base + WAVE + lane < n ? t[base + WAVE + lane] : 0.0;
    const int tlo = (int)(u32)__double_as_longlong(cur), thi = (int)(u32)((u64)__double_as_longlong(cur) >> 32);
    if (__ballot(cur != 0.0)) {                               // + 0.0 is exact (p >= +0): skip all-zero groups
#pragma unroll
      for (int j = 0; j < WAVE; ++j) {
        const u64 bits = ((u64)(u32)__builtin_amdgcn_readlane(thi, j) << 32) | (u64)(u32)__builtin_amdgcn_readlane(tlo, j);
        p += __longlong_as_double((long long)bits);
      }
    }
  }
  if (lane == 0) blocks[c].pointToMin = p;
}

// ---- (f) codeClusterReadMerge (hash10x.c:837-868): components of labels that share a read, renumbered by ascending
// minimum label. One small workgroup per barcode; BIG = blocks with more than MERGE_SMALL_READS read pairs.
constexpr int MERGE_THREADS = 256;
constexpr u32 MERGE_SMALL_READS = 4096;
template <bool BIG>
__global__ __launch_bounds__(MERGE_THREADS)
void read_merge_kernel(h10x_block *__restrict__ blocks, const u64 *__restrict__ blockOff, const u32 *__restrict__ nGood,
                       h10x_clushash *__restrict__ clusHash, u32 codeMin) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ u32 changed;
  const u32 c = codeMin + blockIdx.x;
  if (!nGood[c]) return;
  const h10x_block blk = blocks[c];
  const u32 nSub = blk.nSubCluster, nHash = blk.nHash, nRead = blk.nRead;
  if (!nSub) return;                                         // hash10x.c:840
  if ((nRead > MERGE_SMALL_READS) != BIG) return;
  const int tid = threadIdx.x;
  const u32 nRep = nRead < 65536u ? nRead : 65536u;
  h10x_clushash *ch = clusHash + blockOff[c];
  u8 *readRep = (u8 *)smem;
  u32 *adj = (u32 *)(smem + ((nRep + 15) & ~15u));
  u32 *comp = adj + 256 * 8, *newLab = comp + 256;
  // the block's entries are looked at three times: the first MERGE_KEEP per thread stay in registers (a block of up to
  // 2048 hashes is then read from HBM once — each pass was a round trip of its own in front of a barrier)
  constexpr int MERGE_KEEP = 8;
  h10x_clushash keep[MERGE_KEEP];
#pragma unroll
  for (int j = 0; j < MERGE_KEEP; ++j) { const u32 p = tid + j * MERGE_THREADS; keep[j] = ch[p < nHash ? p : nHash - 1]; }   // (clamped, not predicated: the loads are in flight together; nSub != 0 => nHash >= 1)
#define H10X_FOR_ENTRIES(...)                                                                                          \
  { _Pragma("unroll") for (int j = 0; j < MERGE_KEEP; ++j) { const u32 p = tid + j * MERGE_THREADS; if (p < nHash) { const h10x_clushash e = keep[j]; __VA_ARGS__ } } \
    for (u32 p = tid + MERGE_KEEP * MERGE_THREADS; p < nHash; p += MERGE_THREADS) { const h10x_clushash e = ch[p]; __VA_ARGS__ } }
  for (u32 i = tid; i < (nRep + 3) / 4; i += MERGE_THREADS) ((u32 *)readRep)[i] = 0;
  for (u32 i = tid; i < 256 * 8; i += MERGE_THREADS) adj[i] = 0;
  comp[tid] = tid;
  __syncthreads();
  H10X_FOR_ENTRIES(                                          // any one label of each read is its representative
    if (e.subCluster && e.subCluster <= nSub && e.read < nRep) readRep[e.read] = e.subCluster; )
  __syncthreads();
  H10X_FOR_ENTRIES(
    if (e.subCluster && e.subCluster <= nSub && e.read < nRep) {
      const u32 L = e.subCluster, R = readRep[e.read];
      if (R != L) { atomicOr(&adj[L * 8 + (R >> 5)], 1u << (R & 31)); atomicOr(&adj[R * 8 + (L >> 5)], 1u << (L & 31)); }
    } )
  __syncthreads();
  for (int iter = 0; iter < 256; ++iter) {                   // min-label propagation; <= 255 rounds
    if (tid == 0) changed = 0;
    __syncthreads();
    if (tid >= 1 && tid <= (int)nSub) {
      const u32 mine = comp[tid]; u32 m = mine;
      for (int wd = 0; wd < 8; ++wd) { u32 bits = adj[tid * 8 + wd]; while (bits) { const int b = __ffs((int)bits) - 1; bits &= bits - 1; const u32 cR = *(volatile u32 *)&comp[wd * 32 + b]; m = cR < m ? cR : m; } }
      if (m < mine) { atomicMin(&comp[tid], m); changed = 1; }
    }
    __syncthreads();
    if (!changed) break;
    __syncthreads();
  }
  // renumber components by ascending minimum label: rank of L among the labels that are their component's minimum
  {
    const bool isMin = tid >= 1 && tid <= (int)nSub && comp[tid] == (u32)tid;
    const u64 bal = __ballot(isMin);
    const int lane = tid & (WAVE - 1), wv = tid / WAVE;
    if (lane == 0) newLab[256 + 256 - 4 + wv] = (u32)__popcll(bal);          // scratch slots above the label map (labels <= 255)
    __syncthreads();
    u32 before = 0;
    for (int q = 0; q < wv; ++q) before += newLab[256 + 256 - 4 + q];
    newLab[tid] = before + (u32)__popcll(bal & ((2ULL << lane) - 1));   // inclusive count: rank of L if it is a minimum
    __syncthreads();
    if (tid == 0) blocks[c].nSubCluster = newLab[nSub];
    u32 mapped = 0;
    if (tid >= 1 && tid <= (int)nSub) mapped = newLab[comp[tid]];             // label -> rank of its minimum
    __syncthreads();
    newLab[256 + tid] = mapped;
    __syncthreads();
  }
  H10X_FOR_ENTRIES(
    const u32 L = e.subCluster;
    if (L && L <= nSub) ch[p].subCluster = (u8)newLab[256 + L]; )
#undef H10X_FOR_ENTRIES
}

// launch classes by working-set size: 0 = half a CU's LDS (barcodes with many ranks run their list loop on fewer waves:
// histWaves), 2 = the whole LDS of a CU, 3 = HBM scratch (class 1 is no longer used)
__global__ void cluster_classify_kernel(const h10x_block *__restrict__ blocks, const u32 *__restrict__ nGood, const u32 *__restrict__ entries, u32 codeMin, u32 codeMax,
                                        u32 nBlocks /* LDS entries of first[] (ranked, hashed: barcodes of the data set) */, int ranked /* 1 ranked, 3 translated */, u32 hashMinSlots, u32 maxTrRanks, u32 firstCap, u32 bmWords, u32 waves0, size_t budget0, size_t budgetSmall, size_t budgetBig, u32 bigRanks, u32 estDiv /* translated placement: a block's barcodes are taken to be this fraction of its list entries */,
                                        u32 *__restrict__ list0, u32 *__restrict__ list1, u32 *__restrict__ list2, u32 *__restrict__ list3,
                                        u32 *__restrict__ listBig /* blocks with more ranks than the small replay class holds: counts[9] */, u32 *__restrict__ counts, unsigned long long *__restrict__ work) {
  const u32 c = codeMin + blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & (WAVE - 1);
  const u32 n = c < codeMax ? nGood[c] : 0;
  int cls = -1; u32 nRead = 0;
  if (n) {
    nRead = blocks[c].nRead;
    if (ranked == 3) {                             // translated placement: the table of pass A must hold the barcodes expected in the block's lists
      u32 S, S2, nW; bool compact;
      translatedShape(n, waves0, budget0, hashMinSlots, entries[c], nBlocks, firstCap, S, S2, nW, compact);
      if (n > maxTrRanks) cls = 3;                       // (its handles would not fit the workgroup's HBM slot)
      // (a sixth of the entries, not a seventh as in the ranked placement: the blocks in between would run in the half-CU class on a second table and
      //  with few list-loop waves — they are faster with a CU to themselves: full configs[2] 583 -> 567 ms, 17 % of the blocks in the whole-CU class
      //  instead of 8 %; at a fifth the whole-CU class holds more than half of the CUs and the launches no longer overlap: 890 ms)
      else if (translatedFits(S, S2, rankedFirstEstimateE(nBlocks, n, entries[c], estDiv), entries[c], nBlocks)) cls = 0;
      else { translatedShape(n, CL_THREADS_HUGE / WAVE, budgetBig, hashMinSlots, entries[c], nBlocks, firstCap, S, S2, nW, compact); cls = S ? 2 : 3; }
    }
    else if (histWaves(ranked ? rankedFirstEstimateE(nBlocks, n, entries[c]) : nBlocks, n, waves0, bmWords, budget0)) cls = 0;
    else if (histWaves(ranked ? rankedFirstEstimate(nBlocks, n) : nBlocks, n, CL_THREADS_HUGE / WAVE, bmWords, budgetBig)) cls = 2;
    else cls = 3;
  }
  // the largest barcodes of the main class go to the front of its work queue (list1, handed out before list0): the launch
  // then does not end on a workgroup that drew a big one last. (Ordering the WHOLE queue by size was measured 17 % slower:
  // like-sized workgroups run their phases in step.)
  if (cls == 0 && n > bigRanks) cls = 1;
  for (int s = 32; s; s >>= 1) nRead = max(nRead, (u32)__shfl_xor((int)nRead, s));
  if (lane == 0 && nRead) atomicMax(&counts[8], nRead);
  {                                                          // most ranks of a block per LDS class: the translated placement sizes its handle slots by them
    u32 n0 = (cls == 0 || cls == 1) ? n : 0u, n2 = cls == 2 ? n : 0u;
    for (int s = 32; s; s >>= 1) { n0 = max(n0, (u32)__shfl_xor((int)n0, s)); n2 = max(n2, (u32)__shfl_xor((int)n2, s)); }
    if (lane == 0) { if (n0) atomicMax(&counts[12], n0); if (n2) atomicMax(&counts[13], n2); }
  }
  if (work) {                                            // work of the half-CU classes and of the whole-CU class (list entries + a charge per rank): the launches split the CUs by it
    unsigned long long w0 = (cls == 0 || cls == 1) ? (unsigned long long)entries[c] + 64ull * n : 0ull, w2 = cls == 2 ? (unsigned long long)entries[c] + 64ull * n : 0ull;
    for (int s = 32; s; s >>= 1) { w0 += __shfl_xor(w0, s); w2 += __shfl_xor(w2, s); }
    if (lane == 0) { if (w0) atomicAdd(&work[0], w0); if (w2) atomicAdd(&work[1], w2); }
  }
  {
    const u64 bal = __ballot(n > REPLAY_SMALL);
    if (bal) {
      u32 base = 0;
      if (lane == 0) base = atomicAdd(&counts[9], (u32)__popcll(bal));
      base = (u32)__shfl((int)base, 0);
      if (n > REPLAY_SMALL) listBig[base + (u32)__popcll(bal & ((1ULL << lane) - 1))] = c;
    }
  }
  u32 *const lists[4] = {list0, list1, list2, list3};
#pragma unroll
  for (int k = 0; k < 4; ++k) {                              // one atomic per wave and class
    const u64 bal = __ballot(cls == k);
    if (!bal) continue;
    u32 base = 0;
    if (lane == 0) base = atomicAdd(&counts[k], (u32)__popcll(bal));
    base = (u32)__shfl((int)base, 0);
    if (cls == k) lists[k][base + (u32)__popcll(bal & ((1ULL << lane) - 1))] = c;
  }
}

// one contiguous range of LOCAL block numbers; `more` = a further range of the same command has run before (counters add up)
static int cluster_local_range(Ctx *c, int codeMin, int codeMax, int threshold, bool more) {
  hipStream_t st = c->stream;
  const u32 nGlobal = c->sharded ? c->nBlocksGlobal : c->nBlocks;
  c->tstart(T_CLUSTER);
  const u32 span = (u32)(codeMax - codeMin);
  DevBuf<u32> list0, list1, list2, list3, listBig; DevBuf<u64> zeroed; DevBuf<u64> term;
  H10X_HIP(c, list0.alloc(span)); H10X_HIP(c, list1.alloc(span)); H10X_HIP(c, list2.alloc(span)); H10X_HIP(c, list3.alloc(span)); H10X_HIP(c, listBig.alloc(span));
  // every small counter of the command in one buffer, cleared by one memset: counts[0..3] class sizes, [4] [6] [7] work
  // queue positions, [8] largest nRead, [10] [11] overflowed blocks (lists A, B); stats[0..7] the work counters
  H10X_HIP(c, zeroed.alloc(8 + 8 + 2)); H10X_HIP(c, term.alloc(c->nEntries));   // (+ work of the half-CU and the whole-CU classes)
  H10X_HIP(c, hipMemsetAsync(zeroed.p, 0, (8 + 8 + 2) * 8, st));
  struct { u32 *p; } counts{(u32 *)zeroed.p}; struct { u64 *p; } stats{zeroed.p + 8};   // counts[12] [13]: most ranks of a block in the half-CU classes / the whole-CU class (sizes the handle slots)
  const size_t budgetSmall = c->optClusterLds > 0 ? (size_t)c->optClusterLds : 80 * 1024 - 1024;
  const size_t budgetBig = c->optClusterLds > 0 ? (size_t)c->optClusterLds : 160 * 1024 - 1024;
  const int threads0 = c->optClusterThreads0 == 512 ? 512 : (c->optClusterThreads0 == 768 ? 768 : 1024);   // tuning knobs for class 0
  const size_t budget0 = c->optClusterLds > 0 ? (size_t)c->optClusterLds : (c->optClusterBudget0 > 0 ? (size_t)c->optClusterBudget0 : budgetSmall);
  // Placement of first[] (FirstDense / FirstRanked / SlotTable + FirstSlots): dense in LDS while 2 B per barcode of the data set is
  // small; ranked in LDS while the presence bitmap + prefix (3/16 B per barcode) leaves room for the rest; translated (handles
  // into a table in LDS) up to 2^22 barcodes (beyond that the 10-bit tag needs more slots than LDS has); dense on a per-workgroup
  // HBM slot (L2/MALL resident, atomics + L1-bypassing loads) as the last resort and for blocks whose LDS tables filled up twice.
  const u32 bmWordsAll = (nGlobal + 31) / 32;
  int hashBits = 1; while (hashBits < 32 && (1ull << hashBits) < (unsigned long long)nGlobal) ++hashBits;
  const u32 hashMinSlots = hashBits > 10 ? 4u << (hashBits - 10) : 0u;       // 10-bit tag: buckets >= 2^(b-10)
  // The ranked form reads the lists twice but needs no probing (100 k-barcode set: 39 ms against 56 translated), so it is TRIED wherever the
  // bitmap + prefix leave the whole-CU class any room (up to 100 KB of them: 546 k barcodes) and kept unless the classification then
  // sends more than a few blocks to the HBM-scratch class; up to 48 KB (262 k barcodes) it is taken as before.
  const bool rankedSure = (size_t)bmWordsAll * 6 <= 48 * 1024, rankedTry = !rankedSure && (size_t)bmWordsAll * 6 <= 100 * 1024 && hashBits <= 22;
  int firstMode = (size_t)nGlobal * 2 <= 48 * 1024 ? 0 : (rankedSure || rankedTry ? 1 : (hashBits <= 22 ? 4 : 2));
  if (c->optFirstGlobal == 1) firstMode = 2; else if (c->optFirstGlobal == 2) firstMode = 1;                                          // test knobs
  else if ((c->optFirstGlobal == 3 || c->optFirstGlobal == 4) && hashBits <= 22) firstMode = 4;                                        // (3: round 3's one-pass hashed table, replaced by the translated placement)
  // translated placement: u16 per list on a workgroup's handle slot (a multiple of 64 that holds the longest list), ranks per slot
  u32 hStride = 64; while (hStride < c->maxGoodDepth && hStride < (1u << 24)) hStride <<= 1;   // (a power of two: a handle's position tells its rank)
  const size_t trSlotCapBytes = (size_t)32 << 20;
  const u32 maxTrRanks = (u32)hmin<size_t>(0xFFFFFFFFu, trSlotCapBytes / ((size_t)hStride * 2));
  const bool packed = c->optTrPacked != 0;                   // translated placement: the packed form (several lists per wave instruction, round 5) unless the knob says 0
  // a workgroup's handle slot: the handles of its largest block (ranks x the slot's list stride; + 256: the packed form's class regions start at multiples of 64), then the waves' queues.
  // Sized per launch class by the largest block THAT class holds and by its waves (round 4 took the context's largest block and 16 waves for every class: tens of GB on sets
  // with a few giant barcodes that never run here); blocks handed on by the half-CU class run in the whole-CU class, whose slots therefore cover both
  auto trSlotFor = [&](u32 maxRanks, u32 waves) { return (size_t)hmin<u32>(hmax<u32>(maxRanks, 1u), maxTrRanks) * hStride + 256 + (size_t)waves * TR_QUEUE * 4; };
  const u32 firstCap = c->optFirstCap > 0 ? (u32)c->optFirstCap : 0u;      // test knob only
  u32 hc[14]; u64 hw[2] = {0, 0}; u32 nFirstLds = 0, bmWords = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    nFirstLds = firstMode == 0 ? nGlobal : 0;
    bmWords = firstMode == 1 ? bmWordsAll : 0;
    cluster_classify_kernel<<<divUp(span, 256), 256, 0, st>>>(c->blocks.p, c->nGood.p, c->goodEntries.p, (u32)codeMin, (u32)codeMax,
                                                            firstMode == 1 || firstMode >= 3 ? nGlobal : nFirstLds, firstMode == 1 ? 1 : (firstMode == 4 ? 3 : 0), hashMinSlots, maxTrRanks, firstCap,
                                                            bmWords, (u32)threads0 / WAVE, budget0, budgetSmall, budgetBig, c->optBigRanks > 0 ? (u32)c->optBigRanks : c->meanGood + c->meanGood / 2, c->optTrEstDiv > 0 ? (u32)c->optTrEstDiv : 6u,
                                                            list0.p, list1.p, list2.p, list3.p, listBig.p, counts.p, (unsigned long long *)(zeroed.p + 16));
    H10X_TRY(c->readback(hc, counts.p, 56));
    H10X_TRY(c->readback(hw, zeroed.p + 16, 16));
    H10X_TRY(c->syncReadbacks());
    const u32 classified = hc[0] + hc[1] + hc[2] + hc[3];
    if (attempt || !rankedTry || c->optFirstGlobal || firstMode != 1 || hc[3] <= 16 + classified / 200) break;
    firstMode = 4;                                           // too many blocks without room beside the bitmap: the translated placement after all
    H10X_HIP(c, hipMemsetAsync(counts.p, 0, 16, st));        // the four class sizes
    H10X_HIP(c, hipMemsetAsync(counts.p + 9, 0, 4, st));
    H10X_HIP(c, hipMemsetAsync(zeroed.p + 16, 0, 16, st));
    H10X_HIP(c, hipMemsetAsync(counts.p + 12, 0, 8, st));
  }
  // (the work queue hands barcodes out in the order the classification appended them, i.e. mixed sizes: sorting the
  // queue by descending rank count was measured 17 % SLOWER — workgroups of like size run their phases in step and
  // contend for the same unit at the same time)
  ClusterArgs a{};
  a.blocks = c->blocks.p; a.blockOff = c->blockOff.p; a.clusHash = c->clusHash.p; a.goodPos = c->goodPos.p; a.nGood = c->nGood.p;
  a.goodRow = c->goodRow.p; a.rows = c->rows.p; a.nBlocks = c->nBlocks; a.threshold = threshold;
  a.segs = c->segs; a.nBlocksFirst = nGlobal; a.rowShift = (u32)c->rowShift;
  if (c->sharded && c->optRowsFakeBase) a.rows = c->rows.p - (size_t)c->optRowsFakeBase;   // test knob: rowStart[] carries the same offset (shard_exchangeRows)
  a.narrowFirst = (u32)c->optNarrowFirst; a.tpClassT = c->optTrClassT != 0 ? 1u : 0u;
  a.maxGood = c->maxGood; a.stats = stats.p; a.res = term.p; a.entries = c->goodEntries.p;
  a.firstCap = firstCap; a.hashMinSlots = hashMinSlots;
  a.hashBits = (u32)hashBits; a.hStride = hStride;
  const size_t trStride[3] = {trSlotFor(hc[12], (u32)threads0 / WAVE), trSlotFor(hmax<u32>(hc[12], hc[13]), CL_THREADS_HUGE / WAVE), trSlotFor(hmax<u32>(hc[12], hc[13]), CL_THREADS_HUGE / WAVE)};   // [1]: the whole-CU class's second launch
  // ranked / hashed placement: blocks whose table was too small are re-run — those of the half-CU class (list A) with the
  // whole LDS of a CU, those that fail there as well (list B) with first[] dense on an HBM slot
  DevBuf<u32> ovfA, ovfB; H10X_HIP(c, ovfA.alloc(span)); H10X_HIP(c, ovfB.alloc(span));
  u32 *const ovfCountA = counts.p + 10, *const ovfCountB = counts.p + 11;
  DevBuf<u64> phase;
  if (c->optStamps) { H10X_HIP(c, phase.alloc(8)); H10X_HIP(c, hipMemsetAsync(phase.p, 0, 64, st)); a.phase = phase.p; }
  // HBM working set per workgroup (class 3): first[] + per-rank arrays for the largest barcode
  DevBuf<unsigned char> scratch;
  size_t stride = 0; u32 grid3 = 0;
  if (hc[3]) {
    stride = (workBytes(nGlobal, c->maxGood, CL_THREADS_SMALL / WAVE, 0) + 255) & ~(size_t)255;
    grid3 = hmin<u32>(hc[3], (u32)c->numCU);
    H10X_HIP(c, scratch.alloc(stride * grid3));
    H10X_HIP(c, hipMemsetAsync(scratch.p, 0xFF, stride * grid3, st));    // first[] = unseen everywhere
  }
  // hybrid placement: one first[] slot per resident workgroup of each LDS class
  DevBuf<unsigned char> firstSlots[3];
  const size_t firstStride = (((size_t)nGlobal * 2 + 255) & ~(size_t)255);
  u32 gridOf[3] = {hmin<u32>(hc[0] + hc[1], (u32)c->numCU * (u32)hmax<size_t>(1, (160 * 1024) / (budget0 + 1024))), hmin<u32>(hc[1], (u32)c->numCU * 2), hmin<u32>(hc[2], (u32)c->numCU)};   // class 1 unused
  // Both LDS classes are persistent launches that stay on the CUs they get: a whole-CU workgroup keeps two half-CU workgroups out. Where both have
  // enough blocks to fill the chip, the CUs are split by the classes' work (the whole-CU class does a list entry at ~0.8 x the rate per CU: measured
  // on the million-barcode set, where it used to finish 100 ms behind the main launch), so that the two launches end together.
  u32 secondWind = 0;
  if (firstMode == 4 && gridOf[2] && gridOf[0] >= (u32)c->numCU && hw[0] + hw[1]) {
    const double share2 = 1.25 * (double)hw[1] / (1.25 * (double)hw[1] + (double)hw[0]);
    // (+ 4 CUs: the blocks of that class are the ones with many ranks, which the work figure flatters; its CUs are not lost when it ends early —
    // while it is the smaller side, the main launch brings two workgroups for EVERY CU, and those that find no room wait in the dispatcher
    // for the whole-CU workgroups to leave: 3 Gb set, 2 169 such blocks: 657 ms on one CU behind a main launch of 510 ms before this)
    // Whichever side the split errs on corrects itself: the main launch always brings two workgroups for every CU (they take over the whole-CU class's
    // CUs when that ends first), and behind the main launch, on its stream, the whole-CU class is launched a SECOND time on the same work queue
    // (secondWind): workgroups for every CU, which find the queue empty — or the CUs the main launch has just left free. (Before this the main launch
    // was held to the CUs the split gave it once the whole-CU class had more than half of them: 890 ms instead of 570 on a set classified that way.)
    // (round 6: never fewer than an eighth of the CUs, 32 of 256, while the class has the blocks: where it is a side show — 2 169 blocks of the 3 Gb set, ~130 of a rank's
    // eighth of it — five CUs made it a 2 ms tail whenever the two launches do not run side by side, e.g. when their streams share a hardware queue (seen with eight ranks'
    // forty streams in one process: the rank's --cluster 19.5 instead of 17.4 ms); with 32 CUs it is done in 0.3 ms, and the main launch's workgroups take the CUs over)
    const u32 cu2 = hmin<u32>((u32)c->numCU > 16 ? (u32)c->numCU - 8 : (u32)c->numCU - 1, hmax<u32>(hmax<u32>(1u, (u32)c->numCU / 8u), (u32)(share2 * c->numCU + 0.5) + 4u));
    gridOf[2] = hmin<u32>(hc[2], cu2);
    gridOf[0] = hmin<u32>(gridOf[0], 2 * (u32)c->numCU);
    secondWind = hc[2] > gridOf[2] ? hmin<u32>(hc[2] - gridOf[2], (u32)c->numCU) : 0u;
  }
  DevBuf<u16> trSlots[3];                                  // translated placement: one handle slot per resident workgroup of each LDS class
  // (no room for a class's slots: fewer workgroups — down to one — before giving up; the launches are persistent, a smaller grid is only slower)
  auto trAlloc = [&](int k, u32 &grid) -> int {
    for (;;) {
      if (trSlots[k].alloc(trStride[k] * grid) == hipSuccess) return 0;
      (void)hipGetLastError();
      if (grid <= 1) return c->fail("no device memory for the handle slots of the cluster kernel (%.1f MB per workgroup)", trStride[k] * 2 / 1e6);
      grid = (grid + 1) / 2;
    }
  };
  if (firstMode == 4) for (int k = 0; k < 3; k += 2) if (gridOf[k]) H10X_TRY(trAlloc(k, gridOf[k]));
  if (secondWind) H10X_TRY(trAlloc(1, secondWind));
  if (firstMode == 2) for (int k = 0; k < 3; k += 2) if (gridOf[k]) {       // class 0 serves list 0 AND the front list (hc[1]); class 1 has no launch
    H10X_HIP(c, firstSlots[k].alloc(firstStride * gridOf[k]));
    H10X_HIP(c, hipMemsetAsync(firstSlots[k].p, 0xFF, firstStride * gridOf[k], st));
  }
  // The classes are independent: fork them onto side streams so that the few largest barcodes (long, low
  // parallelism) run beside the many small ones instead of in front of them. Every buffer they touch was
  // allocated before the fork and is released after the join, which is what the block cache requires.
  // a failure between here and the join must not let the buffers above go back to the block cache while a side stream
  // still runs a kernel on them (the cache orders reuse within ONE stream only)
  ForkGuard forkGuard(c);
  c->tstart(T_CLUSTER_K);
  H10X_TRY(c->forkStreams(3));
  if (hc[3]) {
    ClusterArgs g = a; g.list = list3.p; g.nList = hc[3]; g.workCounter = counts.p + 7; g.scratch = scratch.p; g.scratchStride = stride;
    cluster_kernel<false, 2, CL_THREADS_SMALL><<<grid3, CL_THREADS_SMALL, 0, c->aux[0]>>>(g);
  }
#define H10X_LAUNCH_ONE(MODE, K, THREADS, BUDGET, GRID, STREAM)                                                                    \
      { H10X_HIP(c, hipFuncSetAttribute((const void *)cluster_kernel<true, MODE, THREADS, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BUDGET))); \
        cluster_kernel<true, MODE, THREADS, K><<<GRID, THREADS, BUDGET, STREAM>>>(g); }
#define H10X_LAUNCH_LDS(K, THREADS, BUDGET, STREAM, LIST, NLIST, GRID, CNT, OVFCNT, OVF)                                           \
  {                                                                                                                                \
    ClusterArgs g = a; g.list = LIST; g.nList = NLIST; g.workCounter = counts.p + CNT; g.ldsBudget = (u32)(BUDGET);                \
    g.overflow = (OVF); g.overflowCount = (OVFCNT);                                                                                \
    if (firstMode == 0) H10X_LAUNCH_ONE(0, K, THREADS, BUDGET, GRID, STREAM)                                                       \
    else if (firstMode == 1) H10X_LAUNCH_ONE(1, K, THREADS, BUDGET, GRID, STREAM)                                                  \
    else if (firstMode == 4) { g.handles = trSlots[K].p; g.handleStride = trStride[K]; if (packed) H10X_LAUNCH_ONE(5, K, THREADS, BUDGET, GRID, STREAM) else H10X_LAUNCH_ONE(4, K, THREADS, BUDGET, GRID, STREAM) } \
    else { g.scratch = firstSlots[K].p; g.scratchStride = firstStride; H10X_LAUNCH_ONE(2, K, THREADS, BUDGET, GRID, STREAM) }      \
  }
  if (hc[2]) {
    // The whole-CU class must be resident BEFORE the main launch: its workgroups need an empty CU, and once the main
    // launch's persistent workgroups sit on every CU they only get one when those retire — the two launches then run one
    // after the other instead of side by side (seen under rocprofv3: +0.15 ms). Its workgroups report in through pinned
    // host words; the host holds the main launch back until they have (some 10 us), with a time limit as a safeguard.
    if (!c->startFlags) H10X_HIP(c, hipHostMalloc((void **)&c->startFlags, 1024 * sizeof(u32), hipHostMallocDefault));
    memset(c->startFlags, 0, 1024 * sizeof(u32));
    // (only while that class is a side show: when it holds most of the barcodes its persistent workgroups would keep every
    // CU to themselves until their queue is empty, and the two launches share the chip better by racing for the CUs —
    // 1 M-barcode set: 2.20 s against 2.52 s)
    a.started = (gridOf[2] * 2 <= (u32)c->numCU || secondWind) && gridOf[2] <= 1024 ? c->startFlags : nullptr;
    H10X_LAUNCH_LDS(2, CL_THREADS_HUGE, budgetBig, c->aux[1], list2.p, hc[2], gridOf[2], 6, ovfCountB, ovfB.p)
    if (a.started) {
      const auto t0 = std::chrono::steady_clock::now();
      for (;;) {
        u32 up = 0;
        for (u32 i = 0; i < gridOf[2]; ++i) up += __atomic_load_n(&c->startFlags[i], __ATOMIC_RELAXED) ? 1u : 0u;
        if (up == gridOf[2] || std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(500)) break;
      }
    }
    a.started = nullptr;
  }
  // the main launch (class 0) counts its own work and has its own hipEvent bracket on its stream: that is the launch the
  // roofline figure of bench.py is quoted for, and what a rocprofv3 kernel trace reports as cluster_kernel<true, *, 1024, 0>
  if (hc[0] || hc[1]) {
    a.stats = stats.p + 4; a.front = list1.p; a.nFront = hc[1];
    c->tstart(T_CLUSTER_MAIN);
    if (threads0 == 512) H10X_LAUNCH_LDS(0, 512, budget0, st, list0.p, hc[0], gridOf[0], 4, ovfCountA, ovfA.p)
    else if (threads0 == 768) H10X_LAUNCH_LDS(0, 768, budget0, st, list0.p, hc[0], gridOf[0], 4, ovfCountA, ovfA.p)
    else H10X_LAUNCH_LDS(0, CL_THREADS_SMALL, budget0, st, list0.p, hc[0], gridOf[0], 4, ovfCountA, ovfA.p)
    c->tstop(T_CLUSTER_MAIN);
    a.stats = stats.p; a.front = nullptr; a.nFront = 0;
  }
  if (secondWind) {                                          // (see the split above) same work queue, counters and overflow list as the first launch of the class; handle slots of its own
    ClusterArgs g = a; g.list = list2.p; g.nList = hc[2]; g.workCounter = counts.p + 6; g.ldsBudget = (u32)budgetBig;
    g.overflow = ovfB.p; g.overflowCount = ovfCountB; g.handles = trSlots[1].p; g.handleStride = trStride[1];
    if (packed) H10X_LAUNCH_ONE(5, 2, CL_THREADS_HUGE, budgetBig, secondWind, st) else H10X_LAUNCH_ONE(4, 2, CL_THREADS_HUGE, budgetBig, secondWind, st)
  }
  H10X_HIP(c, hipGetLastError());
  H10X_TRY(c->faultAt(4));
  H10X_TRY(c->joinStreams(3));
  u32 nOverflow = 0;
  DevBuf<unsigned char> scratch2;
  if (firstMode == 1 || firstMode == 4) {
    u32 nA = 0, nB = 0;
    H10X_HIP(c, hipMemcpyAsync(&nA, ovfCountA, 4, hipMemcpyDeviceToHost, st));
    H10X_HIP(c, hipStreamSynchronize(st));
    if (nA) {                                                // half-CU tables that were too small: again with the whole LDS of a CU
      H10X_HIP(c, hipMemsetAsync(counts.p + 6, 0, 4, st));
      u32 gridA = hmin<u32>(nA, (u32)c->numCU);
      if (firstMode == 4 && trSlots[2].n < trStride[2] * gridA) H10X_TRY(trAlloc(2, gridA));
      H10X_LAUNCH_LDS(2, CL_THREADS_HUGE, budgetBig, st, ovfA.p, nA, gridA, 6, ovfCountB, ovfB.p)
      H10X_HIP(c, hipGetLastError());
    }
    H10X_HIP(c, hipMemcpyAsync(&nB, ovfCountB, 4, hipMemcpyDeviceToHost, st));
    H10X_HIP(c, hipStreamSynchronize(st));
    nOverflow = nA + nB;
    if (nB) {
      // last resort: first[] dense on a per-workgroup HBM slot and everything else in LDS (the hybrid form); a block that
      // got this far fits the whole-CU budget without its table
      const u32 grid = hmin<u32>(nB, (u32)c->numCU);
      H10X_HIP(c, scratch2.alloc(firstStride * grid));
      H10X_HIP(c, hipMemsetAsync(scratch2.p, 0xFF, firstStride * grid, st));
      H10X_HIP(c, hipMemsetAsync(counts.p + 7, 0, 4, st));
      ClusterArgs g = a; g.list = ovfB.p; g.nList = nB; g.workCounter = counts.p + 7; g.scratch = scratch2.p; g.scratchStride = firstStride;
      g.ldsBudget = (u32)budgetBig;
      H10X_HIP(c, hipFuncSetAttribute((const void *)cluster_kernel<true, 2, CL_THREADS_HUGE, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)budgetBig));
      cluster_kernel<true, 2, CL_THREADS_HUGE, 3><<<grid, CL_THREADS_HUGE, budgetBig, st>>>(g);
      H10X_HIP(c, hipGetLastError());
    }
  }
#undef H10X_LAUNCH_LDS
#undef H10X_LAUNCH_ONE
  // (c) labels, cluster counts and the > 255 clusters cut from the msBest column: one workgroup per barcode, classes by rank count
  DevBuf<unsigned char> replayScratch;
  {
    if (c->clusterRaw.n != 2 * (size_t)c->nBlocks) { H10X_HIP(c, c->clusterRaw.alloc(2 * (size_t)c->nBlocks)); H10X_HIP(c, hipMemsetAsync(c->clusterRaw.p, 0, 8 * (size_t)c->nBlocks, st)); }
    H10X_HIP(c, hipMemsetAsync(c->clusterRaw.p + 2 * (size_t)codeMin, 0, 8 * (size_t)span, st));   // blocks without good hashes say nothing
    ReplayArgs ra{}; ra.raw = c->clusterRaw.p; ra.blocks = c->blocks.p; ra.blockOff = c->blockOff.p; ra.clusHash = c->clusHash.p; ra.goodPos = c->goodPos.p; ra.nGood = c->nGood.p;
    ra.res = term.p; ra.codeMin = (u32)codeMin; ra.span = span;
    // the barcodes with more ranks than the common class holds (hc[9] of them, listed by the classification): workgroups for those only,
    // on a side stream beside the common class (disjoint blocks)
    if (hc[9]) {
      H10X_TRY(c->forkStreams(1));
      ReplayArgs rb = ra; rb.list = listBig.p; rb.span = hc[9];
      const size_t lds = replayBytes(hmin<u32>(REPLAY_MID, c->maxGood));
      H10X_HIP(c, hipFuncSetAttribute((const void *)replay_kernel<true, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      rb.nLo = REPLAY_SMALL; rb.nHi = REPLAY_MID;
      replay_kernel<true, 1024><<<hc[9], 1024, lds, c->aux[0]>>>(rb);
    }
    ra.nLo = 0; ra.nHi = REPLAY_SMALL;
    replay_kernel<true, 256><<<span, 256, replayBytes(hmin<u32>(REPLAY_SMALL, hmax<u32>(c->maxGood, 1))), st>>>(ra);
    ra.list = listBig.p; ra.span = hc[9];
    if (hc[9]) H10X_TRY(c->joinStreams(1));
    if (hc[9] && c->maxGood > REPLAY_MID) {
      const u32 grid = hmin<u32>(hc[9], (u32)c->numCU);
      ra.scratchStride = (replayBytes(c->maxGood) + 255) & ~(size_t)255;
      H10X_HIP(c, replayScratch.alloc(ra.scratchStride * grid));
      ra.scratch = replayScratch.p; ra.nLo = REPLAY_MID; ra.nHi = 0xFFFFFFFFu;
      replay_kernel<false, 1024><<<grid, 1024, 0, st>>>(ra);
    }
    H10X_HIP(c, hipGetLastError());
  }
  // (e) the ordered sums on a side stream beside (f) the read merges: they touch disjoint fields
  H10X_TRY(c->forkStreams(1));
  point_sum_kernel<<<divUp(span, SUM_THREADS / WAVE), SUM_THREADS, 0, c->aux[0]>>>(c->blocks.p, c->blockOff.p, c->nGood.p, (const double *)term.p, (u32)codeMin, (u32)codeMax);
  {
    const size_t ldsSmall = mergeBytes(MERGE_SMALL_READS), ldsBig = mergeBytes(65536);
    read_merge_kernel<false><<<span, MERGE_THREADS, ldsSmall, st>>>(c->blocks.p, c->blockOff.p, c->nGood.p, c->clusHash.p, (u32)codeMin);
    if (hc[8] > MERGE_SMALL_READS) {
      H10X_HIP(c, hipFuncSetAttribute((const void *)read_merge_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBig));
      read_merge_kernel<true><<<span, MERGE_THREADS, ldsBig, st>>>(c->blocks.p, c->blockOff.p, c->nGood.p, c->clusHash.p, (u32)codeMin);
    }
  }
  H10X_HIP(c, hipGetLastError());
  H10X_TRY(c->faultAt(5));
  H10X_TRY(c->joinStreams(1));
  c->tstop(T_CLUSTER_K);
  u64 hs[8];
  H10X_TRY(c->readback(hs, stats.p, 64));
  H10X_TRY(c->syncReadbacks());
  c->tstop(T_CLUSTER);
  if (c->optStamps) { H10X_HIP(c, hipMemcpy(c->ctr.cluster_phase_ticks, phase.p, 64, hipMemcpyDeviceToHost)); }
  forkGuard.done();                                          // everything has been waited for
  if (!more) { memset(c->ctr.cluster_main, 0, sizeof c->ctr.cluster_main); memset(c->ctr.cluster_class_counts, 0, sizeof c->ctr.cluster_class_counts);
               c->ctr.sum_good = c->ctr.sum_good_depth = c->ctr.sum_hash_clustered = c->ctr.clustered_codes = c->ctr.cluster_overflow_blocks = 0; }
  for (int k = 0; k < 4; ++k) { c->ctr.cluster_main[k] += hs[4 + k]; hs[k] += hs[4 + k]; }   // re-runs of overflowed blocks count again (they did the work twice)
  c->ctr.sum_good += hs[0]; c->ctr.sum_good_depth += hs[1]; c->ctr.sum_hash_clustered += hs[2]; c->ctr.clustered_codes += span;
  c->ctr.cluster_first_mode = (uint64_t)firstMode; c->ctr.cluster_overflow_blocks += nOverflow;
  for (int k = 0; k < 4; ++k) c->ctr.cluster_class_counts[k] += hc[k];
  return 0;
}

// the --cluster loop over GLOBAL block numbers [codeMin, codeMax): every segment of this context runs its part
int stageC_cluster(Ctx *c, int codeMin, int codeMax, int threshold) {
  if (!c->haveGood) return c->fail("!! you must set hashDepthRange before cluster");          // hash10x.c:1258
  if (threshold < 1) return c->fail("clusterThreshold %d must be >= 1 (the reference reads an uninitialised msBest otherwise)", threshold);
  if (c->maxGoodDepth > RES_COUNT_MAX) return c->fail("a hash of the depth range lies in %u barcodes: beyond %u, the limit of this build", c->maxGoodDepth, RES_COUNT_MAX);
  const u32 nGlobal = c->sharded ? c->nBlocksGlobal : c->nBlocks;
  if (!codeMin) codeMin = 1;                                                                    // hash10x.c:1243-1244
  if (!codeMax) codeMax = (int)nGlobal;
  if (codeMin < 0 || codeMax > (int)nGlobal) return c->fail("cluster code range %d..%d outside 1..%u", codeMin, codeMax, nGlobal);
  bool any = false;
  for (int k = 0; k < c->segs.n; ++k) {
    const BlockSeg &sg = c->segs.s[k];
    const long first = (long)sg.localStart + (k == 0 ? 1 : 0);                                  // slot 0 of the first segment is nobody's block
    long lo = (long)codeMin - (long)sg.globalBase + (long)sg.localStart, hi = (long)codeMax - (long)sg.globalBase + (long)sg.localStart;
    if (lo < first) lo = first;
    if (hi > (long)sg.localStart + (long)sg.count) hi = (long)sg.localStart + (long)sg.count;
    if (hi <= lo) continue;
    H10X_TRY(cluster_local_range(c, (int)lo, (int)hi, threshold, any));
    any = true;
  }
  if (!any) { memset(c->ctr.cluster_class_counts, 0, sizeof c->ctr.cluster_class_counts); memset(c->ctr.cluster_main, 0, sizeof c->ctr.cluster_main);
              c->ctr.sum_good = c->ctr.sum_good_depth = c->ctr.sum_hash_clustered = c->ctr.clustered_codes = c->ctr.cluster_overflow_blocks = 0; }
  return 0;
}

// ------------------------------------------------------------------------------------------ clusterSplit
// clusterSplitCodes (hash10x.c:956-1013): every sub-cluster becomes a barcode block of its own, appended after the
// original blocks (cluster j of the s-th clustered parent lands at nCodes - 1 + (sub-clusters before it) + j),
// reads renumbered in order of first appearance, labels cleared; the parent keeps its unclustered hashes.
// Not a hot path: one wavefront per parent block, lane 0 replays the reference's single pass over the block.
__global__ void split_count_kernel(const h10x_block *__restrict__ oldB, const u64 *__restrict__ oldOff, const h10x_clushash *__restrict__ ch,
                                   u32 nCodes, const u32 *__restrict__ subBefore /* exclusive scan of nSubCluster */, SegMap segs,
                                   h10x_block *__restrict__ newB, u32 *__restrict__ newNHash) {
  const u32 i = blockIdx.x; if (i >= nCodes) return;
  const h10x_block ob = oldB[i];
  __shared__ u32 cnt[256];
  for (u32 t = threadIdx.x; t < 256; t += blockDim.x) cnt[t] = 0;
  __syncthreads();
  if (!ob.nSubCluster) {
    if (threadIdx.x == 0) { h10x_block b = ob; b.clusHash = 0; newB[i] = b; newNHash[i] = (i >= 1) ? ob.nHash : 0; }
    return;
  }
  const h10x_clushash *e = ch + oldOff[i];
  for (u32 p = threadIdx.x; p < ob.nHash; p += blockDim.x) atomicAdd(&cnt[e[p].subCluster], 1u);
  __syncthreads();
  const u32 ext = nCodes - 1 + subBefore[i];
  for (u32 j = threadIdx.x; j <= ob.nSubCluster && j < 256; j += blockDim.x) {
    h10x_block b; memset(&b, 0, sizeof b);
    if (j == 0) { b.nRead = ob.nRead; b.nHash = cnt[0]; newB[i] = b; newNHash[i] = cnt[0]; }      // hash10x.c:990: nRead kept, rest calloc'd
    else { b.nHash = cnt[j]; b.clusterParent = segs.globalOf(i) + 1; newB[ext + j] = b; newNHash[ext + j] = cnt[j]; }   // nRead filled by the move pass (parent: global number)
  }
}

__global__ void split_move_kernel(const h10x_block *__restrict__ oldB, const u64 *__restrict__ oldOff, const h10x_clushash *__restrict__ ch,
                                  u32 nCodes, const u32 *__restrict__ subBefore, const u64 *__restrict__ readOff /* scan of nRead+1 */,
                                  u32 *__restrict__ readMap /* zeroed */, h10x_block *__restrict__ newB, const u64 *__restrict__ newOff,
                                  h10x_clushash *__restrict__ out) {
  const u32 i = blockIdx.x; if (i >= nCodes) return;
  const h10x_block ob = oldB[i];
  const h10x_clushash *e = ch + oldOff[i];
  if (!ob.nSubCluster) {                                     // *new1++ = *old (hash10x.c:996)
    if (i >= 1) for (u32 p = threadIdx.x; p < ob.nHash; p += blockDim.x) out[newOff[i] + p] = e[p];
    return;
  }
  if (threadIdx.x != 0) return;
  __shared__ u32 cursor[256], nReadNew[256];
  for (u32 j = 0; j <= ob.nSubCluster && j < 256; ++j) { cursor[j] = 0; nReadNew[j] = 0; }
  const u32 ext = nCodes - 1 + subBefore[i];
  u32 *rm = readMap + readOff[i];
  for (u32 p = 0; p < ob.nHash; ++p) {                       // hash10x.c:980-989
    h10x_clushash c = e[p]; const u32 cl = c.subCluster; c.subCluster = 0;
    if (cl && cl <= ob.nSubCluster) {
      if (!rm[c.read]) rm[c.read] = ++nReadNew[cl];
      c.read = (u16)(rm[c.read] - 1);
      out[newOff[ext + cl] + cursor[cl]++] = c;
    } else out[newOff[i] + cursor[0]++] = c;
  }
  for (u32 j = 1; j <= ob.nSubCluster && j < 256; ++j) newB[ext + j].nRead = nReadNew[j];
}

__global__ void split_aux_kernel(const h10x_block *__restrict__ b, u32 n, u32 *__restrict__ nSub, u32 *__restrict__ nReadP1) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { nSub[i] = b[i].nSubCluster; nReadP1[i] = b[i].nSubCluster ? (b[i].nRead > 65535u ? 65536u : b[i].nRead) + 1 : 0; }
}

int stageC_split(Ctx *c) {
  hipStream_t st = c->stream; PrimTemp pt;
  if (!c->haveState) return c->fail("no hash state loaded: use readFQB or readHash first");
  c->tstart(T_SPLIT);
  const u32 nCodes = c->nBlocks;
  DevBuf<u32> nSub, subBefore, nReadP1; DevBuf<u64> readOff;
  H10X_HIP(c, nSub.alloc((size_t)nCodes + 1)); H10X_HIP(c, subBefore.alloc((size_t)nCodes + 1));
  H10X_HIP(c, nReadP1.alloc((size_t)nCodes + 1)); H10X_HIP(c, readOff.alloc((size_t)nCodes + 1));
  H10X_HIP(c, hipMemsetAsync(nSub.p + nCodes, 0, 4, st)); H10X_HIP(c, hipMemsetAsync(nReadP1.p + nCodes, 0, 4, st));
  split_aux_kernel<<<divUp(nCodes, 256), 256, 0, st>>>(c->blocks.p, nCodes, nSub.p, nReadP1.p);
  H10X_TRY(prim_exclusive_scan_u32(c, pt, nSub.p, subBefore.p, (size_t)nCodes + 1));
  H10X_TRY(prim_exclusive_scan_u32_u64(c, pt, nReadP1.p, readOff.p, (size_t)nCodes + 1));
  u32 totalSub = 0; u64 totalRead = 0;
  H10X_HIP(c, hipMemcpyAsync(&totalSub, subBefore.p + nCodes, 4, hipMemcpyDeviceToHost, st));
  H10X_HIP(c, hipMemcpyAsync(&totalRead, readOff.p + nCodes, 8, hipMemcpyDeviceToHost, st));
  H10X_HIP(c, hipStreamSynchronize(st));
  const u64 nNew64 = (u64)nCodes + totalSub;
  if (nNew64 >= (1ULL << 31)) return c->fail("clusterSplit would create %llu barcode blocks", (u64)nNew64);
  const u32 nNew = (u32)nNew64;
  DevBuf<h10x_block> newB; DevBuf<u32> newNHash, readMap; DevBuf<u64> newOff; DevBuf<h10x_clushash> out;
  H10X_HIP(c, newB.alloc(nNew)); H10X_HIP(c, newNHash.alloc((size_t)nNew + 1)); H10X_HIP(c, newOff.alloc((size_t)nNew + 1));
  H10X_HIP(c, readMap.alloc(totalRead + 1)); H10X_HIP(c, out.alloc(c->nEntries));
  H10X_HIP(c, hipMemsetAsync(newB.p, 0, (size_t)nNew * sizeof(h10x_block), st));
  H10X_HIP(c, hipMemsetAsync(newNHash.p, 0, ((size_t)nNew + 1) * 4, st));
  H10X_HIP(c, hipMemsetAsync(readMap.p, 0, (totalRead + 1) * 4, st));
  split_count_kernel<<<nCodes, 256, 0, st>>>(c->blocks.p, c->blockOff.p, c->clusHash.p, nCodes, subBefore.p, c->segs, newB.p, newNHash.p);
  H10X_TRY(prim_exclusive_scan_u32_u64(c, pt, newNHash.p, newOff.p, (size_t)nNew + 1));
  split_move_kernel<<<nCodes, 64, 0, st>>>(c->blocks.p, c->blockOff.p, c->clusHash.p, nCodes, subBefore.p, readOff.p, readMap.p, newB.p, newOff.p, out.p);
  H10X_HIP(c, hipGetLastError());
  H10X_HIP(c, hipStreamSynchronize(st));
  c->blocks.swap(newB); c->blockOff.swap(newOff); c->clusHash.swap(out);
  c->nBlocks = nNew;
  c->haveGood = false; c->goodPos.release(); c->nGood.release(); c->goodEntries.release(); c->goodRow.release();      // lists refer to the old blocks: a new --hashDepthRange is required
  if (c->sharded) H10X_TRY(shard_split(c, subBefore.p, totalSub));   // new blocks get their global numbers; the hash owners rebuild their lists
  else { c->segs.n = 1; c->segs.s[0] = BlockSeg{0, nNew, 0}; H10X_TRY(stageB_buildCSR(c)); }   // hash10x.c:1008-1012: hashCodes rebuilt, hashDepth unchanged
  c->tstop(T_SPLIT);
  return 0;
}

// h10x_warm: the first launch of a kernel loads the code object of its translation unit (HIP loads them on first use); this one is launched ahead of time
__global__ void warm_stageC_kernel() {}
void warm_stageC(hipStream_t st) { warm_stageC_kernel<<<1, 1, 0, st>>>(); }

}  // namespace h10x
