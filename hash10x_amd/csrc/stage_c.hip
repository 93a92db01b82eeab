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
//   (e) one lane adds the quotients in rank order => bit-identical pointToMin
//   (f) read merge : connected components over <= 255 labels linked by shared reads, renumbered by
//                     ascending minimum label.
// Barcodes whose working set exceeds the LDS budget run the same code on a per-workgroup HBM scratch.
#include "common.hpp"
#include "prim.hpp"

namespace h10x {

constexpr int CL_THREADS = 512;
constexpr int CL_WAVES = CL_THREADS / WAVE;
constexpr u16 NONE16 = 0xFFFF;
constexpr int RCHUNK = 4;                                   // register chunks: lists up to 256 entries

// ------------------------------------------------------------------------------------------ depth range
__global__ void within_kernel(const u32 *__restrict__ depth, u32 hashNumber, int lo, int hi, u8 *__restrict__ within) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hashNumber) return;
  const int n = (int)depth[i];
  if (n >= lo && n < hi) within[i] = 1;                     // only ever set (hash10x.c:535)
}
__global__ void within_depth_kernel(const u32 *__restrict__ depth, const u8 *__restrict__ within, u32 hashNumber, u32 *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < hashNumber) out[i] = within[i] ? depth[i] : 0;
}

// per block: keys (depth << 16 | position) of its in-range hashes, appended in any order
__global__ __launch_bounds__(256)
void good_keys_kernel(const h10x_clushash *__restrict__ ch, const u64 *__restrict__ blockOff, const h10x_block *__restrict__ blocks,
                      u32 nBlocks, const u8 *__restrict__ within, const u32 *__restrict__ depth,
                      u64 *__restrict__ key, u32 *__restrict__ nGood, u32 *__restrict__ segEnd) {
  __shared__ u32 sCount;
  for (u32 c = blockIdx.x; c < nBlocks; c += gridDim.x) {
    const u64 o = blockOff[c]; const u32 nHash = blocks[c].nHash;
    __syncthreads();
    if (threadIdx.x == 0) sCount = 0;
    __syncthreads();
    if (nHash <= 65535) {                                    // hash10x.c:748-753: bigger blocks are ignored
      for (u32 base = 0; base < nHash; base += blockDim.x) {
        const u32 p = base + threadIdx.x;
        u32 ix = 0; bool good = false;
        if (p < nHash) { ix = ch[o + p].hash; good = within[ix] != 0; }
        const u64 bal = __ballot(good);
        const int lane = threadIdx.x & (WAVE - 1);
        u32 wb = 0;
        if (lane == 0 && bal) wb = atomicAdd(&sCount, (u32)__popcll(bal));
        wb = __shfl(wb, 0);
        if (good) key[o + wb + (u32)__popcll(bal & ((1ULL << lane) - 1))] = ((u64)depth[ix] << 16) | (u64)p;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) { nGood[c] = sCount; segEnd[c] = (u32)o + sCount; }
  }
}
__global__ void good_pos_kernel(const u64 *__restrict__ key, u64 n, u16 *__restrict__ pos) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) pos[i] = (u16)(key[i] & 0xFFFF);
}
__global__ void offsets32c_kernel(const u64 *__restrict__ off, u32 n, u32 *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (u32)off[i];
}

static int bitsForC(u64 maxValue) { int b = 1; while (b < 64 && (maxValue >> b)) ++b; return b; }

int stageC_depthRange(Ctx *c, int lo, int hi) {
  hipStream_t st = c->stream; PrimTemp pt;
  if (!c->haveState) return c->fail("no hash state loaded: use readFQB or readHash first");
  c->tstart(T_GOOD);
  const u32 U1 = c->hashNumber; const u32 nBlocks = c->nBlocks; const u64 H = c->nEntries;
  if (!(c->haveRange && lo == c->rangeMin && hi == c->rangeMax)) {       // hash10x.c:530
    if (!c->haveRange) { H10X_HIP(c, c->within.alloc(U1)); H10X_HIP(c, hipMemsetAsync(c->within.p, 0, U1, st)); }
    within_kernel<<<divUp(U1, 256), 256, 0, st>>>(c->hashDepth.p, U1, lo, hi, c->within.p);
    c->haveRange = true; c->rangeMin = lo; c->rangeMax = hi;
  }
  // goodHashesBuild (hash10x.c:738-766)
  DevBuf<u64> key, keyS; DevBuf<u32> off32, segEnd, wdepth, red;
  H10X_HIP(c, key.alloc(H)); H10X_HIP(c, keyS.alloc(H)); H10X_HIP(c, off32.alloc((size_t)nBlocks + 1));
  H10X_HIP(c, segEnd.alloc(nBlocks)); H10X_HIP(c, wdepth.alloc(U1)); H10X_HIP(c, red.alloc(2));
  H10X_HIP(c, c->nGood.alloc(nBlocks)); H10X_HIP(c, c->goodPos.alloc(H));
  within_depth_kernel<<<divUp(U1, 256), 256, 0, st>>>(c->hashDepth.p, c->within.p, U1, wdepth.p);
  H10X_TRY(prim_reduce_max_u32(c, pt, wdepth.p, red.p, U1));
  good_keys_kernel<<<hmin<u32>(nBlocks, 16384), 256, 0, st>>>(c->clusHash.p, c->blockOff.p, c->blocks.p, nBlocks, c->within.p,
                                                            c->hashDepth.p, key.p, c->nGood.p, segEnd.p);
  H10X_TRY(prim_reduce_max_u32(c, pt, c->nGood.p, red.p + 1, nBlocks));
  u32 hr[2];
  H10X_HIP(c, hipMemcpyAsync(hr, red.p, 8, hipMemcpyDeviceToHost, st));
  H10X_HIP(c, hipStreamSynchronize(st));
  c->maxGoodDepth = hr[0]; c->maxGood = hr[1];
  offsets32c_kernel<<<divUp((u64)nBlocks + 1, 256), 256, 0, st>>>(c->blockOff.p, nBlocks + 1, off32.p);
  // ascending (depth, position): qsort by depth, stable => ties by position (hash10x.c:726-730,758)
  if (H) {
    H10X_TRY(prim_seg_sort_keys_u64(c, pt, key.p, keyS.p, (u32)H, nBlocks, off32.p, segEnd.p, 0, 16 + bitsForC(hr[0])));
    good_pos_kernel<<<(unsigned)hmin<u64>(divUp(H, 256), 65535u * 2), 256, 0, st>>>(keyS.p, H, c->goodPos.p);
  }
  H10X_HIP(c, hipGetLastError());
  H10X_HIP(c, hipStreamSynchronize(st));
  c->haveGood = true;
  c->tstop(T_GOOD);
  return 0;
}

// ------------------------------------------------------------------------------------------ cluster kernel
struct ClusterArgs {
  h10x_block *blocks; const u64 *blockOff; h10x_clushash *clusHash;
  const u16 *goodPos; const u32 *nGood;
  const u32 *hashDepth; const u64 *rowStart; const u32 *rows;
  const u32 *list; u32 nList; u32 *workCounter;
  u32 nBlocks; int threshold;
  unsigned char *scratch; size_t scratchStride;             // global-mode working set per workgroup
  size_t mergeOffset;                                       // global mode: read-merge tables live behind the rank arrays
  u32 maxGood;
  u64 *stats;                                               // [0] sum good, [1] sum good depth, [2] sum nHash, [3] codes
};

struct PairCT { u32 cnt, tot; };                            // overwritten by the double quotient in phase (d)

// working set layout inside a region (LDS or HBM scratch)
struct Work {
  u16 *first;        // nBlocks entries, NONE16 = unseen
  u32 *gx;           // hash index of each good rank
  PairCT *ct;        // msMax / msTot per rank, later the quotient (as double)
  u16 *best;         // msBest per rank
  u16 *qj;           // founding rank of the cluster the rank joins (NONE16 = no term)
  u8  *lab;          // label per rank
};
__host__ __device__ inline size_t workBytes(u32 nBlocks, u32 n) {
  size_t b = (((size_t)nBlocks * 2 + 15) & ~(size_t)15);
  b += (size_t)n * 8;                                       // ct (8-byte aligned first)
  b += (size_t)n * 4;                                       // gx
  b += (size_t)n * 2 * 2;                                   // best, qj
  b += ((size_t)n + 15) & ~(size_t)15;                      // lab
  return b + 16;
}
__device__ inline Work carve(unsigned char *base, u32 nBlocks, u32 n) {
  Work w; size_t o = 0;
  w.first = (u16 *)(base + o); o += (((size_t)nBlocks * 2 + 15) & ~(size_t)15);
  w.ct = (PairCT *)(base + o); o += (size_t)n * 8;
  w.gx = (u32 *)(base + o); o += (size_t)n * 4;
  w.best = (u16 *)(base + o); o += (size_t)n * 2;
  w.qj = (u16 *)(base + o); o += (size_t)n * 2;
  w.lab = (u8 *)(base + o);
  return w;
}
// read-merge working set: readRep[min(nRead,65536)] bytes, adj[256][8] u32, comp[256] u32, newLab[256] u32
__host__ __device__ inline size_t mergeBytes(u32 nRead) {
  const size_t r = nRead < 65536u ? nRead : 65536u;
  return ((r + 15) & ~(size_t)15) + 256 * 8 * 4 + 256 * 4 + 256 * 4 + 16;
}

// CAS-min on a u16 living in a u32 word (LDS or global)
__device__ __forceinline__ void min_u16(u16 *arr, u32 idx, u32 val) {
  u32 *w = (u32 *)arr + (idx >> 1); const int sh = (idx & 1) * 16;
  u32 old = *(volatile u32 *)w;
  for (;;) {
    if (((old >> sh) & 0xFFFFu) <= val) return;
    const u32 nw = (old & ~(0xFFFFu << sh)) | (val << sh);
    const u32 prev = atomicCAS(w, old, nw);
    if (prev == old) return;
    old = prev;
  }
}

// msBest / msMax / msTot of one barcode list for rank i, one wavefront, list entries in registers
__device__ __forceinline__ void row_mode_regs(const u32 *__restrict__ row, u32 d, u32 code, u32 i, const u16 *first,
                                              u32 &best, u32 &bcnt, u32 &tot) {
  const int lane = threadIdx.x & (WAVE - 1);
  u32 f[RCHUNK]; u64 rem[RCHUNK];
  tot = 0;
#pragma unroll
  for (int r = 0; r < RCHUNK; ++r) {
    const u32 j = r * WAVE + lane; bool valid = false; u32 v = NONE16;
    if (j < d) { const u32 cj = row[j]; if (cj != code) { v = first[cj]; valid = v < i; } }
    f[r] = v; rem[r] = __ballot(valid); tot += (u32)__popcll(rem[r]);
  }
  best = NONE16; bcnt = 0;
#pragma unroll
  for (int r = 0; r < RCHUNK; ++r) {
    while (rem[r]) {
      const int src = __ffsll((long long)rem[r]) - 1;
      const u32 v = (u32)__builtin_amdgcn_readlane((int)f[r], src);
      u32 cnt = 0;
#pragma unroll
      for (int q = 0; q < RCHUNK; ++q) if (q >= r) { const u64 m = __ballot(f[q] == v) & rem[q]; cnt += (u32)__popcll(m); rem[q] &= ~m; }
      if (cnt > bcnt || (cnt == bcnt && v < best)) { bcnt = cnt; best = v; }
    }
  }
}
// same for lists longer than the register budget: re-gather per candidate (rare: depth >= 256)
__device__ void row_mode_long(const u32 *__restrict__ row, u32 d, u32 code, u32 i, const u16 *first, u32 &best, u32 &bcnt, u32 &tot) {
  const int lane = threadIdx.x & (WAVE - 1);
  best = NONE16; bcnt = 0; tot = 0;
  for (u32 a0 = 0; a0 < d; a0 += WAVE) {
    u32 fa = NONE16; bool va = false;
    if (a0 + lane < d) { const u32 cj = row[a0 + lane]; if (cj != code) { fa = first[cj]; va = fa < i; } }
    u64 rem = __ballot(va); tot += (u32)__popcll(rem);
    while (rem) {
      const int src = __ffsll((long long)rem) - 1;
      const u32 v = (u32)__builtin_amdgcn_readlane((int)fa, src);
      rem &= ~__ballot(fa == v);
      u32 cnt = 0; bool seenBefore = false;
      for (u32 b0 = 0; b0 < d; b0 += WAVE) {
        u32 fb = NONE16;
        if (b0 + lane < d) { const u32 cj = row[b0 + lane]; if (cj != code) fb = first[cj]; }
        const u32 m = (u32)__popcll(__ballot(fb == v));
        if (b0 < a0 && m) { seenBefore = true; break; }
        cnt += m;
      }
      if (!seenBefore && (cnt > bcnt || (cnt == bcnt && v < best))) { bcnt = cnt; best = v; }
    }
  }
}
__device__ __forceinline__ u32 row_count(const u32 *__restrict__ row, u32 d, u32 code, u32 target, const u16 *first) {
  const int lane = threadIdx.x & (WAVE - 1);
  u32 cnt = 0;
  for (u32 b0 = 0; b0 < d; b0 += WAVE) {
    bool m = false;
    if (b0 + lane < d) { const u32 cj = row[b0 + lane]; if (cj != code) m = first[cj] == target; }
    cnt += (u32)__popcll(__ballot(m));
  }
  return cnt;
}

template <bool IN_LDS>
__device__ void cluster_one_block(const ClusterArgs &a, u32 code, unsigned char *region, u32 *sh /* small shared ints */) {
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
  const u32 n = a.nGood[code];
  if (n == 0) return;                                        // hash10x.c:780: block left untouched
  const u64 o = a.blockOff[code];
  const u32 nHash = a.blocks[code].nHash, nRead = a.blocks[code].nRead;
  Work w = carve(region, a.nBlocks, n);
  h10x_clushash *ch = a.clusHash + o; const u16 *g = a.goodPos + o;

  // ---- init
  if (IN_LDS) for (u32 i = tid; i < (a.nBlocks + 1) / 2; i += CL_THREADS) ((u32 *)w.first)[i] = 0xFFFFFFFFu;
  u64 myDepth = 0;
  for (u32 i = tid; i < n; i += CL_THREADS) {
    const u32 x = ch[g[i]].hash; w.gx[i] = x; w.lab[i] = 0; w.qj[i] = NONE16; w.best[i] = NONE16;
    w.ct[i].cnt = 0; w.ct[i].tot = 0; myDepth += a.hashDepth[x];
  }
  __syncthreads();

  // ---- (a) first[]: lowest rank >= 1 sharing each other barcode (hash10x.c:794-799, minShare)
  for (u32 i = 1 + wave; i < n; i += CL_WAVES) {
    const u32 x = w.gx[i]; const u32 d = a.hashDepth[x]; const u32 *row = a.rows + a.rowStart[x];
    for (u32 j = lane; j < d; j += WAVE) { const u32 cj = row[j]; if (cj != code) min_u16(w.first, cj, i); }
  }
  __syncthreads();

  // ---- (b) msBest / msMax / msTot per rank (hash10x.c:801-806)
  for (u32 i = 1 + wave; i < n; i += CL_WAVES) {
    const u32 x = w.gx[i]; const u32 d = a.hashDepth[x]; const u32 *row = a.rows + a.rowStart[x];
    u32 best, bcnt, tot;
    if (d <= RCHUNK * WAVE) row_mode_regs(row, d, code, i, w.first, best, bcnt, tot);
    else row_mode_long(row, d, code, i, w.first, best, bcnt, tot);
    if (lane == 0) { w.best[i] = (u16)best; w.ct[i].cnt = bcnt; w.ct[i].tot = tot; }
  }
  __syncthreads();

  // ---- (c) order-dependent replay (hash10x.c:807-822)
  if (tid == 0) {
    u32 nSub = 0, stop = n; u16 *cmin = (u16 *)(sh + 4);       // founding rank of each cluster (clusterMin, hash10x.c:818)
    for (u32 i = 1; i < n; ++i) {
      if ((int)w.ct[i].cnt < a.threshold) continue;
      const u32 b = w.best[i]; u32 L = w.lab[b];
      if (!L) {
        if (++nSub > 255) {                                  // hash10x.c:810-816: abandon; partial pointToMin is kept
          nSub = 0; for (u32 j = 0; j < i; ++j) w.lab[j] = 0;
          stop = i; break;
        }
        L = nSub; w.lab[b] = (u8)L; cmin[L] = (u16)b;
      }
      w.lab[i] = (u8)L; w.qj[i] = cmin[L];
    }
    sh[0] = nSub; sh[1] = stop;
  }
  __syncthreads();
  const u32 nSub = sh[0], stop = sh[1];

  // ---- (d) minShareCount[clusterMin[label]] / (double) msTot per rank (hash10x.c:821)
  for (u32 i = 1 + wave; i < stop; i += CL_WAVES) {
    const u32 qj = w.qj[i];
    if (qj == NONE16) continue;
    u32 q;
    if (qj == w.best[i]) q = w.ct[i].cnt;
    else { const u32 x = w.gx[i]; q = row_count(a.rows + a.rowStart[x], a.hashDepth[x], code, qj, w.first); }
    if (lane == 0) { const double t = (double)(int)q / (double)(int)w.ct[i].tot; *(double *)&w.ct[i] = t; }
  }
  __syncthreads();

  // ---- (e) ordered fp64 sum, labels out
  if (tid == 0) {
    double p = 0.0;
    for (u32 i = 1; i < stop; ++i) if (w.qj[i] != NONE16) p += *(const double *)&w.ct[i];
    a.blocks[code].pointToMin = p; a.blocks[code].nSubCluster = nSub;
  }
  for (u32 i = tid; i < n; i += CL_THREADS) ch[g[i]].subCluster = w.lab[i];     // includes the wipe of hash10x.c:783
  // wave-reduce the per-thread depth sums for the work counters
  for (int s = 32; s; s >>= 1) myDepth += __shfl_down(myDepth, s);
  if (lane == 0) atomicAdd((u64 *)&a.stats[1], (u64)myDepth);
  if (tid == 0) { atomicAdd((u64 *)&a.stats[0], (u64)n); atomicAdd((u64 *)&a.stats[2], (u64)nHash); atomicAdd((u64 *)&a.stats[3], 1ULL); }
  if (!IN_LDS) {                                             // leave first[] clean for the next barcode of this workgroup
    __syncthreads();
    for (u32 i = 1 + wave; i < n; i += CL_WAVES) {
      const u32 x = w.gx[i]; const u32 d = a.hashDepth[x]; const u32 *row = a.rows + a.rowStart[x];
      for (u32 j = lane; j < d; j += WAVE) w.first[row[j]] = NONE16;
    }
  }
  __syncthreads();
  if (!nSub) return;                                         // hash10x.c:840

  // ---- (f) codeClusterReadMerge (hash10x.c:837-868): components of labels that share a read
  const u32 nRep = nRead < 65536u ? nRead : 65536u;
  unsigned char *mregion = IN_LDS ? region : region + a.mergeOffset;   // LDS: the rank arrays are dead by now
  u8 *readRep = (u8 *)mregion;
  u32 *adj = (u32 *)(mregion + ((nRep + 15) & ~15u));
  u32 *comp = adj + 256 * 8, *newLab = comp + 256;
  for (u32 i = tid; i < (nRep + 3) / 4; i += CL_THREADS) ((u32 *)readRep)[i] = 0;
  for (u32 i = tid; i < 256 * 8; i += CL_THREADS) adj[i] = 0;
  if (tid < 256) comp[tid] = tid;
  __threadfence_block();
  __syncthreads();
  for (u32 p = tid; p < nHash; p += CL_THREADS) {            // any one label of each read is its representative
    const h10x_clushash e = ch[p];
    if (e.subCluster && e.subCluster <= nSub && e.read < nRep) readRep[e.read] = e.subCluster;
  }
  __syncthreads();
  for (u32 p = tid; p < nHash; p += CL_THREADS) {
    const h10x_clushash e = ch[p];
    if (e.subCluster && e.subCluster <= nSub && e.read < nRep) {
      const u32 L = e.subCluster, R = readRep[e.read];
      if (R != L) { atomicOr(&adj[L * 8 + (R >> 5)], 1u << (R & 31)); atomicOr(&adj[R * 8 + (L >> 5)], 1u << (L & 31)); }
    }
  }
  __syncthreads();
  for (int iter = 0; iter < 256; ++iter) {                   // min-label propagation; <= 255 rounds
    if (tid == 0) sh[2] = 0;
    __syncthreads();
    if (tid >= 1 && tid <= (int)nSub) {
      u32 m = comp[tid];
      for (int wd = 0; wd < 8; ++wd) { u32 bits = adj[tid * 8 + wd]; while (bits) { const int b = __ffs((int)bits) - 1; bits &= bits - 1; const u32 cR = comp[wd * 32 + b]; m = cR < m ? cR : m; } }
      if (m < comp[tid]) { atomicMin(&comp[tid], m); sh[2] = 1; }
    }
    __syncthreads();
    if (!sh[2]) break;
    __syncthreads();
  }
  if (tid == 0) {                                            // renumber components by ascending minimum label
    u32 alive = 0; newLab[0] = 0;
    for (u32 L = 1; L <= nSub; ++L) { if (comp[L] == L) ++alive; newLab[L] = alive; }   // rank of L if it is a minimum
    for (u32 L = 1; L <= nSub; ++L) comp[L] = newLab[comp[L]];                          // label -> rank of its minimum
    a.blocks[code].nSubCluster = alive;
  }
  __syncthreads();
  for (u32 p = tid; p < nHash; p += CL_THREADS) {
    const u32 L = ch[p].subCluster;
    if (L && L <= nSub) ch[p].subCluster = (u8)comp[L];
  }
  __syncthreads();
}

template <bool IN_LDS>
__global__ __launch_bounds__(CL_THREADS)
void cluster_kernel(ClusterArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ u32 sh[4 + 128];                               // [0..3] scalars, then clusterMin[256] as u16
  unsigned char *region = IN_LDS ? smem : a.scratch + (size_t)blockIdx.x * a.scratchStride;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) sh[3] = atomicAdd(a.workCounter, 1u);
    __syncthreads();
    const u32 wi = sh[3];
    if (wi >= a.nList) break;                                // every wave of the workgroup leaves together
    cluster_one_block<IN_LDS>(a, a.list[wi], region, sh);
  }
}

// which barcodes fit the LDS budget; heaviest-first would be better for the tail, this keeps file order
__global__ void cluster_classify_kernel(const h10x_block *__restrict__ blocks, const u32 *__restrict__ nGood, u32 codeMin, u32 codeMax,
                                        u32 nBlocks, size_t ldsBudget, u32 *__restrict__ listL, u32 *__restrict__ listG, u32 *__restrict__ counts) {
  const u32 c = codeMin + blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= codeMax) return;
  const u32 n = nGood[c];
  if (!n) return;
  const size_t need = max(workBytes(nBlocks, n), mergeBytes(blocks[c].nRead));
  if (need <= ldsBudget) listL[atomicAdd(&counts[0], 1u)] = c; else listG[atomicAdd(&counts[1], 1u)] = c;
}

int stageC_cluster(Ctx *c, int codeMin, int codeMax, int threshold) {
  hipStream_t st = c->stream;
  if (!c->haveGood) return c->fail("!! you must set hashDepthRange before cluster");          // hash10x.c:1258
  if (threshold < 1) return c->fail("clusterThreshold %d must be >= 1 (the reference reads an uninitialised msBest otherwise)", threshold);
  if (!codeMin) codeMin = 1;                                                                    // hash10x.c:1243-1244
  if (!codeMax) codeMax = (int)c->nBlocks;
  if (codeMin < 0 || codeMax > (int)c->nBlocks) return c->fail("cluster code range %d..%d outside 1..%u", codeMin, codeMax, c->nBlocks);
  if (codeMax <= codeMin) return 0;
  c->tstart(T_CLUSTER);
  const u32 span = (u32)(codeMax - codeMin);
  DevBuf<u32> listL, listG, counts; DevBuf<u64> stats;
  H10X_HIP(c, listL.alloc(span)); H10X_HIP(c, listG.alloc(span)); H10X_HIP(c, counts.alloc(4)); H10X_HIP(c, stats.alloc(4));
  H10X_HIP(c, hipMemsetAsync(counts.p, 0, 16, st)); H10X_HIP(c, hipMemsetAsync(stats.p, 0, 32, st));
  const size_t ldsBudget = 64 * 1024 - 1024;
  cluster_classify_kernel<<<divUp(span, 256), 256, 0, st>>>(c->blocks.p, c->nGood.p, (u32)codeMin, (u32)codeMax, c->nBlocks, ldsBudget, listL.p, listG.p, counts.p);
  u32 hc[4];
  H10X_HIP(c, hipMemcpyAsync(hc, counts.p, 16, hipMemcpyDeviceToHost, st));
  H10X_HIP(c, hipStreamSynchronize(st));
  ClusterArgs a{};
  a.blocks = c->blocks.p; a.blockOff = c->blockOff.p; a.clusHash = c->clusHash.p; a.goodPos = c->goodPos.p; a.nGood = c->nGood.p;
  a.hashDepth = c->hashDepth.p; a.rowStart = c->rowStart.p; a.rows = c->rows.p; a.nBlocks = c->nBlocks; a.threshold = threshold;
  a.maxGood = c->maxGood; a.stats = stats.p;
  if (hc[0]) {
    a.list = listL.p; a.nList = hc[0]; a.workCounter = counts.p + 2; a.scratch = nullptr; a.scratchStride = 0;
    H10X_HIP(c, hipFuncSetAttribute((const void *)cluster_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBudget));
    const u32 grid = hmin<u32>(hc[0], (u32)c->numCU * 2);
    c->tstart(T_CLUSTER_K);
    cluster_kernel<true><<<grid, CL_THREADS, ldsBudget, st>>>(a);
    c->tstop(T_CLUSTER_K);
    H10X_HIP(c, hipGetLastError());
  }
  DevBuf<unsigned char> scratch;
  if (hc[1]) {
    // HBM working set per workgroup: first[] + per-rank arrays + read-merge tables for the largest barcode
    const size_t mergeOff = (workBytes(c->nBlocks, c->maxGood) + 255) & ~(size_t)255;
    const size_t stride = mergeOff + ((mergeBytes(65536) + 255) & ~(size_t)255);
    a.mergeOffset = mergeOff;
    const u32 grid = hmin<u32>(hc[1], (u32)c->numCU);
    H10X_HIP(c, scratch.alloc(stride * grid));
    H10X_HIP(c, hipMemsetAsync(scratch.p, 0xFF, stride * grid, st));    // first[] = unseen everywhere
    a.list = listG.p; a.nList = hc[1]; a.workCounter = counts.p + 3; a.scratch = scratch.p; a.scratchStride = stride;
    cluster_kernel<false><<<grid, CL_THREADS, 0, st>>>(a);
    H10X_HIP(c, hipGetLastError());
  }
  u64 hs[4];
  H10X_HIP(c, hipMemcpyAsync(hs, stats.p, 32, hipMemcpyDeviceToHost, st));
  H10X_HIP(c, hipStreamSynchronize(st));
  c->tstop(T_CLUSTER);
  c->ctr.sum_good = hs[0]; c->ctr.sum_good_depth = hs[1]; c->ctr.sum_hash_clustered = hs[2]; c->ctr.clustered_codes = span;
  return 0;
}

int stageC_split(Ctx *c) { return c->fail("clusterSplit is not implemented on the device path yet"); }

}  // namespace h10x
