// shard.hip — the multi-GPU form of --readFQB / --hashDepthRange / --writeHash.
//
// Barcodes are sharded by contiguous ranges of the sorted .fqb (rank r owns barcodes codeBase_r+1 ..); hashes are
// owned by value range (owner(h) = floor(q * N / nKeys), q = h / w the key the entries travel with), so the distinct hashes of owner o all precede those of o+1.
//   1. every rank: mosh extraction of its records (stage A, unchanged)                                  — no traffic
//   2. all-to-all: each (hash, global barcode) entry goes to the hash's owner                           — 12 B per entry
//   3. owner: stable sort by hash => distinct hashes, first barcode, depth, barcode lists (ascending)
//   4. allgather of the per-barcode counts of "first seen here" hashes; with hash ownership monotone this is all
//      an owner needs to number its hashes exactly as the serial hashNumber++ would (rank of (first barcode, hash))
//   5. all-to-all back: the index of every entry, in the order the entries were sent                     — 4 B per entry
//   6. allgather of (index, hash, depth) so that every rank holds hashValue[] / hashDepth[] (and rank 0 can write
//      hashIndex[]); barcode lists travel later, and only for hashes inside the depth range (--hashDepthRange),
//      by an allgather of the filtered lists — clustering then needs no communication at all.
// The result is byte-identical to the single-GPU path (tests: N ranks as threads on one GPU; bench: RCCL).
#include "common.hpp"
#include "prim.hpp"
#include "comm.hpp"
#include <algorithm>

namespace h10x {

__global__ void iota_kernel(u32 *__restrict__ p, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = (u32)i;
}
__global__ void gather_code_kernel(const u32 *__restrict__ entCode, const u32 *__restrict__ perm, u64 n, u32 codeBase, u32 *__restrict__ out) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = entCode[perm[i]] + codeBase;
}
// owner of a hash = the o with lowHash[o] <= h < lowHash[o + 1] (value ranges, so owner o's hashes are all smaller than owner o + 1's)
__global__ void owner_key_kernel(const u64 *__restrict__ entHash, u64 n, const u64 *__restrict__ lowHash /* N+1 */, int N, u32 *__restrict__ key) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const u64 h = entHash[i];
    int lo = 0, hi = N;                                      // largest o with lowHash[o] <= h
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (lowHash[mid] <= h) lo = mid; else hi = mid; }
    key[i] = (u32)lo;
  }
}
__global__ void owner_key_bounds_kernel(const u32 *__restrict__ sKey, u64 n, int N, u64 *__restrict__ bound) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o > N) return;
  if (o == N) { bound[o] = n; return; }
  u64 lo = 0, hi = n;
  while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (sKey[mid] < (u32)o) lo = mid + 1; else hi = mid; }
  bound[o] = lo;
}
__global__ void gather_send_kernel(const u64 *__restrict__ entHash, const u32 *__restrict__ entCode, const u32 *__restrict__ perm, u64 n, u32 codeBase,
                                   u64 *__restrict__ sHash, u32 *__restrict__ sCode) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) { const u32 p = perm[i]; sHash[i] = entHash[p]; sCode[i] = entCode[p] + codeBase; }
}
// Partition by owner, block by block (the entries of a barcode block are contiguous, blockOff[]): one workgroup per block
// counts its entries per owner (pass 0), and after an exclusive scan of the counts in (owner, block) order places them
// (pass 1). An owner's slice then holds the blocks in barcode order — all the owner's stable sort by hash needs, hashes
// being unique inside a block — without a sort and a gather over all entries.
constexpr int PART_MAX_OWNERS = 1024;
template <int PASS>
__global__ __launch_bounds__(256)
void owner_partition_kernel(const u64 *__restrict__ entHash, const u32 *__restrict__ entCode, const u64 *__restrict__ blockOff, u32 nBlocks,
                            const u64 *__restrict__ lowHash /* N+1 */, int N, u32 codeBase, u32 *__restrict__ cnt /* N x nBlocks: counts, then offsets */,
                            u64 *__restrict__ sHash, u32 *__restrict__ sCode, u32 *__restrict__ perm /* perm[pos] = e */, int cb /* > 0: key << cb | global block in sHash, no sCode */,
                            u32 *__restrict__ posOf = nullptr /* instead of perm: posOf[e] = pos — a coalesced store, and what the ClusterHash records gather the replies by */) {
  __shared__ u64 low[PART_MAX_OWNERS + 1];
  __shared__ u32 fill[PART_MAX_OWNERS];
  for (int o = threadIdx.x; o <= N; o += blockDim.x) low[o] = lowHash[o];
  for (u32 b = blockIdx.x + 1; b < nBlocks; b += gridDim.x) {
    const u64 e0 = blockOff[b], e1 = blockOff[b + 1];
    __syncthreads();
    for (int o = threadIdx.x; o < N; o += blockDim.x) fill[o] = PASS ? cnt[(size_t)o * nBlocks + b] : 0u;
    __syncthreads();
    for (u64 e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
      const u64 h = entHash[e];
      int lo = 0, hi = N;                                    // largest o with low[o] <= h
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (low[mid] <= h) lo = mid; else hi = mid; }
      const u32 pos = atomicAdd(&fill[lo], 1u);
      if (PASS) { if (cb) sHash[pos] = (h << cb) | (u64)(entCode[e] + codeBase); else { sHash[pos] = h; sCode[pos] = entCode[e] + codeBase; } if (posOf) posOf[e] = pos; else perm[pos] = (u32)e; }
    }
    if (!PASS) { __syncthreads(); for (int o = threadIdx.x; o < N; o += blockDim.x) cnt[(size_t)o * nBlocks + b] = fill[o]; }
  }
}
__global__ void partition_bounds_kernel(const u32 *__restrict__ off, u32 nBlocks, int N, u64 n, u64 *__restrict__ bound) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o > N) return;
  bound[o] = o == N ? n : (u64)off[(size_t)o * nBlocks];
}
// first position in the ascending array whose hash belongs to owner >= o, for o = 0..N
__global__ void owner_bounds_kernel(const u64 *__restrict__ sHash, u64 n, const u64 *__restrict__ lowHash /* N+1 */, int N, u64 *__restrict__ bound) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o > N) return;
  if (o == N) { bound[o] = n; return; }
  u64 lo = 0, hi = n; const u64 key = lowHash[o];
  while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (sHash[mid] < key) lo = mid + 1; else hi = mid; }
  bound[o] = lo;
}
__global__ void gather_u32_kernel(const u32 *__restrict__ src, const u32 *__restrict__ idx, u64 n, u32 *__restrict__ out) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = src[idx[i]];
}
__global__ void heads_kernel(const u64 *__restrict__ sHash, u64 n, u32 *__restrict__ flags) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) flags[i] = (i == 0 || sHash[i] != sHash[i - 1]) ? 1u : 0u;
}
// the same from packed entries (key << cb | global block), which also yields the owner's barcode lists
__global__ void heads_packed_kernel(const u64 *__restrict__ sKey, int cb, u64 n, u32 *__restrict__ flags, u32 *__restrict__ rows) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  const u64 cmask = ((u64)1 << cb) - 1;
  for (; i < n; i += stride) { const u64 k = sKey[i]; flags[i] = (i == 0 || (k >> cb) != (sKey[i - 1] >> cb)) ? 1u : 0u; rows[i] = (u32)(k & cmask); }
}
__global__ void owner_distinct_kernel(const u64 *__restrict__ sHash, const u32 *__restrict__ rows, const u32 *__restrict__ flags, const u32 *__restrict__ ord,
                                      u64 n, u64 *__restrict__ dHash, u32 *__restrict__ dFirst, u32 *__restrict__ segStart, u32 *__restrict__ iota, int cb /* packed entries */) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) if (flags[i]) {
    const u32 d = ord[i];
    dHash[d] = sHash[i] >> cb; dFirst[d] = rows[i]; segStart[d] = (u32)i; iota[d] = d;   // stable sort => rows[i] is the lowest barcode of the hash
  }
}
// per barcode f: where its run starts in the distinct hashes sorted by first barcode, and how long it is (counting the
// "first seen here" hashes with one atomic per hash put 3 M adds on 10 k words: 0.5 ms; the sort is needed anyway)
__global__ void first_runs_kernel(const u32 *__restrict__ dFirstSorted, u32 U, u32 nB, u32 *__restrict__ myStart, u32 *__restrict__ cntFirst) {
  const u32 f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nB) return;
  u32 lo = 0, hi = U;                                        // first position with dFirstSorted >= f
  while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (dFirstSorted[mid] < f) lo = mid + 1; else hi = mid; }
  u32 lo2 = lo, hi2 = U;                                     // first position with dFirstSorted > f
  while (lo2 < hi2) { const u32 mid = (lo2 + hi2) >> 1; if (dFirstSorted[mid] <= f) lo2 = mid + 1; else hi2 = mid; }
  myStart[f] = lo; cntFirst[f] = lo2 - lo;
}
// per barcode c: how many new hashes it introduces in total and on owners before me
__global__ void first_totals_kernel(const u32 *__restrict__ all /* N x nB */, int N, int me, u32 nB, u32 *__restrict__ total, u32 *__restrict__ before) {
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nB) return;
  u32 t = 0, b = 0;
  for (int o = 0; o < N; ++o) { const u32 v = all[(size_t)o * nB + c]; if (o < me) b += v; t += v; }
  total[c] = t; before[c] = b;
}
// i-th of my distinct hashes in (first barcode, hash) order
__global__ void owner_index_kernel(const u32 *__restrict__ order, const u32 *__restrict__ dFirstSorted, u32 U, const u32 *__restrict__ base /* excl. scan of total */,
                                   const u32 *__restrict__ before, const u32 *__restrict__ myStart /* excl. scan of my cntFirst */, u32 *__restrict__ dIndex) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= U) return;
  const u32 f = dFirstSorted[i];
  dIndex[order[i]] = 1 + base[f] + before[f] + (i - myStart[f]);
}
__global__ void reply_kernel(const u32 *__restrict__ ord /* distinct ordinal per sorted position (exclusive scan of heads) */, const u32 *__restrict__ flags,
                             const u32 *__restrict__ q, u64 n, const u32 *__restrict__ dIndex, u32 *__restrict__ reply) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) reply[q[i]] = dIndex[ord[i] + flags[i] - 1];
}
// Replies by LOOK-UP (round 5): the owner's distinct hashes go into a table of its own — 2^T slots >= twice their number, home = the key's low T bits, linear probing,
// entry = index << 32 | (key >> T) << 8 | displacement (slot and entry together ARE the key: one 8-byte read per probe step; T >= keyBits - 24), the layout of the single-GPU
// index build's private table (stage_b.hip priv_insert_kernel) — and every received entry, in ARRIVAL order, reads its index there. The sort then carries no arrival positions
// (keys only: 16 instead of 24 bytes per entry and pass), and the transposition from hash order back to arrival order is 220 M random READS instead of reply_kernel's
// random 4-byte WRITES, which this chip does at half the rate (1/10 3 Gb set on one rank: 7.7 ms). A key that finds no slot within 256 of its home raises *fail: the caller
// then answers the old way.
constexpr u32 OWNER_MAX_DISP = 256;
constexpr u64 OWNER_EMPTY = ~0ULL;
__global__ void owner_table_insert_kernel(const u64 *__restrict__ dHash, const u32 *__restrict__ dIndex, u32 U, int T, u64 *__restrict__ table, u32 *__restrict__ fail, u32 maxDisp) {
  const u64 mask = ((u64)1 << T) - 1;
  for (u32 d = blockIdx.x * blockDim.x + threadIdx.x; d < U; d += gridDim.x * blockDim.x) {
    const u64 q = dHash[d], low = (q >> T) << 8, val = (u64)dIndex[d] << 32;
    u64 slot = q & mask; u32 disp = 0;
    for (; disp < maxDisp; ++disp, slot = (slot + 1) & mask)
      if (atomicCAS((unsigned long long *)&table[slot], (unsigned long long)OWNER_EMPTY, (unsigned long long)(val | low | disp)) == OWNER_EMPTY) break;
    if (disp == maxDisp) *fail = 1;
  }
}
__global__ void reply_lookup_kernel(const u64 *__restrict__ rKey /* arrival order: key << cb | block */, int cb, u64 n, const u64 *__restrict__ table, int T, u32 *__restrict__ reply, u32 *__restrict__ fail, u32 maxDisp) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  const u64 mask = ((u64)1 << T) - 1;
  for (; i < n; i += stride) {
    const u64 q = rKey[i] >> cb; const u32 want = (u32)((q >> T) << 8);
    u64 slot = q & mask; u32 disp = 0, ix = 0;
    for (; disp < maxDisp; ++disp, slot = (slot + 1) & mask) {
      const u64 e = table[slot];
      if ((u32)e == (want | disp)) { ix = (u32)(e >> 32); break; }
      if (e == OWNER_EMPTY) { disp = maxDisp; break; }         // (never behind an empty slot of its probe sequence)
    }
    if (disp == maxDisp) *fail = 1;
    reply[i] = ix;
  }
}
__global__ void scatter_key_kernel(const u32 *__restrict__ rIdx, const u32 *__restrict__ perm, const u32 *__restrict__ entRead, u64 n, u64 *__restrict__ key) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) { const u32 e = perm[i]; key[e] = ((u64)rIdx[i] << 32) | (u64)(entRead[e] & 0xFFFFu); }
}
__global__ void depth_of_kernel(const u32 *__restrict__ segStart, u32 U, u32 *__restrict__ depth) {
  const u32 d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d < U) depth[d] = segStart[d + 1] - segStart[d];
}
__global__ void scatter_depth_kernel(const u32 *__restrict__ idx, const u32 *__restrict__ depth, u64 n, u32 *__restrict__ hashDepth) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) hashDepth[idx[i]] = depth[i];
}
// hashDepth[] from the owners' (index, depth) runs when every run is SORTED BY INDEX (round 6): lane t takes element t / N of run t % N, so the N runs are walked in step —
// each covers the whole index range evenly (an owner's hashes get indices all over it), the places written at one moment lie within a narrow moving window, and the
// 4-byte writes of a table that no longer fits the Infinity Cache (0.9 GB at 226 M hashes: 23-28 G random writes/s, scratch/r5_scatter_rate.hip) combine in the L2
// instead of each going out as a partial sector. One element after the other of the concatenated runs (scatter_depth_kernel) is a random scatter: 1.05 ms per rank on
// the 1/4 3 Gb set, 2.1 ms on the 1/2 set, every rank scattering ALL U depths.
__global__ void scatter_depth_runs_kernel(const u32 *__restrict__ idx, const u32 *__restrict__ depth, const u64 *__restrict__ runStart /* N + 1 */, int N, u64 maxRun, u32 *__restrict__ hashDepth) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x, total = maxRun * (u64)N;
  for (; t < total; t += stride) {
    const u32 r = (u32)(t % (u64)N); const u64 p = t / (u64)N, a = runStart[r];
    if (a + p < runStart[r + 1]) hashDepth[idx[a + p]] = depth[a + p];
  }
}
__global__ void scatter_value_kernel(const u32 *__restrict__ idx, const u64 *__restrict__ hash /* hash / w */, u64 n, u64 w, u64 *__restrict__ hashValue) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) hashValue[idx[i]] = hash[i] * w;
}

// every collective of this file goes through these two: byte and time accounting per kind of exchange (common.hpp XchgStat)
// beside >= 0: the exchange goes on the context's exchange stream (the caller has called xFork() and will call xJoin() before the next collective and before the result
// is used) and runs beside the kernels of stage timer `beside` on the main stream
static int xchg_a2a(Ctx *c, Comm *cm, XchgId id, const void *dSend, const u64 *sendCnt, const u64 *sendOff, void *dRecv, const u64 *recvCnt, const u64 *recvOff, size_t eb, int beside = -1) {
  XchgStat &x = c->xs[id]; ++x.calls;
  u64 peer = 0;
  for (int p = 0; p < cm->n; ++p) if (p != cm->rank) { x.bytesOut += sendCnt[p] * eb; x.bytesIn += recvCnt[p] * eb; if (sendCnt[p] * eb > peer) peer = sendCnt[p] * eb; }
  x.maxPeerOut += peer;
  x.beside = beside;                                          // (the timers below stay on the MAIN stream: where the host blocks in the call — ranks as threads, TCP — they hold the wait, with RCCL they read ~0)
  Timer &t = c->stageOpen > 0 ? x.tIn : x.t;
  Timer *const ts = c->stageTop < T_COUNT ? &c->stageWait[c->stageTop] : nullptr;
  if (ts) c->tstart(*ts);
  c->tstart(t);
  const int rc = cm->alltoallv(c, dSend, sendCnt, sendOff, dRecv, recvCnt, recvOff, eb, beside >= 0 ? c->xStream : nullptr);
  c->tstop(t);
  if (ts) c->tstop(*ts);
  return rc;
}
static int xchg_hostGather(Ctx *c, Comm *cm, const void *send, void *recv, size_t bytes) {
  XchgStat &x = c->xs[X_HOST_COUNTS]; ++x.calls; x.bytesOut += (u64)bytes * (u64)(cm->n - 1); x.bytesIn += (u64)bytes * (u64)(cm->n - 1); x.maxPeerOut += bytes;
  Timer &t = c->stageOpen > 0 ? x.tIn : x.t;
  Timer *const ts = c->stageTop < T_COUNT ? &c->stageWait[c->stageTop] : nullptr;
  if (ts) c->tstart(*ts);
  c->tstart(t);
  const int rc = cm->allgatherHost(c, send, recv, bytes);
  c->tstop(t);
  if (ts) c->tstop(*ts);
  return rc;
}
static int bitsForS(u64 v) { int b = 1; while (b < 64 && (v >> b)) ++b; return b; }
static unsigned gridFor(u64 n) { return (unsigned)hmin<u64>(divUp(n ? n : 1, 256), 65535u * 2); }

struct ShardInfo { u64 barcodes, entries, records; };

// Where the hash owners' value ranges are cut. A canonical hash is min(hashF, hashR) (seqhash.c:67-68) of two values that are each uniform over the 2^(2k) hash values, so
// its density over x = key / nKeys is 2 (1 - x), NOT flat: equal value ranges (round 5) gave owner 0 of 8 the share 1 - (7/8)^2 = 23.4 % of all entries and distinct hashes
// and owner 7 1.6 % (bench line of round 5: indices_back max / sum = 0.2344; the owners' sort, numbering and list packing fell linearly with the rank). Equal SHARES are the
// quantiles of that density: F(x) = 1 - (1 - x)^2 = o / N  =>  x_o = 1 - sqrt(1 - o / N). Owner ranges stay value ranges (owner o's hashes all precede owner o + 1's), so nothing
// downstream changes; any monotone cut gives the same result, this one balances it. Integer arithmetic only (every rank must cut at the same values): sqrt((N - o) / N) as a
// 0.32 fixed-point number from a 64-bit integer square root.
static u64 isqrt64(u64 v) {
  u64 r = (u64)sqrtl((long double)v);
  while (r > 0xFFFFFFFFull || r * r > v) --r;
  while (r < 0xFFFFFFFFull && (r + 1) * (r + 1) <= v) ++r;
  return r;
}
static void ownerCuts(u64 nKeys, int N, bool equalRanges, u64 *low /* N + 1 */) {
  for (int o = 0; o <= N; ++o) {
    if (equalRanges || o == 0 || o == N) { low[o] = (u64)(((unsigned __int128)o * nKeys + (unsigned)N - 1) / (unsigned)N); continue; }    // ceil(o * nKeys / N)
    const u64 frac = (u64)((((unsigned __int128)(unsigned)(N - o)) << 64) / (unsigned)N);     // (N - o) / N as a 0.64 fixed-point number (o >= 1: below 2^64)
    const u64 root = isqrt64(frac);                                                             // sqrt of it as 0.32
    const u64 upper = (u64)(((unsigned __int128)nKeys * root) >> 32);                           // nKeys sqrt(1 - o / N), rounded down
    low[o] = nKeys - hmin<u64>(upper, nKeys);
  }
  for (int o = 1; o <= N; ++o) if (low[o] < low[o - 1]) low[o] = low[o - 1];                   // (monotone whatever the rounding did)
}

int shard_readFqb(Ctx *c, Comm *cm, const u32 *dRec, u64 nRec) {
  hipStream_t st = c->stream; PrimTemp pt;
  const int N = cm->n, me = cm->rank; const int k = c->prm.k;
  // ---- 1. local mosh extraction. The file's last barcode is never hashed: that is the last barcode of the last
  //         shard that holds any record (with fewer barcodes than ranks the trailing shards are empty). A shard without
  //         records holds no barcode; if every shard is empty, rank 0 keeps the reference's empty block 1 (hash10x.c:200-201).
  std::vector<u64> recsOf((size_t)N); { u64 mineRec = nRec; H10X_TRY(xchg_hostGather(c, cm, &mineRec, recsOf.data(), 8)); }
  bool laterRecords = false, anyRecords = false;
  for (int r = 0; r < N; ++r) { if (recsOf[r]) { anyRecords = true; if (r > me) laterRecords = true; } }
  if (c->optChunk > 0) {
    // the reference's chunk loop runs over the WHOLE file (hash10x.c:202-223): every rank replays it from the run starts of
    // all shards (same verdict everywhere) and keeps the merge points that fall into its own records
    std::vector<u64> st; std::vector<u32> zr;
    H10X_TRY(stageA_runStarts(c, dRec, nRec, st, zr));
    u64 mineN[2] = {(u64)st.size() - 1, (u64)zr.size()}; std::vector<u64> allN((size_t)2 * N);
    H10X_TRY(xchg_hostGather(c, cm, mineN, allN.data(), 16));
    u64 maxRuns = 0, maxZero = 0; for (int r = 0; r < N; ++r) { maxRuns = hmax(maxRuns, allN[2 * r]); maxZero = hmax(maxZero, allN[2 * r + 1]); }
    std::vector<u64> pad((size_t)maxRuns + maxZero + 1, 0), allPad((size_t)N * (maxRuns + maxZero + 1));
    for (size_t i = 0; i + 1 < st.size(); ++i) pad[i] = st[i];
    for (size_t i = 0; i < zr.size(); ++i) pad[maxRuns + i] = zr[i];
    H10X_TRY(xchg_hostGather(c, cm, pad.data(), allPad.data(), pad.size() * 8));
    std::vector<u64> starts; std::vector<u32> zeroRuns; std::vector<u64> recBase((size_t)N + 1, 0);
    for (int r = 0; r < N; ++r) {
      const u64 *p = allPad.data() + (size_t)r * pad.size(); const u32 runBase = (u32)starts.size();
      for (u64 i = 0; i < allN[2 * r]; ++i) starts.push_back(recBase[r] + p[i]);
      for (u64 i = 0; i < allN[2 * r + 1]; ++i) zeroRuns.push_back(runBase + (u32)p[maxRuns + i]);
      recBase[r + 1] = recBase[r] + recsOf[r];
    }
    starts.push_back(recBase[N]);
    std::vector<u64> merges;
    if (replayChunks(starts, zeroRuns, (u64)c->optChunk, merges, c->optChunkEof != 0)) return c->fail("chunkSize too small");       // hash10x.c:206
    c->mergePoints.clear();
    for (u64 m : merges) {
      for (int r = 1; r < N; ++r) if (m == recBase[r] && recsOf[r])
        return c->fail("an all-A barcode run ends at a chunk boundary that is also the boundary of shards %d and %d: the reference merges the two barcodes there (hash10x.c:212); cut the shards elsewhere", r - 1, r);
      if (m >= recBase[me] && m < recBase[me + 1]) c->mergePoints.push_back(m - recBase[me]);
    }
    c->replayDone = true;
  }
  DevBuf<u64> entHash; DevBuf<u32> entCode, entRead;
  H10X_TRY(stageA_run(c, dRec, nRec, entHash, entCode, entRead, laterRecords, anyRecords || me != 0));
  const u64 H = c->nEntries;
  ShardInfo mine{(u64)c->nBlocks - 1, H, nRec}; std::vector<ShardInfo> all((size_t)N);
  H10X_TRY(xchg_hostGather(c, cm, &mine, all.data(), sizeof mine));
  u64 codeBase = 0, totalBarcodes = 0;
  for (int r = 0; r < N; ++r) { if (r < me) codeBase += all[r].barcodes; totalBarcodes += all[r].barcodes; }
  if (totalBarcodes + 1 >= (1ULL << 32)) return c->fail("too many barcodes for this build");
  c->codeBase = (u32)codeBase; c->nBlocksGlobal = (u32)totalBarcodes + 1;
  const u32 nB = c->nBlocksGlobal;

  // ---- 2. my entries by hash owner, each owner's part in block = barcode order (owner_partition_kernel; the owners sort by
  //         hash anyway)
  c->tstart(T_SORT_HASH);
  // key and global block number of an entry in ONE word where they fit (as in the single-GPU index build, common.hpp: 63 bits): 8
  // instead of 12 bytes per entry on the wire, and the owner's barcode lists fall out of the sorted keys without a gather
  int cbG = 1; while (cbG < 32 && ((nB - 1) >> cbG)) ++cbG;
  const int pk = (!c->optNoPack && N <= PART_MAX_OWNERS && c->keyBits + cbG <= 63) ? cbG : 0;
  DevBuf<u64> sHash; DevBuf<u32> perm, sCodeG;
  const bool byPos = N <= PART_MAX_OWNERS;                   // the partition records where every entry went (posOf[e]) instead of which entry went where (perm[pos])
  H10X_HIP(c, sHash.alloc(H)); H10X_HIP(c, perm.alloc(H)); if (!pk) H10X_HIP(c, sCodeG.alloc(H));
  std::vector<u64> lowHash((size_t)N + 1), bound((size_t)N + 1);
  // (the entries hold hash / w, Ctx::keyInv: the owner ranges are cut in that space)
  const u64 nKeys = ((2 * k >= 64 ? ~0ULL : ((u64)1 << (2 * k)) - 1) / (u64)c->prm.w) + 1;
  ownerCuts(nKeys, N, c->optOwnerCut != 0, lowHash.data());
  DevBuf<u64> dLow, dBound; H10X_HIP(c, dLow.alloc((size_t)N + 1)); H10X_HIP(c, dBound.alloc((size_t)N + 1));
  H10X_HIP(c, hipMemcpyAsync(dLow.p, lowHash.data(), ((size_t)N + 1) * 8, hipMemcpyHostToDevice, st));
  if (N <= PART_MAX_OWNERS) {
    const u32 nBl = c->nBlocks; const size_t cells = (size_t)N * nBl;
    DevBuf<u32> cnt, off; H10X_HIP(c, cnt.alloc(cells + 1)); H10X_HIP(c, off.alloc(cells + 1));
    H10X_HIP(c, hipMemsetAsync(cnt.p, 0, (cells + 1) * 4, st));                  // block 0 (unused) and the sentinel stay 0
    const unsigned grid = hmin<u32>(nBl, 16384);
    if (H) owner_partition_kernel<0><<<grid, 256, 0, st>>>(entHash.p, entCode.p, c->blockOff.p, nBl, dLow.p, N, c->codeBase, cnt.p, nullptr, nullptr, nullptr, 0);
    H10X_TRY(prim_exclusive_scan_u32(c, pt, cnt.p, off.p, cells + 1));
    if (H) owner_partition_kernel<1><<<grid, 256, 0, st>>>(entHash.p, entCode.p, c->blockOff.p, nBl, dLow.p, N, c->codeBase, off.p, sHash.p, sCodeG.p, nullptr, pk, perm.p);
    partition_bounds_kernel<<<divUp((u64)N + 1, 256), 256, 0, st>>>(off.p, nBl, N, H, dBound.p);
    H10X_TRY(c->readback(bound.data(), dBound.p, ((size_t)N + 1) * 8));
    H10X_TRY(c->syncReadbacks());
  } else {                                                   // very many ranks: a stable one-pass radix partition on the owner number
    DevBuf<u32> io, oKey, oKeyS; H10X_HIP(c, io.alloc(H)); H10X_HIP(c, oKey.alloc(H)); H10X_HIP(c, oKeyS.alloc(H));
    if (H) { iota_kernel<<<gridFor(H), 256, 0, st>>>(io.p, H); owner_key_kernel<<<gridFor(H), 256, 0, st>>>(entHash.p, H, dLow.p, N, oKey.p); }
    H10X_TRY(prim_sort_pairs_u32_u32(c, pt, oKey.p, oKeyS.p, io.p, perm.p, H, 0, bitsForS((u64)N - 1)));
    if (H) gather_send_kernel<<<gridFor(H), 256, 0, st>>>(entHash.p, entCode.p, perm.p, H, c->codeBase, sHash.p, sCodeG.p);
    owner_key_bounds_kernel<<<divUp((u64)N + 1, 256), 256, 0, st>>>(oKeyS.p, H, N, dBound.p);
    H10X_HIP(c, hipMemcpyAsync(bound.data(), dBound.p, ((size_t)N + 1) * 8, hipMemcpyDeviceToHost, st));
    H10X_HIP(c, hipStreamSynchronize(st));
  }
  std::vector<u64> sendCnt((size_t)N), sendOff((size_t)N), matrix((size_t)N * N), recvCnt((size_t)N), recvOff((size_t)N);
  for (int o = 0; o < N; ++o) { sendOff[o] = bound[o]; sendCnt[o] = bound[o + 1] - bound[o]; }
  H10X_TRY(xchg_hostGather(c, cm, sendCnt.data(), matrix.data(), (size_t)N * 8));
  u64 M = 0;
  for (int r = 0; r < N; ++r) { recvCnt[r] = matrix[(size_t)r * N + me]; recvOff[r] = M; M += recvCnt[r]; }
  if (M >= (1ULL << 32)) return c->fail("%llu entries land on hash owner %d: over this build's 2^32 limit", (u64)M, me);
  DevBuf<u64> rHash; DevBuf<u32> rCode;
  H10X_HIP(c, rHash.alloc(M)); if (!pk) H10X_HIP(c, rCode.alloc(M));
  H10X_TRY(xchg_a2a(c, cm, X_ENTRIES, sHash.p, sendCnt.data(), sendOff.data(), rHash.p, recvCnt.data(), recvOff.data(), 8));
  if (!pk) H10X_TRY(xchg_a2a(c, cm, X_ENTRIES, sCodeG.p, sendCnt.data(), sendOff.data(), rCode.p, recvCnt.data(), recvOff.data(), 4));
  sHash.release(); sCodeG.release();

  // ---- 3. owner side: received runs are in rank (= barcode) order, a stable sort by hash keeps barcodes ascending
  // (packed entries: the replies are looked up, the sort carries nothing — see reply_lookup_kernel; the arrival positions ride along only in the unpacked form and when the knob asks)
  const bool byLookup = pk && c->optReplySort != 1 && c->keyBits <= 56;
  DevBuf<u64> oHash; DevBuf<u32> oQ; H10X_HIP(c, oHash.alloc(M));
  auto sortWithPositions = [&]() -> int {
    H10X_HIP(c, oQ.alloc(M));
    DevBuf<u32> qio; H10X_HIP(c, qio.alloc(M)); if (M) iota_kernel<<<gridFor(M), 256, 0, st>>>(qio.p, M);
    H10X_TRY(prim_sort_pairs_u64_u32(c, pt, rHash.p, oHash.p, qio.p, oQ.p, M, pk, pk + c->keyBits)); H10X_HIP(c, hipStreamSynchronize(st));
    return 0;
  };
  if (byLookup) H10X_TRY(prim_sort_keys_u64(c, pt, rHash.p, oHash.p, M, pk, pk + c->keyBits));
  else H10X_TRY(sortWithPositions());
  H10X_HIP(c, c->oRows.alloc(M));
  if (M && !pk) gather_u32_kernel<<<gridFor(M), 256, 0, st>>>(rCode.p, oQ.p, M, c->oRows.p);
  c->tstop(T_SORT_HASH);
  c->tstart(T_RANK);
  DevBuf<u32> flags, ord; H10X_HIP(c, flags.alloc(M + 1)); H10X_HIP(c, ord.alloc(M + 1));
  if (M && pk) heads_packed_kernel<<<gridFor(M), 256, 0, st>>>(oHash.p, pk, M, flags.p, c->oRows.p);
  else if (M) heads_kernel<<<gridFor(M), 256, 0, st>>>(oHash.p, M, flags.p);
  H10X_HIP(c, hipMemsetAsync(flags.p + M, 0, 4, st));
  H10X_TRY(prim_exclusive_scan_u32(c, pt, flags.p, ord.p, M + 1));
  u32 Uo = 0;
  H10X_TRY(c->readback(&Uo, ord.p + M, 4));
  H10X_TRY(c->syncReadbacks());
  DevBuf<u64> dHash; DevBuf<u32> dFirst, dio, cntFirst;
  H10X_HIP(c, dHash.alloc(Uo)); H10X_HIP(c, dFirst.alloc(Uo)); H10X_HIP(c, c->oSegStart.alloc((size_t)Uo + 1)); H10X_HIP(c, dio.alloc(Uo));
  H10X_HIP(c, cntFirst.alloc(nB));
  if (M) owner_distinct_kernel<<<gridFor(M), 256, 0, st>>>(oHash.p, c->oRows.p, flags.p, ord.p, M, dHash.p, dFirst.p, c->oSegStart.p, dio.p, pk);
  H10X_HIP(c, hipMemsetD32Async((hipDeviceptr_t)(c->oSegStart.p + Uo), (int)(u32)M, 1, st));          // end of the last segment

  // ---- 4. numbering: index = 1 + #(hashes first seen in an earlier barcode) + #(same barcode, smaller hash);
  //         hashes of owners before me are smaller, those after me larger
  DevBuf<u32> allFirst, total, before, base, myStart;
  H10X_HIP(c, allFirst.alloc((size_t)N * nB)); H10X_HIP(c, total.alloc((size_t)nB + 1)); H10X_HIP(c, before.alloc(nB));
  H10X_HIP(c, base.alloc((size_t)nB + 1)); H10X_HIP(c, myStart.alloc(nB));
  // my distinct hashes in (first barcode, hash) order: a stable sort by first barcode of the hash-ordered list; the runs of
  // equal first barcode give the per-barcode counts the numbering needs
  DevBuf<u32> order, dFirstSorted; H10X_HIP(c, order.alloc(Uo)); H10X_HIP(c, dFirstSorted.alloc(Uo)); H10X_HIP(c, c->oIndex.alloc(Uo));
  H10X_TRY(prim_sort_pairs_u32_u32(c, pt, dFirst.p, dFirstSorted.p, dio.p, order.p, Uo, 0, bitsForS(nB)));
  first_runs_kernel<<<divUp(nB, 256), 256, 0, st>>>(dFirstSorted.p, Uo, nB, myStart.p, cntFirst.p);
  { std::vector<u64> sc((size_t)N, nB), so((size_t)N, 0), rc((size_t)N, nB), ro((size_t)N);
    for (int r = 0; r < N; ++r) ro[r] = (u64)r * nB;
    H10X_TRY(xchg_a2a(c, cm, X_FIRST_COUNTS, cntFirst.p, sc.data(), so.data(), allFirst.p, rc.data(), ro.data(), 4)); }   // allgather
  first_totals_kernel<<<divUp(nB, 256), 256, 0, st>>>(allFirst.p, N, me, nB, total.p, before.p);
  H10X_HIP(c, hipMemsetAsync(total.p + nB, 0, 4, st));
  H10X_TRY(prim_exclusive_scan_u32(c, pt, total.p, base.p, (size_t)nB + 1));
  u32 U = 0;
  H10X_TRY(c->readback(&U, base.p + nB, 4));
  H10X_TRY(c->syncReadbacks());
  if ((u64)U + 1 > (((u64)1 << c->prm.B) >> 2) - 2) return c->fail("hashTableSize is too small");     // hash10x.c:149, same verdict on every rank
  if (Uo) owner_index_kernel<<<divUp(Uo, 256), 256, 0, st>>>(order.p, dFirstSorted.p, Uo, base.p, before.p, myStart.p, c->oIndex.p);
  c->oU = Uo; c->oM = M;

  // ---- 5. the index of every entry goes back to the rank that sent it, in the order it was sent
  DevBuf<u32> reply, rIdx; H10X_HIP(c, reply.alloc(M)); H10X_HIP(c, rIdx.alloc(H));
  bool looked = false;
  c->ctr.shard_reply_path = M ? 2 : 0;                       // what answered: 1 the look-up, 2 the scatter, 3 the scatter after a look-up that failed or was not affordable
  if (byLookup && M) {
    int T = 16; while (((u64)1 << T) < 2 * ((u64)Uo + 1) && T < 40) ++T;
    if (T < c->keyBits - 24) T = c->keyBits - 24;
    // knob 3 (tests): a displacement limit of 1 — the first key that finds its home slot taken fails the table, which is how a table that overflows looks
    const u32 maxDisp = c->optReplySort == 3 ? 1u : OWNER_MAX_DISP;
    DevBuf<u64> table; DevBuf<u32> failFlag; u32 failed = 0;
    // the table is an optimisation with a fall-back behind it: where it does not fit what is free at this, the index build's memory peak (rHash, oHash, oRows, flags, ord,
    // reply and rIdx are all live; 8 B x 2^T is 16 GB for 700 M distinct hashes), or its allocation fails, the owner answers by scatter instead of failing --readFQB
    size_t freeB = 0, totalB = 0; const size_t want = ((size_t)1 << T) * 8;
    bool fits = hipMemGetInfo(&freeB, &totalB) == hipSuccess && want <= freeB / 2;
    if (c->optReplySort == 4) fits = false;                   // (knob 4, tests: as if the table did not fit)
    if (fits && table.alloc((size_t)1 << T) != hipSuccess) { (void)hipGetLastError(); fits = false; }
    if (fits) {
      H10X_HIP(c, failFlag.alloc(1));
      H10X_HIP(c, hipMemsetAsync(table.p, 0xFF, want, st)); H10X_HIP(c, hipMemsetAsync(failFlag.p, 0, 4, st));
      if (Uo) owner_table_insert_kernel<<<hmin<u32>(divUp(Uo, 256), 16384), 256, 0, st>>>(dHash.p, c->oIndex.p, Uo, T, table.p, failFlag.p, maxDisp);
      reply_lookup_kernel<<<gridFor(M), 256, 0, st>>>(rHash.p, pk, M, table.p, T, reply.p, failFlag.p, maxDisp);
      H10X_TRY(c->readback(&failed, failFlag.p, 4));
      H10X_TRY(c->syncReadbacks());
      looked = !failed && c->optReplySort != 2;               // (knob 2, tests: a look-up that succeeded, answered by scatter all the same)
    }
    c->ctr.shard_reply_path = looked ? 1 : 3;
  }
  if (M && !looked) {
    if (byLookup) H10X_TRY(sortWithPositions());              // (oHash is written again with what it holds: same keys, stable — ord / flags stand)
    reply_kernel<<<gridFor(M), 256, 0, st>>>(ord.p, flags.p, oQ.p, M, c->oIndex.p, reply.p);
  }
  H10X_TRY(xchg_a2a(c, cm, X_INDEX_BACK, reply.p, recvCnt.data(), recvOff.data(), rIdx.p, sendCnt.data(), sendOff.data(), 4));
  c->hashNumber = U + 1; c->depthBound = nB;
  c->tstop(T_RANK);
  // ---- 6. everyone gets hashDepth[] (the depth filter needs it in full). hashValue[] and the probe table hashIndex[] are
  //         only wanted by --writeHash / the crib: they are built when the shards are gathered (shard_materializeTables),
  //         not on every rank after every --readFQB (at N = 8: 24 M hash values allgathered and inserted into a 512 MB
  //         table per rank, for nothing on the clustering path).
  //         Round 6: the two allgathers go on the exchange stream and run beside the ClusterHash records of step 7 (which need the replies only).
  c->tstart(T_PROBE);
  std::vector<u64> uo((size_t)N); { u64 u = Uo; H10X_TRY(xchg_hostGather(c, cm, &u, uo.data(), 8)); }
  std::vector<u64> sc((size_t)N, Uo), so((size_t)N, 0), rc((size_t)N), ro((size_t)N); u64 Utot = 0;
  for (int r = 0; r < N; ++r) { rc[r] = uo[r]; ro[r] = Utot; Utot += uo[r]; }
  if (Utot != U) return c->fail("sharded index: %llu distinct hashes gathered, %u numbered", (u64)Utot, U);
  DevBuf<u32> dDepth, gIdx, gDepth, sIdx, sDepth;
  H10X_HIP(c, dDepth.alloc(Uo)); H10X_HIP(c, gIdx.alloc(U)); H10X_HIP(c, gDepth.alloc(U));
  if (Uo) depth_of_kernel<<<divUp(Uo, 256), 256, 0, st>>>(c->oSegStart.p, Uo, dDepth.p);
  // my (index, depth) pairs travel sorted by index: every rank then fills hashDepth[] from N runs walked in step (scatter_depth_runs_kernel) instead of by a random scatter
  const bool runsByIndex = c->optOverlap != 0 && N > 1;      // (one rank: its own run IS the whole table in hash order — nothing to walk in step; knob 0 = round 5's form throughout)
  if (runsByIndex) {
    H10X_HIP(c, sIdx.alloc(Uo)); H10X_HIP(c, sDepth.alloc(Uo));
    H10X_TRY(prim_sort_pairs_u32_u32(c, pt, c->oIndex.p, sIdx.p, dDepth.p, sDepth.p, Uo, 0, bitsForS((u64)U + 1)));
  }
  const int besideClus = c->optOverlap ? (int)T_CLUSHASH : -1;
  XGuard xGuard(c);                                          // (an error return below must not park gIdx / gDepth / dDepth while the exchange stream still uses them)
  if (besideClus >= 0) H10X_TRY(c->xFork());
  H10X_TRY(xchg_a2a(c, cm, X_INDEX_DEPTH, runsByIndex ? sIdx.p : c->oIndex.p, sc.data(), so.data(), gIdx.p, rc.data(), ro.data(), 4, besideClus));
  H10X_TRY(xchg_a2a(c, cm, X_INDEX_DEPTH, runsByIndex ? sDepth.p : dDepth.p, sc.data(), so.data(), gDepth.p, rc.data(), ro.data(), 4, besideClus));
  c->tstop(T_PROBE);
  // ---- 7. the ClusterHash records of my blocks from the replies
  c->tstart(T_CLUSHASH);
  if (byPos) H10X_TRY(stageB_finishClusHashFromReplies(c, rIdx.p, perm.p, entRead.p));     // (perm holds posOf[] here)
  else { DevBuf<u64> key; H10X_HIP(c, key.alloc(H));
    if (H) scatter_key_kernel<<<gridFor(H), 256, 0, st>>>(rIdx.p, perm.p, entRead.p, H, key.p);
    H10X_TRY(stageB_finishClusHash(c, key)); }
  c->tstop(T_CLUSHASH);
  c->tstart(T_PROBE);
  H10X_TRY(c->xJoin());
  H10X_HIP(c, c->hashDepth.alloc((size_t)U + 1));
  H10X_HIP(c, hipMemsetAsync(c->hashDepth.p, 0, 4, st));
  if (U && runsByIndex) {
    DevBuf<u64> dRun; H10X_HIP(c, dRun.alloc((size_t)N + 1));
    std::vector<u64> run((size_t)N + 1); u64 maxRun = 0;   // (lives to the end of the function, past the stream synchronisation below)
    for (int r = 0; r < N; ++r) { run[r] = ro[r]; maxRun = hmax(maxRun, rc[r]); }
    run[N] = Utot;
    H10X_HIP(c, hipMemcpyAsync(dRun.p, run.data(), ((size_t)N + 1) * 8, hipMemcpyHostToDevice, st));
    scatter_depth_runs_kernel<<<gridFor(maxRun * (u64)N), 256, 0, st>>>(gIdx.p, gDepth.p, dRun.p, N, maxRun, c->hashDepth.p);
  } else if (U) scatter_depth_kernel<<<gridFor(U), 256, 0, st>>>(gIdx.p, gDepth.p, U, c->hashDepth.p);
  c->oHash.swap(dHash);                                      // this owner's distinct hashes, for shard_materializeTables
  c->hashValue.release(); c->hashIndex.release(); c->tablesPending = true;
  c->tstop(T_PROBE);
  H10X_HIP(c, hipStreamSynchronize(st));
  c->rows.release(); c->rowStart.release();                  // barcode lists arrive with --hashDepthRange
  c->ctr.distinct = U;
  c->comm = cm; c->sharded = true; c->haveState = true;
  c->segs.n = 1; c->segs.s[0] = BlockSeg{0, c->nBlocks, c->codeBase};
  c->rowShift = 0; c->ownerListsStale = false;
  H10X_TRY(shard_refreshLayout(c));
  if (c->nBlocksGlobal != nB) return c->fail("sharded layout: %u blocks counted, %u numbered", c->nBlocksGlobal, nB);
  return 0;
}

// ---------------------------------------------------------------------------------------- who holds which blocks
// Every rank's segments (owned local range, global number of its first block, entries), in file order. The sharded
// --writeHash, the reports and --clusterSplit's numbering all read this table; it is rebuilt whenever blocks move.
struct LayoutMsg { u32 n, pad; u32 localStart[MAX_SEGS], count[MAX_SEGS], globalBase[MAX_SEGS]; u64 entries[MAX_SEGS]; u64 records; };
int shard_refreshLayout(Ctx *c) {
  Comm *cm = c->comm; const int N = cm->n;
  LayoutMsg m; memset(&m, 0, sizeof m);
  m.n = (u32)c->segs.n; m.records = c->nRecords;
  u64 ends[2 * MAX_SEGS];
  for (int k = 0; k < c->segs.n; ++k) {
    const BlockSeg &sg = c->segs.s[k];
    const u32 skip = k == 0 && sg.count ? 1u : 0u;           // slot 0 of the first segment is nobody's block
    m.localStart[k] = sg.localStart + skip; m.count[k] = sg.count - skip; m.globalBase[k] = sg.globalBase + skip;
    H10X_TRY(c->readback(&ends[2 * k], c->blockOff.p + m.localStart[k], 8));
    H10X_TRY(c->readback(&ends[2 * k + 1], c->blockOff.p + m.localStart[k] + m.count[k], 8));
  }
  H10X_TRY(c->syncReadbacks());
  for (int k = 0; k < c->segs.n; ++k) m.entries[k] = ends[2 * k + 1] - ends[2 * k];
  std::vector<LayoutMsg> all((size_t)N);
  H10X_TRY(xchg_hostGather(c, cm, &m, all.data(), sizeof m));
  c->allSegs.clear();
  u64 blocks = 1, entries = 0, records = 0;
  for (int r = 0; r < N; ++r) {
    records += all[r].records;
    for (u32 k = 0; k < all[r].n && k < (u32)MAX_SEGS; ++k) {
      if (!all[r].count[k]) continue;
      c->allSegs.push_back(ShardSegInfo{(u32)r, all[r].localStart[k], all[r].count[k], all[r].globalBase[k], all[r].entries[k]});
      blocks += all[r].count[k]; entries += all[r].entries[k];
    }
  }
  std::sort(c->allSegs.begin(), c->allSegs.end(), [](const ShardSegInfo &a, const ShardSegInfo &b) { return a.globalBase < b.globalBase; });
  u64 expect = 1;
  for (const ShardSegInfo &g : c->allSegs) { if (g.globalBase != expect) return c->fail("sharded layout: block %llu follows block %u", (u64)expect - 1, g.globalBase); expect += g.count; }
  if (blocks >= (1ULL << 32)) return c->fail("too many barcode blocks for this build");
  c->nBlocksGlobal = (u32)blocks; c->nEntriesGlobal = entries; c->nRecordsGlobal = records;
  return 0;
}

// collective: hashValue[] (allgather of every owner's distinct hashes, by index) and the probe table on every rank
int shard_materializeTables(Ctx *c) {
  if (!c->tablesPending) return 0;
  hipStream_t st = c->stream; Comm *cm = c->comm;
  const int N = cm->n; const u32 Uo = c->oU, U = c->hashNumber - 1;
  std::vector<u64> uo((size_t)N); { u64 u = Uo; H10X_TRY(xchg_hostGather(c, cm, &u, uo.data(), 8)); }
  std::vector<u64> sc((size_t)N, Uo), so((size_t)N, 0), rc((size_t)N), ro((size_t)N); u64 Utot = 0;
  for (int r = 0; r < N; ++r) { rc[r] = uo[r]; ro[r] = Utot; Utot += uo[r]; }
  DevBuf<u32> gIdx; DevBuf<u64> gHash;
  H10X_HIP(c, gIdx.alloc(U)); H10X_HIP(c, gHash.alloc(U));
  H10X_TRY(xchg_a2a(c, cm, X_TABLES, c->oIndex.p, sc.data(), so.data(), gIdx.p, rc.data(), ro.data(), 4));
  H10X_TRY(xchg_a2a(c, cm, X_TABLES, c->oHash.p, sc.data(), so.data(), gHash.p, rc.data(), ro.data(), 8));
  H10X_HIP(c, c->hashValue.alloc((size_t)U + 1));
  H10X_HIP(c, hipMemsetAsync(c->hashValue.p, 0, 8, st));
  if (U) scatter_value_kernel<<<gridFor(U), 256, 0, st>>>(gIdx.p, gHash.p, U, (u64)c->prm.w, c->hashValue.p);
  H10X_TRY(stageB_buildProbeTable(c));
  H10X_HIP(c, hipStreamSynchronize(st));
  c->oHash.release(); c->tablesPending = false;
  return 0;
}

// ---------------------------------------------------------------------------------------- lists of in-range hashes
// Lists are sent with their length rounded up to a multiple of 2^shift entries (shift = 0 unless the in-range lists of the
// whole data set hold 2^32 entries or more): cluster_kernel keeps a list's offset as a 32-bit number of 2^shift-entry units.
__global__ void good_len_kernel(const u32 *__restrict__ oIndex, const u32 *__restrict__ segStart, const u8 *__restrict__ within, u32 U,
                                u32 *__restrict__ isGood, u32 *__restrict__ len) {
  const u32 d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d > U) return;
  const bool g = d < U && within[oIndex[d]] != 0;
  isGood[d] = g ? 1u : 0u; len[d] = g ? segStart[d + 1] - segStart[d] : 0u;
}
__global__ void pad_len_kernel(u32 *__restrict__ len, u32 n, u32 pad) {
  const u32 d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d < n) len[d] = (len[d] + pad) & ~pad;
}
// (the in-range hashes are a small part of an owner's distinct hashes — one in ten at yeast scale: the kernels that handle their
// lists walk goodId[], the in-range ordinals in ascending order, not all U hashes)
__global__ void good_ids_kernel(const u32 *__restrict__ isGood, const u32 *__restrict__ pos, u32 U, u32 *__restrict__ goodId) {
  const u32 d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d < U && isGood[d]) goodId[pos[d]] = d;
}
// GW lanes per list (16: four lists to a wave where the lists are short — depth range 6 - 45 on the 3 Gb set: a 64-lane workgroup per list ran a third full, 0.94 ms per rank
// of the 1/4 set at 8 ranks; 64 where they are long)
template <int GW>
__global__ __launch_bounds__(256)
void good_pack_kernel(const u32 *__restrict__ oIndex, const u32 *__restrict__ segStart, const u32 *__restrict__ oRows, const u32 *__restrict__ goodId, u32 nGood,
                      const u64 *__restrict__ off, const u32 *__restrict__ padLen, bool copyRows,
                      u32 *__restrict__ gIdx, u32 *__restrict__ gLen, u32 *__restrict__ gRows) {
  constexpr u32 GPB = 256 / GW;
  const u32 jl = threadIdx.x & (GW - 1);
  for (u32 g = blockIdx.x * GPB + threadIdx.x / GW; g < nGood; g += gridDim.x * GPB) {
    const u32 d = goodId[g];
    const u32 s = segStart[d], n = segStart[d + 1] - s; const u64 o = off[d];
    if (jl == 0) { gIdx[g] = oIndex[d]; gLen[g] = padLen[d]; }
    if (copyRows) for (u32 j = jl; j < n; j += GW) gRows[o + j] = oRows[s + j];
  }
}
// ---- delta-coded lists. A list is ascending global block numbers; between ranks it travels as 16-bit units: the first number in
// two units, then one unit per further entry holding the step from its predecessor — n + 1 units instead of 2 n — unless some
// step of the list does not fit 16 bits, in which case the list goes as it is. The mode is decided per list (no escape codes
// inside a list: the receiver rebuilds the numbers with a wave-wide prefix sum), a list's coded length is rounded up to whole
// 32-bit words, and (coded words | raw << 31) travels beside the list's index and padded length.
__global__ __launch_bounds__(WAVE) void delta_len_kernel(const u32 *__restrict__ segStart, const u32 *__restrict__ oRows, const u32 *__restrict__ goodId, u32 nGood,
                                                          u32 *__restrict__ encLen /* per in-range hash, + a closing 0 */, u32 *__restrict__ gEnc /* the same | raw << 31 */) {
  const int lane = threadIdx.x;
  for (u32 g = blockIdx.x; g < nGood; g += gridDim.x) {
    const u32 d = goodId[g];
    const u32 s = segStart[d], n = segStart[d + 1] - s;
    bool wide = false;
    for (u32 j = 1 + lane; j < n; j += WAVE) wide |= oRows[s + j] - oRows[s + j - 1] > 0xFFFFu;
    const u32 raw = __builtin_amdgcn_ballot_w64(wide) ? 1u : 0u;
    const u32 words = raw ? n : (n + 2) / 2;                 // n + 1 units of 16 bits
    if (lane == 0) { encLen[g] = words; gEnc[g] = words | (raw << 31); }
  }
  if (blockIdx.x == 0 && lane == 0) encLen[nGood] = 0;
}
__global__ __launch_bounds__(WAVE) void delta_pack_kernel(const u32 *__restrict__ segStart, const u32 *__restrict__ oRows, const u32 *__restrict__ goodId, u32 nGood,
                                                           const u64 *__restrict__ encOff, const u32 *__restrict__ gEnc, u32 *__restrict__ enc) {
  const int lane = threadIdx.x;
  for (u32 g = blockIdx.x; g < nGood; g += gridDim.x) {
    const u32 d = goodId[g];
    const u32 s = segStart[d], n = segStart[d + 1] - s; u32 *out = enc + encOff[g];
    if (gEnc[g] >> 31) { for (u32 j = lane; j < n; j += WAVE) out[j] = oRows[s + j]; continue; }
    // word 0 = the first number; word m >= 1 = steps of entries 2m - 1 (low half) and 2m (high half, 0 past the end)
    const u32 words = (n + 2) / 2;
    for (u32 m = lane; m < words; m += WAVE) {
      if (m == 0) { out[0] = oRows[s]; continue; }
      const u32 j = 2 * m - 1;
      const u32 lo = oRows[s + j] - oRows[s + j - 1], hi = j + 1 < n ? oRows[s + j + 1] - oRows[s + j] : 0u;
      out[m] = lo | (hi << 16);
    }
  }
}
// receiver: list i of the concatenated streams -> rows at its (padded) offset
__global__ __launch_bounds__(WAVE) void delta_unpack_kernel(const u32 *__restrict__ aIdx, const u32 *__restrict__ aEnc, const u64 *__restrict__ eOff, const u64 *__restrict__ aOff,
                                                             const u32 *__restrict__ hashDepth, u64 nLists, const u32 *__restrict__ enc, u32 *__restrict__ rows) {
  const int lane = threadIdx.x;
  for (u64 i = blockIdx.x; i < nLists; i += gridDim.x) {
    const u32 n = hashDepth[aIdx[i]]; const u32 *in = enc + eOff[i]; u32 *out = rows + aOff[i];
    if (aEnc[i] >> 31) { for (u32 j = lane; j < n; j += WAVE) out[j] = in[j]; continue; }
    u32 carry = in[0];                                       // value of the entry before this chunk's first step
    if (lane == 0) out[0] = carry;
    for (u32 j0 = 1; j0 < n; j0 += WAVE) {                  // entries j0 .. j0 + 63: step of entry j in unit j + 1
      const u32 j = j0 + lane;
      u32 step = 0;
      if (j < n) { const u32 w = in[(j + 1) >> 1]; step = ((j + 1) & 1) ? (w >> 16) : (w & 0xFFFFu); }
      u32 inc = step;
#pragma unroll
      for (int dd = 1; dd < WAVE; dd <<= 1) { const u32 o = (u32)__shfl_up((int)inc, dd); if (lane >= dd) inc += o; }
      if (j < n) out[j] = carry + inc;
      carry += (u32)__shfl((int)inc, WAVE - 1);
    }
  }
}
__global__ void coded_words_kernel(const u32 *__restrict__ aEnc, u64 n, u32 *__restrict__ words) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) words[i] = aEnc[i] & 0x7FFFFFFFu;
}
__global__ void row_start_kernel(const u32 *__restrict__ gIdx, const u64 *__restrict__ gOff, u64 n, u64 base, u64 *__restrict__ rowStart) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) rowStart[gIdx[i]] = base + gOff[i];
}

int shard_rebuildOwnerLists(Ctx *c);

// after hashWithinRangeBuild: every rank receives the barcode lists of all in-range hashes (allgather of the filtered lists)
int shard_exchangeRows(Ctx *c) {
  hipStream_t st = c->stream; PrimTemp pt; Comm *cm = c->comm;
  const int N = cm->n; const u32 Uo = c->oU;
  if (c->ownerListsStale) H10X_TRY(shard_rebuildOwnerLists(c));
  c->tstart(T_CSR);
  DevBuf<u32> isGood, len, pos; DevBuf<u64> off;
  H10X_HIP(c, isGood.alloc((size_t)Uo + 1)); H10X_HIP(c, len.alloc((size_t)Uo + 1)); H10X_HIP(c, pos.alloc((size_t)Uo + 1)); H10X_HIP(c, off.alloc((size_t)Uo + 1));
  good_len_kernel<<<divUp((u64)Uo + 1, 256), 256, 0, st>>>(c->oIndex.p, c->oSegStart.p, c->within.p, Uo, isGood.p, len.p);
  H10X_TRY(prim_exclusive_scan_u32(c, pt, isGood.p, pos.p, (size_t)Uo + 1));
  H10X_TRY(prim_exclusive_scan_u32_u64(c, pt, len.p, off.p, (size_t)Uo + 1));
  u32 nGoodMine = 0; u64 rawMine = 0;
  H10X_TRY(c->readback(&nGoodMine, pos.p + Uo, 4));
  H10X_TRY(c->readback(&rawMine, off.p + Uo, 8));
  H10X_TRY(c->syncReadbacks());
  // the list alignment: the smallest shift that keeps every offset, in 2^shift-entry units, below 2^32 — the same on every rank
  u64 mine[2] = {nGoodMine, rawMine}; std::vector<u64> all((size_t)2 * N);
  H10X_TRY(xchg_hostGather(c, cm, mine, all.data(), 16));
  u64 totG = 0, totRaw = 0; for (int r = 0; r < N; ++r) { totG += all[2 * r]; totRaw += all[2 * r + 1]; }
  const u64 fake = c->optRowsFakeBase > 0 ? (u64)c->optRowsFakeBase : 0;
  int shift = c->optRowShift >= 0 ? (int)c->optRowShift : 0;
  while (shift < 8 && ((fake + totRaw + totG * ((1ull << shift) - 1)) >> shift) >= (1ull << 32)) ++shift;
  if (((fake + totRaw + totG * ((1ull << shift) - 1)) >> shift) >= (1ull << 32))
    return c->fail("%llu barcode-list entries in the depth range: beyond 2^40, the limit of this build", (u64)totRaw);
  const u32 pad = (1u << shift) - 1;
  if (pad) {
    pad_len_kernel<<<divUp((u64)Uo + 1, 256), 256, 0, st>>>(len.p, Uo + 1, pad);
    H10X_TRY(prim_exclusive_scan_u32_u64(c, pt, len.p, off.p, (size_t)Uo + 1));
    H10X_TRY(c->readback(&mine[1], off.p + Uo, 8));
    H10X_TRY(c->syncReadbacks());
    H10X_TRY(xchg_hostGather(c, cm, mine, all.data(), 16));
  }
  const bool delta = c->optDeltaLists > 0 || (c->optDeltaLists < 0 && N > 1 && cm->slowLinks());
  const u32 nGoodU = (u32)mine[0];
  DevBuf<u32> gIdx, gLen, gRows, goodId;
  H10X_HIP(c, gIdx.alloc(mine[0])); H10X_HIP(c, gLen.alloc(mine[0])); H10X_HIP(c, goodId.alloc(mine[0])); if (!delta) H10X_HIP(c, gRows.alloc(mine[1]));
  if (Uo) good_ids_kernel<<<divUp(Uo, 256), 256, 0, st>>>(isGood.p, pos.p, Uo, goodId.p);
  if (nGoodU) {
    if (c->rangeHiMax && c->rangeHiMax <= 64) good_pack_kernel<16><<<hmin<u32>(divUp(nGoodU, 16), 65535u * 2), 256, 0, st>>>(c->oIndex.p, c->oSegStart.p, c->oRows.p, goodId.p, nGoodU, off.p, len.p, !delta, gIdx.p, gLen.p, gRows.p);
    else good_pack_kernel<64><<<hmin<u32>(divUp(nGoodU, 4), 65535u * 2), 256, 0, st>>>(c->oIndex.p, c->oSegStart.p, c->oRows.p, goodId.p, nGoodU, off.p, len.p, !delta, gIdx.p, gLen.p, gRows.p);
  }
  std::vector<u64> sc((size_t)N), so((size_t)N, 0), rc((size_t)N), ro((size_t)N); u64 nG = 0, nR = 0;
  for (int r = 0; r < N; ++r) { sc[r] = mine[0]; rc[r] = all[2 * r]; ro[r] = nG; nG += rc[r]; }
  DevBuf<u32> aIdx, aLen; H10X_HIP(c, aIdx.alloc(nG)); H10X_HIP(c, aLen.alloc(nG + 1));
  H10X_TRY(xchg_a2a(c, cm, X_LIST_HEADS, gIdx.p, sc.data(), so.data(), aIdx.p, rc.data(), ro.data(), 4));
  H10X_TRY(xchg_a2a(c, cm, X_LIST_HEADS, gLen.p, sc.data(), so.data(), aLen.p, rc.data(), ro.data(), 4));
  for (int r = 0; r < N; ++r) { rc[r] = all[2 * r + 1]; nR += rc[r]; }
  H10X_HIP(c, c->rows.alloc(nR + ROWS_PAD));
  DevBuf<u64> aOff; H10X_HIP(c, aOff.alloc(nG + 1));
  H10X_HIP(c, hipMemsetAsync(aLen.p + nG, 0, 4, st));
  H10X_TRY(prim_exclusive_scan_u32_u64(c, pt, aLen.p, aOff.p, nG + 1));
  if (!delta) {
    // Round 6: the lists themselves are wanted by --cluster only — the good lists need where a list STARTS (rowStart[], from the heads above), not what it holds — so they
    // travel on the exchange stream beside good_block_kernel; stageC_depthRange joins behind its last launch. The send buffer outlives this function in Ctx::xHold.
    const int beside = c->optOverlap ? (int)T_GOOD : -1;
    nR = 0; for (int r = 0; r < N; ++r) { sc[r] = mine[1]; ro[r] = nR; nR += rc[r]; }
    if (beside >= 0) H10X_TRY(c->xFork());
    H10X_TRY(xchg_a2a(c, cm, X_LIST_DATA, gRows.p, sc.data(), so.data(), c->rows.p, rc.data(), ro.data(), 4, beside));
    if (beside >= 0) c->xHold[0].swap(gRows);
  } else {
    // coded lengths, coded stream, the same allgather-shaped exchange, decode into the padded layout (see delta_len_kernel)
    DevBuf<u32> encLen, gEnc, enc, aEnc, encAll, aWords; DevBuf<u64> encOff, eOff;
    H10X_HIP(c, encLen.alloc((size_t)nGoodU + 1)); H10X_HIP(c, gEnc.alloc(mine[0])); H10X_HIP(c, encOff.alloc((size_t)nGoodU + 1));
    delta_len_kernel<<<hmin<u32>(nGoodU + 1, 65535u * 2), WAVE, 0, st>>>(c->oSegStart.p, c->oRows.p, goodId.p, nGoodU, encLen.p, gEnc.p);
    H10X_TRY(prim_exclusive_scan_u32_u64(c, pt, encLen.p, encOff.p, (size_t)nGoodU + 1));
    u64 encMine = 0; H10X_TRY(c->readback(&encMine, encOff.p + nGoodU, 8)); H10X_TRY(c->syncReadbacks());
    std::vector<u64> encTot((size_t)N); H10X_TRY(xchg_hostGather(c, cm, &encMine, encTot.data(), 8));
    H10X_HIP(c, enc.alloc(encMine));
    if (nGoodU) delta_pack_kernel<<<hmin<u32>(nGoodU, 65535u * 2), WAVE, 0, st>>>(c->oSegStart.p, c->oRows.p, goodId.p, nGoodU, encOff.p, gEnc.p, enc.p);
    H10X_HIP(c, aEnc.alloc(nG + 1));
    { u64 o = 0; for (int r = 0; r < N; ++r) { sc[r] = mine[0]; rc[r] = all[2 * r]; ro[r] = o; o += rc[r]; } }
    H10X_TRY(xchg_a2a(c, cm, X_LIST_HEADS, gEnc.p, sc.data(), so.data(), aEnc.p, rc.data(), ro.data(), 4));
    u64 nE = 0; for (int r = 0; r < N; ++r) { sc[r] = encMine; rc[r] = encTot[r]; ro[r] = nE; nE += rc[r]; }
    H10X_HIP(c, encAll.alloc(nE));
    H10X_TRY(xchg_a2a(c, cm, X_LIST_DATA, enc.p, sc.data(), so.data(), encAll.p, rc.data(), ro.data(), 4));
    H10X_HIP(c, eOff.alloc(nG + 1));
    H10X_HIP(c, hipMemsetAsync(aEnc.p + nG, 0, 4, st));
    H10X_HIP(c, aWords.alloc(nG + 1));
    coded_words_kernel<<<gridFor(nG + 1), 256, 0, st>>>(aEnc.p, nG + 1, aWords.p);
    H10X_TRY(prim_exclusive_scan_u32_u64(c, pt, aWords.p, eOff.p, nG + 1));
    if (nG) delta_unpack_kernel<<<(unsigned)hmin<u64>(nG, 65535u * 2), WAVE, 0, st>>>(aIdx.p, aEnc.p, eOff.p, aOff.p, c->hashDepth.p, nG, encAll.p, c->rows.p);
    c->ctr.list_words[0] = nR; c->ctr.list_words[1] = nE;
    H10X_HIP(c, hipStreamSynchronize(st));                   // the coded buffers go out of scope
  }
  H10X_HIP(c, c->rowStart.alloc((size_t)c->hashNumber + 1));
  H10X_HIP(c, hipMemsetAsync(c->rowStart.p, 0, ((size_t)c->hashNumber + 1) * 8, st));
  const u64 fakeAligned = (fake + pad) & ~(u64)pad;
  c->optRowsFakeBase = (int64_t)fakeAligned;                  // what stageC_cluster takes off the rows pointer
  if (nG) row_start_kernel<<<gridFor(nG), 256, 0, st>>>(aIdx.p, aOff.p, nG, fakeAligned, c->rowStart.p);
  H10X_HIP(c, hipStreamSynchronize(st));
  c->rowShift = shift;
  c->tstop(T_CSR);
  return 0;
}

// ---------------------------------------------------------------------------------------- owner lists after --clusterSplit
// clusterSplitCodes ends by rebuilding hashCodes from the blocks (hash10x.c:1008-1012): entries have moved to new blocks,
// hashDepth is unchanged. Sharded: every rank sends (hash index, global block) of each of its entries to the hash's owner,
// which sorts them into its lists again (same lengths, new barcode numbers).
__global__ void owner_of_index_kernel(const u32 *__restrict__ gIdx, const u64 *__restrict__ bound /* N+1 */, int N, u64 U, u16 *__restrict__ idxOwner) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < U; i += stride) {
    int lo = 0, hi = N;                                      // largest o with bound[o] <= i
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (bound[mid] <= i) lo = mid; else hi = mid; }
    idxOwner[gIdx[i]] = (u16)lo;
  }
}
__global__ void ord_of_index_kernel(const u32 *__restrict__ oIndex, u32 Uo, u32 *__restrict__ ordOf) {
  const u32 d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d < Uo) ordOf[oIndex[d]] = d;
}
// PASS 0: entries per owner (device totals); PASS 1: place (index << 32 | global block) into the owner's slice
template <int PASS>
__global__ __launch_bounds__(256)
void entry_owner_kernel(const h10x_clushash *__restrict__ ch, const u64 *__restrict__ blockOff, u32 nBlocks, SegMap segs, const u16 *__restrict__ idxOwner, int N,
                        unsigned long long *__restrict__ cursor /* N: totals (PASS 0) / next free position (PASS 1) */, u64 *__restrict__ out) {
  __shared__ u32 fill[PART_MAX_OWNERS];
  __shared__ unsigned long long base[PART_MAX_OWNERS];
  for (u32 b = blockIdx.x + 1; b < nBlocks; b += gridDim.x) {
    const u64 e0 = blockOff[b], e1 = blockOff[b + 1];
    const u32 g = segs.globalOf(b);
    __syncthreads();
    for (int o = threadIdx.x; o < N; o += blockDim.x) fill[o] = 0;
    __syncthreads();
    for (u64 e = e0 + threadIdx.x; e < e1; e += blockDim.x) atomicAdd(&fill[idxOwner[ch[e].hash]], 1u);
    __syncthreads();
    for (int o = threadIdx.x; o < N; o += blockDim.x) { const u32 n = fill[o]; if (n) base[o] = atomicAdd(&cursor[o], (unsigned long long)n); fill[o] = 0; }
    if (!PASS) continue;
    __syncthreads();
    for (u64 e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
      const u32 ix = ch[e].hash; const int o = idxOwner[ix];
      out[base[o] + atomicAdd(&fill[o], 1u)] = ((u64)ix << 32) | g;
    }
  }
}
__global__ void owner_key_of_kernel(u64 *__restrict__ pairs, u64 n, const u32 *__restrict__ ordOf) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) { const u64 v = pairs[i]; pairs[i] = ((u64)ordOf[(u32)(v >> 32)] << 32) | (v & 0xFFFFFFFFull); }
}
__global__ void low_word_kernel(const u64 *__restrict__ key, u64 n, u32 *__restrict__ out) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = (u32)key[i];
}

int shard_rebuildOwnerLists(Ctx *c) {
  hipStream_t st = c->stream; PrimTemp pt; Comm *cm = c->comm;
  const int N = cm->n, me = cm->rank; const u32 Uo = c->oU, U = c->hashNumber - 1;
  if (N > PART_MAX_OWNERS) return c->fail("clusterSplit on %d ranks: at most %d are supported", N, PART_MAX_OWNERS);
  c->tstart(T_CSR);
  // who owns which index: allgather of every owner's indices
  std::vector<u64> uo((size_t)N); { u64 u = Uo; H10X_TRY(xchg_hostGather(c, cm, &u, uo.data(), 8)); }
  std::vector<u64> sc((size_t)N, Uo), so((size_t)N, 0), rc((size_t)N), ro((size_t)N), bound((size_t)N + 1); u64 Utot = 0;
  for (int r = 0; r < N; ++r) { rc[r] = uo[r]; ro[r] = Utot; bound[r] = Utot; Utot += uo[r]; }
  bound[N] = Utot;
  if (Utot != U) return c->fail("sharded index: %llu distinct hashes gathered, %u numbered", (u64)Utot, U);
  DevBuf<u32> gIdx; DevBuf<u64> dBound; DevBuf<u16> idxOwner; DevBuf<u32> ordOf;
  H10X_HIP(c, gIdx.alloc(U)); H10X_HIP(c, dBound.alloc((size_t)N + 1)); H10X_HIP(c, idxOwner.alloc((size_t)U + 1)); H10X_HIP(c, ordOf.alloc((size_t)U + 1));
  H10X_TRY(xchg_a2a(c, cm, X_OWNER_LISTS, c->oIndex.p, sc.data(), so.data(), gIdx.p, rc.data(), ro.data(), 4));
  H10X_HIP(c, hipMemcpyAsync(dBound.p, bound.data(), ((size_t)N + 1) * 8, hipMemcpyHostToDevice, st));
  H10X_HIP(c, hipMemsetAsync(idxOwner.p, 0, ((size_t)U + 1) * 2, st));
  if (U) owner_of_index_kernel<<<gridFor(U), 256, 0, st>>>(gIdx.p, dBound.p, N, U, idxOwner.p);
  if (Uo) ord_of_index_kernel<<<divUp(Uo, 256), 256, 0, st>>>(c->oIndex.p, Uo, ordOf.p);
  gIdx.release();
  // my entries, by owner
  const u64 H = c->nEntries; const u32 nBl = c->nBlocks;
  DevBuf<unsigned long long> cursor; H10X_HIP(c, cursor.alloc((size_t)N)); H10X_HIP(c, hipMemsetAsync(cursor.p, 0, (size_t)N * 8, st));
  const unsigned grid = hmin<u32>(nBl ? nBl : 1, 16384);
  if (H) entry_owner_kernel<0><<<grid, 256, 0, st>>>(c->clusHash.p, c->blockOff.p, nBl, c->segs, idxOwner.p, N, cursor.p, nullptr);
  std::vector<u64> sendCnt((size_t)N), sendOff((size_t)N), matrix((size_t)N * N), recvCnt((size_t)N), recvOff((size_t)N);
  H10X_HIP(c, hipMemcpyAsync(sendCnt.data(), cursor.p, (size_t)N * 8, hipMemcpyDeviceToHost, st));
  H10X_HIP(c, hipStreamSynchronize(st));
  u64 acc = 0; for (int o = 0; o < N; ++o) { sendOff[o] = acc; acc += sendCnt[o]; }
  if (acc != H) return c->fail("owner lists: %llu of %llu entries placed", (u64)acc, (u64)H);
  H10X_HIP(c, hipMemcpyAsync(cursor.p, sendOff.data(), (size_t)N * 8, hipMemcpyHostToDevice, st));
  DevBuf<u64> sPair; H10X_HIP(c, sPair.alloc(H));
  if (H) entry_owner_kernel<1><<<grid, 256, 0, st>>>(c->clusHash.p, c->blockOff.p, nBl, c->segs, idxOwner.p, N, cursor.p, sPair.p);
  H10X_TRY(xchg_hostGather(c, cm, sendCnt.data(), matrix.data(), (size_t)N * 8));
  u64 M = 0; bool bad = false;
  for (int r = 0; r < N; ++r) { recvCnt[r] = matrix[(size_t)r * N + me]; recvOff[r] = M; M += recvCnt[r]; }
  for (int o = 0; o < N; ++o) { u64 m = 0; for (int r = 0; r < N; ++r) m += matrix[(size_t)r * N + o]; if (m >= (1ULL << 32)) bad = true; }   // the same verdict on every rank
  if (bad) return c->fail("more than 2^32 entries land on one hash owner: over this build's per-GPU limit");
  DevBuf<u64> rPair, rSorted; H10X_HIP(c, rPair.alloc(M)); H10X_HIP(c, rSorted.alloc(M));
  H10X_TRY(xchg_a2a(c, cm, X_OWNER_LISTS, sPair.p, sendCnt.data(), sendOff.data(), rPair.p, recvCnt.data(), recvOff.data(), 8));
  sPair.release();
  if (M != c->oM) return c->fail("owner lists: %llu entries arrived for %llu list slots", (u64)M, (u64)c->oM);
  if (M) owner_key_of_kernel<<<gridFor(M), 256, 0, st>>>(rPair.p, M, ordOf.p);
  H10X_TRY(prim_sort_keys_u64(c, pt, rPair.p, rSorted.p, M, 0, 32 + bitsForS(Uo)));
  H10X_HIP(c, c->oRows.alloc(M));
  if (M) low_word_kernel<<<gridFor(M), 256, 0, st>>>(rSorted.p, M, c->oRows.p);
  H10X_HIP(c, hipStreamSynchronize(st));
  c->ownerListsStale = false;
  c->tstop(T_CSR);
  return 0;
}

// ---------------------------------------------------------------------------------------- --clusterSplit, sharded
// stageC_split has moved the entries locally (children of local parent i at local nCodes - 1 + subBefore[i] + j). What is
// left is collective: the children's GLOBAL numbers — the reference appends them after all existing blocks in the order
// of their parents, i.e. in the order of the existing segments of all ranks by global number — then the layout table and
// the owners' lists.
struct SplitMsg { u32 n, pad; u32 globalBase[MAX_SEGS], children[MAX_SEGS]; };
int shard_split(Ctx *c, const u32 *dSubBefore, u32 totalSubLocal) {
  Comm *cm = c->comm; const int N = cm->n, me = cm->rank;
  const u32 nCodesOld = c->nBlocks - totalSubLocal;
  SplitMsg m; memset(&m, 0, sizeof m); m.n = (u32)c->segs.n;
  u32 edge[2 * MAX_SEGS];
  for (int k = 0; k < c->segs.n; ++k) {
    H10X_TRY(c->readback(&edge[2 * k], dSubBefore + c->segs.s[k].localStart, 4));
    H10X_TRY(c->readback(&edge[2 * k + 1], dSubBefore + c->segs.s[k].localStart + c->segs.s[k].count, 4));
  }
  H10X_TRY(c->syncReadbacks());
  for (int k = 0; k < c->segs.n; ++k) { m.globalBase[k] = c->segs.s[k].globalBase; m.children[k] = edge[2 * k + 1] - edge[2 * k]; }
  std::vector<SplitMsg> all((size_t)N);
  H10X_TRY(xchg_hostGather(c, cm, &m, all.data(), sizeof m));
  struct Par { u32 base, children; int rank, k; };
  std::vector<Par> pars; u64 added = 0; bool tooMany = false;
  for (int r = 0; r < N; ++r) {
    int segsAfter = (int)all[r].n;
    for (u32 k = 0; k < all[r].n; ++k) if (all[r].children[k]) { pars.push_back(Par{all[r].globalBase[k], all[r].children[k], r, (int)k}); ++segsAfter; added += all[r].children[k]; }
    if (segsAfter > MAX_SEGS) tooMany = true;
  }
  if (tooMany) return c->fail("clusterSplit on a sharded context: more than %d block segments on one rank (three splits in a row are supported)", MAX_SEGS);
  if ((u64)c->nBlocksGlobal + added >= (1ULL << 31)) return c->fail("clusterSplit would create %llu barcode blocks", (u64)c->nBlocksGlobal + added);
  std::sort(pars.begin(), pars.end(), [](const Par &a, const Par &b) { return a.base < b.base; });
  u32 next = c->nBlocksGlobal;                               // children are numbered from the old arrayMax on (hash10x.c:961-962)
  const int nOld = c->segs.n;
  for (const Par &p : pars) {
    if (p.rank == me) c->segs.s[c->segs.n++] = BlockSeg{nCodesOld + edge[2 * p.k], p.children, next};
    next += p.children;
  }
  // local order of the new segments = order of their parents' segments = what the loop above appended only if my old
  // segments were already in ascending global order, which they are (segments are created in that order)
  for (int k = nOld + 1; k < c->segs.n; ++k) if (c->segs.s[k].localStart < c->segs.s[k - 1].localStart) return c->fail("clusterSplit: segment order");
  H10X_TRY(shard_refreshLayout(c));
  if (c->nBlocksGlobal != next) return c->fail("clusterSplit: %u blocks numbered, %u laid out", next, c->nBlocksGlobal);
  c->rows.release(); c->rowStart.release(); c->rowShift = 0;   // the in-range lists arrive with the next --hashDepthRange
  c->ownerListsStale = true;
  return shard_rebuildOwnerLists(c);
}

// ---------------------------------------------------------------------------------------- --readHash onto shards
// The tables of a .hash file are on this rank's device (h10x_api.hip upload_state): hashIndex / hashValue / hashDepth of the
// whole set, and the blocks [codeBase + 1, codeBase + nBlocks) of the file with their ClusterHash records (any contiguous cut of
// the file's block order will do — blocks made by an earlier --clusterSplit are blocks like any other). What is missing are the
// hash owners' barcode lists: ownership is by INDEX range here (rank r owns indices 1 + U r / N ...), the list lengths are the
// depths, and the lists are filled by the same exchange that follows a --clusterSplit.
__global__ void iota_from_kernel(u32 *__restrict__ p, u32 n, u32 from) { const u32 i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = from + i; }
__global__ void add_base_kernel(const u64 *__restrict__ in, u32 n, u32 *__restrict__ out) { const u32 i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = (u32)in[i]; }
int shard_adoptLoadedState(Ctx *c, Comm *cm, u32 codeBase, u32 nBlocksGlobal) {
  hipStream_t st = c->stream; PrimTemp pt;
  const int N = cm->n, me = cm->rank; const u32 U = c->hashNumber - 1;
  c->comm = cm; c->sharded = true; c->codeBase = codeBase; c->nBlocksGlobal = nBlocksGlobal;
  c->segs.n = 1; c->segs.s[0] = BlockSeg{0, c->nBlocks, codeBase};
  c->tablesPending = false; c->rowShift = 0;
  const u32 lo = 1 + (u32)((u64)U * (u64)me / (u64)N), hi = 1 + (u32)((u64)U * ((u64)me + 1) / (u64)N), Uo = hi - lo;
  H10X_HIP(c, c->oIndex.alloc(Uo)); H10X_HIP(c, c->oSegStart.alloc((size_t)Uo + 1));
  if (Uo) iota_from_kernel<<<divUp(Uo, 256), 256, 0, st>>>(c->oIndex.p, Uo, lo);
  DevBuf<u64> off; DevBuf<u32> dep; H10X_HIP(c, off.alloc((size_t)Uo + 1)); H10X_HIP(c, dep.alloc((size_t)Uo + 1));
  if (Uo) H10X_HIP(c, hipMemcpyAsync(dep.p, c->hashDepth.p + lo, (size_t)Uo * 4, hipMemcpyDeviceToDevice, st));
  H10X_HIP(c, hipMemsetAsync(dep.p + Uo, 0, 4, st));
  H10X_TRY(prim_exclusive_scan_u32_u64(c, pt, dep.p, off.p, (size_t)Uo + 1));
  u64 M = 0;
  H10X_TRY(c->readback(&M, off.p + Uo, 8));
  H10X_TRY(c->syncReadbacks());
  u64 worst = M; { std::vector<u64> all((size_t)N); H10X_TRY(xchg_hostGather(c, cm, &M, all.data(), 8)); for (u64 x : all) worst = x > worst ? x : worst; }
  if (worst >= (1ULL << 32)) return c->fail("more than 2^32 entries land on one hash owner: over this build's per-GPU limit");   // the same verdict on every rank
  add_base_kernel<<<divUp((u64)Uo + 1, 256), 256, 0, st>>>(off.p, Uo + 1, c->oSegStart.p);
  c->oU = Uo; c->oM = M;
  H10X_TRY(shard_refreshLayout(c));
  if (c->nBlocksGlobal != nBlocksGlobal) return c->fail("sharded --readHash: %u blocks laid out, the file holds %u", c->nBlocksGlobal, nBlocksGlobal);
  c->haveState = true;
  c->ownerListsStale = true;
  return shard_rebuildOwnerLists(c);                         // also the consistency check: every owner must receive exactly its depths' worth of entries
}

// ---------------------------------------------------------------------------------------- gather for a single-GPU continuation
__global__ void blocks_nhash_kernel(const h10x_block *__restrict__ blocks, u32 nBlocks, u32 *__restrict__ nHash) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= nBlocks) nHash[i] = (i >= 1 && i < nBlocks) ? blocks[i].nHash : 0;
}

// collective; afterwards rank 0 holds blocks[] and clusHash[] of the whole data set in file order and is an unsharded
// context (its barcode lists are rebuilt from the gathered blocks). --writeHash and the reports do not need this
// (h10x_shard_segments + h10x_export_slice); it exists for continuing on one GPU.
int shard_gather(Ctx *c) {
  hipStream_t st = c->stream; PrimTemp pt; Comm *cm = c->comm;
  if (!c->sharded) return 0;
  H10X_TRY(shard_materializeTables(c));
  H10X_TRY(shard_refreshLayout(c));
  const int N = cm->n, me = cm->rank;
  // limits first, from the table every rank holds: the same verdict everywhere, before any data moves
  if (c->nEntriesGlobal >= (1ULL << 32)) return c->fail("%llu entries exceed this build's 2^32 limit on the gathering rank", (u64)c->nEntriesGlobal);
  const u64 totB = (u64)c->nBlocksGlobal - 1, totH = c->nEntriesGlobal;
  DevBuf<h10x_block> gBlocks; DevBuf<h10x_clushash> gClus;
  if (me == 0) { H10X_HIP(c, gBlocks.alloc(totB + 1)); H10X_HIP(c, hipMemsetAsync(gBlocks.p, 0, sizeof(h10x_block), st)); H10X_HIP(c, gClus.alloc(totH)); }
  // k-th segment of every rank per round (a rank's segments in local order)
  std::vector<std::vector<ShardSegInfo>> perRank((size_t)N);
  std::vector<u64> entryStart(c->allSegs.size()); { u64 a = 0; for (size_t i = 0; i < c->allSegs.size(); ++i) { entryStart[i] = a; a += c->allSegs[i].entries; } }
  std::vector<std::vector<u64>> perRankEntry((size_t)N);
  for (size_t i = 0; i < c->allSegs.size(); ++i) { perRank[c->allSegs[i].rank].push_back(c->allSegs[i]); perRankEntry[c->allSegs[i].rank].push_back(entryStart[i]); }
  for (int r = 0; r < N; ++r) {                              // local order
    std::vector<size_t> ix(perRank[r].size()); for (size_t i = 0; i < ix.size(); ++i) ix[i] = i;
    std::sort(ix.begin(), ix.end(), [&](size_t a, size_t b) { return perRank[r][a].localStart < perRank[r][b].localStart; });
    std::vector<ShardSegInfo> a; std::vector<u64> b; for (size_t i : ix) { a.push_back(perRank[r][i]); b.push_back(perRankEntry[r][i]); }
    perRank[r].swap(a); perRankEntry[r].swap(b);
  }
  for (int k = 0; k < MAX_SEGS; ++k) {
    std::vector<u64> sc((size_t)N, 0), so((size_t)N, 0), rc((size_t)N, 0), ro((size_t)N, 0), sc2((size_t)N, 0), so2((size_t)N, 0), rc2((size_t)N, 0), ro2((size_t)N, 0);
    bool any = false;
    for (int r = 0; r < N; ++r) if ((size_t)k < perRank[r].size()) {
      any = true;
      if (me == 0) { rc[r] = perRank[r][k].count; ro[r] = perRank[r][k].globalBase; rc2[r] = perRank[r][k].entries; ro2[r] = perRankEntry[r][k]; }
    }
    if (!any) break;
    if ((size_t)k < perRank[me].size()) {
      const ShardSegInfo &g = perRank[me][k];
      u64 e0 = 0; H10X_HIP(c, hipMemcpyAsync(&e0, c->blockOff.p + g.localStart, 8, hipMemcpyDeviceToHost, st)); H10X_HIP(c, hipStreamSynchronize(st));
      sc[0] = g.count; so[0] = g.localStart; sc2[0] = g.entries; so2[0] = e0;
    }
    H10X_TRY(xchg_a2a(c, cm, X_GATHER, c->blocks.p, sc.data(), so.data(), gBlocks.p, rc.data(), ro.data(), sizeof(h10x_block)));
    H10X_TRY(xchg_a2a(c, cm, X_GATHER, c->clusHash.p, sc2.data(), so2.data(), gClus.p, rc2.data(), ro2.data(), sizeof(h10x_clushash)));
  }
  if (me == 0) {
    c->blocks.swap(gBlocks); c->clusHash.swap(gClus);
    c->nBlocks = (u32)totB + 1; c->nEntries = totH; c->nRecords = c->nRecordsGlobal;
    DevBuf<u32> nh; H10X_HIP(c, nh.alloc((size_t)c->nBlocks + 1)); H10X_HIP(c, c->blockOff.alloc((size_t)c->nBlocks + 1));
    blocks_nhash_kernel<<<divUp((u64)c->nBlocks + 1, 256), 256, 0, st>>>(c->blocks.p, c->nBlocks, nh.p);
    H10X_TRY(prim_exclusive_scan_u32_u64(c, pt, nh.p, c->blockOff.p, (size_t)c->nBlocks + 1));
    H10X_HIP(c, hipStreamSynchronize(st));
    c->sharded = false; c->codeBase = 0; c->haveGood = false;  // a full, unsharded state from here on (good lists were per shard)
    c->maxBlockHashes = 0xFFFFFFFFu;                           // other ranks' blocks: unknown
    c->goodPos.release(); c->nGood.release(); c->goodEntries.release(); c->goodRow.release();
    c->segs.n = 1; c->segs.s[0] = BlockSeg{0, c->nBlocks, 0}; c->rowShift = 0; c->optRowsFakeBase = 0; c->allSegs.clear();
    c->oRows.release(); c->oSegStart.release(); c->oIndex.release(); c->oU = 0; c->oM = 0;
    H10X_TRY(stageB_buildCSR(c));                            // the barcode lists of the gathered state (fillHashTable)
  }
  H10X_TRY(cm->barrier(c));
  return 0;
}

// h10x_warm: the first launch of a kernel loads the code object of its translation unit (HIP loads them on first use); this one is launched ahead of time
__global__ void warm_shard_kernel() {}
void warm_shard(hipStream_t st) { warm_shard_kernel<<<1, 1, 0, st>>>(); }

}  // namespace h10x
