// stage_e.hip — what the text reports need, reduced on the device.
//
// --hashStats / --codeStats (hash10x.c:351-402) print histograms; --clusterReport (hash10x.c:870-952) prints one line per
// barcode and one per sub-cluster; --cribSummary (hash10x.c:1017-1061) prints counts per crib type. The reference walks
// its arrays on the host. Here the walks are kernels over the device-resident state and only their results travel:
// histograms, one record per block and per sub-cluster, type counts and two "seen" bitmaps — never clusHash itself
// (12 GB at the 200 M-pair scale). On a sharded context every rank reduces its own blocks; the host layer puts the pieces
// together in file order (h10x_host.c).
#include "common.hpp"
#include "prim.hpp"
#include "comm.hpp"

namespace h10x {

// ------------------------------------------------------------------------------------------ histograms
template <int WHICH /* 0 plain u32 values, 1 blocks[].nHash, 2 blocks[].nSubCluster */>
__global__ void hist_max_kernel(const void *__restrict__ src, u64 first, u64 n, u32 *__restrict__ out) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  u32 m = 0;
  for (; i < n; i += stride) {
    const u32 v = WHICH == 0 ? ((const u32 *)src)[first + i] : (WHICH == 1 ? ((const h10x_block *)src)[first + i].nHash : ((const h10x_block *)src)[first + i].nSubCluster);
    m = v > m ? v : m;
  }
  for (int s = 32; s; s >>= 1) m = max(m, (u32)__shfl_xor((int)m, s));
  if ((threadIdx.x & (WAVE - 1)) == 0 && m) atomicMax(out, m);
}
template <int WHICH>
__global__ void hist_fill_kernel(const void *__restrict__ src, u64 first, u64 n, u32 bins, unsigned long long *__restrict__ hist) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const u32 v = WHICH == 0 ? ((const u32 *)src)[first + i] : (WHICH == 1 ? ((const h10x_block *)src)[first + i].nHash : ((const h10x_block *)src)[first + i].nSubCluster);
    if (v < bins) atomicAdd(&hist[v], 1ull);
  }
}

static int hist_source(Ctx *c, int which, const void *&src, u64 &limit) {
  if (!c->haveState) return c->fail("no hash state loaded: use readFQB or readHash first");
  if (which == 0) { src = c->hashDepth.p; limit = c->hashNumber; }
  else if (which == 1 || which == 2) { src = c->blocks.p; limit = c->nBlocks; }
  else return c->fail("h10x_report_histogram: unknown source %d", which);
  return 0;
}

int stageE_histMax(Ctx *c, int which, u64 first, u64 count, u32 *maxValue) {
  const void *src; u64 limit; H10X_TRY(hist_source(c, which, src, limit));
  if (first > limit || count > limit - first) return c->fail("h10x_report_histogram: range %llu + %llu outside %llu", (u64)first, (u64)count, (u64)limit);
  hipStream_t st = c->stream;
  DevBuf<u32> m; H10X_HIP(c, m.alloc(1)); H10X_HIP(c, hipMemsetAsync(m.p, 0, 4, st));
  if (count) {
    const unsigned g = (unsigned)hmin<u64>(divUp(count, 256), 4096);
    if (which == 0) hist_max_kernel<0><<<g, 256, 0, st>>>(src, first, count, m.p);
    else if (which == 1) hist_max_kernel<1><<<g, 256, 0, st>>>(src, first, count, m.p);
    else hist_max_kernel<2><<<g, 256, 0, st>>>(src, first, count, m.p);
  }
  H10X_TRY(c->readback(maxValue, m.p, 4));
  return c->syncReadbacks();
}

int stageE_histogram(Ctx *c, int which, u64 first, u64 count, u32 bins, u64 *hist) {
  const void *src; u64 limit; H10X_TRY(hist_source(c, which, src, limit));
  if (first > limit || count > limit - first) return c->fail("h10x_report_histogram: range %llu + %llu outside %llu", (u64)first, (u64)count, (u64)limit);
  hipStream_t st = c->stream;
  DevBuf<unsigned long long> d; H10X_HIP(c, d.alloc(bins)); H10X_HIP(c, hipMemsetAsync(d.p, 0, (size_t)bins * 8, st));
  if (count) {
    const unsigned g = (unsigned)hmin<u64>(divUp(count, 256), 4096);
    if (which == 0) hist_fill_kernel<0><<<g, 256, 0, st>>>(src, first, count, bins, d.p);
    else if (which == 1) hist_fill_kernel<1><<<g, 256, 0, st>>>(src, first, count, bins, d.p);
    else hist_fill_kernel<2><<<g, 256, 0, st>>>(src, first, count, bins, d.p);
  }
  H10X_HIP(c, hipMemcpyAsync(hist, d.p, (size_t)bins * 8, hipMemcpyDeviceToHost, st));
  H10X_HIP(c, hipStreamSynchronize(st));
  return 0;
}

// ------------------------------------------------------------------------------------------ --clusterReport
// One workgroup per block. Per sub-cluster (labels 1 .. nSubCluster; a stale label beyond that only counts as clustered):
// hashes, hashes per crib type, the chromosome of its first located hash (crib types htA / htB / hom) with the position
// span of the hashes that agree with it, how many disagree ("OTHER") and the last ten of those in the reference's print
// order (descending position; position 0 ends the reference's linked list and is never printed: hash10x.c:914,941-944).
// Per read the label of its LAST labelled entry (readClus[] is overwritten in entry order, hash10x.c:899).
constexpr int REP_THREADS = 256;
struct RepCluster { u32 n, nt[5], firstLoc, pMin, pMax, nBad, nRead, cur; };
__global__ __launch_bounds__(REP_THREADS)
void cluster_report_kernel(const h10x_block *__restrict__ blocks, const u64 *__restrict__ blockOff, const h10x_clushash *__restrict__ clusHash,
                           const u32 *__restrict__ nGood /* null before --hashDepthRange */, u32 firstBlock, u32 nBlk,
                           const u8 *__restrict__ cribType, const int16_t *__restrict__ cribChr, const u16 *__restrict__ cribPos /* null: no crib */,
                           const u64 *__restrict__ readOff /* per block of the chunk: offset into readLast */, u32 *__restrict__ readLast /* zeroed */,
                           const u64 *__restrict__ clusterOff /* per block of the chunk */, h10x_block_rep *__restrict__ outB, h10x_cluster_rep *__restrict__ outC) {
  __shared__ RepCluster info[256];
  __shared__ u32 sClusHash, sClusRead;
  const u32 bi = blockIdx.x; if (bi >= nBlk) return;
  const u32 c = firstBlock + bi;
  const h10x_block blk = blocks[c];
  const u32 nHash = c ? blk.nHash : 0, nSub = blk.nSubCluster > 255 ? 255 : blk.nSubCluster, nRead = blk.nRead;
  const h10x_clushash *e = clusHash + blockOff[c];
  u32 *rl = readLast + readOff[bi];
  const int tid = threadIdx.x;
  const bool crib = cribType != nullptr;
  for (u32 j = tid; j <= nSub; j += REP_THREADS) { RepCluster z; memset(&z, 0, sizeof z); z.firstLoc = 0xFFFFFFFFu; z.pMin = 0xFFFFFFFFu; z.cur = 0xFFFFFFFFu; info[j] = z; }
  if (tid == 0) { sClusHash = 0; sClusRead = 0; }
  __syncthreads();
  // pass 1: counts, first located hash, last label per read
  u32 myClus = 0;
  for (u32 p = tid; p < nHash; p += REP_THREADS) {
    const h10x_clushash x = e[p]; const u32 cl = x.subCluster;
    if (!cl) continue;
    ++myClus;
    if (x.read < nRead) atomicMax(&rl[x.read], (p << 8) | cl);                // entries beyond 2^24 per block do not occur (nHash > 65535 is never clustered)
    if (cl > nSub) continue;
    atomicAdd(&info[cl].n, 1u);
    if (crib) {
      const u32 t = cribType[x.hash];
      atomicAdd(&info[cl].nt[t < 5 ? t : 0], 1u);
      if (t >= 1 && t <= 3) atomicMin(&info[cl].firstLoc, p);
    }
  }
  for (int s = 32; s; s >>= 1) myClus += (u32)__shfl_down((int)myClus, s);
  if ((tid & (WAVE - 1)) == 0 && myClus) atomicAdd(&sClusHash, myClus);
  __syncthreads();
  // pass 2: span and disagreeing hashes against the chromosome of the first located hash
  if (crib)
    for (u32 p = tid; p < nHash; p += REP_THREADS) {
      const h10x_clushash x = e[p]; const u32 cl = x.subCluster;
      if (!cl || cl > nSub) continue;
      const u32 t = cribType[x.hash];
      if (t < 1 || t > 3) continue;
      const u32 f = info[cl].firstLoc;
      if (p < f) continue;
      const int chr0 = cribChr[e[f].hash];
      if (cribChr[x.hash] == chr0) { const u32 hp = cribPos[x.hash]; atomicMin(&info[cl].pMin, hp); atomicMax(&info[cl].pMax, hp); }
      else atomicAdd(&info[cl].nBad, 1u);
    }
  // reads per label
  u32 myRead = 0;
  for (u32 r = tid; r < nRead; r += REP_THREADS) {
    const u32 v = rl[r];
    if (!v) continue;
    ++myRead;
    const u32 cl = v & 0xFFu;
    if (cl <= nSub) atomicAdd(&info[cl].nRead, 1u);
  }
  for (int s = 32; s; s >>= 1) myRead += (u32)__shfl_down((int)myRead, s);
  if ((tid & (WAVE - 1)) == 0 && myRead) atomicAdd(&sClusRead, myRead);
  __syncthreads();
  h10x_cluster_rep *oc = outC + clusterOff[bi];
  for (u32 j = 1 + tid; j <= nSub; j += REP_THREADS) {
    const RepCluster &r = info[j];
    h10x_cluster_rep o; memset(&o, 0, sizeof o);
    o.n = r.n; o.nRead = r.nRead; for (int t = 0; t < 5; ++t) o.nt[t] = r.nt[t];
    if (crib && r.firstLoc != 0xFFFFFFFFu) { o.chr = cribChr[e[r.firstLoc].hash]; o.pMin = (u16)r.pMin; o.pMax = (u16)r.pMax; }
    o.nBad = r.nBad;
    oc[j - 1] = o;
  }
  // the last ten disagreeing hashes of every cluster, highest position first
  bool anyBad = false;
  for (u32 j = 1; j <= nSub; ++j) if (info[j].nBad) { anyBad = true; break; }        // uniform: read after the barrier
  if (crib && anyBad)
    for (int round = 0; round < 10; ++round) {
      __syncthreads();
      for (u32 j = 1 + tid; j <= nSub; j += REP_THREADS) { info[j].pMax = info[j].cur; info[j].cur = 0; }   // pMax / cur: previous and next pick (spans are written out already)
      __syncthreads();
      for (u32 p = 1 + tid; p < nHash; p += REP_THREADS) {
        const h10x_clushash x = e[p]; const u32 cl = x.subCluster;
        if (!cl || cl > nSub || !info[cl].nBad || p >= info[cl].pMax) continue;
        const u32 t = cribType[x.hash];
        if (t < 1 || t > 3 || p <= info[cl].firstLoc) continue;
        if (cribChr[x.hash] != cribChr[e[info[cl].firstLoc].hash]) atomicMax(&info[cl].cur, p);
      }
      __syncthreads();
      for (u32 j = 1 + tid; j <= nSub; j += REP_THREADS) if (info[j].cur) { oc[j - 1].other[round] = e[info[j].cur].hash; oc[j - 1].nOtherListed = (u16)(round + 1); }
    }
  if (tid == 0) { h10x_block_rep b; b.nGood = nGood ? nGood[c] : 0; b.nClusHash = sClusHash; b.nClusRead = sClusRead; b.reserved = 0; outB[bi] = b; }
}
__global__ void report_sizes_kernel(const h10x_block *__restrict__ blocks, u32 firstBlock, u32 nBlk, u32 *__restrict__ nRead, u32 *__restrict__ nSub) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > nBlk) return;
  if (i == nBlk) { nRead[i] = 0; nSub[i] = 0; return; }
  const h10x_block b = blocks[firstBlock + i];
  nRead[i] = b.nRead; nSub[i] = b.nSubCluster > 255 ? 255 : b.nSubCluster;
}

int stageE_clusterReport(Ctx *c, u32 firstBlock, u32 nBlk, h10x_block_rep *hostB, h10x_cluster_rep *hostC, u64 clusterCap, u64 *nClusters) {
  hipStream_t st = c->stream; PrimTemp pt;
  if (!c->haveState) return c->fail("no hash state loaded: use readFQB or readHash first");
  if (firstBlock > c->nBlocks || nBlk > c->nBlocks - firstBlock) return c->fail("h10x_cluster_report: blocks %u + %u outside %u", firstBlock, nBlk, c->nBlocks);
  if (nClusters) *nClusters = 0;
  if (!nBlk) return 0;
  DevBuf<u32> nRead, nSub; DevBuf<u64> readOff, clusOff;
  H10X_HIP(c, nRead.alloc((size_t)nBlk + 1)); H10X_HIP(c, nSub.alloc((size_t)nBlk + 1)); H10X_HIP(c, readOff.alloc((size_t)nBlk + 1)); H10X_HIP(c, clusOff.alloc((size_t)nBlk + 1));
  report_sizes_kernel<<<divUp((u64)nBlk + 1, 256), 256, 0, st>>>(c->blocks.p, firstBlock, nBlk, nRead.p, nSub.p);
  H10X_TRY(prim_exclusive_scan_u32_u64(c, pt, nRead.p, readOff.p, (size_t)nBlk + 1));
  H10X_TRY(prim_exclusive_scan_u32_u64(c, pt, nSub.p, clusOff.p, (size_t)nBlk + 1));
  u64 totRead = 0, totClus = 0;
  H10X_TRY(c->readback(&totRead, readOff.p + nBlk, 8));
  H10X_TRY(c->readback(&totClus, clusOff.p + nBlk, 8));
  H10X_TRY(c->syncReadbacks());
  if (nClusters) *nClusters = totClus;
  if (totClus > clusterCap) return c->fail("h10x_cluster_report: %llu sub-clusters in these blocks, room for %llu", (u64)totClus, (u64)clusterCap);
  DevBuf<u32> readLast; DevBuf<h10x_block_rep> dB; DevBuf<h10x_cluster_rep> dC;
  H10X_HIP(c, readLast.alloc(totRead + 1)); H10X_HIP(c, dB.alloc(nBlk)); H10X_HIP(c, dC.alloc(totClus + 1));
  H10X_HIP(c, hipMemsetAsync(readLast.p, 0, (totRead + 1) * 4, st));
  cluster_report_kernel<<<nBlk, REP_THREADS, 0, st>>>(c->blocks.p, c->blockOff.p, c->clusHash.p, c->haveGood ? c->nGood.p : nullptr, firstBlock, nBlk,
                                                       c->haveCrib ? c->cribType.p : nullptr, c->haveCrib ? c->cribChr.p : nullptr, c->haveCrib ? c->cribPos.p : nullptr,
                                                       readOff.p, readLast.p, clusOff.p, dB.p, dC.p);
  H10X_HIP(c, hipGetLastError());
  H10X_HIP(c, hipMemcpyAsync(hostB, dB.p, (size_t)nBlk * sizeof(h10x_block_rep), hipMemcpyDeviceToHost, st));
  if (totClus) H10X_HIP(c, hipMemcpyAsync(hostC, dC.p, totClus * sizeof(h10x_cluster_rep), hipMemcpyDeviceToHost, st));
  H10X_HIP(c, hipStreamSynchronize(st));
  return 0;
}

// ------------------------------------------------------------------------------------------ --cribSummary
// entries per crib type in base blocks (clusterParent == 0) and in the blocks --clusterSplit made, and which hashes occur
// in each kind (bitmaps over the hash indices: the caller counts them per type, across ranks when sharded)
__global__ __launch_bounds__(256)
void crib_summary_kernel(const h10x_block *__restrict__ blocks, const u64 *__restrict__ blockOff, const h10x_clushash *__restrict__ ch, u32 nBlocks,
                         const u8 *__restrict__ cribType, unsigned long long *__restrict__ counts /* 5 base, 5 cluster, blocks base, blocks cluster */,
                         u32 *__restrict__ seenBase, u32 *__restrict__ seenCluster) {
  __shared__ unsigned long long loc[10];
  if (threadIdx.x < 10) loc[threadIdx.x] = 0;
  __syncthreads();
  for (u32 b = blockIdx.x + 1; b < nBlocks; b += gridDim.x) {
    const bool isCluster = blocks[b].clusterParent != 0;
    if (threadIdx.x == 0) atomicAdd(&counts[isCluster ? 11 : 10], 1ull);
    const u64 e0 = blockOff[b], e1 = blockOff[b + 1];
    u32 *seen = isCluster ? seenCluster : seenBase;
    for (u64 e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
      const u32 h = ch[e].hash; const u32 t = cribType[h];
      atomicAdd(&loc[(isCluster ? 5 : 0) + (t < 5 ? t : 0)], 1ull);
      const u32 bit = 1u << (h & 31);
      if (!(seen[h >> 5] & bit)) atomicOr(&seen[h >> 5], bit);
    }
  }
  __syncthreads();
  if (threadIdx.x < 10 && loc[threadIdx.x]) atomicAdd(&counts[threadIdx.x], loc[threadIdx.x]);
}

int stageE_cribSummary(Ctx *c, u64 *counts12, u32 *hostSeenBase, u32 *hostSeenCluster) {
  hipStream_t st = c->stream;
  if (!c->haveState) return c->fail("no hash state loaded: use readFQB or readHash first");
  if (!c->haveCrib) return c->fail("no crib: use cribBuild first");
  const size_t words = ((size_t)c->hashNumber + 31) / 32;
  DevBuf<unsigned long long> cnt; DevBuf<u32> sb, sc;
  H10X_HIP(c, cnt.alloc(12)); H10X_HIP(c, sb.alloc(words)); H10X_HIP(c, sc.alloc(words));
  H10X_HIP(c, hipMemsetAsync(cnt.p, 0, 96, st)); H10X_HIP(c, hipMemsetAsync(sb.p, 0, words * 4, st)); H10X_HIP(c, hipMemsetAsync(sc.p, 0, words * 4, st));
  if (c->nBlocks > 1) crib_summary_kernel<<<hmin<u32>(c->nBlocks, 8192), 256, 0, st>>>(c->blocks.p, c->blockOff.p, c->clusHash.p, c->nBlocks, c->cribType.p, cnt.p, sb.p, sc.p);
  H10X_HIP(c, hipGetLastError());
  H10X_HIP(c, hipMemcpyAsync(counts12, cnt.p, 96, hipMemcpyDeviceToHost, st));
  H10X_HIP(c, hipMemcpyAsync(hostSeenBase, sb.p, words * 4, hipMemcpyDeviceToHost, st));
  H10X_HIP(c, hipMemcpyAsync(hostSeenCluster, sc.p, words * 4, hipMemcpyDeviceToHost, st));
  H10X_HIP(c, hipStreamSynchronize(st));
  return 0;
}

// The entries of cribSummary's walk over the blocks (hash10x.c:1030-1046) as one word each, in clusHash order: bit 31 = the block was made by
// --clusterSplit (clusterParent != 0), bits 28-30 the crib type of the hash, bits 0-27 its index (indices stay below 2^28: -B <= 30). The host feeds them
// to a restatement of the reference's HASH objects, whose counts depend on the order of insertion (host/h10x_host.c: RefHash). One wave per block.
__global__ __launch_bounds__(256)
void crib_words_kernel(const h10x_block *__restrict__ blocks, const u64 *__restrict__ blockOff, const h10x_clushash *__restrict__ ch, u32 nBlocks,
                       const u8 *__restrict__ cribType, u64 first, u64 count, u32 *__restrict__ out) {
  const u32 lane = threadIdx.x & (WAVE - 1), wavesPerGrid = gridDim.x * (blockDim.x / WAVE);
  for (u32 b = 1 + blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE; b < nBlocks; b += wavesPerGrid) {
    u64 e0 = blockOff[b], e1 = blockOff[b + 1];
    if (e1 <= first || e0 >= first + count) continue;
    if (e0 < first) e0 = first;
    if (e1 > first + count) e1 = first + count;
    const u32 kind = blocks[b].clusterParent != 0 ? 0x80000000u : 0u;
    for (u64 e = e0 + lane; e < e1; e += WAVE) { const u32 h = ch[e].hash; const u32 t = cribType[h]; out[e - first] = kind | ((t < 5 ? t : 0u) << 28) | h; }
  }
}
int stageE_cribWords(Ctx *c, u64 first, u64 count, u32 *hostOut) {
  hipStream_t st = c->stream;
  if (!c->haveState) return c->fail("no hash state loaded: use readFQB or readHash first");
  if (!c->haveCrib) return c->fail("no crib: use cribBuild first");
  if (first > c->nEntries || count > c->nEntries - first) return c->fail("h10x_crib_words: range %llu + %llu outside %llu", first, count, (u64)c->nEntries);
  if (c->hashNumber > (1u << 28)) return c->fail("h10x_crib_words: %u hash indices do not fit the word's 28 bits", c->hashNumber);
  if (!count) return 0;
  DevBuf<u32> out; H10X_HIP(c, out.alloc(count));
  if (c->nBlocks > 1) crib_words_kernel<<<hmin<u32>(divUp(c->nBlocks, 4), 65535u), 256, 0, st>>>(c->blocks.p, c->blockOff.p, c->clusHash.p, c->nBlocks, c->cribType.p, first, count, out.p);
  H10X_HIP(c, hipGetLastError());
  H10X_HIP(c, hipMemcpyAsync(hostOut, out.p, count * 4, hipMemcpyDeviceToHost, st));
  H10X_HIP(c, hipStreamSynchronize(st));
  return 0;
}

// ------------------------------------------------------------------------------------------ host collectives for launchers
// (sums / maxima of small host arrays and a gather of byte strings to rank 0: what the host layer needs to assemble the
// reports and the sharded .hash header; no-ops on an unsharded context)
int shard_allreduceU64(Ctx *c, u64 *v, u32 n, int op /* 0 sum, 1 max */) {
  if (!c->sharded || !n) return 0;
  Comm *cm = c->comm; const int N = cm->n;
  std::vector<u64> all((size_t)N * n);
  H10X_TRY(cm->allgatherHost(c, v, all.data(), (size_t)n * 8));
  for (u32 i = 0; i < n; ++i) {
    u64 a = op ? 0 : 0;
    for (int r = 0; r < N; ++r) { const u64 x = all[(size_t)r * n + i]; a = op ? (x > a ? x : a) : a + x; }
    v[i] = a;
  }
  return 0;
}
int shard_gatherBytes(Ctx *c, const void *send, u64 nbytes, void *recv, u64 cap, u64 *counts) {
  hipStream_t st = c->stream;
  if (!c->sharded) {
    if (nbytes > cap) return c->fail("h10x_shard_gather_bytes: %llu bytes, room for %llu", (u64)nbytes, (u64)cap);
    if (nbytes) memcpy(recv, send, nbytes);
    if (counts) counts[0] = nbytes;
    return 0;
  }
  Comm *cm = c->comm; const int N = cm->n, me = cm->rank;
  std::vector<u64> all((size_t)N);
  H10X_TRY(cm->allgatherHost(c, &nbytes, all.data(), 8));
  u64 total = 0; for (int r = 0; r < N; ++r) { if (counts) counts[r] = all[r]; total += all[r]; }
  if (total > cap && me == 0) { /* every rank must still take part in the exchange below */ }
  DevBuf<unsigned char> dSend, dRecv;
  H10X_HIP(c, dSend.alloc(nbytes)); if (me == 0) H10X_HIP(c, dRecv.alloc(total));
  if (nbytes) H10X_HIP(c, hipMemcpyAsync(dSend.p, send, nbytes, hipMemcpyHostToDevice, st));
  std::vector<u64> sc((size_t)N, 0), so((size_t)N, 0), rc((size_t)N, 0), ro((size_t)N, 0);
  sc[0] = nbytes;
  if (me == 0) { u64 a = 0; for (int r = 0; r < N; ++r) { rc[r] = all[r]; ro[r] = a; a += all[r]; } }
  H10X_TRY(cm->alltoallv(c, dSend.p, sc.data(), so.data(), dRecv.p, rc.data(), ro.data(), 1));
  if (me == 0) {
    if (total > cap) return c->fail("h10x_shard_gather_bytes: %llu bytes, room for %llu", (u64)total, (u64)cap);
    if (total) H10X_HIP(c, hipMemcpyAsync(recv, dRecv.p, total, hipMemcpyDeviceToHost, st));
  }
  H10X_HIP(c, hipStreamSynchronize(st));
  return 0;
}

// h10x_warm: the first launch of a kernel loads the code object of its translation unit (HIP loads them on first use); this one is launched ahead of time
__global__ void warm_stageE_kernel() {}
void warm_stageE(hipStream_t st) { warm_stageE_kernel<<<1, 1, 0, st>>>(); }

}  // namespace h10x
