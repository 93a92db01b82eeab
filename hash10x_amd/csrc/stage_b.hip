// stage_b.hip — global hash <-> barcode index.
//
// Replaces the order-dependent half of processBlock (hashIndexFind(…,TRUE), ++hashDepth, the sort of
// clusHash by index; hash10x.c:139-152, 174-183) and fillHashTable (hash10x.c:317-347) with an
// order-independent formulation:
//   index(h)   = 1 + rank of (first barcode containing h, h)   — what hashNumber++ yields serially
//   hashDepth  = number of barcodes containing h
//   hashCodes  = those barcodes in ascending order               (one CSR instead of U mallocs)
//   hashIndex  = the open-addressing table sequential insertion in index order would leave behind:
//                key i sits in the first slot of its probe sequence not held by a smaller index, which
//                a parallel atomicMin-and-evict insertion reaches in any order.
// Device-wide radix sorts/scans come from rocPRIM (prim.hip); everything else is hand-written.
#include "common.hpp"
#include "prim.hpp"
#include <rocprim/block/block_radix_sort.hpp>

namespace h10x {

constexpr u32 SLOT_EMPTY = 0xFFFFFFFFu;
static int bitsFor(u64 maxValue) { int b = 1; while (b < 64 && (maxValue >> b)) ++b; return b; }

// ------------------------------------------------------------------------------------------ distinct hashes
// (position i starts a run iff the ordinal steps there: ord has n + 1 entries)
__global__ void seg_scatter_kernel(const u64 *__restrict__ sHash, const u32 *__restrict__ sCode,
                                   const u32 *__restrict__ ord, u64 n, u64 *__restrict__ dHash, u32 *__restrict__ dFirst,
                                   u32 *__restrict__ segStart, u32 *__restrict__ iota) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) if (ord[i + 1] != ord[i]) {
    const u32 d = ord[i];
    dHash[d] = sHash[i]; dFirst[d] = sCode[i]; segStart[d] = (u32)i; iota[d] = d;
  }
}
// the same from packed entries, which also leaves the barcode lists (the block numbers in sorted order) behind. Round 3: a distinct hash
// travels through the sort by first barcode as ONE 16-byte value (hash / w, start and end of its barcode list) — the thread at a run's head
// writes the first two, the thread at its tail the third — so that the index assignment reads its inputs in order instead of gathering
// hash, start and end of every distinct hash from three arrays in hash order (352 MB fetched for 36 MB of output, r2h).
__global__ void seg_scatter_packed_kernel(const u64 *__restrict__ sKey, int cb,
                                          const u32 *__restrict__ ord, u64 n, Val16 *__restrict__ dVal, u32 *__restrict__ dFirst, u32 *__restrict__ rows) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  const u64 cmask = ((u64)1 << cb) - 1;
  for (; i < n; i += stride) {
    const u64 k = sKey[i]; const u32 code = (u32)(k & cmask);
    rows[i] = code;
    const u32 d = ord[i], d1 = ord[i + 1];                   // d1 != d: a run starts here (its ordinal is d)
    if (d1 != d) { dVal[d].a = k >> cb; dVal[d].b = (u32)i; dFirst[d] = code; }
    if (i + 1 == n || ord[i + 2] != d1) dVal[d1 - 1].c = (u32)i + 1;   // a run starts at i + 1 (or the entries end): the run that holds i ends here
  }
}
// i-th distinct hash in (firstBarcode, hash) order gets index i + 1: from the sorted values, in order
__global__ void assign_index_v16_kernel(const Val16 *__restrict__ val, u32 U, u64 w /* val.a holds hash / w */, u64 *__restrict__ hashValue, u32 *__restrict__ hashDepth,
                                        u64 *__restrict__ rowStart) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { hashValue[0] = 0; hashDepth[0] = 0; rowStart[0] = 0; }
  if (i >= U) return;
  const Val16 v = val[i];
  hashValue[i + 1] = v.a * w; hashDepth[i + 1] = v.c - v.b; rowStart[i + 1] = v.b;
}

// i-th distinct hash in (firstBarcode, hash) order gets index i + 1
__global__ void assign_index_kernel(const u32 *__restrict__ order, const u64 *__restrict__ dHash, const u32 *__restrict__ segStart,
                                    u32 U, u64 w /* dHash holds hash / w */, u64 *__restrict__ hashValue, u32 *__restrict__ hashDepth, u64 *__restrict__ rowStart) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { hashValue[0] = 0; hashDepth[0] = 0; rowStart[0] = 0; }
  if (i >= U) return;
  const u32 d = order[i];
  hashValue[i + 1] = dHash[d] * w;
  hashDepth[i + 1] = segStart[d + 1] - segStart[d];
  rowStart[i + 1] = segStart[d];
}

// ------------------------------------------------------------------------------------------ probe table
// hash10x.c:139-152 probe sequence: start hash & mask, odd stride ((hash >> B) & mask) | 1
__global__ void probe_insert_kernel(const u64 *__restrict__ hashValue, u32 hashNumber, int B, u32 *__restrict__ table) {
  const u32 first = blockIdx.x * blockDim.x + threadIdx.x + 1;
  const u64 mask = ((u64)1 << B) - 1;
  for (u32 start = first; start < hashNumber; start += gridDim.x * blockDim.x) {
    u32 cur = start;
    u64 h = hashValue[cur];
    u64 slot = h & mask, step = ((h >> B) & mask) | 1;
    for (;;) {
      const u32 old = atomicMin(&table[slot], cur);
      if (old == SLOT_EMPTY) break;                          // took a free slot
      if (old > cur) {                                       // evicted a later index: it continues from here
        cur = old; h = hashValue[cur]; step = ((h >> B) & mask) | 1;
      }
      slot = (slot + step) & mask;
    }
  }
}
__global__ void probe_finish_kernel(u32 *__restrict__ table, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) if (table[i] == SLOT_EMPTY) table[i] = 0;
}

// The same table with the key beside the index — entry = index << qBits | hash / w, all ones = empty — while that fits 64
// bits (k = 21, w = 31: 38 + B - 2 <= 64 up to B = 28): a look-up then needs ONE random 8-byte read per probe step
// instead of the table word and then hashValue[index] behind it. Built with the same minimum-and-evict walk (entries
// compare by index first), hashIndex[] is its index column; the wide table lives only until the entries are looked up.
constexpr u64 SLOT_EMPTY64 = ~0ULL;
__global__ void probe_insert64_kernel(const u64 *__restrict__ hashValue, u32 hashNumber, int B, u64 w, int qBits, u64 *__restrict__ table) {
  const u32 first = blockIdx.x * blockDim.x + threadIdx.x + 1;
  const u64 mask = ((u64)1 << B) - 1, qmask = ((u64)1 << qBits) - 1;
  for (u32 start = first; start < hashNumber; start += gridDim.x * blockDim.x) {
    u64 h = hashValue[start];
    u64 cur = ((u64)start << qBits) | (h / w);
    u64 slot = h & mask, step = ((h >> B) & mask) | 1;
    for (;;) {
      const u64 old = atomicMin((unsigned long long *)&table[slot], (unsigned long long)cur);
      if (old == SLOT_EMPTY64) break;                        // took a free slot
      if (old > cur) {                                       // evicted a later index: it continues from here
        cur = old; h = (old & qmask) * w; step = ((h >> B) & mask) | 1;
      }
      slot = (slot + step) & mask;
    }
  }
}
// Round 6: where key + index do not fit 64 bits (k = 21, w = 31: B = 29, 30) the wide table still works if an entry holds, instead of the key, what the SLOT does not
// already say: entry = index << SH | (hash >> B) << PB | probe number. The probe sequence of the reference (start hash & mask, stride ((hash >> B) & mask) | 1) makes slot,
// hash >> B and the probe number together the hash: an evicted entry knows its stride and its next probe number without its low bits, a look-up compares
// (hash >> B, probe number) at every step — one random 8-byte read per step in the reference's OWN geometry, and hashIndex[] is the index column of the same table
// (probe_finish64_kernel). Before, B = 29 / 30 built two tables: hashIndex[] by 4-byte minimum-and-evict with hashValue[] re-read on every eviction (6 / 15 ms on
// configs[2] / the 3 Gb set) and the look-up table of the library's own (priv_insert_kernel: 7 / 16 ms). PB = 66 - 2k bits of probe number (24 at k = 21); a probe
// sequence longer than 2^PB - 1 raises *fail and the caller builds the two tables as before.
__global__ void probe_insertP_kernel(const u64 *__restrict__ hashValue, u32 hashNumber, int B, int SH, int PB, u64 *__restrict__ table, u32 *__restrict__ fail) {
  const u32 first = blockIdx.x * blockDim.x + threadIdx.x + 1;
  const u64 mask = ((u64)1 << B) - 1, pmask = ((u64)1 << PB) - 1, umask = ((u64)1 << (SH - PB)) - 1;
  for (u32 start = first; start < hashNumber; start += gridDim.x * blockDim.x) {
    const u64 h = hashValue[start];
    u64 cur = ((u64)start << SH) | ((h >> B) << PB);         // probe number 0
    u64 slot = h & mask, step = ((h >> B) & mask) | 1;
    for (;;) {
      const u64 old = atomicMin((unsigned long long *)&table[slot], (unsigned long long)cur);
      if (old == SLOT_EMPTY64) break;                        // took a free slot
      if (old > cur) { cur = old; step = (((cur >> PB) & umask) & mask) | 1; }   // evicted a later index: it continues from here, with ITS stride and probe number
      if ((cur & pmask) == pmask) { *fail = 1; break; }
      ++cur;                                                 // next probe
      slot = (slot + step) & mask;
    }
  }
}
__global__ void probe_finish64_kernel(const u64 *__restrict__ table64, u64 n, int qBits, u32 *__restrict__ table) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) { const u64 e = table64[i]; table[i] = e == SLOT_EMPTY64 ? 0u : (u32)(e >> qBits); }
}
__global__ void lookup_pack64_kernel(const u64 *__restrict__ entHash /* hash / w, << cb if packed */, const u32 *__restrict__ entRead, u64 n,
                                     const u64 *__restrict__ table64, int B, u64 w, int qBits, int cb, u64 *__restrict__ key) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  const u64 mask = ((u64)1 << B) - 1, qmask = ((u64)1 << qBits) - 1;
  for (; i < n; i += stride) {
    const u64 q = entHash[i] >> cb, h = q * w;
    u64 slot = h & mask; const u64 step = ((h >> B) & mask) | 1;
    u64 e;
    while ((e = table64[slot]) != SLOT_EMPTY64 && (e & qmask) != q) slot = (slot + step) & mask;
    const u32 ix = e == SLOT_EMPTY64 ? 0u : (u32)(e >> qBits);
    key[i] = ((u64)ix << 32) | (u64)(entRead[i] & 0xFFFFu);
  }
}

// Where key and index do not fit one 64-bit word of the reference-shaped table (k = 21, w = 31: B >= 29) the entries are looked up in a table of this
// library's own: 2^T slots >= twice the distinct hashes, home slot = the low T bits of q = hash / w, linear probing, and an entry says how far from home it
// sits — entry = index << 32 | (q >> T) << 8 | displacement, all ones = empty — so slot and entry together ARE q: one random 8-byte read per probe step
// (neighbouring slots share a 64-byte sector) instead of hashIndex[slot] and then hashValue[index] behind it, each a sector of its own: the look-ups of
// 1.47 G entries took 56 ms of the 82 ms of clushash_build at 200 M read pairs. Entries never move; a hash that finds no free slot within 256 of its home
// is simply not in the table, and a look-up that does not find its hash within 256 slots asks the reference-shaped pair instead.
constexpr u32 PRIV_MAX_DISP = 256;
__global__ void priv_insert_kernel(const u64 *__restrict__ hashValue, u32 hashNumber, u64 w, int T, u64 *__restrict__ table) {
  const u32 first = blockIdx.x * blockDim.x + threadIdx.x + 1;
  const u64 mask = ((u64)1 << T) - 1;
  for (u32 i = first; i < hashNumber; i += gridDim.x * blockDim.x) {
    const u64 q = hashValue[i] / w, low = ((q >> T) << 8);
    u64 slot = q & mask;
    for (u32 d = 0; d < PRIV_MAX_DISP; ++d, slot = (slot + 1) & mask)
      if (atomicCAS((unsigned long long *)&table[slot], (unsigned long long)SLOT_EMPTY64, (unsigned long long)(((u64)i << 32) | low | d)) == SLOT_EMPTY64) break;
  }
}
// (e0: the entry at the home slot, read by the caller — who has the home reads of all its entries in flight together)
__device__ __forceinline__ u32 priv_find(const u64 *__restrict__ table, int T, u64 q, const u32 *__restrict__ hashIndex, const u64 *__restrict__ hashValue, int B, u64 w, u64 e0) {
  const u64 mask = ((u64)1 << T) - 1;
  const u32 want = (u32)((q >> T) << 8);
  u64 slot = q & mask;
  for (u32 d = 0; d < PRIV_MAX_DISP; ++d, slot = (slot + 1) & mask) {
    const u64 e = d ? table[slot] : e0;
    if ((u32)e == (want | d)) return (u32)(e >> 32);
    if (e == SLOT_EMPTY64) break;                            // (not behind an empty slot of its probe sequence — unless it was never put in)
  }
  return probe_find(hashIndex, hashValue, B, q * w);
}

// ------------------------------------------------------------------------------------------ clusHash
__global__ void lookup_pack_kernel(const u64 *__restrict__ entHash, const u32 *__restrict__ entRead, u64 n,
                                   const u32 *__restrict__ table, const u64 *__restrict__ hashValue, int B, u64 w /* entHash holds hash / w */, int cb, u64 *__restrict__ key) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const u32 ix = probe_find(table, hashValue, B, (entHash[i] >> cb) * w);
    key[i] = ((u64)ix << 32) | (u64)(entRead[i] & 0xFFFFu);  // ClusterHash.read is U16 (hash10x.c:37,180)
  }
}
__global__ void offsets32_kernel(const u64 *__restrict__ off, u32 n, u32 *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (u32)off[i];
}
__global__ void write_clushash_kernel(const u64 *__restrict__ key, u64 n, h10x_clushash *__restrict__ out) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const u64 v = key[i];
    h10x_clushash c; c.hash = (u32)(v >> 32); c.read = (u16)(v & 0xFFFF); c.subCluster = 0; c.flags = 0;   // zeroed: SURVEY F4
    out[i] = c;
  }
}


// ------------------------------------------------------------------------------------------ clusHash, one workgroup per block
// A block's entries (1500-2000 at 200-250 read pairs per barcode) fit a workgroup: look every entry up (THREADS x IPT probes
// in flight per workgroup, IPT independent ones per lane), sort the block by index in LDS (rocPRIM block radix sort: 8 bits
// per pass) and write the ClusterHash records — one kernel and one pass over the entries instead of look-up kernel, key
// array, device-wide segmented sort and record kernel. Three launch classes by block size (2048 / 4096 / 8192 entries);
// a data set with a larger block takes the device-wide path for everything.
__global__ void block_class_kernel(const u64 *__restrict__ blockOff, u32 nBlocks, u32 *__restrict__ lists, u32 *__restrict__ counts) {
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & (WAVE - 1);
  int cls = -1;
  if (c < nBlocks) { const u64 n = blockOff[c + 1] - blockOff[c]; cls = n <= BLOCK_SORT_CAP0 ? 0 : (n <= BLOCK_SORT_CAP1 ? 1 : (n <= BLOCK_SORT_MAX ? 2 : -1)); }
#pragma unroll
  for (int k = 0; k < 3; ++k) {                              // one atomic per wave and class
    const u64 bal = __ballot(cls == k);
    if (!bal) continue;
    u32 base = 0;
    if (lane == 0) base = atomicAdd(&counts[k], (u32)__popcll(bal));
    base = (u32)__shfl((int)base, 0);
    if (cls == k) lists[(size_t)k * nBlocks + base + (u32)__popcll(bal & ((1ULL << lane) - 1))] = c;
  }
}
int stageB_blockClassLists(Ctx *c, DevBuf<u32> &lists, DevBuf<u32> &counts) {
  H10X_HIP(c, lists.alloc((size_t)3 * c->nBlocks)); H10X_HIP(c, counts.alloc(4));
  H10X_HIP(c, hipMemsetAsync(counts.p, 0, 16, c->stream));
  if (c->nBlocks) block_class_kernel<<<divUp(c->nBlocks, 256), 256, 0, c->stream>>>(c->blockOff.p, c->nBlocks, lists.p, counts.p);
  return 0;
}

// A block's entries ordered by hash index in ONE counting pass (round 5; rocPRIM's block radix sort took four 8-bit passes over the 29-bit indices and was 2/3 of
// clushash_build — the look-ups inside the kernel already ran at the chip's random-read rate). What the keys of a block look like (scratch/r5_bucket_stats.py):
//   * they are distinct (a barcode's mosh set holds a hash once);
//   * the hashes FIRST seen in this barcode got consecutive indices (stage B numbers by first barcode, then hash): 8-17 % of the entries, nearly all of them in the
//     first barcodes of a file, form a run of consecutive integers that ends at the block's largest key — their place follows from their value alone;
//   * the rest is spread over [1, run start) with a bias towards low indices (repeats are met early): cut into NB buckets by a shift, a key shares its bucket with 8
//     others on average (the largest bucket of a block: 20-70 keys).
// So: largest key -> bitmap of (largest - key) -> length of the top run (first clear bit); the other keys are counted into buckets (returning LDS add = arrival rank),
// the counts scanned, the keys dropped into their bucket's range, and every key finds its rank by comparing with the keys of its bucket. Equal keys (cannot happen on
// consistent data: several look-ups that find nothing and return 0, corrupt indices) are ordered by arrival inside a bucket; the top run places by VALUE, so its members are
// counted and a run that holds a key twice is given up (every key then goes through the buckets): the result is a permutation whatever comes in. In: striped or any arrangement, n valid entries (entry e = j * THREADS + tid
// valid iff e < n); out: sorted, striped.
template <int THREADS, int IPT> struct BlockIndexSort {
  static constexpr int CAP = THREADS * IPT;
  static constexpr int NB = 4096, LOGNB = 12;               // (64 KB of static LDS at most: 16 KB of counters beside 6 bytes per entry)
  static constexpr int CPT = NB / THREADS;                   // counters per lane in the scan
  static constexpr int WAVES = THREADS / WAVE;
  struct Storage { u32 cnt[NB + 4]; u32 key[CAP]; u16 val[CAP]; u32 bm[CAP / 32]; u32 waveTot[WAVES]; u32 kmax, firstZero, runCount; };
  __device__ __forceinline__ static void sort(u32 (&k)[IPT], u32 (&v)[IPT], u32 n, Storage &s) {
    const u32 tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
#pragma unroll
    for (int i = 0; i < CPT; ++i) s.cnt[(u32)i * THREADS + tid] = 0;
    if (tid < 4) s.cnt[NB + tid] = 0;
    if (tid < CAP / 32) s.bm[tid] = 0;
    if (tid == 0) { s.kmax = 0; s.firstZero = CAP; s.runCount = 0; }
    u32 m = 0;
#pragma unroll
    for (int j = 0; j < IPT; ++j) if ((u32)j * THREADS + tid < n) m = k[j] > m ? k[j] : m;
#pragma unroll
    for (int d = 32; d; d >>= 1) { const u32 o2 = (u32)__shfl_xor((int)m, d); m = o2 > m ? o2 : m; }
    __syncthreads();
    if (lane == 0) atomicMax(&s.kmax, m);
    __syncthreads();
    const u32 kmax = s.kmax;
#pragma unroll
    for (int j = 0; j < IPT; ++j) if ((u32)j * THREADS + tid < n) { const u32 d = kmax - k[j]; if (d < (u32)CAP) atomicOr(&s.bm[d >> 5], 1u << (d & 31)); }
    __syncthreads();
    if (tid < CAP / 32) { const u32 w = ~s.bm[tid]; if (w) atomicMin(&s.firstZero, tid * 32 + (u32)__ffs((int)w) - 1u); }
    __syncthreads();
    u32 run = s.firstZero;                                    // keys with kmax - key < run are the top run; kmax - run is in no entry
    {                                                         // distinct keys put exactly `run` entries there; more = a key twice: no run then
      u32 mineInRun = 0;
#pragma unroll
      for (int j = 0; j < IPT; ++j) if ((u32)j * THREADS + tid < n && kmax - k[j] < run) ++mineInRun;
#pragma unroll
      for (int d = 32; d; d >>= 1) mineInRun += (u32)__shfl_xor((int)mineInRun, d);
      if (lane == 0 && mineInRun) atomicAdd(&s.runCount, mineInRun);
      __syncthreads();
      if (s.runCount != run) run = 0;
    }
    const u32 lim = run ? (kmax >= run ? kmax - run : 0) : kmax + 1u;   // the other keys are below this
    const int sh = lim > (u32)NB ? (32 - __clz((int)(lim - 1))) - LOGNB : 0;      // (lim - 1) >> sh < NB
    u32 r[IPT];
#pragma unroll
    for (int j = 0; j < IPT; ++j) { r[j] = 0; if ((u32)j * THREADS + tid < n && kmax - k[j] >= run) r[j] = atomicAdd(&s.cnt[k[j] >> sh], 1u); }
    __syncthreads();
    {                                                         // exclusive scan of the counts, in place; cnt[NB] = their sum
      u32 c[CPT], t = 0;
#pragma unroll
      for (int i = 0; i < CPT; ++i) { c[i] = s.cnt[tid * CPT + (u32)i]; t += c[i]; }
      u32 inc = t;
#pragma unroll
      for (int d = 1; d < WAVE; d <<= 1) { const u32 o2 = (u32)__shfl_up((int)inc, d); if (lane >= (u32)d) inc += o2; }
      if (lane == WAVE - 1) s.waveTot[wave] = inc;
      __syncthreads();
      u32 base = inc - t;
      for (u32 q = 0; q < wave; ++q) base += s.waveTot[q];
#pragma unroll
      for (int i = 0; i < CPT; ++i) { s.cnt[tid * CPT + (u32)i] = base; base += c[i]; }
      if (tid == THREADS - 1) s.cnt[NB] = base;
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < IPT; ++j) if ((u32)j * THREADS + tid < n && kmax - k[j] >= run) { r[j] += s.cnt[k[j] >> sh]; s.key[r[j]] = k[j]; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IPT; ++j) if ((u32)j * THREADS + tid < n) {
      const u32 d = kmax - k[j];
      if (d < run) r[j] = n - 1u - d;
      else {
        const u32 b = k[j] >> sh, st = s.cnt[b], en = s.cnt[b + 1];
        u32 rank = 0;
        for (u32 q = st; q < en; ++q) { const u32 x = s.key[q]; rank += (x < k[j] || (x == k[j] && q < r[j])) ? 1u : 0u; }
        r[j] = st + rank;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IPT; ++j) if ((u32)j * THREADS + tid < n) { s.key[r[j]] = k[j]; s.val[r[j]] = (u16)v[j]; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IPT; ++j) { const u32 e = (u32)j * THREADS + tid; if (e < n) { k[j] = s.key[e]; v[j] = s.val[e]; } }
  }
};

template <int THREADS, int IPT, bool LOOKUP>
__global__ __launch_bounds__(THREADS)
void clushash_block_kernel(const u64 *__restrict__ entHash /* hash / w */, const u32 *__restrict__ entRead, const u64 *__restrict__ key /* !LOOKUP: index << 32 | read */,
                           const u64 *__restrict__ blockOff, const u32 *__restrict__ list, const u32 *__restrict__ count /* this class's blocks */,
                           const u64 *__restrict__ table64, int B, u64 w, int qBits, int cb /* packed entries */, int sortBits,
                           h10x_clushash *__restrict__ out, int privT = 0 /* > 0: table64 is the private table of 2^privT slots (priv_find) */,
                           const u32 *__restrict__ hashIndex = nullptr, const u64 *__restrict__ hashValue = nullptr,
                           const u32 *__restrict__ replyIdx = nullptr, const u32 *__restrict__ replyPos = nullptr /* !LOOKUP, sharded: entry e's index is replyIdx[replyPos[e]], its read entRead[e] */,
                           int probeBits = 0 /* > 0: table64 holds index << qBits | (hash >> B) << probeBits | probe number (probe_insertP_kernel) */) {
  using Sort = BlockIndexSort<THREADS, IPT>;
  __shared__ typename Sort::Storage storage;
  const u32 nList = *count;
  for (u32 wi = blockIdx.x; wi < nList; wi += gridDim.x) {
    const u32 c = list[wi];
    const u64 o = blockOff[c]; const u64 n64 = blockOff[c + 1] - o;
    if (n64 == 0) continue;                                  // (slot 0 and empty blocks are on the small class's list)
    const u32 n = (u32)n64;
    u32 k[IPT], v[IPT];
    const u64 mask = ((u64)1 << B) - 1, qmask = qBits >= 64 ? ~0ULL : ((u64)1 << qBits) - 1;
    if (LOOKUP) {
      // the first probe of every entry of the lane is issued before any is looked at: met one entry at a time (a probe loop per entry) a lane waited for IPT random
      // reads in a row — 6 x 3 us of a block's 29 us on the 1/10 3 Gb set, the look-ups then ran at 36 G/s where the chip does 55-59 G/s
      // (entries past the block's end read its last entry: loads without a branch around them — a predicated load is a branch and a wait of its own — and the same
      // table slot for all of them, i.e. one request)
      u64 q[IPT];
#pragma unroll
      for (int j = 0; j < IPT; ++j) { const u32 e = (u32)j * THREADS + threadIdx.x; q[j] = entHash[o + (e < n ? e : n - 1)] >> cb; }
#pragma unroll
      for (int j = 0; j < IPT; ++j) { const u32 e = (u32)j * THREADS + threadIdx.x; v[j] = entRead[o + (e < n ? e : n - 1)] & 0xFFFFu; }   // ClusterHash.read is U16 (hash10x.c:37,180)
      constexpr int LKG = IPT;                            // (groups of 2: the same; the chip's rate of random reads bounds the phase either way)
#pragma unroll
      for (int j0 = 0; j0 < IPT; j0 += LKG) {
        u64 t[LKG];
#pragma unroll
        for (int g = 0; g < LKG; ++g) { const int j = j0 + g < IPT ? j0 + g : IPT - 1; t[g] = table64[privT ? (q[j] & (((u64)1 << privT) - 1)) : ((q[j] * w) & mask)]; }
#pragma unroll
        for (int g = 0; g < LKG; ++g) {
          const int j = j0 + g;
          if (j < IPT) {
            const u32 e = (u32)j * THREADS + threadIdx.x;
            u32 kk;
            if (privT) kk = priv_find(table64, privT, q[j], hashIndex, hashValue, B, w, t[g]);   // (uniform)
            else {
              const u64 h = q[j] * w; u64 slot = h & mask; const u64 step = ((h >> B) & mask) | 1;
              u64 tt = t[g];
              if (probeBits) { u64 want = (h >> B) << probeBits; while (tt != SLOT_EMPTY64 && (tt & qmask) != want) { slot = (slot + step) & mask; ++want; tt = table64[slot]; } }   // (uniform)
              else while (tt != SLOT_EMPTY64 && (tt & qmask) != q[j]) { slot = (slot + step) & mask; tt = table64[slot]; }
              kk = tt == SLOT_EMPTY64 ? 0u : (u32)(tt >> qBits);
            }
            k[j] = e < n ? kk : 0xFFFFFFFFu;
          }
        }
      }
    } else if (replyPos) {
      // sharded --readFQB (round 6): the owners' replies lie in the order the entries were SENT — per owner a slice in block order — and replyPos[e] says where entry e's
      // is. A block's entries read eight slices at consecutive places each: a gather the caches serve, where scatter_key_kernel wrote every index 8 bytes wide to a
      // random place first (1.7 ms per rank on the 1/4 3 Gb set at 8 ranks) and this kernel read the keys back. Loads without a branch around them, as above.
      u32 pp[IPT];
#pragma unroll
      for (int j = 0; j < IPT; ++j) { const u32 e = (u32)j * THREADS + threadIdx.x; pp[j] = replyPos[o + (e < n ? e : n - 1)]; }
#pragma unroll
      for (int j = 0; j < IPT; ++j) { const u32 e = (u32)j * THREADS + threadIdx.x; v[j] = entRead[o + (e < n ? e : n - 1)] & 0xFFFFu; }
#pragma unroll
      for (int j = 0; j < IPT; ++j) { const u32 e = (u32)j * THREADS + threadIdx.x; const u32 kk = replyIdx[pp[j]]; k[j] = e < n ? kk : 0xFFFFFFFFu; }
    } else {
#pragma unroll
      for (int j = 0; j < IPT; ++j) {
        const u32 e = (u32)j * THREADS + threadIdx.x;
        k[j] = 0xFFFFFFFFu; v[j] = 0;
        if (e < n) { const u64 t = key[o + e]; k[j] = (u32)(t >> 32); v[j] = (u32)t & 0xFFFFu; }
      }
    }
    __syncthreads();                                         // the storage of the previous block's sort is free again
    Sort::sort(k, v, n, storage);
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const u32 e = (u32)j * THREADS + threadIdx.x;          // sorted, striped arrangement: a wave's store covers 512 contiguous bytes
      if (e < n) { h10x_clushash r; r.hash = k[j]; r.read = (u16)v[j]; r.subCluster = 0; r.flags = 0; out[o + e] = r; }   // zeroed: SURVEY F4
    }
  }
}

// clusHash of every block by workgroup-local sorts; needs maxBlockHashes <= BLOCK_SORT_MAX. key = nullptr: look the entries up in table64
static int clusHashByBlocks(Ctx *c, const u64 *entHash, const u32 *entRead, const u64 *key, const u64 *table64, int privT = 0, const u32 *replyIdx = nullptr, const u32 *replyPos = nullptr,
                            int probeSH = 0, int probeBits = 0 /* table64 in the probed format: index << probeSH | (hash >> B) << probeBits | probe number */) {
  hipStream_t st = c->stream; const u32 nBlocks = c->nBlocks;
  H10X_HIP(c, c->clusHash.alloc(c->nEntries));
  if (!c->nEntries || nBlocks < 2) return 0;
  int sortBits = bitsFor(c->hashNumber) + 1 > 32 ? 32 : bitsFor(c->hashNumber) + 1;
  const int B = c->prm.B; const u64 w = (u64)c->prm.w; const int qBits = probeBits ? probeSH : c->keyBits;
  DevBuf<u32> lists, counts;
  H10X_TRY(stageB_blockClassLists(c, lists, counts));
  // workgroups per launch: the class's blocks are pulled from its list (count on the device: no round trip); enough workgroups to fill the chip
  const unsigned gridBig = hmin<u32>(nBlocks, (u32)c->numCU * 8), gridSmall = hmin<u32>(nBlocks, 65535u * 4);
#define H10X_CH_LAUNCH(T, I, LOOK, CLS, STREAM) clushash_block_kernel<T, I, LOOK><<<(CLS == 0 ? gridSmall : gridBig), T, 0, STREAM>>>(entHash, entRead, key, c->blockOff.p, \
    lists.p + (size_t)CLS * nBlocks, counts.p + CLS, table64, B, w, qBits, (key || replyPos) ? 0 : c->entCodeBits, sortBits, c->clusHash.p, privT, c->hashIndex.p, c->hashValue.p, replyIdx, replyPos, probeBits)
  const int side = c->maxBlockHashes > BLOCK_SORT_CAP1 ? 2 : (c->maxBlockHashes > BLOCK_SORT_CAP0 ? 1 : 0);
  ForkGuard forkGuard(c);
  if (side) H10X_TRY(c->forkStreams(side));                  // the few large blocks beside the many small ones
  // six entries per lane (512 / 1024 lanes for the 3072 / 6144-entry classes): at twelve the kernel needs 102 registers — four waves per SIMD —
  // and a block's latency chain (loads, look-up, three sort passes, store) has too few neighbours to hide behind
  if (key || replyPos) { H10X_CH_LAUNCH(H10X_BS_T0, H10X_BS_I0, false, 0, st); if (side >= 1) H10X_CH_LAUNCH(H10X_BS_T1, H10X_BS_I1, false, 1, c->aux[0]); if (side >= 2) H10X_CH_LAUNCH(1024, 8, false, 2, c->aux[1]); }
  else { H10X_CH_LAUNCH(H10X_BS_T0, H10X_BS_I0, true, 0, st); if (side >= 1) H10X_CH_LAUNCH(H10X_BS_T1, H10X_BS_I1, true, 1, c->aux[0]); if (side >= 2) H10X_CH_LAUNCH(1024, 8, true, 2, c->aux[1]); }
#undef H10X_CH_LAUNCH
  H10X_TRY(c->faultAt(2));
  if (side) H10X_TRY(c->joinStreams(side));
  forkGuard.done();
  H10X_HIP(c, hipGetLastError());
  return 0;                                                  // no round trip: the buffers go back to the stream-ordered block cache, the next command queues behind
}

int stageB_run(Ctx *c, DevBuf<u64> &entHash, DevBuf<u32> &entCode, DevBuf<u32> &entRead) {
  hipStream_t st = c->stream; PrimTemp pt;
  const u64 H = c->nEntries; const int B = c->prm.B; const u32 nBlocks = c->nBlocks;
  const unsigned gH = (unsigned)hmin<u64>(divUp(H ? H : 1, 256), 65535u * 2);
  const u64 tableSize = (u64)1 << B;

  // ---- sort all (hash, barcode) entries by hash; stable => barcodes ascending inside a hash
  c->tstart(T_SORT_HASH);
  DevBuf<u64> sHash; DevBuf<u32> sCode;
  const int cb = c->entCodeBits;                             // > 0: packed entries (common.hpp), keys-only sort
  H10X_HIP(c, sHash.alloc(H)); H10X_HIP(c, sCode.alloc(H + ROWS_PAD));   // (sCode becomes rows[])
  if (cb) H10X_TRY(prim_sort_keys_u64(c, pt, entHash.p, sHash.p, H, cb, cb + c->keyBits));
  else H10X_TRY(prim_sort_pairs_u64_u32(c, pt, entHash.p, sHash.p, entCode.p, sCode.p, H, 0, c->keyBits));
  c->tstop(T_SORT_HASH);

  // ---- distinct hashes, first barcode, depth
  c->tstart(T_RANK);
  u32 U = 0;
  DevBuf<u64> dHash; DevBuf<u32> dFirst, segStart, iota, order, dFirstSorted; DevBuf<Val16> dVal, dValSorted;
  {
    DevBuf<u32> ord;
    H10X_HIP(c, ord.alloc(H + 2));                           // (+ 1: the packed scatter looks two ahead)
    H10X_TRY(prim_run_ordinals_u64(c, pt, sHash.p, cb, ord.p, H));          // (head flags formed inside the scan)
    H10X_TRY(c->readback(&U, ord.p + H, 4));
    H10X_TRY(c->syncReadbacks());
    // hash10x.c:149: die once hashNumber exceeds 2^(B-2) - 2; hashNumber ends at U + 1
    if ((u64)U + 1 > (tableSize >> 2) - 2) return c->fail("hashTableSize is too small");
    H10X_HIP(c, dFirst.alloc(U)); H10X_HIP(c, dFirstSorted.alloc(U));
    if (cb) { H10X_HIP(c, dVal.alloc(U)); H10X_HIP(c, dValSorted.alloc(U)); }
    else { H10X_HIP(c, dHash.alloc(U)); H10X_HIP(c, segStart.alloc((size_t)U + 1)); H10X_HIP(c, iota.alloc(U)); H10X_HIP(c, order.alloc(U)); }
    if (H && cb) seg_scatter_packed_kernel<<<gH, 256, 0, st>>>(sHash.p, cb, ord.p, H, dVal.p, dFirst.p, sCode.p);
    else if (H) {
      seg_scatter_kernel<<<gH, 256, 0, st>>>(sHash.p, sCode.p, ord.p, H, dHash.p, dFirst.p, segStart.p, iota.p);
      H10X_HIP(c, hipMemsetD32Async((hipDeviceptr_t)(segStart.p + U), (int)(u32)H, 1, st));    // end of the last segment
    }
  }
  // distinct hashes are in ascending hash order; a stable sort by first barcode gives (first, hash) order
  c->hashNumber = U + 1; c->depthBound = nBlocks;             // a hash is met at most once per barcode
  H10X_HIP(c, c->hashValue.alloc((size_t)U + 1)); H10X_HIP(c, c->hashDepth.alloc((size_t)U + 1));
  H10X_HIP(c, c->rowStart.alloc((size_t)U + 2));
  if (cb) {
    H10X_TRY(prim_sort_pairs_u32_v16(c, pt, dFirst.p, dFirstSorted.p, dVal.p, dValSorted.p, U, 0, bitsFor(nBlocks)));
    assign_index_v16_kernel<<<divUp((u64)U + 1, 256), 256, 0, st>>>(dValSorted.p, U, (u64)c->prm.w, c->hashValue.p, c->hashDepth.p, c->rowStart.p);
  } else {
    H10X_TRY(prim_sort_pairs_u32_u32(c, pt, dFirst.p, dFirstSorted.p, iota.p, order.p, U, 0, bitsFor(nBlocks)));
    assign_index_kernel<<<divUp((u64)U + 1, 256), 256, 0, st>>>(order.p, dHash.p, segStart.p, U, (u64)c->prm.w, c->hashValue.p, c->hashDepth.p, c->rowStart.p);
  }
  c->rows.swap(sCode);                                       // barcode lists, grouped by hash (ascending barcodes)
  c->tstop(T_RANK);
  sHash.release(); dHash.release(); dFirst.release(); iota.release(); dFirstSorted.release(); dVal.release(); dValSorted.release();

  const bool forcePriv = (c->optPrivTable == 1 || c->optPrivTable == 3) && c->keyBits <= 40;
  const bool wideTable = !forcePriv && c->keyBits + (B - 2) <= 64 && c->keyBits < 64;   // index < 2^(B-2) (hash10x.c:149)
  DevBuf<u64> table64;
  // the probed format (probe_insertP_kernel) where the classic wide entry does not fit: index bits B - 2, hash >> B in 2k - B bits, the rest — 66 - 2k — for the probe number
  const int hashBits = 2 * c->prm.k, UB = hashBits > B ? hashBits - B : 0, PBroom = 64 - (B - 2) - UB;
  int PB = 0, SH = 0; bool probed = false;
  if (!forcePriv && c->optProbedTable != 2 && (!wideTable || c->optProbedTable == 1 || c->optProbedTable == 3) && PBroom >= 12 && c->keyBits < 64) {
    PB = PBroom > 24 ? 24 : PBroom; if (c->optProbedTable == 3) PB = 1;          // (knob 3, tests: one bit of probe number — the first second collision fails the table)
    SH = UB + PB;
    c->tstart(T_PROBE);
    DevBuf<u32> failFlag; u32 failed = 0;
    H10X_HIP(c, table64.alloc(tableSize)); H10X_HIP(c, c->hashIndex.alloc(tableSize)); H10X_HIP(c, failFlag.alloc(1));
    H10X_HIP(c, hipMemsetAsync(table64.p, 0xFF, tableSize * 8, st)); H10X_HIP(c, hipMemsetAsync(failFlag.p, 0, 4, st));
    if (U) probe_insertP_kernel<<<hmin<u32>(divUp(U, 256), 16384), 256, 0, st>>>(c->hashValue.p, U + 1, B, SH, PB, table64.p, failFlag.p);
    H10X_TRY(c->readback(&failed, failFlag.p, 4));
    H10X_TRY(c->syncReadbacks());
    probed = !failed;
    if (probed) probe_finish64_kernel<<<(unsigned)hmin<u64>(divUp(tableSize, 256), 65535u * 2), 256, 0, st>>>(table64.p, tableSize, SH, c->hashIndex.p);
    else { table64.release(); c->hashIndex.release(); }
    c->tstop(T_PROBE);
    c->ctr.index_table_form = probed ? 2 : 3;
  }
  if (probed) {}
  else if (wideTable) {
    if (c->ctr.index_table_form != 3) c->ctr.index_table_form = 1;   // (3: the probed table failed just above — what is built instead does not change the report)
    c->tstart(T_PROBE);
    H10X_HIP(c, table64.alloc(tableSize)); H10X_HIP(c, c->hashIndex.alloc(tableSize));
    H10X_HIP(c, hipMemsetAsync(table64.p, 0xFF, tableSize * 8, st));
    if (U) probe_insert64_kernel<<<hmin<u32>(divUp(U, 256), 16384), 256, 0, st>>>(c->hashValue.p, U + 1, B, (u64)c->prm.w, c->keyBits, table64.p);
    probe_finish64_kernel<<<(unsigned)hmin<u64>(divUp(tableSize, 256), 65535u * 2), 256, 0, st>>>(table64.p, tableSize, c->keyBits, c->hashIndex.p);
    c->tstop(T_PROBE);
  } else { if (c->ctr.index_table_form != 3) c->ctr.index_table_form = 0; H10X_TRY(stageB_buildProbeTable(c)); }

  // ---- clusHash: look every entry up, order each block by index (hash10x.c:177-183)
  c->tstart(T_CLUSHASH);
  if (H && probed && c->maxBlockHashes <= BLOCK_SORT_MAX) {
    H10X_TRY(clusHashByBlocks(c, entHash.p, entRead.p, nullptr, table64.p, 0, nullptr, nullptr, SH, PB));
    table64.release();
  } else if (H && !probed && wideTable && c->maxBlockHashes <= BLOCK_SORT_MAX) {
    H10X_TRY(clusHashByBlocks(c, entHash.p, entRead.p, nullptr, table64.p));
    table64.release();
  } else if (H && !probed && !wideTable && c->maxBlockHashes <= BLOCK_SORT_MAX && c->keyBits <= 40 && c->optPrivTable != 2) {
    // the look-up table of this library's own (priv_insert_kernel): 2^T slots >= 2 (U + 1), T >= keyBits - 24 so that q >> T fits the entry's 24 bits
    int T = 16; while (((u64)1 << T) < 2 * ((u64)U + 1) && T < 32) ++T;
    if (c->optPrivTable == 3) { T = 4; while (((u64)1 << T) < ((u64)U + 1) / 2 && T < 32) ++T; }   // (test knob: half the hashes find no slot and are looked up the old way)
    if (T < c->keyBits - 24) T = c->keyBits - 24;
    DevBuf<u64> priv;
    H10X_HIP(c, priv.alloc((size_t)1 << T));
    H10X_HIP(c, hipMemsetAsync(priv.p, 0xFF, ((size_t)1 << T) * 8, st));
    if (U) priv_insert_kernel<<<hmin<u32>(divUp(U, 256), 16384), 256, 0, st>>>(c->hashValue.p, U + 1, (u64)c->prm.w, T, priv.p);
    H10X_TRY(clusHashByBlocks(c, entHash.p, entRead.p, nullptr, priv.p, T));
  } else if (H) {
    DevBuf<u64> key; H10X_HIP(c, key.alloc(H));
    if (wideTable && !probed) lookup_pack64_kernel<<<gH, 256, 0, st>>>(entHash.p, entRead.p, H, table64.p, B, (u64)c->prm.w, c->keyBits, cb, key.p);
    else lookup_pack_kernel<<<gH, 256, 0, st>>>(entHash.p, entRead.p, H, c->hashIndex.p, c->hashValue.p, B, (u64)c->prm.w, cb, key.p);
    table64.release();
    H10X_TRY(stageB_finishClusHash(c, key));
  } else H10X_HIP(c, c->clusHash.alloc(0));
  c->tstop(T_CLUSHASH);
  c->ctr.distinct = U;
  c->haveState = true;
  return 0;
}

int stageB_buildProbeTable(Ctx *c) {
  hipStream_t st = c->stream; const int B = c->prm.B; const u64 tableSize = (u64)1 << B; const u32 U = c->hashNumber - 1;
  c->tstart(T_PROBE);
  H10X_HIP(c, c->hashIndex.alloc(tableSize));
  H10X_HIP(c, hipMemsetAsync(c->hashIndex.p, 0xFF, tableSize * 4, st));
  if (U) probe_insert_kernel<<<hmin<u32>(divUp(U, 256), 16384), 256, 0, st>>>(c->hashValue.p, U + 1, B, c->hashIndex.p);
  probe_finish_kernel<<<(unsigned)hmin<u64>(divUp(tableSize, 256), 65535u * 2), 256, 0, st>>>(c->hashIndex.p, tableSize);
  c->tstop(T_PROBE);
  return 0;
}

// the same from the owners' replies of the sharded index build: entry e (block order) has index replyIdx[replyPos[e]] and read entRead[e]
__global__ void reply_key_kernel(const u32 *__restrict__ replyIdx, const u32 *__restrict__ replyPos, const u32 *__restrict__ entRead, u64 n, u64 *__restrict__ key) {
  u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; e < n; e += stride) key[e] = ((u64)replyIdx[replyPos[e]] << 32) | (u64)(entRead[e] & 0xFFFFu);
}
int stageB_finishClusHashFromReplies(Ctx *c, const u32 *replyIdx, const u32 *replyPos, const u32 *entRead) {
  if (c->maxBlockHashes <= BLOCK_SORT_MAX) return clusHashByBlocks(c, nullptr, entRead, nullptr, nullptr, 0, replyIdx, replyPos);
  const u64 H = c->nEntries;                                 // blocks too large for a workgroup-local sort: the keys as an array, then the device-wide segmented sort
  DevBuf<u64> key; H10X_HIP(c, key.alloc(H));
  if (H) reply_key_kernel<<<(unsigned)hmin<u64>(divUp(H, 256), 65535u * 2), 256, 0, c->stream>>>(replyIdx, replyPos, entRead, H, key.p);
  return stageB_finishClusHash(c, key);
}

// key[e] = hash index << 32 | read (U16), entries in block order: sort every block by index, emit ClusterHash
int stageB_finishClusHash(Ctx *c, DevBuf<u64> &key) {
  hipStream_t st = c->stream; PrimTemp pt; const u64 H = c->nEntries; const u32 nBlocks = c->nBlocks;
  const unsigned gH = (unsigned)hmin<u64>(divUp(H ? H : 1, 256), 65535u * 2);
  if (c->maxBlockHashes <= BLOCK_SORT_MAX) return clusHashByBlocks(c, nullptr, nullptr, key.p, nullptr);
  H10X_HIP(c, c->clusHash.alloc(H));
  if (!H) return 0;
  DevBuf<u64> keyS; DevBuf<u32> off32;
  H10X_HIP(c, keyS.alloc(H)); H10X_HIP(c, off32.alloc((size_t)nBlocks + 1));
  offsets32_kernel<<<divUp((u64)nBlocks + 1, 256), 256, 0, st>>>(c->blockOff.p, nBlocks + 1, off32.p);
  H10X_TRY(prim_seg_sort_keys_u64(c, pt, key.p, keyS.p, (u32)H, nBlocks, off32.p, off32.p + 1, 32, 32 + bitsFor(c->hashNumber)));
  write_clushash_kernel<<<gH, 256, 0, st>>>(keyS.p, H, c->clusHash.p);
  H10X_HIP(c, hipGetLastError());
  H10X_HIP(c, hipStreamSynchronize(st));
  return 0;
}

// ------------------------------------------------------------------------------------------ fillHashTable from clusHash
__global__ void entry_code_kernel(const h10x_clushash *__restrict__ ch, const u64 *__restrict__ blockOff, u32 nBlocks,
                                  u32 *__restrict__ keyIdx, u32 *__restrict__ valCode) {
  for (u32 c = blockIdx.x + 1; c < nBlocks; c += gridDim.x) {
    const u64 o = blockOff[c], n = blockOff[c + 1] - o;
    for (u64 i = threadIdx.x; i < n; i += blockDim.x) { keyIdx[o + i] = ch[o + i].hash; valCode[o + i] = c; }
  }
}

// hashCodes for the current clusHash + hashDepth (hash10x.c:317-347): stable sort of (index, barcode)
int stageB_buildCSR(Ctx *c) {
  hipStream_t st = c->stream; PrimTemp pt;
  const u64 H = c->nEntries; const u32 U1 = c->hashNumber;
  c->tstart(T_CSR);
  H10X_HIP(c, c->rowStart.alloc((size_t)U1 + 1));
  H10X_TRY(prim_exclusive_scan_u32_u64(c, pt, c->hashDepth.p, c->rowStart.p, U1));
  u64 lastOff = 0; u32 lastDepth = 0;
  if (U1) {
    H10X_HIP(c, hipMemcpyAsync(&lastOff, c->rowStart.p + (U1 - 1), 8, hipMemcpyDeviceToHost, st));
    H10X_HIP(c, hipMemcpyAsync(&lastDepth, c->hashDepth.p + (U1 - 1), 4, hipMemcpyDeviceToHost, st));
    H10X_HIP(c, hipStreamSynchronize(st));
  }
  if (lastOff + lastDepth != H)
    return c->fail("inconsistent hash state: sum of hashDepth %llu != %llu (barcode,hash) entries", (u64)(lastOff + lastDepth), (u64)H);
  H10X_HIP(c, c->rows.alloc(H + ROWS_PAD));
  if (H) {
    DevBuf<u32> k0, k1, v0;
    H10X_HIP(c, k0.alloc(H)); H10X_HIP(c, k1.alloc(H)); H10X_HIP(c, v0.alloc(H));
    entry_code_kernel<<<hmin<u32>(c->nBlocks, 8192), 256, 0, st>>>(c->clusHash.p, c->blockOff.p, c->nBlocks, k0.p, v0.p);
    H10X_TRY(prim_sort_pairs_u32_u32(c, pt, k0.p, k1.p, v0.p, c->rows.p, H, 0, bitsFor(U1)));
    H10X_HIP(c, hipStreamSynchronize(st));
  }
  c->tstop(T_CSR);
  return 0;
}

// h10x_warm: the first launch of a kernel loads the code object of its translation unit (HIP loads them on first use); this one is launched ahead of time
__global__ void warm_stageB_kernel() {}
void warm_stageB(hipStream_t st) { warm_stageB_kernel<<<1, 1, 0, st>>>(); }

}  // namespace h10x
