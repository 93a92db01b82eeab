// stage_a.hip — .fqb records -> per-barcode unique (mosh hash, lowest read) sets.
//
// Replaces unpackFQB + seqAddHashes + moshRCiterator/moshRCnext + the sort/dedup half of
// processBlock (hash10x.c:108-132, 154-172; seqhash.c:58-80, 154-195) and readFQB's barcode run
// detection (hash10x.c:212-220).
//
// MI355X design: one workgroup per barcode block (mosh_lds_kernel). 32 lanes share a read pair; each lane owns a run of
// consecutive k-mer start positions of one read, fetches the 4 packed dwords that cover its run straight from HBM (the
// next pair's are requested before the current pair is hashed), keeps them as a 128-bit shift register and ROLLS the
// forward and the reverse-complement word (2 bits per step) — what is left per position is the two 64-bit multiplies
// of hashFunc, the minimum and the divisibility test (modulo sampling, SURVEY F1 — there is no window minimum to select).
// Survivors go into an open-addressing hash set in LDS holding (hash << 16 | read), inserted with ds 64-bit
// cmpswap/min atomics, so duplicates inside a barcode collapse on chip and the lowest read index wins.
// Only the unique set (≈ 6-8 entries per pair) is written back to HBM; three table sizes = three launch classes, side by
// side on forked streams. Blocks that cannot use the LDS set (k > 24, > 65535 pairs, or more unique hashes than the
// table holds) take a global-memory path: raw slots in read order + stable device radix sort + unique.
// The entries leave this stage as (hash / w, barcode, read): every mosh is a multiple of w (Ctx::keyInv, common.hpp).
#include "common.hpp"
#include <algorithm>
#include "prim.hpp"

namespace h10x {

constexpr u64 EMPTY64 = ~0ULL;
constexpr u32 NHASH_OVERFLOW = 0xFFFFFFFFu;
constexpr int MOSH_THREADS = 256;
constexpr int REC_TILE = 16;            // records staged per LDS tile
constexpr int SEQ_WORDS = 12;           // 10 packed words + 2 pad words per read

struct MoshConst {
  int k, w, shift1;      // shift1 = 64 - 2k
  int n1, n2;            // k-mers per read 1 (127 bases from base 23) and read 2 (150 bases)
  int run;               // consecutive k-mer slots per lane in mosh_lds_kernel: ceil(n1/run) + ceil(n2/run) <= 32
  u64 factor1;
};
#define H10X_MOSH_DBG(bit) false

// ------------------------------------------------------------------------------------------ helpers
// 2k-bit big-endian word of bases [p, p+k) from MSB-first packed dwords (fq2b.c:33-42 layout;
// the un-justified tail word is consumed as-is, which reproduces the 'A' padding of SURVEY F6)
__device__ __forceinline__ u64 kmer_window(const u32 *seq, int p, int k) {
  const int wi = p >> 4, s = (p & 15) * 2;
  const u64 hi = ((u64)seq[wi] << 32) | (u64)seq[wi + 1];
  const u32 lo = seq[wi + 2];
  const u64 x = s ? ((hi << s) | (u64)(lo >> (32 - s))) : hi;
  return x >> (64 - 2 * k);
}

// reverse complement of a 2k-bit word == what advanceHashRC accumulates in hRC (seqhash.c:75)
__device__ __forceinline__ u64 revcomp_word(u64 x, int k) {
  u64 r = __brevll(~x);                                     // reverse all bits of the complement
  r = ((r & 0xAAAAAAAAAAAAAAAAULL) >> 1) | ((r & 0x5555555555555555ULL) << 1);  // un-swap bit pairs
  return r >> (64 - 2 * k);
}

template <bool W31>
__device__ __forceinline__ bool divisible(u64 h, int w) {
  if (W31) {                                                // 2^30 == 1 (mod 31): fold then 32-bit modulo
    const u32 s = (u32)(h & 0x3FFFFFFFu) + (u32)((h >> 30) & 0x3FFFFFFFu) + (u32)(h >> 60);
    return (s % 31u) == 0;
  }
  return (h % (u64)w) == 0;
}

// canonical hash of k-mer slot t of the record staged at seq (24 words: read 1, read 2)
template <bool W31>
__device__ __forceinline__ bool mosh_of_slot(const u32 *seq, int t, const MoshConst &mc, u64 &h) {
  int p; const u32 *s;
  if (t < mc.n1) { p = 23 + t; s = seq; }                   // hash10x.c:162  &s1[23], 127 bases
  else { p = t - mc.n1; s = seq + SEQ_WORDS; }               // hash10x.c:163  s2, 150 bases
  const u64 f = kmer_window(s, p, mc.k);
  const u64 r = revcomp_word(f, mc.k);
  const u64 hf = (f * mc.factor1) >> mc.shift1;              // seqhash.c:58-59
  const u64 hr = (r * mc.factor1) >> mc.shift1;
  h = hf < hr ? hf : hr;                                     // seqhash.c:67-68
  return divisible<W31>(h, mc.w);
}

// stage the 20 sequence words of `cnt` records (qualities skipped) into LDS, padded to 24 per record
__device__ __forceinline__ void stage_records(const u32 *__restrict__ rec, u64 firstRec, int cnt, u32 *tile) {
  for (int t = threadIdx.x; t < cnt * 20; t += blockDim.x) {
    const int rr = t / 20, wi = t - rr * 20;
    const u32 v = rec[(firstRec + rr) * 30 + (wi < 10 ? wi : wi + 5)];
    tile[rr * 2 * SEQ_WORDS + (wi < 10 ? wi : wi + 2)] = v;
  }
}

// ------------------------------------------------------------------------------------------ block runs
__global__ void head_flags_kernel(const u32 *__restrict__ rec, u64 n, u32 *__restrict__ flags) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) flags[i] = (i == 0 || rec[i * 30] != rec[(i - 1) * 30]) ? 1u : 0u;
}

// code[i] = inclusive scan of flags = 1-based block number of record i
__global__ void block_starts_kernel(const u32 *__restrict__ flags, const u32 *__restrict__ code, u64 n, u32 nBlocks,
                                    u64 *__restrict__ startRec /* nBlocks+1 */) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  if (i == 0) { startRec[0] = 0; startRec[nBlocks] = n; }    // block 0 is unused; the end of the last block
  for (; i < n; i += stride) if (flags[i]) startRec[code[i]] = i;
}

__global__ void clear_heads_kernel(u32 *__restrict__ flags, const u64 *__restrict__ at, u32 n) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[at[i]] = 0;
}
__global__ void zero_runs_kernel(const u32 *__restrict__ rec, const u64 *__restrict__ startRec, u32 nBlocks, u32 *__restrict__ list, u32 cap, u32 *__restrict__ count) {
  const u32 b = blockIdx.x * blockDim.x + threadIdx.x + 1;
  if (b < nBlocks && rec[startRec[b] * 30] == 0) { const u32 p = atomicAdd(count, 1u); if (p < cap) list[p] = b - 1; }
}

// per block: nRead, the LDS table size it gets (0 = global path), and its class list
__global__ void classify_kernel(const u64 *__restrict__ startRec, u32 nBlocks, h10x_block *__restrict__ blocks,
                                u32 *__restrict__ slots, u32 maxSlots, int packedOK,
                                u32 *__restrict__ list0, u32 *__restrict__ list1, u32 *__restrict__ list2, u32 *__restrict__ listF,
                                u32 *__restrict__ counts /* 4: <=4096, 8192, 16384 slots, global path */, int hashLast) {
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & (WAVE - 1);
  int cls = -1; u32 s = 0;
  if (c < nBlocks) {
    h10x_block b; memset(&b, 0, sizeof b);
    if (c > 0) {
      const u64 nr = startRec[c + 1] - startRec[c];
      b.nRead = (u32)nr;
      if (c + 1 < nBlocks || hashLast) {                     // the file's last block is never hashed; a shard's last block is
        u64 want = 256; while (want < nr * 10) want <<= 1;  // expected load <= 0.76 at 7.6 unique / pair (overflow at 0.875 => global path)
        if (packedOK && nr <= 65535 && want <= maxSlots) { s = (u32)want; cls = s <= 4096 ? 0 : (s <= 8192 ? 1 : 2); }
        else cls = 3;
      }
    }
    blocks[c] = b; slots[c] = s;
  }
  u32 *const lists[4] = {list0, list1, list2, listF};
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

// ------------------------------------------------------------------------------------------ LDS path
// packed entry: hash << 16 | read   (needs 2k + 16 <= 64 and read <= 65535)
// 64-bit shifts and compares cost four times a 32-bit one on gfx950 (scratch/valu_rate64.hip: 8 SIMD cycles against 2), so the
// set works on the two halves of an entry: empty = both halves all ones, same hash = high halves equal and low halves equal
// above the 16 read bits.
__device__ __forceinline__ u32 lo32(u64 x) { return (u32)x; }
__device__ __forceinline__ u32 hi32(u64 x) { return (u32)(x >> 32); }
__device__ __forceinline__ u64 mk64(u32 hi, u32 lo) { return ((u64)hi << 32) | lo; }
__device__ __forceinline__ bool entry_empty(u64 v) {
  u32 t = lo32(v) & hi32(v);
  asm("" : "+v"(t));                                         // or the compiler folds this back into one 64-bit compare
  return t == 0xFFFFFFFFu;
}
// One probe = one compare-and-swap against the empty pattern: the returned word says everything (was empty: ours now; same
// hash: keep the lower read with a fire-and-forget min; another hash: next slot) — no look before the swap, one LDS round trip.
__device__ __forceinline__ u32 set_slot_of(u32 hLo, u32 hHi, u32 mask) {
  return (((hLo ^ (hHi * 0x9E3779B1u)) * 0x85EBCA6Bu) >> 15) & mask;   // any spread will do: the order inside a block is not kept
}
__device__ __forceinline__ bool probe_done(u64 *table, u32 slot, u64 cur, u32 pLo, u32 pHi) {
  if (entry_empty(cur)) return true;
  if (hi32(cur) == pHi && ((lo32(cur) ^ pLo) >> 16) == 0) { atomicMin((u64 *)&table[slot], mk64(pHi, pLo)); return true; }
  return false;
}
// Survivors are 1 in w of the slots: inserting them where they turn up would run the probe loop with a quarter of the lanes
// (measured: a third of the launch). Each wave parks them, packed, in a small LDS queue of its own instead — one ballot and
// one write per slot — and empties it 64 at a time with every lane probing for one entry.
constexpr u32 MOSH_QCAP = 96, MOSH_QKEEP = MOSH_QCAP - WAVE;      // entries per wave; at most QKEEP may stay behind before a slot's push of up to 64
typedef __attribute__((address_space(3))) volatile u64 lds_queue_t;   // typed as LDS: a volatile generic pointer compiles to flat accesses, each waited for
// (a queue entry is the hash with the read above it — hash | read << 48, 2k <= 48 — and becomes the set's hash << 16 | read on the way out:
// the push runs for nine slots in ten with two or three lanes, the drain with all 64)
__device__ __forceinline__ void queue_push(lds_queue_t *q, u32 &qn /* wave-uniform */, bool ok, u32 hLo, u32 hHi, u32 read16 /* read << 16 */) {
  const u64 bal = __builtin_amdgcn_ballot_w64(ok);
  if (!bal) return;
  const u32 pre = __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
  if (ok) q[qn + pre] = mk64(hHi | read16, hLo);
  qn += (u32)__popcll(bal);
}
// takes up to 64 entries off the top of the queue; false = the table is full
__device__ __forceinline__ bool queue_drain(u64 *table, u32 mask, lds_queue_t *q, u32 &qn, int lane) {
  const u32 n = qn < (u32)WAVE ? qn : (u32)WAVE;
  qn -= n;
  bool go = (u32)lane < n;
  u64 raw = 0;
  if (go) raw = q[qn + lane];
  const u32 hLo = lo32(raw), hHi = hi32(raw) & 0xFFFFu;
  const u32 pLo = (hLo << 16) | (hi32(raw) >> 16), pHi = __builtin_amdgcn_alignbit(hHi, hLo, 16);
  const u64 packed = mk64(pHi, pLo);
  u32 slot = set_slot_of(hLo, hHi, mask);
  for (u32 probes = 0; __builtin_amdgcn_ballot_w64(go); ++probes) {
    if (probes > mask) return false;
    if (go) {
      const u64 cur = atomicCAS((u64 *)&table[slot], EMPTY64, packed);
      if (probe_done(table, slot, cur, pLo, pHi)) go = false; else slot = (slot + 1) & mask;
    }
  }
  return true;
}

// low 64 bits of x * f from 32-bit halves: one 32 x 32 -> 64 multiply-add and two low multiplies
__device__ __forceinline__ void mul64_lo(u32 xLo, u32 xHi, u32 fLo, u32 fHi, u32 &pLo, u32 &pHi) {
  const u64 p = (u64)xLo * fLo;
  pLo = lo32(p); pHi = hi32(p) + xLo * fHi + xHi * fLo;
}

// min of two 64-bit words held as halves: the borrow of a 32-bit subtract chain picks (a 64-bit compare costs four times a 32-bit operation)
__device__ __forceinline__ void min64_halves(u32 aLo, u32 aHi, u32 bLo, u32 bHi, u32 &mLo, u32 &mHi) {
  asm("v_sub_co_u32 %0, vcc, %2, %4\n\tv_subb_co_u32 %0, vcc, %3, %5, vcc\n\tv_cndmask_b32 %0, %4, %2, vcc\n\tv_cndmask_b32 %1, %5, %3, vcc"
      : "=&v"(mLo), "=&v"(mHi) : "v"(aLo), "v"(aHi), "v"(bLo), "v"(bHi) : "vcc");
}
// 4 p + t in one instruction (the compiler reaches v_lshl_add_u64 only with a shift of 0 here)
__device__ __forceinline__ u64 roll_product(u64 p, u64 t) {
  u64 r;
  asm("v_lshl_add_u64 %0, %1, 2, %2" : "=v"(r) : "v"(p), "v"(t));
  return r;
}
// reverse complement of the 2k-bit word (hi, lo), 32 < 2k < 64, on halves: reverse the bits of the complement, put the bit pairs back in order, shift down
__device__ __forceinline__ void revcomp_halves(u32 lo, u32 hi, int down /* 64 - 2k, 1..31 */, u32 &rLo, u32 &rHi) {
  u32 a = __builtin_bitreverse32(~lo), b = __builtin_bitreverse32(~hi);   // the reversed word is (a, b): a its high half
  a = ((a >> 1) & 0x55555555u) | ((a << 1) & 0xAAAAAAAAu);
  b = ((b >> 1) & 0x55555555u) | ((b << 1) & 0xAAAAAAAAu);
  rLo = __builtin_amdgcn_alignbit(a, b, (u32)down); rHi = a >> down;
}

// One workgroup per barcode block, no staging tile and no barrier in the main loop: 32 lanes share a read pair, each
// lane owns a run of L (8 or 9) consecutive k-mer slots of one read. It fetches the 4 packed dwords that cover its run
// straight from HBM (the next pair's are requested before the current pair is hashed) and ROLLS the forward word (2 bits in
// per step) and the reverse-complement word (seqhash.c:75: shift right, complement of the new base in at the top) instead
// of re-extracting a window per slot; what is left per slot is the two 64-bit multiplies of hashFunc, the min and the
// divisibility test. Survivors (1 in w) are parked, two per lane, and inserted into the LDS hash set once per pair.
// FAST (17 <= k <= 30, the usual range): everything on 32-bit halves, and the PRODUCTS roll, not the words. With F = factor1 and b(i) base i
// of the run, the forward word obeys f(j+1) = 4 f(j) + b(j+k) - b(j) 2^2k, so f(j+1) F = 4 f(j) F + [b(j+k) F - b(j) (F << 2k)] (mod 2^64);
// the reverse-complement word (seqhash.c:75) obeys the mirror image walking BACKWARDS, rc(j) = 4 rc(j+1) + (3 - b(j)) - (3 - b(j+k)) 2^2k.
// Both bracketed terms depend on the same two bases: one 16-entry table in LDS, indexed by b(j) << 2 | b(j+k), holds them side by
// side. A lane multiplies twice per read pair (first forward word, last reverse word), walks the reverse products back into registers,
// then walks forward: per slot two shift-and-adds, the min (taken before the one shift both products share), the divisibility test.
// The quarter-rate multiplies of hashFunc (seqhash.c:58-59) — six per slot before — are gone from the loop; results are the same
// products mod 2^64. Every lane owns a FULL run of L slots (the last lane of a read starts early and re-hashes a few slots of its
// neighbour: the set drops the duplicates), so the loop has no per-slot mask. The general form keeps 64-bit arithmetic.
template <bool W31, int L /* slots per lane: mc.run */, bool FAST>
__global__ __launch_bounds__(1024)
void mosh_lds_kernel(const u32 *__restrict__ rec, const u32 *__restrict__ list, u32 nList,
                     const u64 *__restrict__ startRec, const u32 *__restrict__ slots, const u64 *__restrict__ capOff,
                     MoshConst mc, u32 tableBytes /* of the launch's table class; the waves' queues lie behind */,
                     u64 *__restrict__ stHash, u32 *__restrict__ stRead, u32 *__restrict__ nHashOut) {
  extern __shared__ __align__(16) unsigned char smem[];
  if (blockIdx.x >= nList) return;
  const u32 code = list[blockIdx.x];
  const u32 S = slots[code], mask = S - 1;
  u64 *table = (u64 *)smem;
  lds_queue_t *queue = (lds_queue_t *)(smem + tableBytes) + (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE)) * MOSH_QCAP;   // (a scalar: the push adds its lane offset in one instruction)
  u32 qn = 0;
  __shared__ u32 sOverflow, sCount;
  __shared__ u64 sRoll[32];                                  // FAST: [2 t] forward, [2 t + 1] reverse roll term of t = leaving base << 2 | entering base
  const u64 rec0 = startRec[code];
  const u32 nRead = (u32)(startRec[code + 1] - rec0);

  for (u32 i = threadIdx.x; i < S; i += blockDim.x) table[i] = EMPTY64;
  if (threadIdx.x == 0) { sOverflow = 0; sCount = 0; }
  if (FAST && threadIdx.x < 16) {
    const u64 a = threadIdx.x >> 2, b = threadIdx.x & 3, top = mc.factor1 << (2 * mc.k);   // (2k < 64 here)
    sRoll[2 * threadIdx.x] = b * mc.factor1 - a * top;
    sRoll[2 * threadIdx.x + 1] = (3 - a) * mc.factor1 - (3 - b) * top;
  }
  __syncthreads();

  // this lane's run of slots
  const int lane = threadIdx.x & (WAVE - 1), sub = lane & 31, wv = threadIdx.x / WAVE;
  const int lanes1 = (mc.n1 + L - 1) / L, lanes2 = (mc.n2 + L - 1) / L;                 // lanes1 + lanes2 <= 32 (stageA_run picks L)
  int cnt = 0, pos0 = 0, wordBase = 0;
  if (sub < lanes1) { int t0 = sub * L; if (FAST) t0 = min(t0, mc.n1 - L); cnt = min(L, mc.n1 - t0); pos0 = 23 + t0; }   // hash10x.c:162  &s1[23], 127 bases
  else if (sub < lanes1 + lanes2) { int t0 = (sub - lanes1) * L; if (FAST) t0 = min(t0, mc.n2 - L); cnt = min(L, mc.n2 - t0); pos0 = t0; wordBase = 15; }   // hash10x.c:163  s2, 150 bases
  const int sh = (pos0 & 15) * 2;
  // FAST aligns the run with 32-bit funnel shifts by 32 - sh; a run that starts on a dword edge fetches from one dword earlier, the
  // shift by 32 - 32 then hands back the second word of each pair — no special case (the dword before a read is inside the record)
  const int wi = wordBase + (pos0 >> 4) - (FAST && sh == 0 ? 1 : 0);
  const u32 unal = (u32)(32 - sh) & 31u;
  const int k2 = 2 * mc.k, down = 64 - k2;
  const u32 pairsPerRound = (blockDim.x / WAVE) * 2;
  const u32 fLo = lo32(mc.factor1), fHi = hi32(mc.factor1);
  u32 r = (u32)wv * 2 + (u32)(lane >> 5);
  uint4 nx = make_uint4(0, 0, 0, 0);
  const u32 *pn = rec + (rec0 + r) * 30 + wi;                // this lane's words of its next read pair: a running pointer (one 64-bit add per pair instead of the product)
  if (r < nRead && cnt > 0) nx = make_uint4(pn[0], pn[1], pn[2], pn[3]);
  for (; __builtin_amdgcn_ballot_w64(r < nRead); r += pairsPerRound) {
    const uint4 cw = nx;
    const u32 rn = r + pairsPerRound;
    pn += (size_t)pairsPerRound * 30;
    if (rn < nRead && cnt > 0 && !H10X_MOSH_DBG(512)) nx = make_uint4(pn[0], pn[1], pn[2], pn[3]);
    if (H10X_MOSH_DBG(512)) nx = make_uint4(cw.y * 0x9E3779B1u + r, cw.x ^ 0x85EBCA6Bu, cw.w + cw.x, cw.z * 5u + 1u);
    const bool live = r < nRead && !H10X_MOSH_DBG(32);
    if (FAST) {
      // the run's bases, MSB first (fq2b.c:33-42; the un-justified tail word is consumed as is: SURVEY F6): base i at bits [95 - 2i, 94 - 2i] of (x0, x1, x2)
      const u32 x0 = __builtin_amdgcn_alignbit(cw.x, cw.y, unal), x1 = __builtin_amdgcn_alignbit(cw.y, cw.z, unal), x2 = __builtin_amdgcn_alignbit(cw.z, cw.w, unal);
      const u32 inw = __builtin_amdgcn_alignbit(x1, x2, (u32)down);     // base k + j at bits [31 - 2j, 30 - 2j]   (2k - 32 = 32 - down bits into x1)
      u32 ta[L - 1];                                                    // byte offset of slot j's pair of roll terms: (b(j) << 2 | b(j+k)) * 16
      if constexpr (L == 8) {
        // both streams hold base j at the same bits: the even bases of the two side by side make the nibbles of one word, the odd ones of another;
        // a nibble at the top of a byte IS the offset (x 16), picked with one byte-select AND (15 instructions for the seven instead of 23)
        const u32 ze = (x0 & 0xCCCCCCCCu) | ((inw >> 2) & 0x33333333u), zo = ((x0 << 2) & 0xCCCCCCCCu) | (inw & 0x33333333u);
        const u32 zes = ze << 4, zos = zo << 4;
        ta[0] = (ze >> 24) & 0xF0u; ta[1] = (zo >> 24) & 0xF0u; ta[2] = (zes >> 24) & 0xF0u; ta[3] = (zos >> 24) & 0xF0u;
        ta[4] = (ze >> 16) & 0xF0u; ta[5] = (zo >> 16) & 0xF0u; ta[6] = (zes >> 16) & 0xF0u;
#pragma unroll
        for (int j = 0; j < L - 1; ++j) asm("" : "+v"(ta[j]));
      } else {
#pragma unroll
        for (int j = 0; j < L - 1; ++j) { ta[j] = (((x0 >> (30 - 2 * j)) & 3u) << 6) | (((inw >> (30 - 2 * j)) & 3u) << 4); asm("" : "+v"(ta[j])); }   // (or its two halves are kept and joined twice)
      }
      // last window -> its reverse complement -> product; then back to slot 0
      u64 pr[L];
      {
        const u32 g0 = __builtin_amdgcn_alignbit(x0, x1, 32 - 2 * (L - 1)), g1 = __builtin_amdgcn_alignbit(x1, x2, 32 - 2 * (L - 1));
        u32 rl, rh;
        revcomp_halves(__builtin_amdgcn_alignbit(g0, g1, (u32)down), g0 >> down, down, rl, rh);
        u32 pl, ph;
        mul64_lo(rl, rh, fLo, fHi, pl, ph);
        pr[L - 1] = mk64(ph, pl);
      }
#pragma unroll
      for (int j = L - 2; j >= 0; --j) pr[j] = roll_product(pr[j + 1], *(const u64 *)((const char *)sRoll + ta[j] + 8));
      u64 pf;
      { u32 pl, ph; mul64_lo(__builtin_amdgcn_alignbit(x0, x1, (u32)down), x0 >> down, fLo, fHi, pl, ph); pf = mk64(ph, pl); }
      u32 thr = live && cnt > 0 ? 0xFFFFFFFFu / 31u + 1u : 0u;          // x divisible by 31 <=> x * (1 / 31 mod 2^32) <= (2^32 - 1) / 31
      asm("" : "+v"(thr));                                              // (kept a per-lane bound: as a condition of its own it costs two more instructions per slot)
      const u32 r16 = r << 16;
#pragma unroll
      for (int j = 0; j < L; ++j) {
        u64 t = 0;
        if (j < L - 1) t = *(const u64 *)((const char *)sRoll + ta[j]);   // asked for before the slot's test, used after it
        u32 mL, mH;
        min64_halves(lo32(pf), hi32(pf), lo32(pr[j]), hi32(pr[j]), mL, mH);       // seqhash.c:67-68, the shift of :58-59 comes after the min: it is monotone
        const u32 hL = __builtin_amdgcn_alignbit(mH, mL, (u32)down), hH = mH >> down;
        bool ok;
        if (W31) ok = ((hL & 0x3FFFFFFFu) + __builtin_amdgcn_alignbit(hH, hL, 30)) * 0xBDEF7BDFu < thr;   // 2^30 == 1 (mod 31); h < 2^60
        else ok = (mk64(hH, hL) % (u64)mc.w) == 0 && thr != 0;
        queue_push(queue, qn, ok, hL, hH, r16);
        if (H10X_MOSH_DBG(16)) { if (qn > MOSH_QKEEP) qn = 0; }
        else if (qn > MOSH_QKEEP && !queue_drain(table, mask, queue, qn, lane)) sOverflow = 1;
        if (j < L - 1) pf = roll_product(pf, t);
      }
      continue;
    }
    // 128-bit shift register, bases MSB first (fq2b.c:33-42; the un-justified tail word is consumed as is: SURVEY F6)
    u64 hi = ((u64)cw.x << 32) | cw.y, lo = ((u64)cw.z << 32) | cw.w;
    if (sh) { hi = (hi << sh) | (lo >> (64 - sh)); lo <<= sh; }
    u64 f = hi >> down;
    u64 rc = revcomp_word(f, mc.k);
#pragma unroll
    for (int j = 0; j < L; ++j) {
      const u64 hf = (f * mc.factor1) >> mc.shift1;          // seqhash.c:58-59
      const u64 hr = (rc * mc.factor1) >> mc.shift1;
      const u64 h = hf < hr ? hf : hr;                       // seqhash.c:67-68
      queue_push(queue, qn, divisible<W31>(h, mc.w) && live && j < cnt, lo32(h), hi32(h), r << 16);
      if (qn > MOSH_QKEEP && !queue_drain(table, mask, queue, qn, lane)) sOverflow = 1;
      hi = (hi << 2) | (lo >> 62); lo <<= 2;                 // next base in
      f = hi >> down;
      rc = (rc >> 2) | ((u64)(3u - ((u32)f & 3u)) << (k2 - 2));
    }
  }
  while (qn) if (!queue_drain(table, mask, queue, qn, lane)) { sOverflow = 1; break; }
  __syncthreads();

  // compact the set to this block's staging slice (order inside a block is irrelevant downstream)
  const u64 out0 = capOff[code];
  const u32 cap = S - (S >> 3);                              // > 87.5 % full => treat as overflow
  for (u32 base = 0; base < (H10X_MOSH_DBG(64) ? 0u : S); base += blockDim.x) {
    const u32 i = base + threadIdx.x;
    const u64 v = i < S ? table[i] : EMPTY64;
    const bool valid = !entry_empty(v);
    const u64 bal = __ballot(valid);
    const int lane = threadIdx.x & (WAVE - 1);
    u32 wbase = 0;
    if (lane == 0 && bal) wbase = atomicAdd(&sCount, (u32)__popcll(bal));
    wbase = __shfl(wbase, 0);
    if (valid) {
      const u32 pos = wbase + (u32)__popcll(bal & ((1ULL << lane) - 1));
      if (pos < cap) { stHash[out0 + pos] = mk64(hi32(v) >> 16, __builtin_amdgcn_alignbit(hi32(v), lo32(v), 16)); stRead[out0 + pos] = lo32(v) & 0xFFFFu; }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 n = sCount;
    if (sOverflow || n > cap) n = NHASH_OVERFLOW;
    else if (n == 0) { stHash[out0] = 0; stRead[out0] = 0; n = 1; }   // hash10x.c:167-174 quirk
    nHashOut[code] = n;
  }
}

// ------------------------------------------------------------------------------------------ global path
// raw slot (read r, k-mer t) -> key = canonical hash if it is a mosh, else all-ones; slots are in read
// order, so a stable sort by key leaves the lowest read first in every run of equal hashes
template <bool W31>
__global__ __launch_bounds__(MOSH_THREADS)
void mosh_raw_kernel(const u32 *__restrict__ rec, u64 rec0, u32 nRead, MoshConst mc, u64 *__restrict__ rawKey, u32 *__restrict__ rawSlot) {
  __shared__ u32 tile[REC_TILE * 2 * SEQ_WORDS];
  const int nk = mc.n1 + mc.n2;
  for (u32 i = threadIdx.x; i < REC_TILE * 2 * SEQ_WORDS; i += blockDim.x) tile[i] = 0;
  for (u32 r0 = blockIdx.x * REC_TILE; r0 < nRead; r0 += gridDim.x * REC_TILE) {
    const int cnt = (int)min((u32)REC_TILE, nRead - r0);
    __syncthreads();
    stage_records(rec, rec0 + r0, cnt, tile);
    __syncthreads();
    for (int rr = 0; rr < cnt; ++rr)
      for (int t = threadIdx.x; t < nk; t += blockDim.x) {
        u64 h; const bool ok = mosh_of_slot<W31>(tile + rr * 2 * SEQ_WORDS, t, mc, h);
        const u64 slot = (u64)(r0 + rr) * nk + t;
        rawKey[slot] = ok ? h : EMPTY64;
        rawSlot[slot] = (u32)slot;
      }
  }
}

__global__ void unique_flags_kernel(const u64 *__restrict__ key, u64 n, u32 *__restrict__ flags) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) flags[i] = (key[i] != EMPTY64 && (i == 0 || key[i] != key[i - 1])) ? 1u : 0u;
}
__global__ void unique_scatter_kernel(const u64 *__restrict__ key, const u32 *__restrict__ slot, const u32 *__restrict__ flags,
                                      const u32 *__restrict__ pos, u64 n, u32 nk, u64 *__restrict__ outHash, u32 *__restrict__ outRead) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) if (flags[i]) { outHash[pos[i]] = key[i]; outRead[pos[i]] = slot[i] / nk; }
}

// ------------------------------------------------------------------------------------------ gather
__global__ void set_nhash_kernel(h10x_block *__restrict__ blocks, const u32 *__restrict__ nHash, u32 nBlocks) {
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nBlocks) blocks[c].nHash = nHash[c];
}

struct SrcPtr { const u64 *hash; const u32 *read; };
__global__ void count_overflow_kernel(const u32 *__restrict__ nHash, u32 nBlocks, u32 *__restrict__ count) {
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  const u64 bal = __ballot(c < nBlocks && nHash[c] == NHASH_OVERFLOW);
  if ((threadIdx.x & (WAVE - 1)) == 0 && bal) atomicAdd(count, (u32)__popcll(bal));
}

__global__ void compact_entries_kernel(const u64 *__restrict__ stHash, const u32 *__restrict__ stRead, const u64 *__restrict__ capOff,
                                       const SrcPtr *__restrict__ fb /* per block, null hash = staging */,
                                       const u32 *__restrict__ nHash, const u64 *__restrict__ blockOff, u32 nBlocks,
                                       u64 keyInv, int keyShift, int codeBits /* > 0: packed entries (Ctx::entCodeBits), entCode unused */,
                                       u64 *__restrict__ entHash, u32 *__restrict__ entCode, u32 *__restrict__ entRead) {
  for (u32 c = blockIdx.x + 1; c < nBlocks; c += gridDim.x) {
    const u32 n = nHash[c]; if (!n) continue;
    const u64 *sh; const u32 *sr;
    if (fb[c].hash) { sh = fb[c].hash; sr = fb[c].read; } else { sh = stHash + capOff[c]; sr = stRead + capOff[c]; }
    const u64 o = blockOff[c];
    if (codeBits) for (u32 i = threadIdx.x; i < n; i += blockDim.x) { entHash[o + i] = (((sh[i] >> keyShift) * keyInv) << codeBits) | c; entRead[o + i] = sr[i]; }
    else for (u32 i = threadIdx.x; i < n; i += blockDim.x) { entHash[o + i] = (sh[i] >> keyShift) * keyInv; entCode[o + i] = c; entRead[o + i] = sr[i]; }   // hash / w (Ctx::keyInv)
  }
}

// ------------------------------------------------------------------------------------------ record sort
// What the reference leaves to an external tool (README.md:26 `bsort -k 4 -r 120`): order the 120-byte records by
// their first four BYTES (byte 0 most significant = the byte-swapped barcode word), so that equal barcodes are
// contiguous. Stable: records of a barcode keep their file order (bsort is not stable; hash10x only needs the runs).
__global__ void sort_keys_kernel(const u32 *__restrict__ rec, u64 n, u32 *__restrict__ key, u32 *__restrict__ idx) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) { key[i] = __builtin_bswap32(rec[i * 30]); idx[i] = (u32)i; }
}
// one lane per dword: a record's 30 dwords are read from one 120-byte stretch and written fully coalesced
__global__ void gather_records_kernel(const u32 *__restrict__ rec, const u32 *__restrict__ idx, u64 n, u32 *__restrict__ out) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x, total = n * 30;
  for (; t < total; t += stride) { const u64 j = t / 30; const u32 w = (u32)(t - j * 30); out[t] = rec[(u64)idx[j] * 30 + w]; }
}
int stageA_sortRecords(Ctx *c, const u32 *dIn, u64 n, u32 *dOut) {
  hipStream_t st = c->stream; PrimTemp pt;
  if (n >= (1ULL << 32)) return c->fail("record sort: %llu records exceed this build's 2^32 limit", (u64)n);
  if (!n) return 0;
  DevBuf<u32> key, keyS, idx, idxS;
  H10X_HIP(c, key.alloc(n)); H10X_HIP(c, keyS.alloc(n)); H10X_HIP(c, idx.alloc(n)); H10X_HIP(c, idxS.alloc(n));
  const unsigned g = (unsigned)hmin<u64>(divUp(n, 256), 65535u * 4);
  sort_keys_kernel<<<g, 256, 0, st>>>(dIn, n, key.p, idx.p);
  H10X_TRY(prim_sort_pairs_u32_u32(c, pt, key.p, keyS.p, idx.p, idxS.p, n, 0, 32));
  gather_records_kernel<<<(unsigned)hmin<u64>(divUp(n * 30, 256), 65535u * 16), 256, 0, st>>>(dIn, idxS.p, n, dOut);
  H10X_HIP(c, hipGetLastError());
  H10X_HIP(c, hipStreamSynchronize(st));
  return 0;
}

// ------------------------------------------------------------------------------------------ the reference's chunk loop
// readFQB reads chunkSize - (records of the open block) records at a time (hash10x.c:202-211), so a chunk always ends
// chunkSize records behind the start of the block that was open when it began. Two things follow from the chunk
// boundaries and nothing else: a block with chunkSize or more records dies ("chunkSize too small"), and — `if (!barcode)
// barcode = u[0]` at every chunk start (hash10x.c:212) — a run of the all-A barcode (word 0) that ends exactly at a chunk
// boundary swallows the run behind it. Both are replayed here from the run starts alone.
int replayChunks(const std::vector<u64> &starts, const std::vector<u32> &zeroRuns, u64 chunk, std::vector<u64> &merges, bool eofPass) {
  merges.clear();
  const size_t R = starts.size() - 1; const u64 total = starts[R];
  if (!R || !total) return 0;
  std::vector<u32> zr(zeroRuns); std::sort(zr.begin(), zr.end());
  auto isZero = [&](size_t r) { return std::binary_search(zr.begin(), zr.end(), (u32)r); };
  auto runOf = [&](u64 pos) { return (size_t)(std::upper_bound(starts.begin(), starts.begin() + R, pos) - starts.begin()) - 1; };
  std::vector<size_t> swallowed;                             // runs without a head of their own (ascending)
  u64 pos = 0, head = 0; bool open = false, barcodeZero = true;
  while (pos < total) {
    const u64 carry = open ? pos - head : 0;
    if (chunk <= carry) return 1;
    if (barcodeZero && open) {                               // barcode = u[0]: the first record of this chunk joins the open block whatever it is
      const size_t r = runOf(pos);
      if (starts[r] == pos) { merges.push_back(pos); swallowed.push_back(r); }
    }
    if (!open) { head = pos; open = true; }
    const u64 end = head + chunk < total ? head + chunk : total;
    size_t q = runOf(end - 1);
    barcodeZero = isZero(q);                                 // the barcode the loop holds at the end of the chunk
    while (std::binary_search(swallowed.begin(), swallowed.end(), q)) --q;
    head = starts[q];
    pos = end;
  }
  // Unless -N ended the loop (hash10x.c:202: `while (!N || nReads < N)`), the reference comes round once more at end of file and
  // tests `chunkSize - b->nRead <= 0` BEFORE the fread that finds the end (hash10x.c:205-208): a file whose last block holds
  // exactly chunkSize records dies there
  if (eofPass && open && total - head >= chunk) return 1;
  return 0;
}

int stageA_runStarts(Ctx *c, const u32 *dRec, u64 nRec, std::vector<u64> &starts, std::vector<u32> &zeroRuns) {
  hipStream_t st = c->stream; PrimTemp pt;
  starts.assign(1, 0); zeroRuns.clear();
  if (!nRec) return 0;
  DevBuf<u32> flags, code; DevBuf<u64> startRec;
  H10X_HIP(c, flags.alloc(nRec)); H10X_HIP(c, code.alloc(nRec));
  const unsigned g = (unsigned)hmin<u64>(divUp(nRec, 256), 65535u * 4);
  head_flags_kernel<<<g, 256, 0, st>>>(dRec, nRec, flags.p);
  H10X_TRY(prim_inclusive_scan_u32(c, pt, flags.p, code.p, nRec));
  u32 nRuns = 0;
  H10X_TRY(c->readback(&nRuns, code.p + (nRec - 1), 4));
  H10X_TRY(c->syncReadbacks());
  const u32 nBlocks = nRuns + 1;
  H10X_HIP(c, startRec.alloc((size_t)nBlocks + 1));
  block_starts_kernel<<<g, 256, 0, st>>>(flags.p, code.p, nRec, nBlocks, startRec.p);
  constexpr u32 ZCAP = 4096;
  DevBuf<u32> zl, zc; H10X_HIP(c, zl.alloc(ZCAP)); H10X_HIP(c, zc.alloc(1)); H10X_HIP(c, hipMemsetAsync(zc.p, 0, 4, st));
  zero_runs_kernel<<<divUp(nBlocks, 256), 256, 0, st>>>(dRec, startRec.p, nBlocks, zl.p, ZCAP, zc.p);
  starts.resize((size_t)nRuns + 1);
  H10X_HIP(c, hipMemcpyAsync(starts.data(), startRec.p + 1, ((size_t)nRuns + 1) * 8, hipMemcpyDeviceToHost, st));
  u32 nz = 0;
  H10X_TRY(c->readback(&nz, zc.p, 4));
  H10X_TRY(c->syncReadbacks());
  if (nz > ZCAP) return c->fail("%u separate runs of the all-A barcode: the input is not grouped by barcode", nz);
  zeroRuns.resize(nz);
  if (nz) { H10X_HIP(c, hipMemcpyAsync(zeroRuns.data(), zl.p, (size_t)nz * 4, hipMemcpyDeviceToHost, st)); H10X_HIP(c, hipStreamSynchronize(st)); }
  return 0;
}

// ------------------------------------------------------------------------------------------ driver
int stageA_run(Ctx *c, const u32 *dRec, u64 nRec, DevBuf<u64> &entHash, DevBuf<u32> &entCode, DevBuf<u32> &entRead, bool hashLast, bool emptyIsNoBlock) {
  hipStream_t st = c->stream;
  PrimTemp pt;
  const int k = c->prm.k;
  MoshConst mc; mc.k = k; mc.w = c->prm.w; mc.shift1 = 64 - 2 * k; mc.factor1 = c->prm.factor1;
  mc.n1 = 127 - k + 1 > 0 ? 127 - k + 1 : 0;               // len < k => no k-mer (seqhash.c:162)
  mc.n2 = 150 - k + 1 > 0 ? 150 - k + 1 : 0;
  mc.run = 8; while ((mc.n1 + mc.run - 1) / mc.run + (mc.n2 + mc.run - 1) / mc.run > 32) ++mc.run;   // 8 at k = 21, 9 for k <= 7
  const bool w31 = c->prm.w == 31;

  // ---- barcode runs (hash10x.c:212-220)
  c->tstart(T_RUNS);
  u32 nBlocks = 2;                                           // empty input: block 1 with nRead 0 (hash10x.c:200-201)
  DevBuf<u64> startRec;
  if (nRec) {
    DevBuf<u32> flags, code;
    H10X_HIP(c, flags.alloc(nRec)); H10X_HIP(c, code.alloc(nRec));
    const unsigned g = (unsigned)hmin<u64>(divUp(nRec, 256), 65535u * 4);
    if (c->optChunk > 0 && !c->replayDone) {                 // the reference's chunk loop over this (whole) file
      std::vector<u64> starts; std::vector<u32> zeroRuns;
      H10X_TRY(stageA_runStarts(c, dRec, nRec, starts, zeroRuns));
      if (replayChunks(starts, zeroRuns, (u64)c->optChunk, c->mergePoints, c->optChunkEof != 0)) return c->fail("chunkSize too small");   // hash10x.c:206
    }
    c->replayDone = false;
    head_flags_kernel<<<g, 256, 0, st>>>(dRec, nRec, flags.p);
    if (!c->mergePoints.empty()) {
      DevBuf<u64> at; H10X_HIP(c, at.alloc(c->mergePoints.size()));
      H10X_HIP(c, hipMemcpyAsync(at.p, c->mergePoints.data(), c->mergePoints.size() * 8, hipMemcpyHostToDevice, st));
      clear_heads_kernel<<<divUp(c->mergePoints.size(), 256), 256, 0, st>>>(flags.p, at.p, (u32)c->mergePoints.size());
      H10X_HIP(c, hipStreamSynchronize(st));                 // the host vector is released next
      c->mergePoints.clear();
    }
    H10X_TRY(prim_inclusive_scan_u32(c, pt, flags.p, code.p, nRec));
    u32 nRuns = 0;
    H10X_TRY(c->readback(&nRuns, code.p + (nRec - 1), 4));
    H10X_TRY(c->syncReadbacks());
    nBlocks = nRuns + 1;
    H10X_HIP(c, startRec.alloc((size_t)nBlocks + 1));
    block_starts_kernel<<<g, 256, 0, st>>>(flags.p, code.p, nRec, nBlocks, startRec.p);
  } else {
    if (emptyIsNoBlock) nBlocks = 1;                         // a shard without records holds no barcode at all (only the unused block 0)
    H10X_HIP(c, startRec.alloc(3));
    H10X_HIP(c, hipMemsetAsync(startRec.p, 0, 24, st));
  }
  c->nBlocks = nBlocks; c->nRecords = nRec;
  H10X_HIP(c, c->blocks.alloc(nBlocks));

  // ---- classes
  u32 maxSlots = 16384;                                      // LDS table classes: 4096 / 8192 / 16384 slots = 32 / 64 / 128 KB
  if (c->optMaxSlots > 0) maxSlots = (u32)c->optMaxSlots;
  const int packedOK = (2 * k + 16 <= 64) ? 1 : 0;
  DevBuf<u32> slots, list0, list1, list2, listF, counts, nHash;
  H10X_HIP(c, slots.alloc(nBlocks)); H10X_HIP(c, list0.alloc(nBlocks)); H10X_HIP(c, list1.alloc(nBlocks)); H10X_HIP(c, list2.alloc(nBlocks));
  H10X_HIP(c, listF.alloc(nBlocks)); H10X_HIP(c, counts.alloc(4)); H10X_HIP(c, nHash.alloc((size_t)nBlocks + 1));
  H10X_HIP(c, hipMemsetAsync(counts.p, 0, 16, st));
  H10X_HIP(c, hipMemsetAsync(nHash.p, 0, ((size_t)nBlocks + 1) * 4, st));
  classify_kernel<<<divUp(nBlocks, 256), 256, 0, st>>>(startRec.p, nBlocks, c->blocks.p, slots.p, maxSlots, packedOK,
                                                      list0.p, list1.p, list2.p, listF.p, counts.p, hashLast ? 1 : 0);
  DevBuf<u64> capOff; H10X_HIP(c, capOff.alloc((size_t)nBlocks + 1));
  H10X_TRY(prim_exclusive_scan_u32_u64(c, pt, slots.p, capOff.p, nBlocks));
  u32 hc[4]; u64 capTotal = 0; u32 lastSlots = 0;
  H10X_TRY(c->readback(hc, counts.p, 16));
  H10X_TRY(c->readback(&capTotal, capOff.p + (nBlocks - 1), 8));
  H10X_TRY(c->readback(&lastSlots, slots.p + (nBlocks - 1), 4));
  H10X_TRY(c->syncReadbacks());
  capTotal += lastSlots;
  c->tstop(T_RUNS);

  // ---- LDS path
  DevBuf<u64> stHash; DevBuf<u32> stRead;
  H10X_HIP(c, stHash.alloc(capTotal)); H10X_HIP(c, stRead.alloc(capTotal));
  c->tstart(T_MOSH);
  // the three table classes are independent: side by side on forked streams, the few blocks with many read pairs
  // (largest tables, most lanes per workgroup) first
  ForkGuard forkGuard(c);                                    // (declared behind every buffer the side streams touch)
  H10X_TRY(c->forkStreams(2));
#define H10X_MOSH_LAUNCH(W, LL, FK, TH, STREAM)                                                                                    \
    { if (lds > 48 * 1024) H10X_HIP(c, hipFuncSetAttribute((const void *)mosh_lds_kernel<W, LL, FK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      mosh_lds_kernel<W, LL, FK><<<n, TH, lds, STREAM>>>(dRec, list, n, startRec.p, slots.p, capOff.p, mc, (u32)tableBytes, stHash.p, stRead.p, nHash.p); }
#define H10X_MOSH_CLASS(TH, STREAM)                                                                                                \
    { if (fastK && w31) H10X_MOSH_LAUNCH(true, 8, true, TH, STREAM) else if (fastK) H10X_MOSH_LAUNCH(false, 8, true, TH, STREAM)   \
      else if (w31 && mc.run == 8) H10X_MOSH_LAUNCH(true, 8, false, TH, STREAM) else if (w31) H10X_MOSH_LAUNCH(true, 9, false, TH, STREAM) \
      else if (mc.run == 8) H10X_MOSH_LAUNCH(false, 8, false, TH, STREAM) else H10X_MOSH_LAUNCH(false, 9, false, TH, STREAM) }
  // workgroups of 512 lanes for the 32 KB tables too: four of them fill a CU's 32 wave slots (256-lane ones left half empty:
  // 1.20 -> 1.06 ms on the yeast-scale set); the 64 KB class gains nothing from 1024 lanes (measured)
  const bool fastK = k >= 17 && k <= 30 && mc.run == 8;
  for (int cls = 2; cls >= 0; --cls) {
    const u32 n = hc[cls]; if (!n) continue;
    const u32 *list = cls == 0 ? list0.p : cls == 1 ? list1.p : list2.p;
    const size_t tableBytes = (size_t)hmin<u32>(4096u << cls, maxSlots) * 8;
    const size_t lds = tableBytes + (size_t)((cls == 2 ? 1024 : 512) / WAVE) * MOSH_QCAP * 8;        // the table, then one survivor queue per wave
    if (cls == 2) H10X_MOSH_CLASS(1024, c->aux[0]) else if (cls == 1) H10X_MOSH_CLASS(512, c->aux[1]) else H10X_MOSH_CLASS(512, st)
    H10X_HIP(c, hipGetLastError());
  }
#undef H10X_MOSH_CLASS
#undef H10X_MOSH_LAUNCH
  H10X_TRY(c->faultAt(1));
  H10X_TRY(c->joinStreams(2));
  forkGuard.done();
  c->tstop(T_MOSH);

  // ---- global path: class F plus any block whose LDS set overflowed. The usual case — no such block — is recognised
  // from one counter that comes back together with the entry total (no copy of the per-block arrays, no extra round trip).
  DevBuf<u32> nOverflow; H10X_HIP(c, nOverflow.alloc(1)); H10X_HIP(c, hipMemsetAsync(nOverflow.p, 0, 4, st));
  count_overflow_kernel<<<divUp(nBlocks, 256), 256, 0, st>>>(nHash.p, nBlocks, nOverflow.p);
  H10X_HIP(c, c->blockOff.alloc((size_t)nBlocks + 1));
  H10X_TRY(prim_exclusive_scan_u32_u64(c, pt, nHash.p, c->blockOff.p, (size_t)nBlocks + 1));
  u64 H = 0, lastStart = 0; u32 hOverflow = 0, hMaxHash = 0xFFFFFFFFu;
  DevBuf<u32> dMaxHash; H10X_HIP(c, dMaxHash.alloc(1));
  H10X_TRY(prim_reduce_max_u32(c, pt, nHash.p, dMaxHash.p, nBlocks));       // an overflowed block reads as the largest value: "unknown" until repaired
  H10X_TRY(c->readback(&hMaxHash, dMaxHash.p, 4));
  H10X_TRY(c->readback(&H, c->blockOff.p + nBlocks, 8));
  H10X_TRY(c->readback(&lastStart, startRec.p + (nBlocks - 1), 8));
  H10X_TRY(c->readback(&hOverflow, nOverflow.p, 4));
  H10X_TRY(c->syncReadbacks());
  const bool anyFallback = hc[3] || hOverflow;
  std::vector<u32> hNHash, hListF(hc[3]);
  std::vector<u64> hStart;
  if (anyFallback) {
    hNHash.resize(nBlocks + 1); hStart.resize(nBlocks + 1);
    H10X_HIP(c, hipMemcpyAsync(hNHash.data(), nHash.p, (size_t)nBlocks * 4, hipMemcpyDeviceToHost, st));
    H10X_HIP(c, hipMemcpyAsync(hStart.data(), startRec.p, ((size_t)nBlocks + 1) * 8, hipMemcpyDeviceToHost, st));
    if (hc[3]) H10X_HIP(c, hipMemcpyAsync(hListF.data(), listF.p, (size_t)hc[3] * 4, hipMemcpyDeviceToHost, st));
    H10X_HIP(c, hipStreamSynchronize(st));
    for (u32 b = 1; b + (hashLast ? 0 : 1) < nBlocks; ++b) if (hNHash[b] == NHASH_OVERFLOW) hListF.push_back(b);
  }
  std::vector<SrcPtr> hFb;
  std::vector<DevBuf<u64> *> keepH; std::vector<DevBuf<u32> *> keepR;
  c->ctr.fallback_blocks = hListF.size();
  if (!hListF.empty()) {
    hFb.assign(nBlocks, SrcPtr{nullptr, nullptr});
    c->tstart(T_FALLBACK);
    const u32 nk = (u32)(mc.n1 + mc.n2);
    for (u32 b : hListF) {
      const u32 nr = (u32)(hStart[b + 1] - hStart[b]);
      const u64 nSlot = (u64)nr * nk;
      DevBuf<u64> *oh = new DevBuf<u64>(); DevBuf<u32> *orr = new DevBuf<u32>(); keepH.push_back(oh); keepR.push_back(orr);
      u32 nUniq = 0;
      if (nSlot >= (1ULL << 32)) return c->fail("barcode block %u too large for the global mosh path (%llu k-mer slots)", b, (u64)nSlot);
      if (nSlot) {
        DevBuf<u64> k0, k1; DevBuf<u32> v0, v1, fl, ps;
        H10X_HIP(c, k0.alloc(nSlot)); H10X_HIP(c, k1.alloc(nSlot)); H10X_HIP(c, v0.alloc(nSlot)); H10X_HIP(c, v1.alloc(nSlot));
        H10X_HIP(c, fl.alloc(nSlot + 1)); H10X_HIP(c, ps.alloc(nSlot + 1));
        const unsigned g = (unsigned)hmin<u64>(divUp(nr, REC_TILE), 4096);
        if (w31) mosh_raw_kernel<true><<<g, MOSH_THREADS, 0, st>>>(dRec, hStart[b], nr, mc, k0.p, v0.p);
        else     mosh_raw_kernel<false><<<g, MOSH_THREADS, 0, st>>>(dRec, hStart[b], nr, mc, k0.p, v0.p);
        H10X_TRY(prim_sort_pairs_u64_u32(c, pt, k0.p, k1.p, v0.p, v1.p, nSlot, 0, 64));
        const unsigned g2 = (unsigned)hmin<u64>(divUp(nSlot, 256), 65535);
        unique_flags_kernel<<<g2, 256, 0, st>>>(k1.p, nSlot, fl.p);
        H10X_HIP(c, hipMemsetAsync(fl.p + nSlot, 0, 4, st));
        H10X_TRY(prim_exclusive_scan_u32(c, pt, fl.p, ps.p, nSlot + 1));
        H10X_HIP(c, hipMemcpyAsync(&nUniq, ps.p + nSlot, 4, hipMemcpyDeviceToHost, st));
        H10X_HIP(c, hipStreamSynchronize(st));
        H10X_HIP(c, oh->alloc(nUniq ? nUniq : 1)); H10X_HIP(c, orr->alloc(nUniq ? nUniq : 1));
        if (nUniq) unique_scatter_kernel<<<g2, 256, 0, st>>>(k1.p, v1.p, fl.p, ps.p, nSlot, nk, oh->p, orr->p);
        H10X_HIP(c, hipStreamSynchronize(st));
      } else { H10X_HIP(c, oh->alloc(1)); H10X_HIP(c, orr->alloc(1)); }
      if (!nUniq) {                                           // no mosh in the whole block: one {hash 0, read 0} entry
        H10X_HIP(c, hipMemsetAsync(oh->p, 0, 8, st)); H10X_HIP(c, hipMemsetAsync(orr->p, 0, 4, st)); nUniq = 1;
      }
      hNHash[b] = nUniq; hFb[b] = SrcPtr{oh->p, orr->p};
    }
    H10X_HIP(c, hipMemcpyAsync(nHash.p, hNHash.data(), (size_t)nBlocks * 4, hipMemcpyHostToDevice, st));
    c->tstop(T_FALLBACK);
  }

  // ---- gather into one (block-ordered) entry list
  c->tstart(T_COMPACT);
  DevBuf<SrcPtr> dFb; H10X_HIP(c, dFb.alloc(nBlocks));
  if (hFb.empty()) H10X_HIP(c, hipMemsetAsync(dFb.p, 0, (size_t)nBlocks * sizeof(SrcPtr), st));       // every block comes from the staging area
  else {
    H10X_HIP(c, hipMemcpyAsync(dFb.p, hFb.data(), (size_t)nBlocks * sizeof(SrcPtr), hipMemcpyHostToDevice, st));
    H10X_TRY(prim_exclusive_scan_u32_u64(c, pt, nHash.p, c->blockOff.p, (size_t)nBlocks + 1));          // again, with the repaired counts
    H10X_HIP(c, hipMemcpyAsync(&H, c->blockOff.p + nBlocks, 8, hipMemcpyDeviceToHost, st));
    H10X_HIP(c, hipStreamSynchronize(st));
  }
  if (H >= (1ULL << 32)) return c->fail("%llu (barcode,hash) entries exceed this build's 2^32 per-GPU limit", (u64)H);
  c->nEntries = H;
  if (!hNHash.empty()) { hMaxHash = 0; for (u32 b = 0; b < nBlocks; ++b) hMaxHash = hNHash[b] > hMaxHash ? hNHash[b] : hMaxHash; }   // fallback path: the repaired counts
  c->maxBlockHashes = hMaxHash;
  set_nhash_kernel<<<divUp(nBlocks, 256), 256, 0, st>>>(c->blocks.p, nHash.p, nBlocks);
  H10X_HIP(c, entHash.alloc(H)); H10X_HIP(c, entRead.alloc(H));
  {                                                          // sort key = hash / w (common.hpp, Ctx::keyInv)
    u64 m = (u64)c->prm.w; int s = 0; while (!(m & 1)) { m >>= 1; ++s; }
    u64 inv = m; for (int it = 0; it < 6; ++it) inv *= 2 - m * inv;          // Newton: m^-1 mod 2^64 (m odd: 3 correct bits double each step)
    const int k2 = 2 * c->prm.k; const u64 maxH = k2 >= 64 ? ~0ULL : ((u64)1 << k2) - 1, maxQ = maxH / (u64)c->prm.w;
    int b = 1; while (b < 64 && (maxQ >> b)) ++b;
    c->keyInv = inv; c->keyShift = s; c->keyBits = b;
    int cb = 1; while (cb < 32 && ((nBlocks - 1) >> cb)) ++cb;
    // (63, not 64: rocPRIM 4.2's radix_sort_keys mis-sorts inputs of some thousand to a million keys when the bit range starts above
    // bit 0 and ends at bit 64 — scratch/sort_bits_check.hip; found by tests/soak.py at k = 31, w = 5 with 15 barcodes)
    c->entCodeBits = (c->wantPacked && b + cb <= 63) ? cb : 0;
  }
  if (!c->entCodeBits) H10X_HIP(c, entCode.alloc(H));
  if (H) compact_entries_kernel<<<hmin<u32>(nBlocks, 8192), 256, 0, st>>>(stHash.p, stRead.p, capOff.p, dFb.p, nHash.p, c->blockOff.p, nBlocks,
                                                                        c->keyInv, c->keyShift, c->entCodeBits, entHash.p, entCode.p, entRead.p);
  H10X_HIP(c, hipGetLastError());
  if (!hFb.empty()) H10X_HIP(c, hipStreamSynchronize(st));   // the host copy of the fallback pointers is still being read
  c->tstop(T_COMPACT);
  for (auto *p : keepH) delete p;
  for (auto *p : keepR) delete p;

  u64 hashedPairs = hashLast ? nRec : (nBlocks >= 2 ? lastStart : 0);  // all blocks but the file's last
  c->ctr.pairs = nRec; c->ctr.kmers = hashedPairs * (u64)(mc.n1 + mc.n2); c->ctr.entries = H;
  return 0;
}

// h10x_warm: the first launch of a kernel loads the code object of its translation unit (HIP loads them on first use); this one is launched ahead of time
__global__ void warm_stageA_kernel() {}
void warm_stageA(hipStream_t st) { warm_stageA_kernel<<<1, 1, 0, st>>>(); }

}  // namespace h10x
