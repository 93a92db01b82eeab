// common.hpp — context, device buffers, error plumbing and timers shared by the stage files.
// gfx950 only; wavefront = 64.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <cstdlib>
#include <ctime>
#include <string>
#include <vector>
#include <map>
#include <mutex>
#include "../../include/h10x.h"

namespace h10x {

constexpr int WAVE = 64;
typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t  u8;

static_assert(sizeof(h10x_block) == 32, "ClusterBlock layout (hash10x.c:62-70)");
static_assert(sizeof(h10x_clushash) == 8, "ClusterHash layout (hash10x.c:35-43)");

// ---- device buffer with explicit ownership ------------------------------------------------------
// A command allocates dozens of temporaries; hipMalloc/hipFree cost driver round trips and implicit device
// syncs. Blocks are therefore recycled through a per-device cache: release() parks the block, alloc() takes
// the smallest parked block that fits (within 2x). This is safe without events because every context issues
// all of its work on ONE stream and blocks are cached per stream: a block is only ever handed to work enqueued
// after the work that used it.
// (hipMallocAsync/hipFreeAsync were tried first and dropped: on ROCm 7.2 recycled blocks were handed out while
// still in use — nondeterministic results on the second --readFQB of a process.)
// Debug knobs: H10X_NOPOOL=1 (plain hipMalloc/hipFree), H10X_POISON=1 (fill every block handed out with 0xA5).
struct DevCache {
  std::multimap<size_t, void *> parked; size_t parkedBytes = 0;
  static std::mutex &mu() { static std::mutex m; return m; }
  // one cache per (device, stream): a parked block may only go back to work issued on the stream that last used it;
  // (device, nullptr) holds blocks whose stream has been synchronised (context destroyed): anyone may take those
  static DevCache &of(int device, hipStream_t st) { static std::map<std::pair<int, hipStream_t>, DevCache> m; return m[{device, st}]; }
  static bool disabled() { static int v = -1; if (v < 0) v = getenv("H10X_NOPOOL") != nullptr; return v != 0; }
  static unsigned long long &freshCalls() { static unsigned long long v = 0; return v; }   // blocks the cache could not serve (h10x_alloc_stats)
  static unsigned long long &freshBytes() { static unsigned long long v = 0; return v; }
  static int poison() { static int p = -1; if (p < 0) { const char *e = getenv("H10X_POISON"); p = e ? atoi(e) : 0; } return p; }
  void *takeLocked(size_t bytes, size_t &got) {
    auto it = parked.lower_bound(bytes);
    if (it == parked.end() || it->first > 2 * bytes + (1u << 20)) return nullptr;
    void *p = it->second; got = it->first; parkedBytes -= got; parked.erase(it);
    return p;
  }
  static void *take(int device, hipStream_t st, size_t bytes, size_t &got) {
    std::lock_guard<std::mutex> g(mu());
    void *p = of(device, st).takeLocked(bytes, got);
    if (!p && st) p = of(device, nullptr).takeLocked(bytes, got);
    return p;
  }
  static void park(int device, hipStream_t st, void *p, size_t bytes) {
    std::lock_guard<std::mutex> g(mu()); DevCache &c = of(device, st); c.parked.emplace(bytes, p); c.parkedBytes += bytes;
  }
  static void retireStream(int device, hipStream_t st) {     // call after the stream has been synchronised
    std::lock_guard<std::mutex> g(mu());
    DevCache &c = of(device, st), &idle = of(device, nullptr);
    for (auto &kv : c.parked) { idle.parked.emplace(kv.first, kv.second); idle.parkedBytes += kv.first; }
    c.parked.clear(); c.parkedBytes = 0;
  }
  static void trim(int device) {                             // give every parked block of the device back to the driver
    std::lock_guard<std::mutex> g(mu());
    (void)hipDeviceSynchronize();
    static std::map<std::pair<int, hipStream_t>, DevCache> *dummy = nullptr; (void)dummy;
    for (hipStream_t st : streamsOf(device)) { DevCache &c = of(device, st); for (auto &kv : c.parked) (void)hipFree(kv.second); c.parked.clear(); c.parkedBytes = 0; }
  }
  static std::vector<hipStream_t> &streamsOf(int device) { static std::map<int, std::vector<hipStream_t>> m; auto &v = m[device]; if (v.empty()) v.push_back(nullptr); return v; }
  static void noteStream(int device, hipStream_t st) { std::lock_guard<std::mutex> g(mu()); auto &v = streamsOf(device); for (auto x : v) if (x == st) return; v.push_back(st); }
};
struct AllocScope {                                       // set by every C-ABI entry point for its thread
  static hipStream_t &stream() { static thread_local hipStream_t s = nullptr; return s; }
  static int &device() { static thread_local int d = 0; return d; }
};
template <typename T> struct DevBuf {
  T *p = nullptr; size_t n = 0; size_t cap = 0; int dev = 0; hipStream_t st = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete; DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) { if (DevCache::disabled()) (void)hipFree(p); else DevCache::park(dev, st, p, cap); }
    p = nullptr; n = 0; cap = 0;
  }
  hipError_t alloc(size_t count) {                       // contents undefined
    release();
    if (!count) count = 1;
    size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
    dev = AllocScope::device(); st = AllocScope::stream();
    hipError_t e = hipSuccess;
    void *q = DevCache::disabled() ? nullptr : DevCache::take(dev, st, bytes, cap);
    if (!q) {
      static const bool log = getenv("H10X_ALLOC_LOG") != nullptr;   // debug knob: every block the cache could not serve, with the time hipMalloc took
      timespec t0, t1; if (log) clock_gettime(CLOCK_MONOTONIC, &t0);
      cap = bytes; e = hipMalloc(&q, bytes);
      if (e != hipSuccess && !DevCache::disabled()) { DevCache::trim(dev); e = hipMalloc(&q, bytes); }   // give parked blocks back and retry
      if (e == hipSuccess) { __atomic_fetch_add(&DevCache::freshCalls(), 1, __ATOMIC_RELAXED); __atomic_fetch_add(&DevCache::freshBytes(), (unsigned long long)bytes, __ATOMIC_RELAXED); }
      if (log) { clock_gettime(CLOCK_MONOTONIC, &t1); fprintf(stderr, "h10x alloc: %.1f MB fresh, %.3f ms\n", bytes / 1e6, (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) / 1e6); }
    }
    if (e == hipSuccess) { p = (T *)q; n = count; } else { p = nullptr; cap = 0; }
    if (e == hipSuccess && DevCache::poison()) (void)hipMemsetAsync(p, 0xA5, bytes, AllocScope::stream());
    return e;
  }
  void swap(DevBuf &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); std::swap(dev, o.dev); std::swap(st, o.st); }
  size_t bytes() const { return n * sizeof(T); }
};

// ---- timers ---------------------------------------------------------------------------------------
enum TimerId { T_RUNS = 0, T_MOSH, T_FALLBACK, T_COMPACT, T_SORT_HASH, T_RANK, T_PROBE, T_CLUSHASH, T_CSR,
               T_GOOD, T_CLUSTER, T_CLUSTER_K, T_CLUSTER_MAIN, T_SPLIT, T_COUNT };
static const char *const kTimerNames[T_COUNT] = {
  "block_runs", "mosh_extract", "mosh_fallback", "compact_entries", "sort_by_hash", "index_rank",
  "probe_table", "clushash_build", "csr_build", "good_hashes", "cluster", "cluster_kernel", "cluster_main", "cluster_split" };

struct Timer { hipEvent_t a = nullptr, b = nullptr; double ms = 0; uint64_t launches = 0; bool pending = false; };

// ---- the exchanges of the sharded path, one entry per KIND of collective (shard.hip goes through xchg_a2a / xchg_hostGather): calls, bytes that left this rank for
// other ranks and arrived from them (the part a rank keeps is a device copy, not counted), the largest share one peer got — what a single xGMI link carries, since
// the all-to-alls are grouped point-to-point sends — and, with timing on, the time from the call to its completion on the context's stream (the wait for slower
// ranks included: a rank's compute is its stage timers less these). h10x_exchange_get; bench.py --scaling strong and --virtual-ranks print them.
enum XchgId { X_HOST_COUNTS = 0, X_ENTRIES, X_FIRST_COUNTS, X_INDEX_BACK, X_INDEX_DEPTH, X_LIST_HEADS, X_LIST_DATA, X_TABLES, X_OWNER_LISTS, X_GATHER, X_COUNT };
static const char *const kXchgNames[X_COUNT] = {
  "host_counts (small host allgathers: sizes, counts)", "entries_to_hash_owners (all-to-all)", "first_seen_counts (allgather)", "indices_back (all-to-all)",
  "index_depth (allgather)", "list_heads (allgather: index + length of the in-range hashes)", "list_data (allgather: the in-range barcode lists)",
  "hash_tables (allgather: hashValue, for --writeHash / the crib)", "owner_lists_rebuild (after --clusterSplit)", "gather_to_rank0" };
struct XchgStat { u64 calls = 0, bytesOut = 0, bytesIn = 0, maxPeerOut = 0; Timer t, tIn; int beside = -1; };   // beside: the stage timer whose kernels run while this exchange is on the exchange stream (-1: on the main stream, nothing beside it)   // tIn: the calls made inside a stage timer's bracket (their time is part of that stage's figure)

// ---- local block number -> global block number ------------------------------------------------------
// An unsharded context numbers its blocks as the reference does (one segment, identity). A shard owns a contiguous range
// of the file's barcodes (segment 0: local l = global codeBase + l, slot 0 unused), and every --clusterSplit appends, per
// segment that held parents, one segment for the blocks it created: the reference numbers those after ALL existing
// blocks, in the order of their parents (hash10x.c:961-1003), so in a sharded run they sit behind the blocks of every rank.
// rows[] (the barcode lists) is allocated with this many entries of slack behind the last list: the cluster kernels read a list's first two
// 64-entry chunks without looking at its length (a load no lane skips keeps the compiler's wait counts exact: stage_c.hip descLoadU)
constexpr size_t ROWS_PAD = 128;
constexpr int MAX_SEGS = 8;
struct BlockSeg { u32 localStart, count, globalBase; };     // global = globalBase + (l - localStart)
struct SegMap {
  int n; BlockSeg s[MAX_SEGS];
  __host__ __device__ u32 globalOf(u32 l) const {
    for (int k = n - 1; k > 0; --k) if (l >= s[k].localStart) return s[k].globalBase + (l - s[k].localStart);
    return s[0].globalBase + l;
  }
};
struct ShardSegInfo { u32 rank, localStart, count, globalBase; u64 entries; };   // one segment of one rank (allgathered: shard.hip)

// ---- the context ------------------------------------------------------------------------------------
struct Ctx {
  h10x_params prm{};
  // Every mosh is a multiple of w (and the stand-in entry of a block without moshes is 0): the entries travel and are
  // sorted as q = hash / w, log2(w) fewer key bits (k = 21, w = 31: 38 instead of 42 = one radix pass less), and are
  // multiplied back where hashValue[] is written or the probe table is searched. Exact division: w = 2^keyShift * m,
  // q = (hash >> keyShift) * keyInv with keyInv = m^-1 mod 2^64. Set by stageA_run.
  u64 keyInv = 1; int keyShift = 0; int keyBits = 64;
  // Single-GPU index build: an entry's key and block number travel in ONE word, (hash / w) << entCodeBits | block, so the
  // device-wide sort moves 8 bytes per entry and pass instead of 12 (keys only, on bits [entCodeBits, entCodeBits + keyBits)).
  // 0 = separate arrays (sharded path, or the two do not fit 63 bits). wantPacked is set by the caller of stageA_run.
  bool wantPacked = false; int entCodeBits = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t aux[3] = {nullptr, nullptr, nullptr};   // side streams for independent launches (fork/join around them)
  hipEvent_t evFork = nullptr, evJoin[3] = {nullptr, nullptr, nullptr};
  int numCU = 256;
  bool ownStream = false;     // stream created (and destroyed) by the context
  std::string err;

  // persistent state == the reference's globals (hash10x.c:85-96), device resident
  DevBuf<u32> hashIndex;      // 2^B, 0 = empty
  DevBuf<u64> hashValue;      // hashNumber
  DevBuf<u32> hashDepth;      // hashNumber
  DevBuf<u64> rowStart;       // hashNumber+1 : offsets of each hash's barcode list in rows[]
  DevBuf<u32> rows;           // H : hashCodes lists (ascending barcode within a list)
  DevBuf<h10x_block> blocks;  // nBlocks
  DevBuf<u64> blockOff;       // nBlocks+1 : clusHash offset of each block
  DevBuf<h10x_clushash> clusHash;  // H
  u32 hashNumber = 1, nBlocks = 0;
  u32 maxBlockHashes = 0xFFFFFFFFu;   // no block of this context holds more entries (0xFFFFFFFF = unknown): blocks up to 8192 entries are sorted in LDS
  u64 nEntries = 0, nRecords = 0;
  bool haveState = false;

  // depth range / good hashes (hash10x.c:525-539, 722-766)
  DevBuf<u8>  within;         // hashNumber
  DevBuf<u16> goodPos;        // H : per block [blockOff[c], blockOff[c]+nGood[c])
  DevBuf<u32> nGood;          // nBlocks
  DevBuf<u64> goodRow;        // H : per good hash of a block, in rank order: (offset of its barcode list in rows[] >> rowShift, its length) — stage_c.hip good_rows_kernel
  DevBuf<u32> goodEntries;    // nBlocks : sum of the depths of a block's good hashes = entries of its barcode lists
  DevBuf<u32> clusterRaw;     // 2 x nBlocks, after --cluster: clusters before the read merge (bit 31: given up), good hashes labelled — the --verbose figures of codeClusterFind
  bool haveRange = false, haveGood = false; int rangeMin = 0, rangeMax = 0;
  u32 rangeHiMax = 0;             // largest upper limit of the ranges set so far: an in-range depth is below it
  u32 depthBound = 0xFFFFFFFFu;   // no hashDepth[] value exceeds this: barcodes of the data set after --readFQB, the largest value read after --readHash
  u32 maxGoodDepth = 0, maxGood = 0, meanGood = 0;

  // crib (hash10x.c:406-521): per hash index, where the two truth genomes hold it
  DevBuf<u32> cribCount[2];   // occurrences in genome 1 / 2
  DevBuf<u64> cribFirst[2];   // (sequence number << 32 | position) of the first occurrence
  DevBuf<int16_t> cribChr; DevBuf<u16> cribPos; DevBuf<u8> cribType;   // the merged crib[] / cribType[] of cribBuild
  DevBuf<u32> cribHist;       // 4 depth histograms (err, het, hom, mul), cribHistDim entries each, then the 4 arrayMax values
  u32 cribHistDim = 0; int cribGenomes = 0; bool haveCrib = false;

  // sharded (multi-GPU) operation: this context owns barcodes codeBase+1 .. codeBase+nBlocks-1 of nBlocksGlobal-1 and,
  // as hash owner, the barcode lists of the hashes in its hash range (shard.hip)
  struct Comm *comm = nullptr; bool sharded = false;
  u32 codeBase = 0, nBlocksGlobal = 0;
  SegMap segs{1, {{0, 0, 0}}};                               // this context's blocks in global numbering (identity when unsharded)
  std::vector<ShardSegInfo> allSegs;                         // sharded: every rank's segments, ascending globalBase = file order
  u64 nEntriesGlobal = 0, nRecordsGlobal = 0;                // sharded: whole data set
  int rowShift = 0;                                          // rows[]: lists start at multiples of 2^rowShift entries, rowStart[] in entries (shard.hip: > 2^32 list entries)
  bool ownerListsStale = false;                              // sharded: oRows predates a --clusterSplit
  DevBuf<u32> oRows, oSegStart, oIndex; u32 oU = 0; u64 oM = 0;
  DevBuf<u64> oHash;          // this owner's distinct hashes (by oIndex) until the global tables are built
  bool tablesPending = false; // sharded: hashValue[] / hashIndex[] not built yet (shard_materializeTables, collective)

  // options / measurement
  int64_t optMaxSlots = 0;    // testing knob: cap stage-A LDS table
  int64_t optClusterLds = 0;  // testing knob: LDS budget of cluster_kernel (forces the HBM-scratch path when small)
  int64_t optFirstGlobal = 0; // testing knob: 1 = first[] on HBM slots, 2 = ranked first[] in LDS, even when the dense table would fit
  int64_t optFirstCap = 0;    // testing knob: entries of the ranked first[]
  int64_t optClusterThreads0 = 0, optClusterBudget0 = 0;   // tuning knobs: lanes and LDS bytes of the first cluster class
  int64_t optBigRanks = 0;    // tuning knob: rank count above which a barcode goes to the front of the main work queue (0 = 1.5 x the mean)
  int64_t optPrivTable = 0;   // entry look-ups of the index build: 0 = the table of this library's own where key + index do not fit the 64-bit reference-shaped one, 1 = always (tests), 2 = never, 3 = always and too small (tests: the fall-back)
  int64_t optProbedTable = 0; // index build, the wide table's "probed" entry format (index | hash >> B | probe number: stage_b.hip probe_insertP_kernel): 0 = where the classic entry (index | hash / w) does not fit 64 bits (-B 29 / 30 at k = 21), 1 = always (tests), 2 = never (round 5: two tables), 3 = always with ONE bit of probe number (tests: the table fails and the two tables are built)
  int64_t optNoPack = 0;      // index build with separate key / block arrays even where the packed form fits (A/B, tests)
  int64_t optNarrowFirst = 0; // first[] of the cluster kernel at 2 bytes per entry in every block (default: 4 where the block's working set leaves room)
  int64_t optTrEstDiv = 0;    // tuning knob: translated placement, classification: a block's barcodes estimated as entries / this (0 = 6)
  int64_t optTrClassT = -1;   // packed translated placement: -1 / 1 = lists of 65 .. 96 entries run two to a unit (a whole chunk each + one shared by their tails: class T), 0 = as class D, a unit of two chunks each (A/B, tests)
  int64_t optTrPacked = -1;   // translated placement of the cluster kernel: -1 / 1 = packed (several lists per wave instruction), 0 = one list per wave instruction (round 4's form: A/B, tests)
  int64_t optStamps = 0;      // diagnostic: per-phase wall-clock stamps in cluster_kernel
  int64_t optChunk = 0;       // -c <chunkSize> of the reference's readFQB loop (hash10x.c:202-223): 0 = no chunk semantics (no "chunkSize too small", no
                              // all-A-barcode quirk); set by the session layer for --readFQB
  int64_t optChunkEof = 1;    // the records given end the file (no -N cut): the reference's loop then makes one more pass, which dies if the last
                              // block holds chunkSize records (hash10x.c:205-208); 0 = -N ended the loop
  std::vector<u64> mergePoints; bool replayDone = false;     // records whose barcode change does NOT start a block (chunk replay; consumed by stageA_run)
  int64_t optRowShift = -1;   // testing knob: force the list alignment of the sharded rows[] (-1 = as small as the offsets allow)
  int64_t optReplySort = 0;   // sharded index build, how an owner answers: 0 = by look-up in a table of its distinct hashes (packed entries), 1 = by scattering from the sorted order (the sort carries arrival positions; round 4's form),
                              // tests of the look-up's fall-back: 2 = look up, then answer by scatter all the same, 3 = a table whose displacement limit is 1 (it FAILS), 4 = as if the table did not fit the free memory
  int64_t optOverlap = 1;     // sharded path: 1 = exchanges whose result a later stage needs run on the exchange stream beside the main stream's kernels (the in-range lists beside the good lists, hashDepth[] beside the ClusterHash records), 0 = every exchange on the main stream (round 5)
  int64_t optOwnerCut = 0;    // sharded index build, where the hash owners' value ranges are cut: 0 = at the quantiles of the canonical-hash density 2 (1 - x) (equal shares), 1 = equal value ranges (round 5: owner 0 of 8 holds 23.4 %)
  int64_t optDeltaLists = -1; // in-range barcode lists travel delta-coded between ranks: -1 = where bytes are dear (more than one rank on the host-staged TCP backend; not over xGMI: DESIGN 5), 0 never, 1 always (tests)
  int64_t optRowsFakeBase = 0; // testing knob: list offsets start at this many entries (multiple of 2^rowShift) in front of the real array: 64-bit offsets on small inputs
  // streaming ingest (h10x_ingest_fqb): the record image grows on the device as the chunks arrive
  DevBuf<u32> ingestBuf; u64 ingestRecords = 0, ingestCap = 0; bool ingestAsync = false;   // ingestAsync: chunks came through h10x_ingest_fqb_async (the closing call then checks the count)
  static constexpr int INGEST_SLOTS = 8; hipEvent_t ingestEv[INGEST_SLOTS] = {};   // h10x_ingest_fqb_async: one event per caller's buffer
  int64_t optFaultInject = 0; // test knob: the fork/join region with this number (1 mosh classes, 2 clusHash classes, 3 good-list classes, 4 cluster
                              // classes, 5 sums beside merges) fails once between its fork and its join, then the knob clears itself
  int faultAt(int region) { if (optFaultInject != region) return 0; optFaultInject = 0; return fail("injected fault in fork/join region %d", region); }
  bool timing = false;
  u32 *startFlags = nullptr;   // pinned host words a side-stream kernel's workgroups set when they start (stageC_cluster)
  Timer timers[T_COUNT];
  XchgStat xs[X_COUNT];
  h10x_counters ctr{};

  int fail(const char *fmt, ...) {
    char buf[1024]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    err = buf; return -1;
  }
  // Small device -> host read-backs (counts, totals: the sizes the next launches are made with) land in a pinned, coherent
  // mailbox. readback() only notes what is wanted; syncReadbacks() queues ONE small kernel that stores the values and then a
  // sequence number into the mailbox, and the host spins on that number — a round trip of a few microseconds where a
  // device-to-host copy followed by hipStreamSynchronize cost 25-40 (the step makes eight of them, each with the GPU idle).
  // (prim.hip; a read-back that does not fit the mailbox is copied the ordinary way.)
  // NOTE: the source is read when syncReadbacks() runs, not when readback() is called: it must stay valid and unmodified until then (a read-back
  // that falls back to the ordinary copy captures it at the call instead — callers sync right behind their readback() calls, so both agree).
  static constexpr size_t MAIL_BYTES = 4096, MAIL_ITEMS = 12;
  unsigned char *mail = nullptr; size_t mailUsed = 0; u32 mailSeq = 0; bool mailDirect = false;
  struct PendingRead { void *dst; const void *dev; size_t off, n; };
  std::vector<PendingRead> pendingReads;
  int readback(void *dst, const void *dev, size_t n) {
    if (!mail) {
      if (hipHostMalloc((void **)&mail, MAIL_BYTES + 64, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { mail = nullptr; return fail("hipHostMalloc of the read-back mailbox failed"); }
      memset(mail, 0, MAIL_BYTES + 64);                    // (the sequence number lives behind the values)
    }
    const size_t off = mailUsed, step = (n + 7) & ~(size_t)7;
    if (off + step > MAIL_BYTES || pendingReads.size() >= MAIL_ITEMS) {
      mailDirect = true;
      return hipMemcpyAsync(dst, dev, n, hipMemcpyDeviceToHost, stream) == hipSuccess ? 0 : fail("hipMemcpyAsync (device to host) failed");
    }
    mailUsed += step; pendingReads.push_back(PendingRead{dst, dev, off, n});
    return 0;
  }
  int syncReadbacks();
  int forkStreams(int n) {                               // side streams wait for everything issued on the main stream so far
    if (!evFork) {
      if (hipEventCreateWithFlags(&evFork, hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate failed");
      for (int i = 0; i < 3; ++i) {
        if (hipStreamCreateWithFlags(&aux[i], hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate failed");
        if (hipEventCreateWithFlags(&evJoin[i], hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate failed");
      }
    }
    if (hipEventRecord(evFork, stream) != hipSuccess) return fail("hipEventRecord failed");
    for (int i = 0; i < n; ++i) if (hipStreamWaitEvent(aux[i], evFork, 0) != hipSuccess) return fail("hipStreamWaitEvent failed");
    return 0;
  }
  // The exchange stream: a collective whose result a LATER stage needs (the in-range barcode lists: wanted by --cluster, not by the good lists; hashDepth[] of the other
  // owners: wanted by --hashDepthRange, not by the ClusterHash records) is queued here and runs beside the main stream's kernels. xFork(): the exchange stream waits for
  // what the main stream has queued so far (the send buffers); xJoin(): the main stream waits for the exchange, and the buffers parked in xHold — send buffers that must
  // outlive the function that filled them — go back to the block cache (whose reuse order is the main stream's: not before this point).
  hipStream_t xStream = nullptr; hipEvent_t evXFork = nullptr, evXDone = nullptr; bool xOpen = false;
  DevBuf<u32> xHold[4];
  int xFork() {
    if (!xStream) {
      if (hipStreamCreateWithFlags(&xStream, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate failed");
      if (hipEventCreateWithFlags(&evXFork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&evXDone, hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate failed");
    }
    if (hipEventRecord(evXFork, stream) != hipSuccess || hipStreamWaitEvent(xStream, evXFork, 0) != hipSuccess) return fail("hipStreamWaitEvent failed");
    xOpen = true;
    return 0;
  }
  int xJoin() {
    if (!xOpen) return 0;
    xOpen = false;
    if (hipEventRecord(evXDone, xStream) != hipSuccess || hipStreamWaitEvent(stream, evXDone, 0) != hipSuccess) return fail("hipStreamWaitEvent failed");
    for (auto &b : xHold) b.release();
    return 0;
  }
  int joinStreams(int n) {                               // the main stream waits for the side streams
    for (int i = 0; i < n; ++i) {
      if (hipEventRecord(evJoin[i], aux[i]) != hipSuccess) return fail("hipEventRecord failed");
      if (hipStreamWaitEvent(stream, evJoin[i], 0) != hipSuccess) return fail("hipStreamWaitEvent failed");
    }
    return 0;
  }
  void tstart(Timer &x) {
    if (!timing) return;
    if (!x.a) { (void)hipEventCreate(&x.a); (void)hipEventCreate(&x.b); }
    flush(x);
    (void)hipEventRecord(x.a, stream);
  }
  void tstop(Timer &x) {
    if (!timing) return;
    (void)hipEventRecord(x.b, stream); x.pending = true; ++x.launches;
  }
  void flush(Timer &x) {
    if (x.pending) { float ms = 0; (void)hipEventSynchronize(x.b); (void)hipEventElapsedTime(&ms, x.a, x.b); x.ms += ms; x.pending = false; }
  }
  // Stage timers open right now, innermost last (the exchange wrappers of shard.hip charge their waits to the innermost one). A STACK: when a nested timer closes
  // (cluster > cluster_kernel > cluster_main) the parent is the innermost again; a command that returned early between a tstart and its tstop cannot leave stages open
  // for the context's lifetime — stageReset() runs on every command entry and in h10x_timing_reset.
  static constexpr int STAGE_DEPTH = 8;
  TimerId stageStack[STAGE_DEPTH]; int stageOpen = 0;
  TimerId stageTop = T_COUNT; // the innermost open stage (T_COUNT: none)
  Timer stageWait[T_COUNT];   // per stage: the exchange waits inside its bracket (a stage's compute = its timer less this)
  void stageReset() { stageOpen = 0; stageTop = T_COUNT; }
  void tstart(TimerId t) { if (stageOpen < STAGE_DEPTH) stageStack[stageOpen] = t; ++stageOpen; stageTop = t; tstart(timers[t]); }
  void tstop(TimerId t) {
    if (stageOpen > 0) --stageOpen;
    stageTop = stageOpen > 0 && stageOpen <= STAGE_DEPTH ? stageStack[stageOpen - 1] : T_COUNT;
    tstop(timers[t]);
  }
  void flush(TimerId t) { flush(timers[t]); }
};

// A fork/join region: side streams run kernels on buffers that go back to the per-stream block cache when the function returns (the
// cache orders reuse within ONE stream only). If the function leaves early — an error between fork and join — the guard waits for every
// stream before those destructors run; after a successful join the main stream is ordered behind the side streams and done() disarms it.
struct ForkGuard {
  Ctx *c; bool armed = true;
  explicit ForkGuard(Ctx *c_) : c(c_) {}
  ForkGuard(const ForkGuard &) = delete; ForkGuard &operator=(const ForkGuard &) = delete;
  ~ForkGuard() { if (!armed) return; for (int i = 0; i < 3; ++i) if (c->aux[i]) (void)hipStreamSynchronize(c->aux[i]); (void)hipStreamSynchronize(c->stream); }
  void done() { armed = false; }
};

// An exchange on the exchange stream whose function leaves early: wait for it before the buffers it uses go back to the block cache
struct XGuard {
  Ctx *c;
  explicit XGuard(Ctx *c_) : c(c_) {}
  XGuard(const XGuard &) = delete; XGuard &operator=(const XGuard &) = delete;
  ~XGuard() { if (!c->xOpen) return; (void)hipStreamSynchronize(c->xStream); c->xOpen = false; for (auto &b : c->xHold) b.release(); }
};

#define H10X_HIP(ctx, call)                                                                      \
  do { hipError_t e__ = (call);                                                                  \
       if (e__ != hipSuccess) return (ctx)->fail("HIP error %s at %s:%d: %s", hipGetErrorName(e__), __FILE__, __LINE__, #call); } while (0)

#define H10X_TRY(expr) do { int rc__ = (expr); if (rc__) return rc__; } while (0)

static inline unsigned divUp(u64 a, u64 b) { return (unsigned)((a + b - 1) / b); }
template <typename T> static inline T hmin(T a, T b) { return a < b ? a : b; }
template <typename T> static inline T hmax(T a, T b) { return a > b ? a : b; }

// hashIndexFind(hash, FALSE) (hash10x.c:139-152): start hash & mask, odd stride ((hash >> B) & mask) | 1; 0 = absent
__device__ __forceinline__ u32 probe_find(const u32 *__restrict__ table, const u64 *__restrict__ hashValue, int B, u64 h) {
  const u64 mask = ((u64)1 << B) - 1;
  u64 slot = h & mask; const u64 step = ((h >> B) & mask) | 1;
  u32 ix;
  while ((ix = table[slot]) && hashValue[ix] != h) slot = (slot + step) & mask;
  return ix;
}

// stage entry points (one per translation unit)
int stageA_run(Ctx *c, const u32 *dRecords, u64 nRecords,
               DevBuf<u64> &entHash, DevBuf<u32> &entCode, DevBuf<u32> &entRead, bool hashLast = false, bool emptyIsNoBlock = false);
int stageA_sortRecords(Ctx *c, const u32 *dIn, u64 nRecords, u32 *dOut);
// barcode runs of a record image: starts[r] = first record of run r (r = 0 .. R-1), starts[R] = nRecords; zeroRuns = runs whose barcode word is 0
int stageA_runStarts(Ctx *c, const u32 *dRec, u64 nRec, std::vector<u64> &starts, std::vector<u32> &zeroRuns);
// the reference's chunk loop replayed over the runs of the whole file: 1 = "chunkSize too small"; merges = records (file positions) that
// start a run but not a block (an all-A barcode's run ended exactly at a chunk boundary: hash10x.c:212)
int replayChunks(const std::vector<u64> &starts, const std::vector<u32> &zeroRuns, u64 chunk, std::vector<u64> &merges, bool eofPass);   // stable sort of .fqb records by their first 4 bytes
int stageB_run(Ctx *c, DevBuf<u64> &entHash, DevBuf<u32> &entCode, DevBuf<u32> &entRead);
void warm_prim(hipStream_t), warm_stageA(hipStream_t), warm_stageB(hipStream_t), warm_stageC(hipStream_t), warm_stageD(hipStream_t), warm_stageE(hipStream_t), warm_shard(hipStream_t);   // h10x_warm
int stageB_buildCSR(Ctx *c);                   // rows/rowStart from clusHash + hashDepth (fillHashTable)
int stageB_finishClusHash(Ctx *c, DevBuf<u64> &key);   // key[e] = index << 32 | read16 in block order -> clusHash sorted per block
int stageB_finishClusHashFromReplies(Ctx *c, const u32 *replyIdx, const u32 *replyPos, const u32 *entRead);   // sharded: entry e's index is replyIdx[replyPos[e]]
// workgroup-local sorts of a block's entries (clushash_block_kernel, good_block_kernel): three launch classes, side by side on
// forked streams — 256 lanes x 12 items (blocks up to 3072 entries: 98 % of them at 250 read pairs per barcode), 512 x 12, 1024 x 8
#ifndef H10X_BS_T0
#define H10X_BS_T0 512     // lanes x entries per lane of the workgroup-local sorts of the first two classes (CAP0 = T0 x I0, CAP1 = T1 x I1)
#define H10X_BS_I0 6
#define H10X_BS_T1 1024
#define H10X_BS_I1 6
#endif
constexpr u32 BLOCK_SORT_CAP0 = H10X_BS_T0 * H10X_BS_I0, BLOCK_SORT_CAP1 = H10X_BS_T1 * H10X_BS_I1, BLOCK_SORT_MAX = 8192;
int stageB_buildProbeTable(Ctx *c);            // hashIndex[] from hashValue[1..hashNumber)
// the blocks of the three workgroup-local-sort classes as lists: lists[k * nBlocks ..] holds counts[k] block numbers (entries up to CAP0 — incl. the
// empty blocks and slot 0 —, up to CAP1, up to BLOCK_SORT_MAX); larger blocks are in none. A class's kernel then starts workgroups for ITS blocks
// only: launched over all blocks, the two large classes spent as long dispatching workgroups that had nothing to do as the small class spent working.
int stageB_blockClassLists(Ctx *c, DevBuf<u32> &lists, DevBuf<u32> &counts);
int stageC_depthRange(Ctx *c, int min, int max);
int stageC_cluster(Ctx *c, int codeMin, int codeMax, int threshold);
int stageC_split(Ctx *c);
int stageD_cribGenome(Ctx *c, const u8 *hostCodes, const u64 *seqStart, u32 nSeq, int which, u64 *nPresent, u64 *nAbsent);
int stageD_cribFinish(Ctx *c);
int stageE_histMax(Ctx *c, int which, u64 first, u64 count, u32 *maxValue);
int stageE_histogram(Ctx *c, int which, u64 first, u64 count, u32 bins, u64 *hist);
int stageE_clusterReport(Ctx *c, u32 firstBlock, u32 nBlk, h10x_block_rep *hostB, h10x_cluster_rep *hostC, u64 clusterCap, u64 *nClusters);
int stageE_cribSummary(Ctx *c, u64 *counts12, u32 *hostSeenBase, u32 *hostSeenCluster);
int stageE_cribWords(Ctx *c, u64 first, u64 count, u32 *hostOut);
int shard_allreduceU64(Ctx *c, u64 *v, u32 n, int op);
int shard_gatherBytes(Ctx *c, const void *send, u64 nbytes, void *recv, u64 cap, u64 *counts);
int shard_exchangeRows(Ctx *c);
int shard_split(Ctx *c, const u32 *dSubBefore, u32 totalSubLocal);   // collective part of a sharded --clusterSplit: new segments, layout, owner lists
int shard_refreshLayout(Ctx *c);                // collective: allSegs / global totals from every rank's current segments
int shard_materializeTables(Ctx *c);            // collective: hashValue[] + hashIndex[] on every rank (deferred by the sharded --readFQB)                 // sharded --hashDepthRange: allgather the in-range barcode lists

}  // namespace h10x
