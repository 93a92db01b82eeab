// common.hpp — context, device buffers, error plumbing and timers shared by the stage files.
// gfx950 only; wavefront = 64.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include <map>
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
// Allocations are stream-ordered (hipMallocAsync on the calling context's stream, pool kept warm), so the
// dozens of temporaries of a command cost no driver round trips or implicit device syncs after warm-up.
struct AllocScope {                                       // set by every C-ABI entry point for its thread
  static hipStream_t &stream() { static thread_local hipStream_t s = nullptr; return s; }
  static bool &async() { static thread_local bool a = false; return a; }
  static int poison() { static int p = -1; if (p < 0) { const char *e = getenv("H10X_POISON"); p = e ? atoi(e) : 0; } return p; }   // debug: fill fresh buffers
};
// debug (H10X_ALLOC_CHECK=1): registry of live allocations, reports overlapping ranges handed out by the allocator
struct AllocRegistry {
  static std::map<uintptr_t, size_t> &live() { static std::map<uintptr_t, size_t> m; return m; }
  static bool on() { static int v = -1; if (v < 0) { const char *e = getenv("H10X_ALLOC_CHECK"); v = e ? atoi(e) : 0; } return v != 0; }
  static void add(void *p, size_t bytes) {
    if (!on()) return;
    const uintptr_t a = (uintptr_t)p;
    for (auto &kv : live()) if (a < kv.first + kv.second && kv.first < a + bytes)
      fprintf(stderr, "H10X_ALLOC_CHECK: new [%p,+%zu) overlaps live [%p,+%zu)\n", p, bytes, (void *)kv.first, kv.second);
    live()[a] = bytes;
  }
  static void del(void *p) { if (on()) live().erase((uintptr_t)p); }
};
template <typename T> struct DevBuf {
  T *p = nullptr; size_t n = 0; bool viaPool = false; hipStream_t st = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete; DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) { AllocRegistry::del(p); if (viaPool) (void)hipFreeAsync(p, st); else (void)hipFree(p); }
    p = nullptr; n = 0;
  }
  hipError_t alloc(size_t count) {                       // contents undefined
    release();
    if (!count) count = 1;
    hipError_t e;
    if (AllocScope::async()) { st = AllocScope::stream(); viaPool = true; e = hipMallocAsync((void **)&p, count * sizeof(T), st); }
    else { viaPool = false; e = hipMalloc((void **)&p, count * sizeof(T)); }
    if (e == hipSuccess) { n = count; AllocRegistry::add(p, count * sizeof(T)); } else p = nullptr;
    if (e == hipSuccess && AllocScope::poison()) (void)hipMemsetAsync(p, 0xA5, count * sizeof(T), AllocScope::stream());
    return e;
  }
  void swap(DevBuf &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(viaPool, o.viaPool); std::swap(st, o.st); }
  size_t bytes() const { return n * sizeof(T); }
};

// ---- timers ---------------------------------------------------------------------------------------
enum TimerId { T_RUNS = 0, T_MOSH, T_FALLBACK, T_COMPACT, T_SORT_HASH, T_RANK, T_PROBE, T_CLUSHASH, T_CSR,
               T_GOOD, T_CLUSTER, T_CLUSTER_K, T_SPLIT, T_COUNT };
static const char *const kTimerNames[T_COUNT] = {
  "block_runs", "mosh_extract", "mosh_fallback", "compact_entries", "sort_by_hash", "index_rank",
  "probe_table", "clushash_build", "csr_build", "good_hashes", "cluster", "cluster_kernel", "cluster_split" };

struct Timer { hipEvent_t a = nullptr, b = nullptr; double ms = 0; uint64_t launches = 0; bool pending = false; };

// ---- the context ------------------------------------------------------------------------------------
struct Ctx {
  h10x_params prm{};
  int device = 0;
  hipStream_t stream = nullptr;
  int numCU = 256;
  bool ownStream = false;     // stream created (and destroyed) by the context
  bool poolOK = false;        // stream-ordered allocator usable on this device
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
  u64 nEntries = 0, nRecords = 0;
  bool haveState = false;

  // depth range / good hashes (hash10x.c:525-539, 722-766)
  DevBuf<u8>  within;         // hashNumber
  DevBuf<u16> goodPos;        // H : per block [blockOff[c], blockOff[c]+nGood[c])
  DevBuf<u32> nGood;          // nBlocks
  bool haveRange = false, haveGood = false; int rangeMin = 0, rangeMax = 0;
  u32 maxGoodDepth = 0, maxGood = 0;

  // options / measurement
  int64_t optMaxSlots = 0;    // testing knob: cap stage-A LDS table
  int64_t optClusterLds = 0;  // testing knob: LDS budget of cluster_kernel (forces the HBM-scratch path when small)
  int64_t optStamps = 0;      // diagnostic: per-phase wall-clock stamps in cluster_kernel
  bool timing = false;
  Timer timers[T_COUNT];
  h10x_counters ctr{};

  int fail(const char *fmt, ...) {
    char buf[1024]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    err = buf; return -1;
  }
  void tstart(TimerId t) {
    if (!timing) return;
    Timer &x = timers[t];
    if (!x.a) { (void)hipEventCreate(&x.a); (void)hipEventCreate(&x.b); }
    flush(t);
    (void)hipEventRecord(x.a, stream);
  }
  void tstop(TimerId t) {
    if (!timing) return;
    Timer &x = timers[t];
    (void)hipEventRecord(x.b, stream); x.pending = true; ++x.launches;
  }
  void flush(TimerId t) {
    Timer &x = timers[t];
    if (x.pending) { float ms = 0; (void)hipEventSynchronize(x.b); (void)hipEventElapsedTime(&ms, x.a, x.b); x.ms += ms; x.pending = false; }
  }
};

#define H10X_HIP(ctx, call)                                                                      \
  do { hipError_t e__ = (call);                                                                  \
       if (e__ != hipSuccess) return (ctx)->fail("HIP error %s at %s:%d: %s", hipGetErrorName(e__), __FILE__, __LINE__, #call); } while (0)

#define H10X_TRY(expr) do { int rc__ = (expr); if (rc__) return rc__; } while (0)

static inline unsigned divUp(u64 a, u64 b) { return (unsigned)((a + b - 1) / b); }
template <typename T> static inline T hmin(T a, T b) { return a < b ? a : b; }
template <typename T> static inline T hmax(T a, T b) { return a > b ? a : b; }

// stage entry points (one per translation unit)
int stageA_run(Ctx *c, const u32 *dRecords, u64 nRecords,
               DevBuf<u64> &entHash, DevBuf<u32> &entCode, DevBuf<u32> &entRead);
int stageB_run(Ctx *c, DevBuf<u64> &entHash, DevBuf<u32> &entCode, DevBuf<u32> &entRead);
int stageB_buildCSR(Ctx *c);                   // rows/rowStart from clusHash + hashDepth (fillHashTable)
int stageC_depthRange(Ctx *c, int min, int max);
int stageC_cluster(Ctx *c, int codeMin, int codeMax, int threshold);
int stageC_split(Ctx *c);

}  // namespace h10x
