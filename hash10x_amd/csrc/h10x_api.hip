// h10x_api.hip — the C ABI of include/h10x.h over the stage drivers. No torch, no CPU fallback.
#include "common.hpp"
#include "comm.hpp"
#include "build_id.h"
#include <cstdlib>
#include <ctime>
#include <new>

using namespace h10x;

// every entry point: select the device and route DevBuf allocations to this device's block cache
static int enter(Ctx &c) {
  if (hipSetDevice(c.device) != hipSuccess) return c.fail("hipSetDevice(%d) failed", c.device);
  AllocScope::stream() = c.stream; AllocScope::device() = c.device;
  DevCache::noteStream(c.device, c.stream);
  c.pendingReads.clear(); c.mailUsed = 0; c.mailDirect = false;                    // read-backs a failed call left behind point into its dead frame
  c.stageReset();                                                                   // (and so do the stage timers a failed call left open)
  return 0;                                                                         // (an exchange still on the exchange stream is joined by the command that needs its result: stageC_cluster, shard_*)
}

struct h10x_ctx { Ctx c; };

extern "C" {

int h10x_abi_version(void) { return H10X_ABI_VERSION; }
const char *h10x_build_id(void) { return H10X_BUILD_ID; }

int h10x_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

uint64_t h10x_factor1_from_seed(int32_t seed) {
  // hash10x.c:1101 srandom(r); seqhash.c:29 (random() << 32) | random() | 0x01 — first draw is the high word
  // glibc's srandom() / random() share one process-wide state: rank threads initialising at the same time would draw from
  // each other's sequence. random_r on a private state of the default size (128 bytes = TYPE_3) yields the same numbers.
  struct random_data rd; char state[128]; int32_t a = 0, b = 0;
  memset(&rd, 0, sizeof rd); memset(state, 0, sizeof state);
  initstate_r((unsigned)seed, state, sizeof state, &rd);
  random_r(&rd, &a); random_r(&rd, &b);
  return ((uint64_t)(uint32_t)a << 32) | (uint64_t)(uint32_t)b | 1;
}

static int create_fail(char *err, int errlen, const char *fmt, ...) {
  if (err && errlen > 0) { va_list ap; va_start(ap, fmt); vsnprintf(err, (size_t)errlen, fmt, ap); va_end(ap); }
  return -1;
}

int h10x_create(h10x_ctx **out, const h10x_params *p, int device, void *stream, char *err, int errlen) {
  if (!out || !p) return create_fail(err, errlen, "h10x_create: null argument");
  *out = nullptr;
  // the reference's die() conditions (hash10x.c:1103,1107-1108; seqhash.c:24-25)
  if (p->k <= 0 || p->w <= 0) return create_fail(err, errlen, "k %d, w %d must be > 0; run without args for usage", p->k, p->w);
  if (p->k >= 32) return create_fail(err, errlen, "seqhash k %d must be between 1 and 32\n", p->k);
  if (p->B < 20 || p->B > 30) return create_fail(err, errlen, "hashTableBits %d out of range 20-30", p->B);
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return create_fail(err, errlen, "no HIP device available: libh10x_hip has no CPU fallback");
  if (device < 0 || device >= n) return create_fail(err, errlen, "HIP device %d out of range 0..%d", device, n - 1);
  if (hipSetDevice(device) != hipSuccess) return create_fail(err, errlen, "hipSetDevice(%d) failed", device);
  static hipDeviceProp_t propCache[64]; static bool propOK[64];          // hipGetDeviceProperties costs milliseconds
  if (device >= 64) return create_fail(err, errlen, "HIP device %d out of range", device);
  if (!propOK[device]) {
    if (hipGetDeviceProperties(&propCache[device], device) != hipSuccess) return create_fail(err, errlen, "hipGetDeviceProperties failed");
    propOK[device] = true;
  }
  const hipDeviceProp_t &prop = propCache[device];
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return create_fail(err, errlen, "device %d is %s: this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
  h10x_ctx *h = new (std::nothrow) h10x_ctx();
  if (!h) return create_fail(err, errlen, "out of host memory");
  h->c.prm = *p; h->c.device = device; h->c.stream = (hipStream_t)stream; h->c.numCU = prop.multiProcessorCount;
  if (!stream) {
    // never the legacy NULL stream (no implicit syncs with other streams); the block cache of common.hpp relies
    // on all work of a context being issued on this one stream
    if (hipStreamCreateWithFlags(&h->c.stream, hipStreamNonBlocking) != hipSuccess) { delete h; return create_fail(err, errlen, "hipStreamCreate failed"); }
    h->c.ownStream = true;
  }
  *out = h;
  return 0;
}

void h10x_destroy(h10x_ctx *h) {
  if (!h) return;
  (void)enter(h->c);
  (void)hipStreamSynchronize(h->c.stream);
  for (auto &t : h->c.timers) { if (t.a) (void)hipEventDestroy(t.a); if (t.b) (void)hipEventDestroy(t.b); }
  for (int i = 0; i < 3; ++i) { if (h->c.aux[i]) { (void)hipStreamSynchronize(h->c.aux[i]); (void)hipStreamDestroy(h->c.aux[i]); } if (h->c.evJoin[i]) (void)hipEventDestroy(h->c.evJoin[i]); }
  if (h->c.evFork) (void)hipEventDestroy(h->c.evFork);
  if (h->c.xStream) { (void)hipStreamSynchronize(h->c.xStream); (void)hipStreamDestroy(h->c.xStream); h->c.xStream = nullptr; }
  if (h->c.evXFork) (void)hipEventDestroy(h->c.evXFork);
  if (h->c.evXDone) (void)hipEventDestroy(h->c.evXDone);
  for (auto &e : h->c.ingestEv) if (e) (void)hipEventDestroy(e);
  if (h->c.startFlags) (void)hipHostFree(h->c.startFlags);
  if (h->c.mail) (void)hipHostFree(h->c.mail);
  hipStream_t own = h->c.ownStream ? h->c.stream : nullptr; const hipStream_t used = h->c.stream; const int dev = h->c.device;
  delete h;                                                  // parks every buffer of the context
  (void)hipStreamSynchronize(used);
  DevCache::retireStream(dev, used);                         // synchronised: its blocks may now serve any stream
  if (own) (void)hipStreamDestroy(own);
}

const char *h10x_last_error(const h10x_ctx *h) { return h ? h->c.err.c_str() : "null context"; }

static void reset_state(Ctx &c) {
  c.haveState = false; c.haveRange = false; c.haveGood = false; c.rangeMin = c.rangeMax = 0; c.depthBound = 0xFFFFFFFFu; c.rangeHiMax = 0;
  c.within.release(); c.goodPos.release(); c.nGood.release(); c.goodEntries.release(); c.goodRow.release();
  // the tables of the state being replaced go back to the block cache NOW, not when their successors are swapped in: the second --readFQB of a
  // context then finds every block of the first one parked and allocates nothing (kept until the swap, rows[] and clusHash had no twin in the
  // cache, took larger blocks, and those were allocated afresh behind them: 50-150 ms of hipMalloc inside the sort at 200 M read pairs)
  c.hashIndex.release(); c.hashValue.release(); c.hashDepth.release(); c.rowStart.release(); c.rows.release();
  c.blocks.release(); c.blockOff.release(); c.clusHash.release(); c.clusterRaw.release();
  c.hashNumber = 1; c.nBlocks = 0; c.nEntries = 0; c.nRecords = 0; c.maxBlockHashes = 0xFFFFFFFFu;
  c.sharded = false; c.codeBase = 0; c.nBlocksGlobal = 0; c.oRows.release(); c.oSegStart.release(); c.oIndex.release(); c.oU = 0; c.oM = 0;
  c.oHash.release(); c.tablesPending = false;
  c.segs.n = 1; c.segs.s[0] = BlockSeg{0, 0, 0}; c.allSegs.clear(); c.nEntriesGlobal = c.nRecordsGlobal = 0; c.rowShift = 0; c.ownerListsStale = false;
  memset(&c.ctr, 0, sizeof c.ctr);
  c.haveCrib = false; c.cribGenomes = 0; c.cribChr.release(); c.cribPos.release(); c.cribType.release(); c.cribHist.release();
  for (int g = 0; g < 2; ++g) { c.cribCount[g].release(); c.cribFirst[g].release(); }
}

int h10x_read_fqb_device(h10x_ctx *h, const uint32_t *dRec, uint64_t n) {
  if (!h) return -1;
  Ctx &c = h->c;
  H10X_TRY(enter(c));
  if (n && !dRec) return c.fail("h10x_read_fqb_device: null records");
  reset_state(c);
  static const bool prof = getenv("H10X_HOSTPROF") != nullptr;
  timespec t0, t1, t2; clock_gettime(CLOCK_MONOTONIC, &t0);
  DevBuf<u64> entHash; DevBuf<u32> entCode, entRead;
  c.wantPacked = !c.optNoPack;
  { const int rcA = stageA_run(&c, dRec, n, entHash, entCode, entRead); c.wantPacked = false; if (rcA) return rcA; }   // (the flag must not outlive a failed run)
  c.segs.n = 1; c.segs.s[0] = BlockSeg{0, c.nBlocks, 0};
  clock_gettime(CLOCK_MONOTONIC, &t1);
  H10X_TRY(stageB_run(&c, entHash, entCode, entRead));
  clock_gettime(CLOCK_MONOTONIC, &t2);
  if (prof) fprintf(stderr, "hostprof: stageA %.3f ms, stageB %.3f ms (wall)\n", 1e3 * (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_nsec - t0.tv_nsec),
                    1e3 * (t2.tv_sec - t1.tv_sec) + 1e-6 * (t2.tv_nsec - t1.tv_nsec));
  return 0;
}

int h10x_read_fqb(h10x_ctx *h, const uint32_t *hostRec, uint64_t n) {
  if (!h) return -1;
  Ctx &c = h->c;
  H10X_TRY(enter(c));
  if (n && !hostRec) return c.fail("h10x_read_fqb: null records");
  DevBuf<u32> d;
  H10X_HIP(&c, d.alloc(n * 30));
  if (n) H10X_HIP(&c, hipMemcpyAsync(d.p, hostRec, n * 120, hipMemcpyHostToDevice, c.stream));
  H10X_HIP(&c, hipStreamSynchronize(c.stream));
  return h10x_read_fqb_device(h, d.p, n);
}

__global__ void block_offsets_kernel(const h10x_block *__restrict__ blocks, u32 nBlocks, u32 *__restrict__ nHash) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= nBlocks) nHash[i] = (i >= 1 && i < nBlocks) ? blocks[i].nHash : 0;   // block 0 owns no clusHash (hash10x.c:256)
}

// --readHash takes the file at its word: before the tables are used as indices, look for values that would send a kernel
// out of bounds (a truncated, padded or hand-made file). bad[0..3]: hash index of an entry / table slot beyond hashNumber,
// more than 255 sub-clusters in a block, a label beyond its block's count.
__global__ void validate_entries_kernel(const h10x_clushash *__restrict__ ch, const u64 *__restrict__ blockOff, const h10x_block *__restrict__ blocks, u32 nBlocks,
                                        u32 hashNumber, u32 *__restrict__ bad) {
  for (u32 b = blockIdx.x + 1; b < nBlocks; b += gridDim.x) {
    const u64 e0 = blockOff[b], e1 = blockOff[b + 1]; const u32 nSub = blocks[b].nSubCluster;
    if (threadIdx.x == 0 && nSub > 255) atomicAdd(&bad[2], 1u);
    for (u64 e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
      const h10x_clushash x = ch[e];
      if (x.hash >= hashNumber) atomicAdd(&bad[0], 1u);
      if (x.subCluster > nSub) atomicAdd(&bad[3], 1u);
    }
  }
}
__global__ void validate_table_kernel(const u32 *__restrict__ table, u64 n, u32 hashNumber, u32 *__restrict__ bad) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; const u64 stride = (u64)gridDim.x * blockDim.x;
  u32 mine = 0;
  for (; i < n; i += stride) if (table[i] >= hashNumber) ++mine;
  if (mine) atomicAdd(&bad[1], mine);
}

}  // extern "C"

#include "prim.hpp"

extern "C" {

}  // extern "C"
namespace h10x { int shard_adoptLoadedState(Ctx *c, Comm *cm, u32 codeBase, u32 nBlocksGlobal); }
// the tables of a parsed .hash file (or of one rank's share of it: blocks = slot 0 + the rank's blocks) onto the device, validated
static int upload_state(Ctx &c, const uint32_t *hashIndex, uint32_t hashNumber, const uint64_t *hashValue,
                        const uint32_t *hashDepth, const h10x_block *blocks, uint32_t nBlocks, const h10x_clushash *clusHash) {
  if (!hashIndex || !hashValue || !hashDepth || !blocks || nBlocks < 1 || hashNumber < 1) return c.fail("h10x_load_state: bad argument");
  reset_state(c);
  hipStream_t st = c.stream;
  const u64 tableSize = (u64)1 << c.prm.B;
  u64 H = 0;
  u32 maxHash = 0;
  for (u32 i = 1; i < nBlocks; ++i) { H += blocks[i].nHash; maxHash = blocks[i].nHash > maxHash ? blocks[i].nHash : maxHash; }
  if (H >= (1ULL << 32)) return c.fail("%llu (barcode,hash) entries exceed this build's 2^32 per-GPU limit", (u64)H);
  if (H && !clusHash) return c.fail("h10x_load_state: null clusHash");
  c.hashNumber = hashNumber; c.nBlocks = nBlocks; c.nEntries = H; c.maxBlockHashes = maxHash;
  H10X_HIP(&c, c.hashIndex.alloc(tableSize)); H10X_HIP(&c, c.hashValue.alloc(hashNumber)); H10X_HIP(&c, c.hashDepth.alloc(hashNumber));
  H10X_HIP(&c, c.blocks.alloc(nBlocks)); H10X_HIP(&c, c.clusHash.alloc(H)); H10X_HIP(&c, c.blockOff.alloc((size_t)nBlocks + 1));
  H10X_HIP(&c, hipMemcpyAsync(c.hashIndex.p, hashIndex, tableSize * 4, hipMemcpyHostToDevice, st));
  H10X_HIP(&c, hipMemcpyAsync(c.hashValue.p, hashValue, (size_t)hashNumber * 8, hipMemcpyHostToDevice, st));
  H10X_HIP(&c, hipMemcpyAsync(c.hashDepth.p, hashDepth, (size_t)hashNumber * 4, hipMemcpyHostToDevice, st));
  { u32 mx = 0; for (uint32_t i = 0; i < hashNumber; ++i) mx = hashDepth[i] > mx ? hashDepth[i] : mx; c.depthBound = mx; }
  H10X_HIP(&c, hipMemcpyAsync(c.blocks.p, blocks, (size_t)nBlocks * sizeof(h10x_block), hipMemcpyHostToDevice, st));
  if (H) H10X_HIP(&c, hipMemcpyAsync(c.clusHash.p, clusHash, H * sizeof(h10x_clushash), hipMemcpyHostToDevice, st));
  PrimTemp pt; DevBuf<u32> nh; H10X_HIP(&c, nh.alloc((size_t)nBlocks + 1));
  block_offsets_kernel<<<divUp((u64)nBlocks + 1, 256), 256, 0, st>>>(c.blocks.p, nBlocks, nh.p);
  H10X_TRY(prim_exclusive_scan_u32_u64(&c, pt, nh.p, c.blockOff.p, (size_t)nBlocks + 1));
  {
    DevBuf<u32> bad; H10X_HIP(&c, bad.alloc(4)); H10X_HIP(&c, hipMemsetAsync(bad.p, 0, 16, st));
    if (H) validate_entries_kernel<<<hmin<u32>(nBlocks, 8192), 256, 0, st>>>(c.clusHash.p, c.blockOff.p, c.blocks.p, nBlocks, hashNumber, bad.p);
    validate_table_kernel<<<(unsigned)hmin<u64>(divUp(tableSize, 256), 8192), 256, 0, st>>>(c.hashIndex.p, tableSize, hashNumber, bad.p);
    u32 hb[4];
    H10X_TRY(c.readback(hb, bad.p, 16));
    H10X_TRY(c.syncReadbacks());
    if (hb[0] || hb[1]) return c.fail("corrupt hash file: %u clusHash entries and %u hashIndex slots point beyond hashNumber %u", hb[0], hb[1], hashNumber);
    if (hb[2]) return c.fail("corrupt hash file: %u blocks with more than 255 sub-clusters", hb[2]);
    // (a label above its block's nSubCluster is what re-clustering a clustered file can leave behind in the reference too: tolerated, see DESIGN)
  }
  c.ctr.entries = H; c.ctr.distinct = hashNumber - 1;
  return 0;
}
extern "C" {
int h10x_load_state(h10x_ctx *h, const uint32_t *hashIndex, uint32_t hashNumber, const uint64_t *hashValue,
                    const uint32_t *hashDepth, const h10x_block *blocks, uint32_t nBlocks, const h10x_clushash *clusHash) {
  if (!h) return -1;
  Ctx &c = h->c;
  H10X_TRY(enter(c));
  H10X_TRY(upload_state(c, hashIndex, hashNumber, hashValue, hashDepth, blocks, nBlocks, clusHash));
  c.segs.n = 1; c.segs.s[0] = BlockSeg{0, nBlocks, 0};
  H10X_TRY(stageB_buildCSR(&c));                              // fillHashTable (hash10x.c:1210)
  c.haveState = true;
  return 0;
}
int h10x_shard_load_state(h10x_ctx *h, const uint32_t *hashIndex, uint32_t hashNumber, const uint64_t *hashValue, const uint32_t *hashDepth,
                          const h10x_block *localBlocks, uint32_t nLocalBlocks, const h10x_clushash *localClusHash, uint32_t codeBase, uint32_t nBlocksGlobal) {
  if (!h) return -1;
  Ctx &c = h->c;
  H10X_TRY(enter(c));
  if (!c.comm) return c.fail("h10x_shard_load_state: no communicator attached");
  Comm *cm = c.comm;
  // upload_state validates THIS rank's cut of the file only: every rank learns the verdict of all before the first
  // collective of the adoption, or a rank that fails alone would leave the others waiting for it for ever
  const int rcUp = upload_state(c, hashIndex, hashNumber, hashValue, hashDepth, localBlocks, nLocalBlocks, localClusHash);
  c.comm = cm;
  u64 bad = rcUp ? 1 : 0; std::vector<u64> all((size_t)cm->n);
  const std::string mine = c.err;
  if (cm->allgatherHost(&c, &bad, all.data(), 8)) return -1;
  if (rcUp) { c.err = mine; return rcUp; }
  for (int r = 0; r < cm->n; ++r) if (all[(size_t)r]) return c.fail("rank %d failed to load its part of the hash file", r);
  return shard_adoptLoadedState(&c, cm, codeBase, nBlocksGlobal);
}
/* collective over the attached communicator, with or without a loaded state: do all ranks say ok? (a rank that failed on its own
   — a short read, no memory — says so here instead of leaving the others in the next collective) */
int h10x_shard_agree(h10x_ctx *h, int ok, int *allOk) {
  if (!h || !allOk) return -1;
  Ctx &c = h->c;
  H10X_TRY(enter(c));
  if (!c.comm) { *allOk = ok ? 1 : 0; return 0; }
  u64 bad = ok ? 0 : 1; std::vector<u64> all((size_t)c.comm->n);
  const std::string mine = c.err;
  if (c.comm->allgatherHost(&c, &bad, all.data(), 8)) return -1;
  c.err = mine;
  *allOk = 1;
  for (int r = 0; r < c.comm->n; ++r) if (all[(size_t)r]) *allOk = 0;
  return 0;
}

int h10x_depth_range(h10x_ctx *h, int32_t lo, int32_t hi) {
  if (!h) return -1;
  H10X_TRY(enter(h->c));
  return stageC_depthRange(&h->c, lo, hi);
}

int h10x_cluster(h10x_ctx *h, int32_t codeMin, int32_t codeMax, int32_t threshold) {
  if (!h) return -1;
  H10X_TRY(enter(h->c));
  return stageC_cluster(&h->c, codeMin, codeMax, threshold);
}

int h10x_cluster_split(h10x_ctx *h) {
  if (!h) return -1;
  H10X_TRY(enter(h->c));
  return stageC_split(&h->c);
}

int h10x_get_sizes(h10x_ctx *h, h10x_sizes *out) {
  if (!h || !out) return -1;
  Ctx &c = h->c;
  if (!c.haveState) return c.fail("no hash state loaded: use readFQB or readHash first");
  out->B = c.prm.B; out->hashNumber = c.hashNumber; out->nBlocks = c.nBlocks; out->reserved = 0;
  out->nClusHash = c.nEntries; out->nRecords = c.nRecords;
  return 0;
}

int h10x_export(h10x_ctx *h, uint32_t *hashIndex, uint64_t *hashValue, uint32_t *hashDepth, h10x_block *blocks, h10x_clushash *clusHash) {
  if (!h) return -1;
  Ctx &c = h->c;
  H10X_TRY(enter(c));
  if (!c.haveState) return c.fail("no hash state loaded: use readFQB or readHash first");
  if (c.sharded) return c.fail("this context holds one shard: call h10x_shard_gather and export on rank 0");
  hipStream_t st = c.stream;
  if (hashIndex) H10X_HIP(&c, hipMemcpyAsync(hashIndex, c.hashIndex.p, ((size_t)1 << c.prm.B) * 4, hipMemcpyDeviceToHost, st));
  if (hashValue) H10X_HIP(&c, hipMemcpyAsync(hashValue, c.hashValue.p, (size_t)c.hashNumber * 8, hipMemcpyDeviceToHost, st));
  if (hashDepth) H10X_HIP(&c, hipMemcpyAsync(hashDepth, c.hashDepth.p, (size_t)c.hashNumber * 4, hipMemcpyDeviceToHost, st));
  if (blocks) H10X_HIP(&c, hipMemcpyAsync(blocks, c.blocks.p, (size_t)c.nBlocks * sizeof(h10x_block), hipMemcpyDeviceToHost, st));
  if (clusHash && c.nEntries) H10X_HIP(&c, hipMemcpyAsync(clusHash, c.clusHash.p, c.nEntries * sizeof(h10x_clushash), hipMemcpyDeviceToHost, st));
  H10X_HIP(&c, hipStreamSynchronize(st));
  return 0;
}

int h10x_device_mem_info(int device, uint64_t *freeBytes, uint64_t *totalBytes) {
  if (hipSetDevice(device) != hipSuccess) return -1;
  size_t f = 0, t = 0;
  if (hipMemGetInfo(&f, &t) != hipSuccess) return -1;
  if (freeBytes) *freeBytes = f;
  if (totalBytes) *totalBytes = t;
  return 0;
}
void *h10x_device_malloc(int device, uint64_t bytes) {
  void *p = nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
  return p;
}
int h10x_device_free(int device, void *ptr) { return (hipSetDevice(device) == hipSuccess && hipFree(ptr) == hipSuccess) ? 0 : -1; }
int h10x_device_upload(int device, void *dst, const void *src, uint64_t bytes) {
  // hipMemcpy from pageable memory may return once the bytes are STAGED, before the DMA has landed; the contexts of this
  // library run on non-blocking streams that do not wait for the default stream, so wait here (seen as rare wrong records
  // at the end of an upload when several rank threads shared one device)
  return (hipSetDevice(device) == hipSuccess && hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess && hipDeviceSynchronize() == hipSuccess) ? 0 : -1;
}
int h10x_device_synchronize(int device) { return (hipSetDevice(device) == hipSuccess && hipDeviceSynchronize() == hipSuccess) ? 0 : -1; }

}  // extern "C"
#include "comm.hpp"
namespace h10x { Comm *comm_impl(h10x_comm *c); int shard_readFqb(Ctx *c, Comm *cm, const u32 *dRec, u64 nRec); int shard_gather(Ctx *c); }
extern "C" {

int h10x_shard_attach(h10x_ctx *h, h10x_comm *comm) {
  if (!h || !comm) return -1;
  h->c.comm = comm_impl(comm);
  return 0;
}
int h10x_shard_read_fqb_device(h10x_ctx *h, const uint32_t *dRec, uint64_t n) {
  if (!h) return -1;
  Ctx &c = h->c;
  H10X_TRY(enter(c));
  if (!c.comm) return c.fail("h10x_shard_read_fqb: no communicator attached");
  if (n && !dRec) return c.fail("h10x_shard_read_fqb_device: null records");
  Comm *cm = c.comm;
  c.wantPacked = false;                                      // (the sharded path sends unpacked entries: shard.hip)
  reset_state(c);
  c.comm = cm;
  return shard_readFqb(&c, cm, dRec, n);
}
/* ---- streaming ingest: the reference reads the file in chunks of chunkSize records (hash10x.c:202-223) and so can a caller of this
   library — a chunk is copied to the device and appended to the context's record image, which grows geometrically (or to the size
   announced by h10x_ingest_reserve); the closing call hashes the image exactly as h10x_read_fqb_device does and gives the memory back.
   The host never holds more than one chunk. */
static int ingest_append(Ctx &c, const uint32_t *hostRec, uint64_t n) {
  if (n && !hostRec) return c.fail("h10x_ingest_fqb: null records");
  const u64 need = c.ingestRecords + n;
  if (need > c.ingestCap) {
    u64 cap = c.ingestCap ? c.ingestCap * 2 : (u64)1 << 16;
    if (cap < need) cap = need;
    DevBuf<u32> bigger;
    H10X_HIP(&c, bigger.alloc(cap * 30));
    if (c.ingestRecords) H10X_HIP(&c, hipMemcpyAsync(bigger.p, c.ingestBuf.p, c.ingestRecords * 120, hipMemcpyDeviceToDevice, c.stream));
    H10X_HIP(&c, hipStreamSynchronize(c.stream));            // (the old block goes back to the cache only after the copy)
    c.ingestBuf.swap(bigger); c.ingestCap = cap;
  }
  if (n) {
    H10X_HIP(&c, hipMemcpyAsync(c.ingestBuf.p + c.ingestRecords * 30, hostRec, n * 120, hipMemcpyHostToDevice, c.stream));
    H10X_HIP(&c, hipStreamSynchronize(c.stream));            // the caller may reuse its chunk buffer when this returns
    c.ingestRecords += n;
  }
  return 0;
}
static void ingest_drop(Ctx &c) { (void)hipStreamSynchronize(c.stream); c.ingestBuf.release(); c.ingestRecords = 0; c.ingestCap = 0; c.ingestAsync = false; }
// closing call of an ingest fed by h10x_ingest_fqb_async: the image was announced as a whole (the uploads were queued into it where it stood) — fewer
// records than announced means the caller lost a chunk, and hashing the short image would pass silently
static int ingest_close_check(Ctx &c) {
  if (!c.ingestAsync || c.ingestRecords == c.ingestCap) { c.ingestAsync = false; return 0; }
  const u64 got = c.ingestRecords, want = c.ingestCap;
  ingest_drop(c);
  return c.fail("ingest closed with %llu records where h10x_ingest_reserve announced %llu", got, want);
}
// Device code is loaded on first use, a translation unit at a time (the sorts' alone is 21 MB): called from a thread of its own while the caller reads
// its input, this takes that out of the first command's time. Stream and device of its own choosing; touches no context.
int h10x_warm(int device) {
  if (hipSetDevice(device) != hipSuccess) return -1;
  hipStream_t st = nullptr;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return -1;
  warm_stageA(st); warm_prim(st); warm_stageB(st); warm_stageC(st); warm_stageD(st); warm_stageE(st); warm_shard(st);   // in the order a run needs them
  const hipError_t e = hipStreamSynchronize(st);
  (void)hipStreamDestroy(st);
  return e == hipSuccess && hipGetLastError() == hipSuccess ? 0 : -1;
}
void h10x_alloc_stats(uint64_t *calls, uint64_t *bytes) {
  if (calls) *calls = __atomic_load_n(&DevCache::freshCalls(), __ATOMIC_RELAXED);
  if (bytes) *bytes = __atomic_load_n(&DevCache::freshBytes(), __ATOMIC_RELAXED);
}
void *h10x_pinned_alloc(size_t bytes) { void *p = nullptr; return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr; }
void h10x_pinned_free(void *p) { if (p) (void)hipHostFree(p); }
int h10x_ingest_fqb_async(h10x_ctx *h, const uint32_t *pinnedRec, uint64_t n, int slot) {
  if (!h) return -1;
  Ctx &c = h->c;
  H10X_TRY(enter(c));
  if (slot < 0 || slot >= Ctx::INGEST_SLOTS) return c.fail("h10x_ingest_fqb_async: slot %d outside 0..%d", slot, Ctx::INGEST_SLOTS - 1);
  if (n && !pinnedRec) return c.fail("h10x_ingest_fqb_async: null records");
  if (c.ingestRecords + n > c.ingestCap) return c.fail("h10x_ingest_fqb_async: %llu records beyond the %llu announced by h10x_ingest_reserve (the image cannot move while uploads are queued)",
                                                     (u64)(c.ingestRecords + n), (u64)c.ingestCap);
  // a failure from here on drops the image like the synchronous call does (h10x_ingest_fqb): queued uploads are waited for first, the image must not
  // go back to the block cache under a running DMA
  auto giveUp = [&](const char *what) { ingest_drop(c); return c.fail("h10x_ingest_fqb_async: %s failed", what); };
  if (!c.ingestEv[slot] && hipEventCreateWithFlags(&c.ingestEv[slot], hipEventDisableTiming) != hipSuccess) return giveUp("hipEventCreate");
  if (n && hipMemcpyAsync(c.ingestBuf.p + c.ingestRecords * 30, pinnedRec, n * 120, hipMemcpyHostToDevice, c.stream) != hipSuccess) return giveUp("hipMemcpyAsync (host to device)");
  if (hipEventRecord(c.ingestEv[slot], c.stream) != hipSuccess) return giveUp("hipEventRecord");
  c.ingestRecords += n; c.ingestAsync = true;
  return 0;
}
int h10x_ingest_wait(h10x_ctx *h, int slot) {
  if (!h) return -1;
  Ctx &c = h->c;
  H10X_TRY(enter(c));
  if (slot < 0 || slot >= Ctx::INGEST_SLOTS) return c.fail("h10x_ingest_wait: slot %d outside 0..%d", slot, Ctx::INGEST_SLOTS - 1);
  if (c.ingestEv[slot]) H10X_HIP(&c, hipEventSynchronize(c.ingestEv[slot]));
  return 0;
}
int h10x_ingest_reserve(h10x_ctx *h, uint64_t n_records) {
  if (!h) return -1;
  Ctx &c = h->c;
  H10X_TRY(enter(c));
  if (!n_records) { ingest_drop(c); return 0; }              // 0 = give up the image of an ingest that will not be closed
  if (c.ingestRecords) return c.fail("h10x_ingest_reserve: an ingest is under way");
  ingest_drop(c);
  H10X_HIP(&c, c.ingestBuf.alloc(n_records * 30));
  c.ingestCap = n_records;
  return 0;
}
int h10x_ingest_fqb(h10x_ctx *h, const uint32_t *hostRec, uint64_t n, int final_chunk) {
  if (!h) return -1;
  Ctx &c = h->c;
  H10X_TRY(enter(c));
  if (const int rc = ingest_append(c, hostRec, n)) { ingest_drop(c); return rc; }
  if (!final_chunk) return 0;
  H10X_TRY(ingest_close_check(c));
  DevBuf<u32> image; image.swap(c.ingestBuf);                // released when this returns, whatever happens
  const u64 total = c.ingestRecords; c.ingestRecords = 0; c.ingestCap = 0;
  return h10x_read_fqb_device(h, image.p, total);
}
int h10x_shard_ingest_fqb(h10x_ctx *h, const uint32_t *hostRec, uint64_t n, int final_chunk) {
  if (!h) return -1;
  Ctx &c = h->c;
  H10X_TRY(enter(c));
  if (const int rc = ingest_append(c, hostRec, n)) { ingest_drop(c); return rc; }
  if (!final_chunk) return 0;
  H10X_TRY(ingest_close_check(c));
  DevBuf<u32> image; image.swap(c.ingestBuf);
  const u64 total = c.ingestRecords; c.ingestRecords = 0; c.ingestCap = 0;
  return h10x_shard_read_fqb_device(h, image.p, total);
}
int h10x_shard_read_fqb(h10x_ctx *h, const uint32_t *hostRec, uint64_t n) {
  if (!h) return -1;
  Ctx &c = h->c;
  H10X_TRY(enter(c));
  DevBuf<u32> d;
  H10X_HIP(&c, d.alloc(n * 30));
  if (n) H10X_HIP(&c, hipMemcpyAsync(d.p, hostRec, n * 120, hipMemcpyHostToDevice, c.stream));
  H10X_HIP(&c, hipStreamSynchronize(c.stream));
  return h10x_shard_read_fqb_device(h, d.p, n);
}
int h10x_shard_gather(h10x_ctx *h) { if (!h) return -1; H10X_TRY(enter(h->c)); if (!h->c.comm) return h->c.fail("no communicator attached"); return shard_gather(&h->c); }
int h10x_shard_barrier(h10x_ctx *h) { if (!h) return -1; H10X_TRY(enter(h->c)); if (!h->c.comm) return 0; return h->c.comm->barrier(&h->c); }
int h10x_shard_allreduce_max(h10x_ctx *h, double *v) { if (!h || !v) return -1; H10X_TRY(enter(h->c)); if (!h->c.comm) return 0; return h->c.comm->allreduceMaxHost(&h->c, v); }

int h10x_shard_allreduce_sum_u64(h10x_ctx *h, uint64_t *v, uint32_t n) { if (!h || !v) return -1; H10X_TRY(enter(h->c)); return shard_allreduceU64(&h->c, (u64 *)v, n, 0); }
int h10x_shard_allreduce_max_u64(h10x_ctx *h, uint64_t *v, uint32_t n) { if (!h || !v) return -1; H10X_TRY(enter(h->c)); return shard_allreduceU64(&h->c, (u64 *)v, n, 1); }
int h10x_shard_gather_bytes(h10x_ctx *h, const void *send, uint64_t nbytes, void *recv, uint64_t cap, uint64_t *counts) {
  if (!h) return -1; H10X_TRY(enter(h->c)); return shard_gatherBytes(&h->c, send, nbytes, recv, cap, (u64 *)counts);
}
int h10x_report_max(h10x_ctx *h, int which, uint64_t first, uint64_t count, uint32_t *maxValue) {
  if (!h || !maxValue) return -1; H10X_TRY(enter(h->c)); return stageE_histMax(&h->c, which, first, count, maxValue);
}
int h10x_report_histogram(h10x_ctx *h, int which, uint64_t first, uint64_t count, uint32_t bins, uint64_t *hist) {
  if (!h || !hist || !bins) return -1; H10X_TRY(enter(h->c)); return stageE_histogram(&h->c, which, first, count, bins, (u64 *)hist);
}
int h10x_cluster_report(h10x_ctx *h, uint32_t firstBlock, uint32_t nBlocks, h10x_block_rep *blocks, h10x_cluster_rep *clusters, uint64_t clusterCap, uint64_t *nClusters) {
  if (!h || (nBlocks && !blocks)) return -1; H10X_TRY(enter(h->c)); return stageE_clusterReport(&h->c, firstBlock, nBlocks, blocks, clusters, clusterCap, (u64 *)nClusters);
}
int h10x_crib_summary(h10x_ctx *h, uint64_t counts[12], uint32_t *seenBase, uint32_t *seenCluster) {
  if (!h || !counts || !seenBase || !seenCluster) return -1; H10X_TRY(enter(h->c)); return stageE_cribSummary(&h->c, (u64 *)counts, seenBase, seenCluster);
}

int h10x_crib_words(h10x_ctx *h, uint64_t first, uint64_t count, uint32_t *words) {
  if (!h || (count && !words)) return -1; H10X_TRY(enter(h->c)); return stageE_cribWords(&h->c, first, count, words);
}

int h10x_timing_enable(h10x_ctx *h, int on) { if (!h) return -1; h->c.timing = on != 0; return 0; }
int h10x_timing_count(const h10x_ctx *) { return T_COUNT; }
const char *h10x_timing_name(const h10x_ctx *, int i) { return (i >= 0 && i < T_COUNT) ? kTimerNames[i] : ""; }
int h10x_timing_get(h10x_ctx *h, int i, double *ms, uint64_t *launches) {
  if (!h || i < 0 || i >= T_COUNT) return -1;
  h->c.flush((TimerId)i);
  if (ms) *ms = h->c.timers[i].ms;
  if (launches) *launches = h->c.timers[i].launches;
  return 0;
}
int h10x_timing_reset(h10x_ctx *h) {
  if (!h) return -1;
  for (int i = 0; i < T_COUNT; ++i) { h->c.flush((TimerId)i); h->c.timers[i].ms = 0; h->c.timers[i].launches = 0; h->c.flush(h->c.stageWait[i]); h->c.stageWait[i].ms = 0; h->c.stageWait[i].launches = 0; }
  for (int i = 0; i < X_COUNT; ++i) { XchgStat &x = h->c.xs[i]; h->c.flush(x.t); h->c.flush(x.tIn); x.t.ms = x.tIn.ms = 0; x.t.launches = x.tIn.launches = 0; x.calls = x.bytesOut = x.bytesIn = x.maxPeerOut = 0; }
  h->c.stageReset();
  return 0;
}
int h10x_timing_wait_get(h10x_ctx *h, int i, double *ms) {
  if (!h || i < 0 || i >= T_COUNT) return -1;
  h->c.flush(h->c.stageWait[i]);
  if (ms) *ms = h->c.stageWait[i].ms;
  return 0;
}
int h10x_exchange_count(void) { return X_COUNT; }
const char *h10x_exchange_name(int i) { return (i >= 0 && i < X_COUNT) ? kXchgNames[i] : ""; }
int h10x_exchange_get(h10x_ctx *h, int i, uint64_t *calls, uint64_t *bytesOut, uint64_t *bytesIn, uint64_t *maxPeerOut, double *ms, double *msInStages) {
  if (!h || i < 0 || i >= X_COUNT) return -1;
  XchgStat &x = h->c.xs[i]; h->c.flush(x.t); h->c.flush(x.tIn);
  if (msInStages) *msInStages = x.tIn.ms;
  if (calls) *calls = x.calls;
  if (bytesOut) *bytesOut = x.bytesOut;
  if (bytesIn) *bytesIn = x.bytesIn;
  if (maxPeerOut) *maxPeerOut = x.maxPeerOut;
  if (ms) *ms = x.t.ms + x.tIn.ms;
  return 0;
}
int h10x_exchange_beside(h10x_ctx *h, int i) { return (!h || i < 0 || i >= X_COUNT) ? -1 : h->c.xs[i].beside; }
int h10x_sort_fqb_device(h10x_ctx *h, const uint32_t *dIn, uint64_t n, uint32_t *dOut) {
  if (!h) return -1;
  Ctx &c = h->c; H10X_TRY(enter(c));
  if (n && (!dIn || !dOut)) return c.fail("h10x_sort_fqb_device: null buffer");
  return stageA_sortRecords(&c, dIn, n, dOut);
}
int h10x_sort_fqb(h10x_ctx *h, const uint32_t *in, uint64_t n, uint32_t *out) {
  if (!h) return -1;
  Ctx &c = h->c; H10X_TRY(enter(c));
  if (n && (!in || !out)) return c.fail("h10x_sort_fqb: null buffer");
  DevBuf<u32> dIn, dOut;
  H10X_HIP(&c, dIn.alloc(n * 30)); H10X_HIP(&c, dOut.alloc(n * 30));
  if (n) H10X_HIP(&c, hipMemcpyAsync(dIn.p, in, n * 120, hipMemcpyHostToDevice, c.stream));
  H10X_TRY(stageA_sortRecords(&c, dIn.p, n, dOut.p));
  if (n) H10X_HIP(&c, hipMemcpyAsync(out, dOut.p, n * 120, hipMemcpyDeviceToHost, c.stream));
  H10X_HIP(&c, hipStreamSynchronize(c.stream));
  return 0;
}

int h10x_crib_genome(h10x_ctx *h, const uint8_t *codes, const uint64_t *seqStart, uint32_t nSeq, int which, uint64_t *nPresent, uint64_t *nAbsent) {
  if (!h) return -1;
  Ctx &c = h->c; H10X_TRY(enter(c));
  if (!seqStart || (seqStart[nSeq] && !codes)) return c.fail("h10x_crib_genome: null argument");
  return stageD_cribGenome(&c, codes, (const u64 *)seqStart, nSeq, which, (u64 *)nPresent, (u64 *)nAbsent);
}
int h10x_crib_finish(h10x_ctx *h) { if (!h) return -1; Ctx &c = h->c; H10X_TRY(enter(c)); return stageD_cribFinish(&c); }
int h10x_crib_sizes(h10x_ctx *h, uint32_t *histDim, uint32_t arrayMax[4]) {
  if (!h) return -1;
  Ctx &c = h->c; H10X_TRY(enter(c));
  if (!c.haveCrib) return c.fail("no crib: use cribBuild first");
  if (histDim) *histDim = c.cribHistDim;
  if (arrayMax) { H10X_HIP(&c, hipMemcpyAsync(arrayMax, c.cribHist.p + (size_t)4 * c.cribHistDim, 16, hipMemcpyDeviceToHost, c.stream)); H10X_HIP(&c, hipStreamSynchronize(c.stream)); }
  return 0;
}
int h10x_crib_export(h10x_ctx *h, int16_t *chr, uint16_t *pos, uint8_t *type, uint32_t *hist) {
  if (!h) return -1;
  Ctx &c = h->c; H10X_TRY(enter(c));
  if (!c.haveCrib) return c.fail("no crib: use cribBuild first");
  const size_t U1 = c.hashNumber;
  if (chr) H10X_HIP(&c, hipMemcpyAsync(chr, c.cribChr.p, U1 * 2, hipMemcpyDeviceToHost, c.stream));
  if (pos) H10X_HIP(&c, hipMemcpyAsync(pos, c.cribPos.p, U1 * 2, hipMemcpyDeviceToHost, c.stream));
  if (type) H10X_HIP(&c, hipMemcpyAsync(type, c.cribType.p, U1, hipMemcpyDeviceToHost, c.stream));
  if (hist) H10X_HIP(&c, hipMemcpyAsync(hist, c.cribHist.p, (size_t)4 * c.cribHistDim * 4, hipMemcpyDeviceToHost, c.stream));
  H10X_HIP(&c, hipStreamSynchronize(c.stream));
  return 0;
}
int h10x_export_ngood(h10x_ctx *h, uint32_t *nGood) {
  if (!h || !nGood) return -1;
  Ctx &c = h->c; H10X_TRY(enter(c));
  if (!c.haveGood) return c.fail("!! you must set hashDepthRange before clusterReport");
  H10X_HIP(&c, hipMemcpyAsync(nGood, c.nGood.p, (size_t)c.nBlocks * 4, hipMemcpyDeviceToHost, c.stream));
  H10X_HIP(&c, hipStreamSynchronize(c.stream));
  return 0;
}

int h10x_get_counters(h10x_ctx *h, h10x_counters *out) { if (!h || !out) return -1; *out = h->c.ctr; return 0; }

// ---- sharded contexts: where the blocks are, and copy-out by slices (no gather) ----
int h10x_shard_info(h10x_ctx *h, h10x_shard_info_t *out) {
  if (!h || !out) return -1;
  Ctx &c = h->c;
  if (!c.haveState) return c.fail("no hash state loaded: use readFQB or readHash first");
  memset(out, 0, sizeof *out);
  out->rank = c.sharded ? c.comm->rank : 0; out->nranks = c.sharded ? c.comm->n : 1;
  out->hashNumber = c.hashNumber; out->B = c.prm.B;
  if (c.sharded) { out->nBlocksGlobal = c.nBlocksGlobal; out->nEntriesGlobal = c.nEntriesGlobal; out->nRecordsGlobal = c.nRecordsGlobal; out->nSegs = (uint32_t)c.allSegs.size(); }
  else { out->nBlocksGlobal = c.nBlocks; out->nEntriesGlobal = c.nEntries; out->nRecordsGlobal = c.nRecords; out->nSegs = c.nBlocks > 1 ? 1 : 0; }
  return 0;
}
int h10x_shard_segments(h10x_ctx *h, h10x_shard_seg *out, uint32_t cap) {
  if (!h || !out) return -1;
  Ctx &c = h->c; H10X_TRY(enter(c));
  if (!c.haveState) return c.fail("no hash state loaded: use readFQB or readHash first");
  if (!c.sharded) {
    if (c.nBlocks <= 1) return 0;
    if (cap < 1) return c.fail("h10x_shard_segments: room for %u segments, 1 needed", cap);
    out[0] = h10x_shard_seg{0, 1, c.nBlocks - 1, 1, c.nEntries, 0, 0};
    return 0;
  }
  if (cap < c.allSegs.size()) return c.fail("h10x_shard_segments: room for %u segments, %zu needed", cap, c.allSegs.size());
  u64 g = 0;
  for (size_t i = 0; i < c.allSegs.size(); ++i) {
    const ShardSegInfo &s = c.allSegs[i];
    u64 local = 0;                                           // where the segment's entries start in this rank's clusHash (own segments only)
    if ((int)s.rank == c.comm->rank) H10X_HIP(&c, hipMemcpy(&local, c.blockOff.p + s.localStart, 8, hipMemcpyDeviceToHost));
    out[i] = h10x_shard_seg{s.rank, s.localStart, s.count, s.globalBase, s.entries, local, g};
    g += s.entries;
  }
  return 0;
}
int h10x_shard_prepare_export(h10x_ctx *h) {
  if (!h) return -1;
  Ctx &c = h->c; H10X_TRY(enter(c));
  if (!c.haveState) return c.fail("no hash state loaded: use readFQB or readHash first");
  if (!c.sharded) return 0;
  return shard_materializeTables(&c);
}
int h10x_export_slice(h10x_ctx *h, int table, uint64_t first, uint64_t count, void *dst) {
  if (!h) return -1;
  Ctx &c = h->c; H10X_TRY(enter(c));
  if (!c.haveState) return c.fail("no hash state loaded: use readFQB or readHash first");
  if (!count) return 0;
  if (!dst) return c.fail("h10x_export_slice: null destination");
  const void *src = nullptr; size_t eb = 0; u64 limit = 0;
  switch (table) {
    case H10X_TABLE_HASHINDEX: src = c.hashIndex.p; eb = 4; limit = (u64)1 << c.prm.B; break;
    case H10X_TABLE_HASHVALUE: src = c.hashValue.p; eb = 8; limit = c.hashNumber; break;
    case H10X_TABLE_HASHDEPTH: src = c.hashDepth.p; eb = 4; limit = c.hashNumber; break;
    case H10X_TABLE_BLOCKS:    src = c.blocks.p; eb = sizeof(h10x_block); limit = c.nBlocks; break;
    case H10X_TABLE_CLUSHASH:  src = c.clusHash.p; eb = sizeof(h10x_clushash); limit = c.nEntries; break;
    case H10X_TABLE_CLUSTER_RAW: if (!c.clusterRaw.p || c.clusterRaw.n != 2 * (size_t)c.nBlocks) return c.fail("no --cluster has run on these blocks"); src = c.clusterRaw.p; eb = 8; limit = c.nBlocks; break;
    case H10X_TABLE_NGOOD:     if (!c.haveGood) return c.fail("!! you must set hashDepthRange before clusterReport"); src = c.nGood.p; eb = 4; limit = c.nBlocks; break;
    default: return c.fail("h10x_export_slice: unknown table %d", table);
  }
  if ((table == H10X_TABLE_HASHINDEX || table == H10X_TABLE_HASHVALUE) && c.tablesPending) return c.fail("h10x_export_slice: call h10x_shard_prepare_export first (collective)");
  if (!src || first > limit || count > limit - first) return c.fail("h10x_export_slice: table %d, range %llu + %llu outside %llu", table, (u64)first, (u64)count, (u64)limit);
  H10X_HIP(&c, hipMemcpyAsync(dst, (const char *)src + first * eb, count * eb, hipMemcpyDeviceToHost, c.stream));
  H10X_HIP(&c, hipStreamSynchronize(c.stream));
  return 0;
}

int h10x_set_option(h10x_ctx *h, const char *name, int64_t value) {
  if (!h || !name) return -1;
  if (!strcmp(name, "stage_a_max_slots")) {
    if (value && (value < 256 || value > 16384 || (value & (value - 1)))) return h->c.fail("stage_a_max_slots must be a power of two in 256..16384 (or 0)");
    h->c.optMaxSlots = value; return 0;
  }
  if (!strcmp(name, "cluster_lds_budget")) { h->c.optClusterLds = value; return 0; }
  if (!strcmp(name, "cluster_first_global")) { h->c.optFirstGlobal = value; return 0; }
  if (!strcmp(name, "cluster_first_cap")) { h->c.optFirstCap = value; return 0; }
  if (!strcmp(name, "cluster_threads0")) { h->c.optClusterThreads0 = value; return 0; }
  if (!strcmp(name, "cluster_budget0")) { h->c.optClusterBudget0 = value; return 0; }
  if (!strcmp(name, "cluster_big_ranks")) { h->c.optBigRanks = value; return 0; }
  if (!strcmp(name, "cluster_narrow_first")) { h->c.optNarrowFirst = value; return 0; }
  if (!strcmp(name, "cluster_tr_packed")) { h->c.optTrPacked = value; return 0; }
  if (!strcmp(name, "cluster_tr_class_t")) { h->c.optTrClassT = value; return 0; }
  if (!strcmp(name, "cluster_tr_est_div")) { h->c.optTrEstDiv = value; return 0; }
  if (!strcmp(name, "index_no_pack")) { h->c.optNoPack = value; return 0; }
  if (!strcmp(name, "index_probed_table")) { if (value < 0 || value > 3) return h->c.fail("index_probed_table must be 0..3"); h->c.optProbedTable = value; return 0; }
  if (!strcmp(name, "index_priv_table")) { if (value < 0 || value > 3) return h->c.fail("index_priv_table must be 0..3"); h->c.optPrivTable = value; return 0; }
  if (!strcmp(name, "cluster_stamps")) { h->c.optStamps = value; return 0; }
  if (!strcmp(name, "chunk_size")) { if (value < 0) return h->c.fail("chunk_size must be >= 0"); h->c.optChunk = value; return 0; }
  if (!strcmp(name, "chunk_eof_pass")) { h->c.optChunkEof = value ? 1 : 0; return 0; }
  if (!strcmp(name, "fault_inject")) { h->c.optFaultInject = value; return 0; }
  if (!strcmp(name, "shard_row_shift")) { if (value < -1 || value > 8) return h->c.fail("shard_row_shift must be -1..8"); h->c.optRowShift = value; return 0; }
  if (!strcmp(name, "shard_reply_sort")) { if (value < 0 || value > 4) return h->c.fail("shard_reply_sort must be 0..4"); h->c.optReplySort = value; return 0; }
  if (!strcmp(name, "shard_overlap")) { h->c.optOverlap = value ? 1 : 0; return 0; }
  if (!strcmp(name, "shard_owner_cut")) { if (value < 0 || value > 1) return h->c.fail("shard_owner_cut must be 0 or 1"); h->c.optOwnerCut = value; return 0; }
  if (!strcmp(name, "shard_delta_lists")) { if (value < -1 || value > 1) return h->c.fail("shard_delta_lists must be -1, 0 or 1"); h->c.optDeltaLists = value; return 0; }
  if (!strcmp(name, "shard_rows_fake_base")) { if (value < 0) return h->c.fail("shard_rows_fake_base must be >= 0"); h->c.optRowsFakeBase = value; return 0; }
  return h->c.fail("unknown option %s", name);
}

}  // extern "C"
