/* h10x_host.c — see h10x_host.h. Plain C; links libh10x_hip.so. No compute here: files, Array bookkeeping, text. */
#define _GNU_SOURCE
#include "h10x_host.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <limits.h>
#include <time.h>
#include <fcntl.h>
#include <pthread.h>
#include <unistd.h>
#include <sys/types.h>
#include <sys/stat.h>
#include <sys/mman.h>

#define ARRAY_MAGIC 8918274                            /* array.h:56 */
typedef struct { int32_t magic, pad0; uint64_t base; int32_t dim, size, max, pad1; } array_hdr;   /* array.h:41-50 */

enum { N_KNOBS = 21 };
struct h10x_session {
  int k, w, r, B, N, chunk, ct, device;                /* params (hash10x.c:25-33) */
  int timing;                                          /* measurement hook: enable hipEvent timers on every new context */
  int knob[N_KNOBS];                                        /* test / tuning knobs forwarded to h10x_set_option (names in knobName[]) */
  int16_t *cribChr; uint16_t *cribPos; uint8_t *cribType; uint32_t *cribDepth; uint32_t cribN;   /* per-hash host copies for the report texts */
  h10x_ctx *ctx;
  int ctxK, ctxW, ctxR, ctxB, ctxDev;                   /* parameters the live context was created with */
  /* Array bookkeeping of the reference for the two arrays that are dumped raw into .hash */
  int depthDim, depthMax, blocksDim, blocksMax;
  uint32_t *depthTail; int depthTailFrom;              /* entries [hashNumber, dim) as read from a file (normally zero) */
  char err[1024];
};
static const char *const knobName[N_KNOBS] = {"cluster_stamps", "cluster_lds_budget", "cluster_first_global", "cluster_first_cap",
                                              "cluster_big_ranks", "cluster_threads0", "cluster_budget0", "shard_row_shift", "shard_rows_fake_base", "stage_a_max_slots", "cluster_narrow_first", "index_no_pack", "shard_delta_lists", "index_priv_table", "cluster_tr_packed", "cluster_tr_est_div", "cluster_tr_class_t", "shard_reply_sort", "shard_owner_cut", "shard_overlap", "index_probed_table"};

static int fail(h10x_session *s, const char *fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(s->err, sizeof s->err, fmt, ap); va_end(ap);
  return -1;
}
static int fail_ctx(h10x_session *s) { snprintf(s->err, sizeof s->err, "%s", h10x_last_error(s->ctx)); return -1; }

h10x_session *h10x_session_new(void) {
  if (h10x_abi_version() != H10X_ABI_VERSION) {          /* libh10x_hip.so built from another include/h10x.h than this file: structs and tables would not line up */
    fprintf(stderr, "h10x: libh10x_hip.so speaks ABI %d, this host layer was built for %d: rebuild both (make -C hash10x_amd/csrc && make -C hash10x_amd/host)\n", h10x_abi_version(), H10X_ABI_VERSION);
    return 0;
  }
  h10x_session *s = (h10x_session *)calloc(1, sizeof *s);
  if (!s) return 0;
  s->k = 21; s->w = 31; s->r = 17; s->B = 28; s->N = 0; s->chunk = 100000; s->ct = 5; s->device = 0;
  for (int i = 0; i < N_KNOBS; ++i)                    /* by name: the table above may be reordered */
    if (!strcmp(knobName[i], "shard_delta_lists") || !strcmp(knobName[i], "shard_row_shift") || !strcmp(knobName[i], "cluster_tr_packed") || !strcmp(knobName[i], "cluster_tr_class_t")) s->knob[i] = -1;   /* delta-coded list exchange when there is more than one rank; list alignment as small as the offsets allow */
    else if (!strcmp(knobName[i], "shard_overlap")) s->knob[i] = 1;   /* exchanges beside compute */
  s->blocksDim = 1200;                                 /* arrayCreate(1200, ClusterBlock), hash10x.c:1151 */
  return s;
}
static void crib_free(h10x_session *s) { free(s->cribChr); free(s->cribPos); free(s->cribType); free(s->cribDepth); s->cribChr = 0; s->cribPos = 0; s->cribType = 0; s->cribDepth = 0; s->cribN = 0; }
void h10x_session_free(h10x_session *s) {
  if (!s) return;
  if (s->ctx) h10x_destroy(s->ctx);
  crib_free(s); free(s->depthTail); free(s);
}
const char *h10x_session_error(const h10x_session *s) { return s->err; }
h10x_ctx *h10x_session_ctx(h10x_session *s) { return s->ctx; }

static int *param_slot(h10x_session *s, const char *n) {
  static const char *const names[] = {"k", "w", "r", "B", "N", "c", "ct", "device", "timing", 0};
  int *const slots[] = {&s->k, &s->w, &s->r, &s->B, &s->N, &s->chunk, &s->ct, &s->device, &s->timing};
  for (int i = 0; names[i]; ++i) if (!strcmp(n, names[i])) return slots[i];
  for (int i = 0; i < N_KNOBS; ++i) if (!strcmp(n, knobName[i])) return &s->knob[i];
  return 0;
}
int h10x_session_set(h10x_session *s, const char *name, int value) {
  int *p = param_slot(s, name); if (!p) return fail(s, "unknown parameter %s", name);
  *p = value;
  if (s->ctx && p >= s->knob && p < s->knob + N_KNOBS && h10x_set_option(s->ctx, name, value)) return fail_ctx(s);   /* a knob reaches a live context at once */
  return 0;
}
int h10x_session_get(const h10x_session *s, const char *name) {
  int *p = param_slot((h10x_session *)s, name); return p ? *p : 0;
}

/* arrayExtend (array.c:144-170) entered with n == dim when an Array is touched at ascending indices */
int h10x_host_array_dim(int dim, int size, int64_t last) {
  while (last >= dim) dim = dim * size < (1 << 23) ? dim * 2 : dim + 1024 + (1 << 23) / size;
  return dim;
}

/* readFQB's chunk loop (hash10x.c:202-223) on a record image in host memory: how many records it consumes, or -1 with
   "chunkSize too small". (The library replays the same loop from the run starts alone when the "chunk_size" option is set;
   this host form serves callers that hold the records and want the verdict before any upload.) */
int64_t h10x_host_check_chunks(const uint32_t *rec, uint64_t total, int N, int chunk, char *err, int errlen) {
  const uint64_t limit = N > 0 && (uint64_t)N < total ? (uint64_t)N : total;
  uint64_t pos = 0, runLen = 0; uint32_t cur = 0;
  while (pos < limit) {
    if (runLen >= (uint64_t)chunk) { if (err) snprintf(err, (size_t)errlen, "chunkSize too small"); return -1; }   /* hash10x.c:206 */
    uint64_t end = pos + ((uint64_t)chunk - runLen); if (end > limit) end = limit;
    if (!cur) cur = rec[30 * pos];                                                      /* hash10x.c:212 */
    for (; pos < end; ++pos) { const uint32_t b = rec[30 * pos]; if (b == cur) ++runLen; else { cur = b; runLen = 1; } }
  }
  /* unless -N ended the loop (`while (!N || nReads < N)`), the reference comes round once more and tests the chunk size before
     the fread that finds the end of the file (hash10x.c:205-208) */
  if (!(N > 0 && (uint64_t)N <= total) && runLen >= (uint64_t)chunk) { if (err) snprintf(err, (size_t)errlen, "chunkSize too small"); return -1; }
  return (int64_t)limit;
}

int h10x_host_partition(const uint32_t *rec, uint64_t n, int nParts, uint64_t *cut) {
  if (nParts < 1 || !cut) return -1;
  cut[0] = 0; cut[nParts] = n;
  for (int g = 1; g < nParts; ++g) {
    uint64_t p = (uint64_t)(((__uint128_t)n * (unsigned)g) / (unsigned)nParts);
    if (p < cut[g - 1]) p = cut[g - 1];
    while (p > 0 && p < n && rec[30 * p] == rec[30 * (p - 1)]) ++p;     /* move forward to the next run boundary */
    cut[g] = p;
  }
  return 0;
}

/* the same cuts for a file, reading only the barcode words around each cut */
int h10x_host_partition_file(const char *path, uint64_t n, int nParts, uint64_t *cut, char *err, int errlen) {
  if (nParts < 1 || !cut) return -1;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) { if (err) snprintf(err, (size_t)errlen, "failed to open fqb file %s", path); return -1; }
  cut[0] = 0; cut[nParts] = n;
  for (int g = 1; g < nParts; ++g) {
    uint64_t p = (uint64_t)(((__uint128_t)n * (unsigned)g) / (unsigned)nParts);
    if (p < cut[g - 1]) p = cut[g - 1];
    uint32_t prev = 0, here = 0;
    if (p > 0 && p < n && pread(fd, &prev, 4, (off_t)((p - 1) * 120)) != 4) { close(fd); if (err) snprintf(err, (size_t)errlen, "file read problem"); return -1; }
    while (p > 0 && p < n) {
      if (pread(fd, &here, 4, (off_t)(p * 120)) != 4) { close(fd); if (err) snprintf(err, (size_t)errlen, "file read problem"); return -1; }
      if (here != prev) break;
      ++p;
    }
    cut[g] = p;
  }
  close(fd);
  return 0;
}

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return 1e3 * t.tv_sec + 1e-6 * t.tv_nsec; }
static int hostprof(void) { static int v = -1; if (v < 0) v = getenv("H10X_HOSTPROF") != 0; return v; }

/* measurement / test knobs follow the context */
static int apply_options(h10x_session *s) {
  h10x_timing_enable(s->ctx, s->timing);
  for (int i = 0; i < N_KNOBS; ++i) if (h10x_set_option(s->ctx, knobName[i], s->knob[i])) return fail_ctx(s);
  if (h10x_set_option(s->ctx, "chunk_size", 0)) return fail_ctx(s);
  return 0;
}

/* initialise() (hash10x.c:1099-1118): a fresh context with the currently latched parameters */
static int session_init(h10x_session *s) {
  double t0 = now_ms();
  /* same hasher/table/device as the live context: keep it (its stream and warm memory pool); the next
     read/load call resets every table, which is all initialise() does to the state */
  const int reuse = s->ctx && s->ctxK == s->k && s->ctxW == s->w && s->ctxR == s->r && s->ctxB == s->B && s->ctxDev == s->device && s->k > 0 && s->w > 0;
  if (!reuse) {
    if (s->ctx) { h10x_destroy(s->ctx); s->ctx = 0; }
    double t1 = now_ms();
    h10x_params p; memset(&p, 0, sizeof p);
    p.k = s->k; p.w = s->w; p.B = s->B;
    if (s->k > 0 && s->w > 0) p.factor1 = h10x_factor1_from_seed(s->r);
    if (h10x_create(&s->ctx, &p, s->device, 0, s->err, (int)sizeof s->err)) return -1;
    s->ctxK = s->k; s->ctxW = s->w; s->ctxR = s->r; s->ctxB = s->B; s->ctxDev = s->device;
    if (hostprof()) fprintf(stderr, "hostprof: destroy %.3f ms, create %.3f ms\n", t1 - t0, now_ms() - t1);
  } else h10x_timing_reset(s->ctx);                    /* timers are per initialise() */
  s->depthDim = 1 << 20; s->depthMax = 0;             /* arrayCreate(1 << 20, U32), hash10x.c:1114 */
  free(s->depthTail); s->depthTail = 0;
  crib_free(s);
  return apply_options(s);
}

/* Array dims as --readFQB of the WHOLE file leaves them (global sizes on a sharded context) */
static int after_readFQB(h10x_session *s) {
  h10x_shard_info_t z; if (h10x_shard_info(s->ctx, &z)) return fail_ctx(s);
  /* hashDepth: touched at indices 1 .. hashNumber-1 in ascending order of first touch (hash10x.c:178) */
  if (z.hashNumber > 1) { s->depthDim = h10x_host_array_dim(1 << 20, 4, (int64_t)z.hashNumber - 1); s->depthMax = (int)z.hashNumber; }
  else { s->depthDim = 1 << 20; s->depthMax = 0; }
  /* clusterBlocks: arrayp(…,1) then one more per barcode (hash10x.c:200,218); main() creates it once */
  s->blocksDim = h10x_host_array_dim(s->blocksDim > 0 ? s->blocksDim : 1200, 32, (int64_t)z.nBlocksGlobal - 1);
  s->blocksMax = (int)z.nBlocksGlobal;
  return 0;
}

/* for callers that drive the library's C ABI themselves between the two (streaming ingest): initialise() with the latched parameters,
   and the Array dims a finished --readFQB leaves */
int h10x_session_begin(h10x_session *s) { return session_init(s) ? -1 : 0; }
int h10x_session_after_read(h10x_session *s) { return after_readFQB(s); }

/* --readFQB on an image in host memory: -N truncates (hash10x.c:202,207); the chunk loop's two side effects (the
   "chunkSize too small" death and the all-A-barcode quirk) are replayed inside the library from the run starts */
int h10x_session_readFQB_mem(h10x_session *s, const uint32_t *rec, uint64_t n) {
  if (session_init(s)) return -1;
  const int cutByN = s->N > 0 && (uint64_t)s->N <= n;                                 /* -N ends the reference's loop before its pass at end of file */
  if (s->N > 0 && (uint64_t)s->N < n) n = (uint64_t)s->N;
  if (h10x_set_option(s->ctx, "chunk_size", s->chunk) || h10x_set_option(s->ctx, "chunk_eof_pass", !cutByN)) return fail_ctx(s);
  if (h10x_read_fqb(s->ctx, rec, n)) return fail_ctx(s);
  return after_readFQB(s);
}

/* records already in HBM (pipelines, bench): -N only, no chunk semantics */
int h10x_session_readFQB_dev(h10x_session *s, const uint32_t *devRec, uint64_t n) {
  if (session_init(s)) return -1;
  if (s->N > 0 && (uint64_t)s->N < n) n = (uint64_t)s->N;
  if (h10x_read_fqb_device(s->ctx, devRec, n)) return fail_ctx(s);
  return after_readFQB(s);
}

/* records [first, first + n) of a file through the library's streaming ingest (h10x_ingest_fqb / h10x_shard_ingest_fqb), 64 MiB at a
   time: the file never sits in host memory, and the closing call is the --readFQB itself (collective when sharded). A rank that fails
   here while sharded still makes the closing call's counterpart impossible for the others, so the verdict is agreed on first. */
/* ---- --readFQB's fread loop (hash10x.c:202-209) as a pipeline of page-locked slabs, round 6. What scratch/r6_io_rate.c measured on a GPU box's memory-backed storage
   (profiles/r6g_io_rate.log): one thread copies out of the page cache at 9.5 GB/s, sixteen at 136 GB/s — reads scale, and PCIe takes ~55 GB/s — so the file is read by a
   pool of READERS persistent threads, PIECE bytes at a time (out of a mapping of the file: see stream_records), into NSLAB slabs of SLAB bytes; the calling thread alone talks to the library (the C ABI is single-threaded):
   it queues a slab's upload (h10x_ingest_fqb_async, in file order) as soon as all its pieces are in, and hands a slab back to the readers when its upload has landed.
   Round 5 read 16 MiB slabs on 4 threads created and joined per slab, one slab at a time: 5.2 GB/s, 4.6 s of the 9.5 s configs[2] needs end to end. */
enum { IN_PIECE = 4 << 20, IN_SLAB_DEFAULT = ((64 << 20) / 120 / 30) * 120 * 30, IN_NSLAB = 6, IN_MAXREADERS = 32 };   /* whole records per slab; H10X_SLAB_MIB (tests): slabs of that many MiB, so that small files cycle through the slots */
typedef struct {
  int fd; uint64_t base, bytes;                       /* the byte range of the file to read */
  uint64_t slabBytes;                                 /* bytes per slab: a whole number of 120-byte records */
  const char *map;                                    /* the whole file mapped (MAP_SHARED), or 0: pread */
  uint64_t nSlabs, piecesPerSlab;
  int S;                                              /* slabs in use: slab k lives in slot k % S */
  char *slab[IN_NSLAB];
  pthread_mutex_t mu; pthread_cond_t cv;
  uint64_t nextPiece;                                 /* next piece (global number: slab * piecesPerSlab + piece) a reader takes */
  uint64_t freeBelow;                                 /* slabs below this number may be written to (their slot's previous upload has landed) */
  uint32_t done[IN_NSLAB];                            /* pieces of the slab now in slot k that are in */
  int failed, stop;
} InPipe;
static uint64_t inpipe_slab_bytes(const InPipe *q, uint64_t k) { const uint64_t at = k * q->slabBytes; return q->bytes - at < q->slabBytes ? q->bytes - at : q->slabBytes; }
static void *inpipe_reader(void *a) {
  InPipe *q = (InPipe *)a;
  for (;;) {
    pthread_mutex_lock(&q->mu);
    const uint64_t p = q->nextPiece; const uint64_t k = p / q->piecesPerSlab;
    if (q->stop || q->failed || k >= q->nSlabs) { pthread_mutex_unlock(&q->mu); return 0; }
    while (k >= q->freeBelow && !q->stop && !q->failed) pthread_cond_wait(&q->cv, &q->mu);
    if (q->stop || q->failed) { pthread_mutex_unlock(&q->mu); return 0; }
    if (q->nextPiece != p) { pthread_mutex_unlock(&q->mu); continue; }           /* somebody else took it while this thread waited */
    q->nextPiece = p + 1;
    pthread_mutex_unlock(&q->mu);
    const uint64_t slabBytes = inpipe_slab_bytes(q, k), off = (p % q->piecesPerSlab) * (uint64_t)IN_PIECE;
    int ok = 1;
    if (off < slabBytes) {
      const uint64_t len = slabBytes - off < IN_PIECE ? slabBytes - off : IN_PIECE; uint64_t got = 0;
      char *dst = q->slab[k % (uint64_t)q->S] + off;
      if (q->map) {
        const char *src = q->map + q->base + k * q->slabBytes + off;
        memcpy(dst, src, len); got = len;
      }
      while (got < len) { const ssize_t r = pread(q->fd, dst + got, len - got, (off_t)(q->base + k * q->slabBytes + off + got)); if (r <= 0) break; got += (uint64_t)r; }
      ok = got == len;
    }
    pthread_mutex_lock(&q->mu);
    if (!ok) q->failed = 1;
    ++q->done[k % (uint64_t)q->S];
    pthread_cond_broadcast(&q->cv);
    pthread_mutex_unlock(&q->mu);
  }
}
typedef struct { void *p; size_t n; } UnmapJob;
static void *unmap_job(void *a) { UnmapJob *u = (UnmapJob *)a; munmap(u->p, u->n); free(u); return 0; }
/* The device code loads on a thread of its own (h10x_warm: HIP loads a translation unit's code object on first use, ~1.2 s for the library's 50 MB — with the file read
   at PCIe speed that, not the read, is what a --readFQB of a cached file waits for). Started when --readFQB has validated its parameters and has its context, once per process and device (ranks may be threads); joined where the first kernels are about
   to run. (Not at process start: a die() — exit(-1), hash10x.c's way — while that thread is inside the runtime's initialisation ends in the allocator's abort
   instead of exit code 255: test_cli_matches_reference_commands_and_reports, round 6; and it bought nothing measurable.) */
static int warmDevs[64]; static pthread_t warmThreads[64]; static volatile int warmState[64];   /* 0 not started, 1 running, 2 joined */
static void *warm_job(void *a) { (void)h10x_warm(*(int *)a); return 0; }           /* a failure here shows up in the first command proper */
void h10x_host_warm_start(int device) {
  if (device < 0 || device >= 64 || getenv("H10X_NOWARM")) return;
  if (!__sync_bool_compare_and_swap(&warmState[device], 0, 1)) return;              /* (started before — running or long joined: a swap that stored 1 over "joined" sent a second --readFQB of the process into joining a dead thread) */
  warmDevs[device] = device;
  if (pthread_create(&warmThreads[device], 0, warm_job, &warmDevs[device]) != 0) warmState[device] = 2;
}
static void warm_join(int device) {
  if (device < 0 || device >= 64) return;
  if (__sync_bool_compare_and_swap(&warmState[device], 1, 3)) { pthread_join(warmThreads[device], 0); warmState[device] = 2; }
  else while (warmState[device] == 3) usleep(200);                                  /* another rank's thread is joining it */
}
static int stream_records(h10x_session *s, const char *path, uint64_t first, uint64_t n, int sharded) {
  int rc = 0;
  struct timespec t0, t1, t2, t3; clock_gettime(CLOCK_MONOTONIC, &t0);
  h10x_host_warm_start(s->device);                                                   /* (once per process and device) */
  const uint64_t bytes = n * 120;
  InPipe q; memset(&q, 0, sizeof q);
  q.fd = open(path, O_RDONLY); q.base = first * 120; q.bytes = bytes;
  /* The readers copy out of a MAP_SHARED mapping of the file rather than pread it: the FIRST read of a file fresh in memory-backed storage goes through the page cache's LRU
     lists under a lock when it is a read() — 16 threads: 15.3 GB/s the first time, 150-250 the second (scratch/r6_read_existing.c, profiles/r6n_read_existing.log,
     r6o_read_mmap.log) — and at 146 GB/s through a mapping (MADV_POPULATE_READ first: as slow as the read()). pread where the file cannot be mapped (a pipe, a special file), or H10X_NO_MMAP is set.
     (A file that shrinks under the mapping ends in SIGBUS where read() would have come back short: the reference's fread loop dies on a short file either way.) */
  size_t mapLen = 0;
  if (q.fd >= 0 && bytes && !getenv("H10X_NO_MMAP")) {
    struct stat sb;
    if (fstat(q.fd, &sb) == 0 && S_ISREG(sb.st_mode) && (uint64_t)sb.st_size >= q.base + bytes) {
      void *m = mmap(0, (size_t)(q.base + bytes), PROT_READ, MAP_SHARED, q.fd, 0);
      if (m != MAP_FAILED) { q.map = (const char *)m; mapLen = (size_t)(q.base + bytes); }
    }
  }
  { const char *e = getenv("H10X_SLAB_MIB"); const long mib = e ? atol(e) : 0; q.slabBytes = mib >= 1 && mib <= 1024 ? ((uint64_t)mib << 20) / 3600 * 3600 : (uint64_t)IN_SLAB_DEFAULT; }
  const uint64_t SLABB = q.slabBytes;
  q.nSlabs = (bytes + SLABB - 1) / SLABB; q.piecesPerSlab = (SLABB + IN_PIECE - 1) / IN_PIECE;
  const int nSlabBuf = q.nSlabs < IN_NSLAB ? (int)(q.nSlabs ? q.nSlabs : 1) : IN_NSLAB;
  int pinned = 1;
  for (int k = 0; k < nSlabBuf && pinned; ++k) if (!(q.slab[k] = (char *)h10x_pinned_alloc(bytes < SLABB ? (bytes ? bytes : 1) : SLABB))) pinned = 0;
  if (!pinned) { for (int k = 0; k < IN_NSLAB; ++k) { h10x_pinned_free(q.slab[k]); q.slab[k] = 0; } q.slab[0] = (char *)malloc(bytes < SLABB ? (bytes ? bytes : 1) : SLABB); }
  q.S = pinned ? nSlabBuf : 1;                                                        /* without page-locked memory: one slab, one synchronous copy at a time, as before */
  if (q.fd < 0) rc = fail(s, "failed to open fqb file %s", path);                     /* hash10x.c:1201 */
  else if (!q.slab[0]) rc = fail(s, "out of memory for a %d MiB read buffer", (int)(SLABB >> 20));
  else if (h10x_ingest_reserve(s->ctx, n)) rc = fail_ctx(s);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  pthread_t th[IN_MAXREADERS]; int nTh = 0; double waitRead = 0, waitUpload = 0, inCalls = 0;   /* ms the calling thread waited for the readers / for uploads to land / spent queueing uploads */
  if (!rc && q.nSlabs) {
    pthread_mutex_init(&q.mu, 0); pthread_cond_init(&q.cv, 0);
    q.freeBelow = (uint64_t)q.S;                                                      /* = slabs whose upload has landed + S */
    long cpus = sysconf(_SC_NPROCESSORS_ONLN); if (cpus < 1) cpus = 1;
    const char *e = getenv("H10X_READERS"); int want = e ? atoi(e) : 16;
    if (want < 1) want = 1;
    if (want > IN_MAXREADERS) want = IN_MAXREADERS;
    if (want > cpus) want = (int)cpus;
    if ((uint64_t)want > q.nSlabs * q.piecesPerSlab) want = (int)(q.nSlabs * q.piecesPerSlab);
    for (int i = 0; i < want; ++i) if (pthread_create(&th[nTh], 0, inpipe_reader, &q) == 0) ++nTh;
    if (!nTh) rc = fail(s, "could not start a reader thread");
    for (uint64_t k = 0; !rc && k < q.nSlabs; ++k) {
      const int slot = (int)(k % (uint64_t)q.S); const uint64_t want120 = inpipe_slab_bytes(&q, k) / 120;
      const double tw0 = now_ms();
      pthread_mutex_lock(&q.mu);
      while (q.done[slot] < q.piecesPerSlab && !q.failed) pthread_cond_wait(&q.cv, &q.mu);
      const int bad = q.failed; q.done[slot] = 0;
      pthread_mutex_unlock(&q.mu);
      waitRead += now_ms() - tw0;
      if (bad) { rc = fail(s, "file read problem"); break; }                          /* hash10x.c:209 */
      const double tc0 = now_ms();
      if (pinned ? h10x_ingest_fqb_async(s->ctx, (const uint32_t *)q.slab[slot], want120, slot)
                 : (sharded ? h10x_shard_ingest_fqb(s->ctx, (const uint32_t *)q.slab[0], want120, 0) : h10x_ingest_fqb(s->ctx, (const uint32_t *)q.slab[0], want120, 0))) { rc = fail_ctx(s); break; }
      /* a slab goes back to the readers when its upload has landed. Uploads are queued in order: with several slots the calling thread waits for the PREVIOUS slab's
         (this one is on its way while the next is read); with one slot for this one's */
      uint64_t landed; const double tw1 = now_ms(); inCalls += tw1 - tc0;
      if (q.S > 1) { landed = k; if (k >= 1 && h10x_ingest_wait(s->ctx, (int)((k - 1) % (uint64_t)q.S))) { rc = fail_ctx(s); break; } }
      else { landed = k + 1; if (pinned && h10x_ingest_wait(s->ctx, slot)) { rc = fail_ctx(s); break; } }
      waitUpload += now_ms() - tw1;
      pthread_mutex_lock(&q.mu);
      q.freeBelow = landed + (uint64_t)q.S;
      pthread_cond_broadcast(&q.cv);
      pthread_mutex_unlock(&q.mu);
    }
    pthread_mutex_lock(&q.mu); q.stop = 1; pthread_cond_broadcast(&q.cv); pthread_mutex_unlock(&q.mu);
    for (int i = 0; i < nTh; ++i) pthread_join(th[i], 0);
    pthread_mutex_destroy(&q.mu); pthread_cond_destroy(&q.cv);
  }
  /* the mapping goes away on a thread of its own, beside the hashing and the commands that follow: unmapping 24 GB is 6 M page-table entries, 0.8 s of one thread
     (dropping them piece by piece from the reader threads — MADV_DONTNEED — made the readers as slow as read(): the per-page cost is the kernel's either way) */
  if (q.map) {
    UnmapJob *u = (UnmapJob *)malloc(sizeof *u); pthread_t ut;
    if (u) { u->p = (void *)q.map; u->n = mapLen; }
    if (!u || pthread_create(&ut, 0, unmap_job, u) != 0) { munmap((void *)q.map, mapLen); free(u); }
    else pthread_detach(ut);
  }
  if (q.fd >= 0) close(q.fd);
  if (pinned) { for (int k = 0; k < IN_NSLAB; ++k) { if (q.slab[k]) h10x_ingest_wait(s->ctx, k); h10x_pinned_free(q.slab[k]); } } else free(q.slab[0]);
  const double tj0 = now_ms(); warm_join(s->device); const double waitWarm = now_ms() - tj0;
  if (sharded) { int allOk = 0; if (h10x_shard_agree(s->ctx, !rc, &allOk)) return fail_ctx(s); if (!allOk && !rc) rc = fail(s, "another rank failed to read its part of %s", path); }
  if (rc) { h10x_ingest_reserve(s->ctx, 0); return rc; }
  clock_gettime(CLOCK_MONOTONIC, &t2);
  if (sharded ? h10x_shard_ingest_fqb(s->ctx, 0, 0, 1) : h10x_ingest_fqb(s->ctx, 0, 0, 1)) return fail_ctx(s);
  clock_gettime(CLOCK_MONOTONIC, &t3);
  if (getenv("H10X_INGEST_TIMING"))                                                    /* where a --readFQB spends its wall time */
    fprintf(stderr, "  ingest of %.2f GB: buffers + image %.3f s, read + upload %.3f s (%.1f GB/s, %d readers%s; the calling thread waited %.3f s for readers, %.3f s for uploads, queued uploads for %.3f s, waited %.3f s for the device code to load), hashing + index %.3f s\n", (double)bytes / 1e9,
            (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec), (double)(t2.tv_sec - t1.tv_sec) + 1e-9 * (double)(t2.tv_nsec - t1.tv_nsec),
            (double)bytes / 1e9 / ((double)(t2.tv_sec - t1.tv_sec) + 1e-9 * (double)(t2.tv_nsec - t1.tv_nsec) + 1e-9), nTh, mapLen ? " on a mapping" : ", pread", waitRead * 1e-3, waitUpload * 1e-3, inCalls * 1e-3, waitWarm * 1e-3, (double)(t3.tv_sec - t2.tv_sec) + 1e-9 * (double)(t3.tv_nsec - t2.tv_nsec));
  return 0;
}
static int file_records(h10x_session *s, const char *path, uint64_t *n, int *cutByN) {
  struct stat sb;
  if (stat(path, &sb)) return fail(s, "failed to open fqb file %s", path);
  *n = (uint64_t)sb.st_size / 120;                                                   /* fread(u,120,…) ignores a partial tail */
  if (cutByN) *cutByN = s->N > 0 && (uint64_t)s->N <= *n;                            /* -N ends the reference's loop before its pass at end of file */
  if (s->N > 0 && (uint64_t)s->N < *n) *n = (uint64_t)s->N;
  return 0;
}

int h10x_session_readFQB(h10x_session *s, const char *path) {
  uint64_t n = 0; int cutByN = 0;
  if (file_records(s, path, &n, &cutByN)) return -1;
  if (session_init(s)) return -1;
  if (h10x_set_option(s->ctx, "chunk_size", s->chunk) || h10x_set_option(s->ctx, "chunk_eof_pass", !cutByN)) return fail_ctx(s);
  if (stream_records(s, path, 0, n, 0)) return -1;
  return after_readFQB(s);
}

/* ---- sharded sessions: one per rank; every call below is collective ---- */
int h10x_session_shardReadFQB_mem(h10x_session *s, h10x_comm *comm, const uint32_t *rec, uint64_t n) {
  if (session_init(s)) return -1;
  if (h10x_shard_attach(s->ctx, comm)) return fail(s, "h10x_shard_attach failed");
  if (h10x_shard_read_fqb(s->ctx, rec, n)) return fail_ctx(s);
  return after_readFQB(s);
}
int h10x_session_shardReadFQB_dev(h10x_session *s, h10x_comm *comm, const uint32_t *devRec, uint64_t n) {
  if (session_init(s)) return -1;
  if (h10x_shard_attach(s->ctx, comm)) return fail(s, "h10x_shard_attach failed");
  if (h10x_shard_read_fqb_device(s->ctx, devRec, n)) return fail_ctx(s);
  return after_readFQB(s);
}
/* this rank's records [first, first + n) of the (N-truncated) file, streamed into HBM; chunk semantics of the whole file */
int h10x_session_shardReadFQB_file(h10x_session *s, h10x_comm *comm, const char *path, uint64_t first, uint64_t n) {
  if (session_init(s)) return -1;
  if (h10x_shard_attach(s->ctx, comm)) return fail(s, "h10x_shard_attach failed");
  uint64_t nFile = 0; int cutByN = 0;
  if (file_records(s, path, &nFile, &cutByN)) return -1;
  if (h10x_set_option(s->ctx, "chunk_size", s->chunk) || h10x_set_option(s->ctx, "chunk_eof_pass", !cutByN)) return fail_ctx(s);
  if (stream_records(s, path, first, n, 1)) return -1;
  return after_readFQB(s);
}
int h10x_session_shardGather(h10x_session *s) {
  if (!s->ctx) return fail(s, "no hash state loaded: use readFQB or readHash first");
  if (h10x_shard_gather(s->ctx)) return fail_ctx(s);
  return 0;
}

int h10x_session_hashDepthRange(h10x_session *s, int min, int max) {
  if (!s->ctx) return fail(s, "no hash state loaded: use readFQB or readHash first");
  return h10x_depth_range(s->ctx, min, max) ? fail_ctx(s) : 0;
}
int h10x_session_cluster(h10x_session *s, int codeMin, int codeMax) {
  if (!s->ctx) return fail(s, "!! you must set hashDepthRange before cluster");
  return h10x_cluster(s->ctx, codeMin, codeMax, s->ct) ? fail_ctx(s) : 0;
}
int h10x_session_clusterSplit(h10x_session *s) {
  if (!s->ctx) return fail(s, "no hash state loaded: use readFQB or readHash first");
  if (h10x_cluster_split(s->ctx)) return fail_ctx(s);
  h10x_shard_info_t z; if (h10x_shard_info(s->ctx, &z)) return fail_ctx(s);
  s->blocksDim = s->blocksMax = (int)z.nBlocksGlobal;                                /* arrayCreate(n) + arrayMax = n, hash10x.c:961-962 */
  return 0;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * --writeHash (hash10x.c:244-267 + arrayWrite, array.c:213-218). The file's layout is known from the sizes alone
 * (SURVEY App. B), so nothing is gathered: rank 0 creates the file at its final size and writes the headers, then every
 * rank pwrites its share — a slice of each replicated table, and the blocks / ClusterHash records of its own segments at
 * their place in file order. One rank (unsharded) is the same code. Heap-pointer fields are written as 0.
 * ------------------------------------------------------------------------------------------------------------------- */
typedef struct { int fd; const char *src; uint64_t at, len; int ok; } WriteJob;
static void *write_job(void *a) {
  WriteJob *j = (WriteJob *)a; uint64_t done = 0;
  while (done < j->len) { const ssize_t w = pwrite(j->fd, j->src + done, j->len - done > (1u << 30) ? (1u << 30) : j->len - done, (off_t)(j->at + done)); if (w <= 0) break; done += (uint64_t)w; }
  j->ok = done == j->len; return 0;
}
/* n bytes to the file at `at`. ONE pwrite stream per file: writers of one file take the inode's lock in turn, and on a GPU box's memory-backed storage one thread
   writes a new file at 9.0 GB/s, two at 8.9, four at 4.1, eight at 4.4 (scratch/r6_io_rate.c, profiles/r6g_io_rate.log; a mapping of the file filled by memcpy: 2 - 4.6) —
   round 5 split every slice over four threads and had two slices in the air: 3.7 GB/s. What matters is that the one stream never waits for the device: Writer below. */
static pthread_mutex_t fileWriteMu = PTHREAD_MUTEX_INITIALIZER;                      /* (ranks of one process write the same file: one at a time) */
static int put(int fd, const void *p, uint64_t n, uint64_t at) {
  WriteJob job = {fd, (const char *)p, at, n, 0};
  if (n >= (1u << 20)) { pthread_mutex_lock(&fileWriteMu); write_job(&job); pthread_mutex_unlock(&fileWriteMu); }
  else write_job(&job);
  return job.ok ? 0 : -1;
}
/* --writeHash as a pipeline: two page-locked host buffers of 128 MiB; while one is being written to the file (put, on a thread of its own) the next slice comes off
   the device into the other (~50 GB/s against the file's 9): the write stream is busy all the time. */
typedef struct { int fd; void *buf; uint64_t n, at; int rc, busy; pthread_t th; double profWait, profExport; /* H10X_HOSTPROF: where --writeHash spends its time (per call: ranks of one process write side by side) */ } Writer;
static void *writer_job(void *a) { Writer *w = (Writer *)a; w->rc = put(w->fd, w->buf, w->n, w->at); return 0; }
static int writer_wait(Writer *w) { if (w->busy) { pthread_join(w->th, 0); w->busy = 0; } const int rc = w->rc; w->rc = 0; return rc; }
static int writer_start(Writer *w, int fd, uint64_t n, uint64_t at) {
  w->fd = fd; w->n = n; w->at = at; w->rc = 0;
  if (pthread_create(&w->th, 0, writer_job, w) == 0) { w->busy = 1; return 0; }
  return put(fd, w->buf, n, at);                                                     /* no thread: write here */
}
/* elements [first, first + count) of a device table to the file at byte `at`, through the two bounded host buffers (*turn: which is next) */
static int put_table(h10x_session *s, int fd, int table, size_t elem, uint64_t first, uint64_t count, uint64_t at, Writer w[2], int *turn, size_t bufBytes, const char *what) {
  const uint64_t step = bufBytes / elem;
  for (uint64_t done = 0; done < count; ) {
    const uint64_t n = count - done < step ? count - done : step;
    Writer *const k = &w[*turn]; *turn ^= 1;
    const double tw = hostprof() ? now_ms() : 0;
    if (writer_wait(k)) return fail(s, "%s", what);                                  /* the buffer's previous slice is in the file */
    const double te = hostprof() ? now_ms() : 0;
    if (h10x_export_slice(s->ctx, table, first + done, n, k->buf)) return fail_ctx(s);
    if (hostprof()) { k->profWait += te - tw; k->profExport += now_ms() - te; }
    if (table == H10X_TABLE_BLOCKS) for (uint64_t i = 0; i < n; ++i) ((h10x_block *)k->buf)[i].clusHash = 0;
    if (writer_start(k, fd, n * elem, at + done * elem)) return fail(s, "%s", what);
    done += n;
  }
  return 0;
}
int h10x_session_writeHash(h10x_session *s, const char *path) {
  if (!s->ctx) return fail(s, "no hash state loaded: use readFQB or readHash first");
  h10x_shard_info_t z; if (h10x_shard_info(s->ctx, &z)) return fail_ctx(s);
  if (h10x_shard_prepare_export(s->ctx)) return fail_ctx(s);                          /* collective: hashValue / hashIndex of the whole set */
  const uint64_t T = (uint64_t)1 << z.B;
  const uint64_t depthDim = (uint64_t)s->depthDim, blocksDim = (uint64_t)s->blocksDim > z.nBlocksGlobal ? (uint64_t)s->blocksDim : z.nBlocksGlobal;
  const uint64_t oIndex = 16, oNumber = oIndex + 4 * T, oValue = oNumber + 4, oDepthHdr = oValue + 8 * (uint64_t)z.hashNumber, oDepth = oDepthHdr + 32,
                 oBlocksHdr = oDepth + 4 * depthDim, oBlocks = oBlocksHdr + 32, oClus = oBlocks + 32 * blocksDim, total = oClus + 8 * z.nEntriesGlobal;
  const double tp0 = hostprof() ? now_ms() : 0;
  int fd = -1, rc = 0, turn = 0; enum { BUF = 128 << 20 }; int bufPinned = 1;
  Writer w[2]; memset(w, 0, sizeof w);
  w[0].buf = h10x_pinned_alloc(BUF); w[1].buf = h10x_pinned_alloc(BUF);             /* page-locked: the slices come off the device at DMA speed */
  if (!w[0].buf || !w[1].buf) { bufPinned = 0; h10x_pinned_free(w[0].buf); h10x_pinned_free(w[1].buf); w[0].buf = malloc(BUF); w[1].buf = malloc(BUF); }
  h10x_shard_seg *segs = (h10x_shard_seg *)calloc((size_t)z.nSegs + 1, sizeof *segs);
  /* (a failure of one rank up to here is carried into the agreement below: nobody skips a collective) */
  const double tp01 = hostprof() ? now_ms() : 0;
  if (!w[0].buf || !w[1].buf || !segs) rc = fail(s, "out of host memory for .hash export");
  else if (h10x_shard_segments(s->ctx, segs, z.nSegs + 1)) rc = fail_ctx(s);
  if (!rc && z.rank == 0) {
    fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) rc = fail(s, "failed to open hash file %s", path);
    else if (ftruncate(fd, (off_t)total)) rc = fail(s, "write fail 1");              /* unwritten parts read as zeros: Array tails beyond max */
    else {
      struct { char magic[4]; uint32_t version; uint16_t chs, cbs; int32_t B; } head = {{'1', '0', 'X', 'H'}, 2, 8, 32, z.B};
      const array_hdr hd = {ARRAY_MAGIC, 0, 0, (int32_t)depthDim, 4, s->depthMax, 0}, hb = {ARRAY_MAGIC, 0, 0, (int32_t)blocksDim, 32, s->blocksMax, 0};
      if (put(fd, &head, 16, 0)) rc = fail(s, "write fail 1");
      else if (put(fd, &z.hashNumber, 4, oNumber)) rc = fail(s, "failed to write hashNumber");
      else if (put(fd, &hd, 32, oDepthHdr)) rc = fail(s, "failed to write hashDepth array");
      else if (put(fd, &hb, 32, oBlocksHdr)) rc = fail(s, "failed to write clusterBlocks array");
      else if (s->depthTail && depthDim > (uint64_t)s->depthTailFrom &&              /* bytes beyond max travel unchanged from --readHash */
               put(fd, s->depthTail, 4 * (depthDim - (uint64_t)s->depthTailFrom), oDepth + 4 * (uint64_t)s->depthTailFrom)) rc = fail(s, "failed to write hashDepth array");
    }
  }
  {                                                                                  /* the file exists (or rank 0 failed) before anybody else opens it */
    uint64_t bad = rc ? 1 : 0;
    if (h10x_shard_allreduce_max_u64(s->ctx, &bad, 1)) { rc = fail_ctx(s); goto done; }
    if (bad) { if (!rc) rc = fail(s, "another rank could not prepare the export of %s", path); goto done; }
  }
  const double tp02 = hostprof() ? now_ms() : 0;
  if (z.rank != 0 && (fd = open(path, O_WRONLY)) < 0) rc = fail(s, "failed to open hash file %s", path);
  if (!rc) {                                                                         /* replicated tables: rank r writes the r-th part of each */
    const uint64_t R = (uint64_t)z.nranks, me = (uint64_t)z.rank;
    const uint64_t i0 = T * me / R, i1 = T * (me + 1) / R, v0 = (uint64_t)z.hashNumber * me / R, v1 = (uint64_t)z.hashNumber * (me + 1) / R;
    const uint64_t dLimit = (uint64_t)z.hashNumber < depthDim ? z.hashNumber : depthDim, d0 = dLimit * me / R, d1 = dLimit * (me + 1) / R;
    rc = put_table(s, fd, H10X_TABLE_HASHINDEX, 4, i0, i1 - i0, oIndex + 4 * i0, w, &turn, BUF, "write fail 2");
    if (!rc) rc = put_table(s, fd, H10X_TABLE_HASHVALUE, 8, v0, v1 - v0, oValue + 8 * v0, w, &turn, BUF, "failed to write hashValue");
    if (!rc) rc = put_table(s, fd, H10X_TABLE_HASHDEPTH, 4, d0, d1 - d0, oDepth + 4 * d0, w, &turn, BUF, "failed to write hashDepth array");
  }
  for (uint32_t i = 0; !rc && i < z.nSegs; ++i) {                                    /* my blocks and their ClusterHash records */
    if ((int)segs[i].rank != z.rank) continue;
    rc = put_table(s, fd, H10X_TABLE_BLOCKS, 32, segs[i].localStart, segs[i].count, oBlocks + 32 * (uint64_t)segs[i].globalBase, w, &turn, BUF, "failed to write clusterBlocks array");
    if (!rc) rc = put_table(s, fd, H10X_TABLE_CLUSHASH, 8, segs[i].localEntryStart, segs[i].entries, oClus + 8 * segs[i].globalEntryStart, w, &turn, BUF, "write fail 3");
  }
  for (int k = 0; k < 2; ++k) if (writer_wait(&w[k]) && !rc) rc = fail(s, "write fail 3");   /* the last slices are in the file */
  const double tp1 = hostprof() ? now_ms() : 0;
  if (fd >= 0 && close(fd) && !rc) rc = fail(s, "write fail 3");
  if (hostprof()) fprintf(stderr, "hostprof: writeHash %.1f ms to the last slice (buffers %.1f ms, file + headers %.1f ms, device -> host %.1f ms, waiting for the file %.1f ms), close %.1f ms, %.2f GB\n",
                          tp1 - tp0, tp01 - tp0, tp02 - tp01, w[0].profExport + w[1].profExport, w[0].profWait + w[1].profWait, now_ms() - tp1, (double)total / 1e9);
  fd = -1;
  {
    uint64_t bad = rc ? 1 : 0;
    if (h10x_shard_allreduce_max_u64(s->ctx, &bad, 1)) { rc = fail_ctx(s); goto done; }
    if (bad && !rc) rc = fail(s, "another rank failed to write %s", path);
  }
done:
  for (int k = 0; k < 2; ++k) (void)writer_wait(&w[k]);
  if (fd >= 0) close(fd);
  for (int k = 0; k < 2; ++k) { if (bufPinned) h10x_pinned_free(w[k].buf); else free(w[k].buf); }
  free(segs);
  return rc;
}

/* readHashFile (hash10x.c:269-315) + arrayRead (array.c:220-238), then the upload that replaces
   fillHashTable's input state */
int h10x_session_readHash(h10x_session *s, const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) return fail(s, "failed to open hash file %s", path);
  if (session_init(s)) { fclose(f); return -1; }
  int rc = 0;
  char name[5] = {0}; uint32_t version = 0; uint16_t chs = 0, cbs = 0; int32_t B = 0;
  uint32_t *hashIndex = 0, *depth = 0; uint64_t *hashValue = 0; h10x_block *blocks = 0; h10x_clushash *ch = 0;
  uint32_t hashNumber = 0; array_hdr h, hb;
  const uint64_t T = (uint64_t)1 << s->B;
  if (fread(name, 4, 1, f) != 1 || fread(&version, 4, 1, f) != 1 || fread(&chs, 2, 1, f) != 1 || fread(&cbs, 2, 1, f) != 1) { rc = fail(s, "read fail 0"); goto done; }
  if (strcmp(name, "10XH")) { rc = fail(s, "not a 10X hash file"); goto done; }
  if (version > 2) { rc = fail(s, "hash file version mismatch: file %d > code %d", version, 2); goto done; }
  if (chs != 8) { rc = fail(s, "ClusterHash structure size mismatch: file %d != code %d", chs, 8); goto done; }
  if (cbs != 32) { rc = fail(s, "ClusterBlock structure size mismatch: file %d != code %d", cbs, 32); goto done; }
  if (fread(&B, 4, 1, f) != 1) { rc = fail(s, "read fail 1"); goto done; }
  if (B != s->B) { rc = fail(s, "incompatible hash table size: rerun with -B %d", B); goto done; }
  hashIndex = (uint32_t *)malloc(T * 4);
  if (!hashIndex || fread(hashIndex, 4, T, f) != T) { rc = fail(s, "read fail 2"); goto done; }
  if (version == 1) {                                                               /* hashValue stored as an Array */
    if (fread(&h, 32, 1, f) != 1 || h.dim < h.max || h.max < 0) { rc = fail(s, "failed to read hashValue array"); goto done; }
    hashValue = (uint64_t *)malloc((size_t)h.dim * 8 + 8);
    if (!hashValue || fread(hashValue, 8, (size_t)h.dim, f) != (size_t)h.dim) { rc = fail(s, "failed to read hashValue array"); goto done; }
    hashNumber = (uint32_t)h.max;
  } else {
    if (fread(&hashNumber, 4, 1, f) != 1) { rc = fail(s, "failed to read hashNumber"); goto done; }
    if (hashNumber > (T >> 2)) { rc = fail(s, "failed to read hashValue"); goto done; }
    hashValue = (uint64_t *)malloc((size_t)hashNumber * 8 + 8);
    if (!hashValue || fread(hashValue, 8, hashNumber, f) != hashNumber) { rc = fail(s, "failed to read hashValue"); goto done; }
  }
  if (fread(&h, 32, 1, f) != 1 || h.size != 4 || h.dim < 0) { rc = fail(s, "failed to read hashDepth array"); goto done; }
  depth = (uint32_t *)calloc((size_t)(h.dim > (int)hashNumber ? h.dim : (int)hashNumber) + 1, 4);
  if (!depth || fread(depth, 4, (size_t)h.dim, f) != (size_t)h.dim) { rc = fail(s, "failed to read hashDepth array"); goto done; }
  if (fread(&hb, 32, 1, f) != 1 || hb.size != 32 || hb.dim < hb.max || hb.max < 1) { rc = fail(s, "failed to read clusterBlocks array"); goto done; }
  blocks = (h10x_block *)calloc((size_t)hb.dim + 1, 32);
  if (!blocks || fread(blocks, 32, (size_t)hb.dim, f) != (size_t)hb.dim) { rc = fail(s, "failed to read clusterBlocks array"); goto done; }
  {
    uint64_t nCh = 0;
    for (int i = 1; i < hb.max; ++i) nCh += blocks[i].nHash;
    ch = (h10x_clushash *)malloc(nCh ? nCh * 8 : 8);
    if (!ch || (nCh && fread(ch, 8, nCh, f) != nCh)) { rc = fail(s, "read fail 3"); goto done; }
    if (hashNumber < 1) hashNumber = 1;
    if (h10x_load_state(s->ctx, hashIndex, hashNumber, hashValue, depth, blocks, (uint32_t)hb.max, ch)) { rc = fail_ctx(s); goto done; }
  }
  s->depthDim = h.dim; s->depthMax = h.max; s->blocksDim = hb.dim; s->blocksMax = hb.max;
  free(s->depthTail); s->depthTail = 0;
  if (h.dim > (int)hashNumber) {                                                    /* keep the bytes the file carries beyond hashNumber */
    s->depthTailFrom = (int)hashNumber;
    s->depthTail = (uint32_t *)malloc((size_t)(h.dim - (int)hashNumber) * 4);
    memcpy(s->depthTail, depth + hashNumber, (size_t)(h.dim - (int)hashNumber) * 4);
  }
done:
  fclose(f);
  free(hashIndex); free(hashValue); free(depth); free(blocks); free(ch);
  return rc;
}

/* --readHash onto N ranks (collective): every rank reads the replicated tables and the block array, takes a contiguous cut of the
   blocks balanced by ClusterHash records, and preads its own records. Same checks and messages as the single-GPU reader. */
int h10x_session_shardReadHash(h10x_session *s, h10x_comm *comm, const char *path) {
  const int rank = h10x_comm_rank(comm), nranks = h10x_comm_size(comm);
  FILE *f = fopen(path, "rb");
  if (!f) return fail(s, "failed to open hash file %s", path);
  if (session_init(s)) { fclose(f); return -1; }
  if (h10x_shard_attach(s->ctx, comm)) { fclose(f); return fail(s, "h10x_shard_attach failed"); }
  int rc = 0, agreed = 0;
  struct { char magic[4]; uint32_t version; uint16_t chs, cbs; int32_t B; } head;
  uint32_t *hashIndex = 0, *depth = 0; uint64_t *hashValue = 0; h10x_block *blocks = 0; h10x_clushash *ch = 0;
  uint32_t hashNumber = 0; array_hdr h, hb;
  const uint64_t T = (uint64_t)1 << s->B;
  if (fread(&head, 16, 1, f) != 1) { rc = fail(s, "read fail 0"); goto done; }
  if (memcmp(head.magic, "10XH", 4)) { rc = fail(s, "not a 10X hash file"); goto done; }
  if (head.version != 2) { rc = fail(s, "hash file version %d: only version 2 files can be read onto several GPUs", (int)head.version); goto done; }
  if (head.chs != 8) { rc = fail(s, "ClusterHash structure size mismatch: file %d != code %d", head.chs, 8); goto done; }
  if (head.cbs != 32) { rc = fail(s, "ClusterBlock structure size mismatch: file %d != code %d", head.cbs, 32); goto done; }
  if (head.B != s->B) { rc = fail(s, "incompatible hash table size: rerun with -B %d", head.B); goto done; }
  hashIndex = (uint32_t *)malloc(T * 4);
  if (!hashIndex || fread(hashIndex, 4, T, f) != T) { rc = fail(s, "read fail 2"); goto done; }
  if (fread(&hashNumber, 4, 1, f) != 1) { rc = fail(s, "failed to read hashNumber"); goto done; }
  if (hashNumber > (T >> 2)) { rc = fail(s, "failed to read hashValue"); goto done; }
  hashValue = (uint64_t *)malloc((size_t)hashNumber * 8 + 8);
  if (!hashValue || fread(hashValue, 8, hashNumber, f) != hashNumber) { rc = fail(s, "failed to read hashValue"); goto done; }
  if (fread(&h, 32, 1, f) != 1 || h.size != 4 || h.dim < 0) { rc = fail(s, "failed to read hashDepth array"); goto done; }
  depth = (uint32_t *)calloc((size_t)(h.dim > (int)hashNumber ? h.dim : (int)hashNumber) + 1, 4);
  if (!depth || fread(depth, 4, (size_t)h.dim, f) != (size_t)h.dim) { rc = fail(s, "failed to read hashDepth array"); goto done; }
  if (fread(&hb, 32, 1, f) != 1 || hb.size != 32 || hb.dim < hb.max || hb.max < 1) { rc = fail(s, "failed to read clusterBlocks array"); goto done; }
  blocks = (h10x_block *)calloc((size_t)hb.dim + 1, 32);
  if (!blocks || fread(blocks, 32, (size_t)hb.dim, f) != (size_t)hb.dim) { rc = fail(s, "failed to read clusterBlocks array"); goto done; }
  {
    const off_t chBase = ftello(f);
    uint64_t total = 0; for (int i = 1; i < hb.max; ++i) total += blocks[i].nHash;
    /* cut b_r = first block of rank r: the first block at which the records before it reach r / nranks of the total */
    uint32_t b0 = 1, b1 = (uint32_t)hb.max; uint64_t e0 = 0, e1 = total, acc = 0; int next = 1;
    if (rank == 0) { b0 = 1; e0 = 0; }
    for (int i = 1; i <= hb.max; ++i) {
      while (next <= nranks - 1 && (i == hb.max || acc >= (uint64_t)(((__uint128_t)total * (unsigned)next) / (unsigned)nranks))) {
        if (next == rank) { b0 = (uint32_t)i; e0 = acc; }
        if (next == rank + 1) { b1 = (uint32_t)i; e1 = acc; }
        ++next;
      }
      if (i < hb.max) acc += blocks[i].nHash;
    }
    const uint32_t nLocal = b1 - b0 + 1;                                             /* slot 0 + my blocks */
    h10x_block *mine = (h10x_block *)calloc((size_t)nLocal, 32);
    ch = (h10x_clushash *)malloc(e1 > e0 ? (e1 - e0) * 8 : 8);
    if (!mine || !ch) { free(mine); rc = fail(s, "out of host memory for the blocks of rank %d", rank); goto done; }
    memcpy(mine + 1, blocks + b0, (size_t)(b1 - b0) * 32);
    if (e1 > e0 && (fseeko(f, chBase + (off_t)(e0 * 8), SEEK_SET) || fread(ch, 8, e1 - e0, f) != e1 - e0)) { free(mine); rc = fail(s, "read fail 3"); goto done; }
    if (hashNumber < 1) hashNumber = 1;
    /* every rank has read its own part by now: one that failed alone (a file cut short inside its records) must not leave
       the others waiting in the collective load */
    { int allOk = 0; if (h10x_shard_agree(s->ctx, 1, &allOk)) { free(mine); rc = fail_ctx(s); goto out; }
      if (!allOk) { free(mine); rc = fail(s, "another rank failed to read its part of %s", path); goto out; } }
    agreed = 1;
    if (h10x_shard_load_state(s->ctx, hashIndex, hashNumber, hashValue, depth, mine, nLocal, ch, b0 - 1, (uint32_t)hb.max)) rc = fail_ctx(s);
    free(mine);
    if (rc) goto done;
  }
  s->depthDim = h.dim; s->depthMax = h.max; s->blocksDim = hb.dim; s->blocksMax = hb.max;
  free(s->depthTail); s->depthTail = 0;
  if (h.dim > (int)hashNumber) {                                                    /* keep the bytes the file carries beyond hashNumber */
    s->depthTailFrom = (int)hashNumber;
    s->depthTail = (uint32_t *)malloc((size_t)(h.dim - (int)hashNumber) * 4);
    memcpy(s->depthTail, depth + hashNumber, (size_t)(h.dim - (int)hashNumber) * 4);
  }
done:
  if (rc && !agreed) { int allOk = 0; (void)h10x_shard_agree(s->ctx, 0, &allOk); }    /* my failure, told to the ranks that wait for the verdict */
out:
  fclose(f);
  free(hashIndex); free(hashValue); free(depth); free(blocks); free(ch);
  return rc;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * --hashStats / --codeStats (hash10x.c:351-402). The histograms are filled on the device (h10x_report_histogram) and
 * summed over the ranks; what is left is the summary line and the table. The arithmetic keeps the reference's types so
 * that the printed figures agree digit for digit: counts are ints in an Array of int, the running sums are U64, the four
 * thresholds are doubles cut to int (hash10x.c:358), and i * count is an int product.
 * ------------------------------------------------------------------------------------------------------------------- */
static int cut_to_int(double v) { return v >= -2147483648.0 && v < 2147483648.0 ? (int)v : INT_MIN; }   /* what cvttsd2si leaves for an out-of-range value */
typedef struct { uint64_t count, mass; } Running;
/* smallest bin >= 1 whose running value exceeds the threshold (bin 0 cannot hold a median: `!median` stays true there) */
static int first_over(const Running *run, int bins, int threshold, int useMass) {
  for (int i = 1; i < bins; ++i) if ((useMass ? run[i].mass : run[i].count) > (uint64_t)(int64_t)threshold) return i;
  return 0;
}
static void write_histogram(FILE *f, const char *prefix, const uint64_t *hist, int bins) {
  Running *run = (Running *)calloc((size_t)bins + 1, sizeof *run);
  uint64_t count = 0, mass = 0; int mode = 0, massMode = 0; uint64_t best = 0, bestMass = 0;
  for (int i = 0; i < bins; ++i) {
    const int n = (int)hist[i]; const int in = (int)((uint32_t)i * (uint32_t)n);      /* int product, as in the reference */
    count += (uint64_t)(int64_t)n; mass += (uint64_t)(int64_t)in;
    run[i].count = count; run[i].mass = mass;
    if ((uint64_t)(int64_t)n > best) { best = (uint64_t)(int64_t)n; mode = i; }
    if ((uint64_t)(int64_t)in > bestMass) { bestMass = (uint64_t)(int64_t)in; massMode = i; }
  }
  for (int i = 0; i < bins; ++i) fprintf(f, "%s_HIST %6d %d %.4f %.4f\n", prefix, i, (int)hist[i], run[i].count / (double)count, run[i].mass / (double)mass);
  const int median = first_over(run, bins, cut_to_int(count * 0.5), 0), p99 = first_over(run, bins, cut_to_int(count * 0.99), 0);
  const int n50 = first_over(run, bins, cut_to_int(mass * 0.5), 1), n99 = first_over(run, bins, cut_to_int(mass * 0.99), 1);
  fprintf(f, "%s_STATS MEAN %.1f  MODE %d  MEDIAN %d  PERCENT99 %d  MASS_MODE %d  N50 %d  N99 %d\n", prefix, mass / (double)count, mode, median, p99, massMode, n50, n99);
  free(run);
}
/* histogram of one source over this rank's part, summed over all ranks; *bins = largest value + 1 */
static int gather_histogram(h10x_session *s, int which, uint64_t first, uint64_t count, uint64_t **hist, int *bins) {
  uint32_t mx = 0;
  if (h10x_report_max(s->ctx, which, first, count, &mx)) return fail_ctx(s);
  uint64_t top = mx;
  if (h10x_shard_allreduce_max_u64(s->ctx, &top, 1)) return fail_ctx(s);
  *bins = (int)top + 1;
  *hist = (uint64_t *)calloc((size_t)*bins, 8);
  if (!*hist) return fail(s, "out of host memory for a histogram of %d bins", *bins);
  if (h10x_report_histogram(s->ctx, which, first, count, (uint32_t)*bins, *hist) || h10x_shard_allreduce_sum_u64(s->ctx, *hist, (uint32_t)*bins)) { free(*hist); *hist = 0; return fail_ctx(s); }
  return 0;
}

int h10x_session_hashStats(h10x_session *s, FILE *f) {         /* hashDepthHist: arrayMax(hashDepth) entries, index 0 included */
  h10x_shard_info_t z;
  if (!s->ctx || h10x_shard_info(s->ctx, &z) || !s->depthMax) { if (f) fprintf(stderr, "  no hash list to print stats for\n"); return 0; }
  const uint64_t have = (uint64_t)s->depthMax < z.hashNumber ? (uint64_t)s->depthMax : z.hashNumber;
  uint64_t *hist; int bins;
  /* hashDepth is the same on every rank: rank 0's copy is the histogram, the others contribute nothing */
  if (gather_histogram(s, 0, 0, z.rank == 0 ? have : 0, &hist, &bins)) return -1;
  if ((uint64_t)s->depthMax > have) hist[0] += (uint64_t)s->depthMax - have;           /* entries of the Array beyond hashNumber are zeros */
  if (f) write_histogram(f, "HASH_COUNT", hist, bins);
  free(hist);
  return 0;
}

int h10x_session_codeStats(h10x_session *s, FILE *f) {         /* codeSizeHist: every block of the Array, block 0 included */
  h10x_shard_info_t z;
  if (!s->ctx || h10x_shard_info(s->ctx, &z) || !z.nBlocksGlobal) { if (f) fprintf(stderr, "  no barcodes to print stats for\n"); return 0; }
  h10x_shard_seg *segs = (h10x_shard_seg *)calloc((size_t)z.nSegs + 1, sizeof *segs);
  if (!segs || h10x_shard_segments(s->ctx, segs, z.nSegs + 1)) { free(segs); return segs ? fail_ctx(s) : fail(s, "out of host memory"); }
  int rc = 0;
  for (int which = 1; which <= 2 && !rc; ++which) {
    /* one pair of collectives per source: each rank histograms all its segments locally first */
    uint64_t *sum = 0; int bins = 1; uint32_t mx = 0;
    for (uint32_t i = 0; i < z.nSegs && !rc; ++i) if ((int)segs[i].rank == z.rank) { uint32_t m = 0; if (h10x_report_max(s->ctx, which, segs[i].localStart, segs[i].count, &m)) rc = fail_ctx(s); if (m > mx) mx = m; }
    uint64_t top = mx;
    if (!rc && h10x_shard_allreduce_max_u64(s->ctx, &top, 1)) rc = fail_ctx(s);
    if (rc) break;
    bins = (int)top + 1;
    sum = (uint64_t *)calloc((size_t)bins, 8); uint64_t *part = (uint64_t *)calloc((size_t)bins, 8);
    if (!sum || !part) { free(sum); free(part); rc = fail(s, "out of host memory for a histogram of %d bins", bins); break; }
    for (uint32_t i = 0; i < z.nSegs && !rc; ++i) if ((int)segs[i].rank == z.rank) {
      if (h10x_report_histogram(s->ctx, which, segs[i].localStart, segs[i].count, (uint32_t)bins, part)) rc = fail_ctx(s);
      for (int b = 0; b < bins; ++b) sum[b] += part[b];
    }
    if (!rc && h10x_shard_allreduce_sum_u64(s->ctx, sum, (uint32_t)bins)) rc = fail_ctx(s);
    if (!rc) {
      sum[0] += 1;                                                                     /* block 0: nHash 0, nSubCluster 0 */
      if (f && which == 1) write_histogram(f, "CODE_SIZE", sum, bins);
      if (f && which == 2 && bins > 1) write_histogram(f, "CODE_CLUSTER", sum, bins);   /* arrayMax(clusterHist) > 1 */
    }
    free(sum); free(part);
  }
  free(segs);
  return rc;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * crib: --cribBuild, --clusterReport, --cribSummary (hash10x.c:406-521, 870-952, 1017-1061). The genomes are hashed and
 * looked up on the device (h10x_crib_genome / h10x_crib_finish) and the per-cluster figures are reduced there
 * (h10x_cluster_report); what is left here is the FASTA reader and the text.
 * ------------------------------------------------------------------------------------------------------------------- */
static const char *const cribTypeName[5] = {"err", "htA", "htB", "hom", "mul"};     /* hash10x.c:417 */

/* One genome as readSequence() + dna2indexConv deliver it to cribAddGenome (readseq.c:66-152, 513-522, hash10x.c:432):
   '>' anywhere starts a record, blanks / tabs / newlines are skipped, ACGT in either case are 0..3, N is 0, any other
   character is a "Bad char": the reference then stops reading the file, so do we (same message on stderr). A record
   without bases also ends the file. Returns the number of sequences; codes / start are malloc'ed. */
static int read_crib_fasta(const char *path, uint8_t **codesOut, uint64_t **startOut, uint32_t *nSeqOut, char *err, int errlen) {
  FILE *f = fopen(path, "r");
  if (!f) { snprintf(err, (size_t)errlen, "failed to open .fa file %s", path); return -1; }
  static signed char conv[256]; static int convInit = 0;
  if (!convInit) {
    memset(conv, -2, sizeof conv);
    conv['A'] = conv['a'] = 0; conv['C'] = conv['c'] = 1; conv['G'] = conv['g'] = 2; conv['T'] = conv['t'] = 3; conv['N'] = conv['n'] = 0;
    conv[' '] = conv['\t'] = -1; conv['\n'] = -3;
    convInit = 1;
  }
  size_t cap = 1 << 20, n = 0, capS = 64; uint32_t nSeq = 0;
  uint8_t *codes = (uint8_t *)malloc(cap); uint64_t *start = (uint64_t *)malloc(capS * 8);
  int line = 1, c;
  start[0] = 0;
  for (;;) {
    c = getc(f);
    if (c == '>') { while ((c = getc(f)) != EOF && c != '\n') {} ++line; }             /* id and description are not used */
    else if (c != EOF) ungetc(c, f);
    const size_t n0 = n; int bad = 0;
    while ((c = getc(f)) != EOF) {
      if (c == '>') { ungetc(c, f); break; }
      const int v = conv[c & 0xFF];
      if (v == -3) { ++line; continue; }
      if (v == -1) continue;
      if (v < 0) { fprintf(stderr, "Bad char 0x%x = '%c' at line %d, base %d\n", c, c, line, (int)(n - n0)); bad = 1; break; }
      if (n + 1 >= cap) { cap *= 2; codes = (uint8_t *)realloc(codes, cap); if (!codes) { fclose(f); snprintf(err, (size_t)errlen, "out of memory reading %s", path); return -1; } }
      codes[n++] = (uint8_t)v;
    }
    if (bad) { n = n0; break; }
    if (n == n0) break;                                                               /* readSequence returned 0: end of input */
    if (nSeq + 2 >= capS) { capS *= 2; start = (uint64_t *)realloc(start, capS * 8); }
    start[++nSeq] = n;
  }
  fclose(f);
  *codesOut = codes; *startOut = start; *nSeqOut = nSeq;
  return 0;
}

/* printArrayStats (hash10x.c:456-468) over a depth histogram with the reference's arrayMax */
static void crib_array_stats(FILE *f, const uint32_t *a, uint32_t arrayMax) {
  int sum = 0, min = -1, max = (int)arrayMax - 1; double total = 0;
  for (uint32_t i = 0; i < arrayMax; ++i) if (a[i]) { sum += (int)a[i]; total += (double)((int)a[i] * (int)i); if (min == -1) min = (int)i; }
  fprintf(f, "  %d mean %.1f min %d max %d\n", sum, total / sum, min, max);
}

/* out = NULL: take part (collective on a sharded context) without printing */
int h10x_session_cribBuild(h10x_session *s, const char *fa1, const char *fa2, FILE *out, int printTables) {
  if (!s->ctx) return fail(s, "no hash state loaded: use readFQB or readHash first");
  crib_free(s);
  const char *paths[2] = {fa1, fa2};
  /* the reference opens both files before it reads either (hash10x.c:1235-1236) */
  for (int g = 0; g < 2; ++g) { FILE *t = fopen(paths[g], "r"); if (!t) return fail(s, "failed to open .fa file %s", paths[g]); fclose(t); }
  for (int g = 0; g < 2; ++g) {
    uint8_t *codes = 0; uint64_t *start = 0; uint32_t nSeq = 0; uint64_t nPresent = 0, nAbsent = 0;
    if (read_crib_fasta(paths[g], &codes, &start, &nSeq, s->err, (int)sizeof s->err)) return -1;
    const int rc = h10x_crib_genome(s->ctx, codes, start, nSeq, g, &nPresent, &nAbsent);
    free(codes); free(start);
    if (rc) return fail_ctx(s);
    if (out) fprintf(out, "  read %d known and %d unknown hashes from %d sequences in crib genome\n", (int)nPresent, (int)nAbsent, (int)nSeq);
    if (out && out != stdout) printf("  read %d known and %d unknown hashes from %d sequences in crib genome\n", (int)nPresent, (int)nAbsent, (int)nSeq);
  }
  if (h10x_crib_finish(s->ctx)) return fail_ctx(s);
  if (!out) return 0;
  uint32_t dim = 0, amax[4];
  if (h10x_crib_sizes(s->ctx, &dim, amax)) return fail_ctx(s);
  uint32_t *hist = (uint32_t *)calloc((size_t)4 * dim + 4, 4);
  if (!hist || h10x_crib_export(s->ctx, 0, 0, 0, hist)) { free(hist); return hist ? fail_ctx(s) : fail(s, "out of host memory"); }
  const uint32_t *aErr = hist, *aHet = hist + dim, *aHom = hist + 2 * (size_t)dim, *aMul = hist + 3 * (size_t)dim;
  fprintf(out, "  crib matches\n");
  fprintf(out, "    hom  "); crib_array_stats(out, aHom, amax[2]);
  fprintf(out, "    het  "); crib_array_stats(out, aHet, amax[1]);
  fprintf(out, "    mul "); crib_array_stats(out, aMul, amax[3]);
  fprintf(out, "    err "); crib_array_stats(out, aErr, amax[0]);
  if (printTables) {                                                                 /* hash10x.c:502-509 */
    fprintf(out, "CRIB_TABLE      i   err         het          hom         mul\n");
    for (uint32_t i = 1; i < 256; ++i)
      fprintf(out, "CRIB_TABLE      %4d%12d%12d%12d%12d\n", (int)i, amax[0] > i ? (int)aErr[i] : 0, amax[1] > i ? (int)aHet[i] : 0,
              amax[2] > i ? (int)aHom[i] : 0, amax[3] > i ? (int)aMul[i] : 0);
  }
  free(hist);
  return 0;
}

/* per-hash host copies for the report texts (crib[] / cribType[] / hashDepth[]: 9 bytes per distinct hash), fetched once per crib */
static int crib_fetch(h10x_session *s, uint32_t hashNumber) {
  if (s->cribChr && s->cribN == hashNumber) return 0;
  crib_free(s);
  s->cribChr = (int16_t *)malloc((size_t)hashNumber * 2); s->cribPos = (uint16_t *)malloc((size_t)hashNumber * 2);
  s->cribType = (uint8_t *)malloc(hashNumber); s->cribDepth = (uint32_t *)malloc((size_t)hashNumber * 4);
  if (!s->cribChr || !s->cribPos || !s->cribType || !s->cribDepth) { crib_free(s); return fail(s, "out of host memory for the crib"); }
  if (h10x_crib_export(s->ctx, s->cribChr, s->cribPos, s->cribType, 0) || h10x_export_slice(s->ctx, H10X_TABLE_HASHDEPTH, 0, hashNumber, s->cribDepth)) { crib_free(s); return fail_ctx(s); }
  s->cribN = hashNumber;
  return 0;
}

/* a growing text buffer */
typedef struct { char *p; size_t n, cap; } Text;
static int text_room(Text *t, size_t more) {
  if (t->n + more <= t->cap) return 0;
  size_t nc = t->cap ? t->cap : 1 << 16; while (nc < t->n + more) nc *= 2;
  char *q = (char *)realloc(t->p, nc); if (!q) return -1;
  t->p = q; t->cap = nc; return 0;
}
static int text_printf(Text *t, const char *fmt, ...) {
  if (text_room(t, 512)) return -1;
  va_list ap; va_start(ap, fmt); const int w = vsnprintf(t->p + t->n, t->cap - t->n, fmt, ap); va_end(ap);
  if (w < 0) return -1;
  t->n += (size_t)w; return 0;
}
static int text_bytes(Text *t, const void *p, size_t n) { if (text_room(t, n)) return -1; memcpy(t->p + t->n, p, n); t->n += n; return 0; }

/* the lines of one block (hash10x.c:917-947) from its device-side figures */
static int report_block_lines(const h10x_session *s, Text *t, uint32_t code, const h10x_block *b, const h10x_block_rep *br, const h10x_cluster_rep *cl, int haveCrib) {
  if (text_printf(t, "  CLUSTER_SUMMARY %d nRead %d nHash %d nGoodHash %d nClusHash %d nClusRead %d nSubCluster %d\n",
                  (int)code, (int)b->nRead, (int)(code ? b->nHash : 0), (int)br->nGood, (int)br->nClusHash, (int)br->nClusRead, (int)b->nSubCluster)) return -1;
  const uint32_t nSub = b->nSubCluster > 255 ? 255 : b->nSubCluster;
  for (uint32_t j = 1; j <= nSub; ++j) {
    const h10x_cluster_rep *r = &cl[j - 1];
    if (!r->n) continue;
    if (text_printf(t, "    CODE_CLUSTER %d %d : %d reads %d hashes", (int)code, (int)j, (int)r->nRead, (int)r->n)) return -1;
    if (haveCrib) {
      if (text_printf(t, " %d hom, %d htA, %d htB, %d mul, %d err", (int)r->nt[3], (int)r->nt[1], (int)r->nt[2], (int)r->nt[4], (int)r->nt[0])) return -1;
      if (r->chr && text_printf(t, "  chr %d pos %d %d", (int)r->chr, (int)r->pMin, (int)r->pMax - (int)r->pMin + 1)) return -1;
      if (r->nBad) {
        if (text_printf(t, "  OTHER %d", (int)r->nBad)) return -1;
        for (uint32_t q = 0; q < r->nOtherListed && q < 10; ++q) {                   /* cribText (hash10x.c:511-521) */
          const uint32_t x = r->other[q]; const int ty = s->cribType[x];
          if (text_printf(t, " %d:%s", (int)x, cribTypeName[ty])) return -1;
          if (ty > 0 && ty < 4 && text_printf(t, "_%d.%d", (int)s->cribChr[x], (int)s->cribPos[x])) return -1;
          if (text_printf(t, "-%d", (int)s->cribDepth[x])) return -1;
        }
      }
    }
    if (text_bytes(t, "\n", 1)) return -1;
  }
  return 0;
}

/* codeClusterReport (hash10x.c:870-952). Every rank turns its blocks of [codeMin, codeMax) into text (figures from
   h10x_cluster_report, in runs of at most 64 k blocks) and hands rank 0 one package per run: {first global block, blocks,
   text bytes} + text + the blocks' pointToMin and nGoodHash, which rank 0 adds up in file order for MIN_POINT_DENSITY (the
   reference's double sum is order dependent). out = NULL on the other ranks. */
typedef struct { uint32_t first, count; uint64_t textBytes; } RunHead;
static RunHead run_head(const char *p) { RunHead h; memcpy(&h, p, sizeof h); return h; }   /* packages sit at any byte offset */
/* The per-barcode lines of the reference's --verbose --cluster (hash10x.c:827-834 codeClusterFind, :867 codeClusterReadMerge) for the blocks
   [codeMin, codeMax) just clustered, and its "too many clusters" notes (hash10x.c:813-814, printed whether verbose or not): from the blocks, the
   good-hash counts and the two figures per block the replay kernel left (H10X_TABLE_CLUSTER_RAW). Sharded: every rank formats its own blocks,
   rank 0 prints the runs by first block. out / err may be null (nothing printed there). */
int h10x_session_clusterVerbose(h10x_session *s, int codeMin, int codeMax, FILE *out, FILE *err) {
  if (!s->ctx) return fail(s, "no hash state loaded: use readFQB or readHash first");
  h10x_shard_info_t z; if (h10x_shard_info(s->ctx, &z)) return fail_ctx(s);
  if (!codeMin) codeMin = 1;                                                          /* hash10x.c:1243-1244 */
  if (!codeMax) codeMax = (int)z.nBlocksGlobal;
  enum { RUN = 65536 };
  uint64_t wantLines = out != 0;                                                      /* (rank 0 holds the stream: every rank must format if it prints) */
  if (h10x_shard_allreduce_max_u64(s->ctx, &wantLines, 1)) return fail_ctx(s);
  int rc = 0; Text pack[2] = {{0, 0, 0}, {0, 0, 0}}, lines[2] = {{0, 0, 0}, {0, 0, 0}};
  h10x_shard_seg *segs = (h10x_shard_seg *)calloc((size_t)z.nSegs + 1, sizeof *segs);
  h10x_block *blocks = (h10x_block *)malloc((size_t)RUN * sizeof *blocks);
  uint32_t *good = (uint32_t *)malloc((size_t)RUN * 4), *raw = (uint32_t *)malloc((size_t)RUN * 8);
  if (!segs || !blocks || !good || !raw) { rc = fail(s, "out of host memory for the verbose lines"); goto done; }
  if (h10x_shard_segments(s->ctx, segs, z.nSegs + 1)) { rc = fail_ctx(s); goto done; }
  for (uint32_t i = 0; i < z.nSegs && !rc; ++i) {
    if ((int)segs[i].rank != z.rank) continue;
    int64_t g0 = segs[i].globalBase, g1 = (int64_t)segs[i].globalBase + segs[i].count;
    if (g0 < codeMin) g0 = codeMin;
    if (g1 > codeMax) g1 = codeMax;
    for (int64_t g = g0; g < g1 && !rc; g += RUN) {
      const uint32_t n = (uint32_t)(g1 - g < RUN ? g1 - g : RUN), local = segs[i].localStart + (uint32_t)(g - segs[i].globalBase);
      if (h10x_export_slice(s->ctx, H10X_TABLE_CLUSTER_RAW, local, n, raw)) { rc = fail_ctx(s); break; }
      if (!wantLines) {                                                               /* not verbose: only the notes, i.e. only runs with a block that was given up (8 bytes per block looked at) */
        int any = 0; for (uint32_t b = 0; b < n && !any; ++b) any = raw[2 * b] >> 31;
        if (!any) continue;
      }
      if (h10x_export_slice(s->ctx, H10X_TABLE_BLOCKS, local, n, blocks) || h10x_export_slice(s->ctx, H10X_TABLE_NGOOD, local, n, good)) { rc = fail_ctx(s); break; }
      lines[0].n = lines[1].n = 0;
      for (uint32_t b = 0; b < n && !rc; ++b) {
        const int code = (int)(g + b); const h10x_block *k = &blocks[b];
        if (!good[b]) {                                                               /* codeClusterFind returns at once; codeClusterReadMerge still speaks if labels of an earlier clustering stand */
          if (k->nSubCluster && text_printf(&lines[0], " then %d merged clusters\n", (int)k->nSubCluster)) rc = -1;
          continue;
        }
        const uint32_t rawN = raw[2 * b] & 0x7FFFFFFFu;
        if (raw[2 * b] >> 31) { if (text_printf(&lines[1], "    code %d with %d good hashes has too many clusters\n", code, (int)good[b])) rc = -1; }
        if (rawN) { if (text_printf(&lines[0], "  code %d with %d reads %d hashes, %d good hashes, of which %d cluster into %d raw then %d merged clusters\n",
                                    code, (int)k->nRead, (int)k->nHash, (int)good[b], (int)raw[2 * b + 1], (int)rawN, (int)k->nSubCluster)) rc = -1; }
        else if (text_printf(&lines[0], "  code %d with %d reads %d hashes, %d good hashes, 0 clusters\n", code, (int)k->nRead, (int)k->nHash, (int)good[b])) rc = -1;
      }
      if (rc) { rc = fail(s, "out of host memory for the verbose lines"); break; }
      for (int w = 0; w < 2; ++w) {
        const RunHead h = {(uint32_t)g, n, lines[w].n};
        if (text_bytes(&pack[w], &h, sizeof h) || text_bytes(&pack[w], lines[w].p, lines[w].n)) { rc = fail(s, "out of host memory for the verbose lines"); break; }
      }
    }
  }
  for (int w = 0; w < 2; ++w) {                                                       /* to rank 0, in any order; printed by first block */
    uint64_t bad = rc ? 1 : 0, total = pack[w].n;
    if (h10x_shard_allreduce_max_u64(s->ctx, &bad, 1) || h10x_shard_allreduce_sum_u64(s->ctx, &total, 1)) { rc = fail_ctx(s); goto done; }
    if (bad) { if (!rc) rc = fail(s, "the verbose lines failed on another rank"); goto done; }
    char *all = z.rank == 0 ? (char *)malloc(total + 1) : 0; uint64_t *counts = (uint64_t *)calloc((size_t)z.nranks, 8);
    if ((z.rank == 0 && !all) || !counts) { free(all); free(counts); rc = fail(s, "out of host memory for the verbose lines"); goto done; }
    if (h10x_shard_gather_bytes(s->ctx, pack[w].p, pack[w].n, all, total, counts)) { free(all); free(counts); rc = fail_ctx(s); goto done; }
    FILE *to = w ? err : out;
    if (z.rank == 0 && to) {
      size_t nRuns = 0, capRuns = 64; const char **runs = (const char **)malloc(capRuns * sizeof *runs);
      for (uint64_t o = 0; o < total; ) {
        const RunHead h = run_head(all + o);
        if (nRuns == capRuns) { capRuns *= 2; runs = (const char **)realloc(runs, capRuns * sizeof *runs); }
        runs[nRuns++] = all + o;
        o += sizeof h + h.textBytes;
      }
      for (size_t a = 1; a < nRuns; ++a) {                                           /* few runs: insertion sort by first block */
        const char *r = runs[a]; size_t b = a;
        while (b > 0 && run_head(runs[b - 1]).first > run_head(r).first) { runs[b] = runs[b - 1]; --b; }
        runs[b] = r;
      }
      for (size_t a = 0; a < nRuns; ++a) { const RunHead h = run_head(runs[a]); fwrite(runs[a] + sizeof h, 1, h.textBytes, to); }
      free(runs);
    }
    free(all); free(counts);
  }
done:
  free(segs); free(blocks); free(good); free(raw); free(pack[0].p); free(pack[1].p); free(lines[0].p); free(lines[1].p);
  return rc;
}

int h10x_session_clusterReport(h10x_session *s, int codeMin, int codeMax, FILE *out) {
  if (!s->ctx) return fail(s, "no hash state loaded: use readFQB or readHash first");
  h10x_shard_info_t z; if (h10x_shard_info(s->ctx, &z)) return fail_ctx(s);
  if (!codeMax) codeMax = (int)z.nBlocksGlobal;                                      /* hash10x.c:1263-1264: codeMin 0 stays 0 (block 0 is reported, empty) */
  if (codeMin < 0 || codeMax > (int)z.nBlocksGlobal) return fail(s, "clusterReport code range %d..%d outside 0..%u", codeMin, codeMax, z.nBlocksGlobal);
  uint32_t histDim = 0; const int haveCrib = h10x_crib_sizes(s->ctx, &histDim, 0) == 0;
  if (haveCrib) { if (crib_fetch(s, z.hashNumber)) return -1; } else crib_free(s);
  enum { RUN = 65536 };
  int rc = 0; Text pack = {0, 0, 0}, lines = {0, 0, 0};
  h10x_shard_seg *segs = (h10x_shard_seg *)calloc((size_t)z.nSegs + 1, sizeof *segs);
  h10x_block *blocks = (h10x_block *)malloc((size_t)RUN * sizeof *blocks);
  h10x_block_rep *brep = (h10x_block_rep *)malloc((size_t)RUN * sizeof *brep);
  h10x_cluster_rep *crep = 0; uint64_t crepCap = 0;
  double *p2m = (double *)malloc((size_t)RUN * 8); uint32_t *good = (uint32_t *)malloc((size_t)RUN * 4);
  if (!segs || !blocks || !brep || !p2m || !good) { rc = fail(s, "out of host memory for clusterReport"); goto done; }
  if (h10x_shard_segments(s->ctx, segs, z.nSegs + 1)) { rc = fail_ctx(s); goto done; }
  if (z.rank == 0 && codeMin == 0 && codeMax > 0) {                                  /* block 0 belongs to no segment: all zeros */
    const RunHead h = {0, 1, 0}; const double zero = 0.0; const uint32_t g0 = 0; Text one = {0, 0, 0};
    h10x_block b0; memset(&b0, 0, sizeof b0); h10x_block_rep r0; memset(&r0, 0, sizeof r0);
    if (report_block_lines(s, &one, 0, &b0, &r0, 0, haveCrib)) { free(one.p); rc = fail(s, "out of host memory for clusterReport"); goto done; }
    RunHead hh = h; hh.textBytes = one.n;
    if (text_bytes(&pack, &hh, sizeof hh) || text_bytes(&pack, one.p, one.n) || text_bytes(&pack, &zero, 8) || text_bytes(&pack, &g0, 4)) { free(one.p); rc = fail(s, "out of host memory for clusterReport"); goto done; }
    free(one.p);
  }
  for (uint32_t i = 0; i < z.nSegs && !rc; ++i) {
    if ((int)segs[i].rank != z.rank) continue;
    int64_t g0 = segs[i].globalBase, g1 = (int64_t)segs[i].globalBase + segs[i].count;
    if (g0 < codeMin) g0 = codeMin;
    if (g1 > codeMax) g1 = codeMax;
    for (int64_t g = g0; g < g1 && !rc; g += RUN) {
      const uint32_t n = (uint32_t)(g1 - g < RUN ? g1 - g : RUN), local = segs[i].localStart + (uint32_t)(g - segs[i].globalBase);
      if (h10x_export_slice(s->ctx, H10X_TABLE_BLOCKS, local, n, blocks)) { rc = fail_ctx(s); break; }
      uint64_t nClus = 0; for (uint32_t b = 0; b < n; ++b) nClus += blocks[b].nSubCluster > 255 ? 255 : blocks[b].nSubCluster;
      if (nClus + 1 > crepCap) { free(crep); crepCap = (nClus + 1) * 2; crep = (h10x_cluster_rep *)malloc(crepCap * sizeof *crep); if (!crep) { rc = fail(s, "out of host memory for clusterReport"); break; } }
      uint64_t got = 0;
      if (h10x_cluster_report(s->ctx, local, n, brep, crep, crepCap, &got)) { rc = fail_ctx(s); break; }
      lines.n = 0; uint64_t at = 0;
      for (uint32_t b = 0; b < n; ++b) {
        if (report_block_lines(s, &lines, (uint32_t)g + b, &blocks[b], &brep[b], crep + at, haveCrib)) { rc = fail(s, "out of host memory for clusterReport"); break; }
        at += blocks[b].nSubCluster > 255 ? 255 : blocks[b].nSubCluster;
        p2m[b] = blocks[b].pointToMin; good[b] = brep[b].nGood;
      }
      const RunHead h = {(uint32_t)g, n, lines.n};
      if (!rc && (text_bytes(&pack, &h, sizeof h) || text_bytes(&pack, lines.p, lines.n) || text_bytes(&pack, p2m, (size_t)n * 8) || text_bytes(&pack, good, (size_t)n * 4)))
        rc = fail(s, "out of host memory for clusterReport");
    }
  }
  {                                                                                  /* to rank 0, in any order; printed by first block */
    uint64_t bad = rc ? 1 : 0, total = pack.n;
    if (h10x_shard_allreduce_max_u64(s->ctx, &bad, 1) || h10x_shard_allreduce_sum_u64(s->ctx, &total, 1)) { rc = fail_ctx(s); goto done; }
    if (bad) { if (!rc) rc = fail(s, "clusterReport failed on another rank"); goto done; }
    char *all = z.rank == 0 ? (char *)malloc(total + 1) : 0; uint64_t *counts = (uint64_t *)calloc((size_t)z.nranks, 8);
    if ((z.rank == 0 && !all) || !counts) { free(all); free(counts); rc = fail(s, "out of host memory for clusterReport"); goto done; }
    if (h10x_shard_gather_bytes(s->ctx, pack.p, pack.n, all, total, counts)) { free(all); free(counts); rc = fail_ctx(s); goto done; }
    if (z.rank == 0 && out) {
      size_t nRuns = 0, capRuns = 64; const char **runs = (const char **)malloc(capRuns * sizeof *runs);
      for (uint64_t o = 0; o < total; ) {
        const RunHead h = run_head(all + o);
        if (nRuns == capRuns) { capRuns *= 2; runs = (const char **)realloc(runs, capRuns * sizeof *runs); }
        runs[nRuns++] = all + o;
        o += sizeof h + h.textBytes + (uint64_t)h.count * 12;
      }
      for (size_t a = 1; a < nRuns; ++a) {                                           /* few runs: insertion sort by first block */
        const char *r = runs[a]; size_t b = a;
        while (b > 0 && run_head(runs[b - 1]).first > run_head(r).first) { runs[b] = runs[b - 1]; --b; }
        runs[b] = r;
      }
      uint64_t totalGood = 0; double totalPoint = 0.0;
      for (size_t a = 0; a < nRuns; ++a) {
        const RunHead h = run_head(runs[a]);
        const char *text = runs[a] + sizeof h;
        fwrite(text, 1, h.textBytes, out);
        for (uint32_t b = 0; b < h.count; ++b) {
          double p; uint32_t g; memcpy(&p, text + h.textBytes + (size_t)b * 8, 8); memcpy(&g, text + h.textBytes + (size_t)h.count * 8 + (size_t)b * 4, 4);
          totalGood += g; totalPoint += p;
        }
      }
      if (totalGood) fprintf(out, "  MIN_POINT_DENSITY %.3f\n", totalPoint / totalGood);
      free(runs);
    }
    free(all); free(counts);
  }
done:
  free(pack.p); free(lines.p); free(segs); free(blocks); free(brep); free(crep); free(p2m); free(good);
  return rc;
}

/* ---- the reference's HASH object (hash.c) as far as cribSummary uses it — hashCreate(1 << 20), hashAdd of int keys, hashCount — restated with its two oddities,
   because the second figure of every type in the summary is its hashCount():
   (1) when a table doubles (hash.c:123-160) the bounce stride is worked out once, for the first key that bounces, and reused for every later key of that doubling: keys
       re-inserted off their own probe sequence are not found by a later hashAdd of the same key, which inserts them a second time, so the count runs ahead of the
       number of distinct keys (3 Gb-shaped set at 1/10: 1 460 918 "htA hashes" where 1 340 473 are distinct) — by how much depends on the ORDER of the insertions;
   (2) the marker of a removed key, (INT_MAX - 1) ^ INT_MAX, is 1: a slot holding key 1 counts as free (keys are hash index ^ INT_MAX, hash.h:41: no index below
       2^28 maps to it, the rule is kept for the restatement's sake).
   No table doubles below 524 288 keys, where the figures are plain distinct counts (what the small-set tests compare). */
typedef struct { int nbits, n, guard; unsigned mask; int *keys; } RefHash;
static int rh_init(RefHash *h) { h->nbits = 20; h->mask = (1u << 20) - 1; h->guard = 1 << 19; h->n = 0; h->keys = (int *)calloc((size_t)1 << 20, sizeof(int)); return h->keys ? 0 : -1; }
static inline long rh_fold(int key, int step, int times, unsigned mask) { long v = key; int x = key >> step; while (times--) { v ^= x; x >>= step; } return v & (long)mask; }
static inline long rh_home(const RefHash *h, int key) { return rh_fold(key, 5, 12, h->mask); }             /* HASH_FUNC: sizeof(long) * 8 / 5 folds */
static inline long rh_stride(const RefHash *h, int key) { return rh_fold(key, 7, 9, h->mask) | 1; }        /* DELTA: odd, so coprime to the table size */
static int rh_double(RefHash *h) {
  const size_t oldSize = (size_t)1 << h->nbits;
  int *old = h->keys, *fresh = (int *)calloc(oldSize * 2, sizeof(int));
  if (!fresh) return -1;
  ++h->nbits; h->mask = (1u << h->nbits) - 1; h->guard = 1 << (h->nbits - 1); h->keys = fresh;
  long stride = 0;                                                                  /* (1): one stride for the whole doubling */
  for (size_t i = 0; i < oldSize; ++i) {
    const int k = old[i];
    if (!k || k == 1) continue;                                                     /* (2): key 1 is dropped as "removed" */
    long p = rh_home(h, k);
    while (fresh[p]) { if (!stride) stride = rh_stride(h, k); p = (p + stride) & (long)h->mask; }
    fresh[p] = k; --h->guard;
  }
  free(old);
  return 0;
}
static int rh_add(RefHash *h, int key) {
  if (!h->guard && rh_double(h)) return -1;
  long p = rh_home(h, key), stride = 0;
  for (;;) {
    const int cur = h->keys[p];
    if (cur == 0 || cur == 1) { if (!cur) --h->guard; h->keys[p] = key; ++h->n; return 0; }
    if (cur == key) return 0;
    if (!stride) stride = rh_stride(h, key);
    p = (p + stride) & (long)h->mask;
  }
}
/* test hook (tests/test_host_cpu.py, against the reference's own hash.c through oracle/_ref/refhash): hashCount() after hashAdd(HASH_INT(key)) of n keys; -1 without memory */
int h10x_host_refhash_count(const int32_t *keys, uint64_t n) {
  RefHash h; if (rh_init(&h)) return -1;
  int rc = 0;
  for (uint64_t i = 0; i < n && !rc; ++i) rc = rh_add(&h, (int)((uint32_t)keys[i] ^ 0x7FFFFFFFu));
  const int count = rc ? -1 : h.n;
  free(h.keys);
  return count;
}
typedef struct { RefHash *h; const uint32_t *w; uint64_t n; uint32_t sel; int rc; } RhJob;
static void *rh_job(void *a) {
  RhJob *j = (RhJob *)a;
  for (uint64_t i = 0; i < j->n; ++i)                                               /* HASH_INT(x): the key is x ^ INT_MAX (hash.h:41) */
    if ((j->w[i] & 0xF0000000u) == j->sel && rh_add(j->h, (int)((j->w[i] & 0x0FFFFFFFu) ^ 0x7FFFFFFFu))) { j->rc = -1; break; }
  return 0;
}

/* --cribSummary (hash10x.c:1017-1061): entries per crib type in base blocks and in the blocks --clusterSplit made — counted on the device — and, per type, the
   reference's hashCount() of the hash indices met there: the records go through RefHash in the order of the reference's walk (blocks in file order, rank after rank
   on shards: every rank's words travel to rank 0, 128 MB at a time; ten tables, a thread each). */
int h10x_session_cribSummary(h10x_session *s, FILE *out) {
  if (!s->ctx) return fail(s, "no hash state loaded: use readFQB or readHash first");
  uint32_t dim = 0;
  if (h10x_crib_sizes(s->ctx, &dim, 0)) { if (out) fprintf(stderr, "cribSummary requires crib\n"); return 0; }
  h10x_shard_info_t z; if (h10x_shard_info(s->ctx, &z)) return fail_ctx(s);
  const size_t words = ((size_t)z.hashNumber + 31) / 32;
  uint32_t *seen = (uint32_t *)calloc(2 * words + 1, 4); uint64_t counts[12]; int rc = 0;
  if (!seen) return fail(s, "out of host memory for cribSummary");
  if (h10x_crib_summary(s->ctx, counts, seen, seen + words)) { free(seen); return fail_ctx(s); }
  free(seen);
  if (out) fprintf(stderr, "made hash objects\n");
  if (h10x_shard_allreduce_sum_u64(s->ctx, counts, 12)) return fail_ctx(s);
  enum { CH = 32 << 20 };
  RefHash tab[10]; memset(tab, 0, sizeof tab);
  h10x_shard_seg *segs = (h10x_shard_seg *)calloc((size_t)z.nSegs + 1, sizeof *segs);
  uint32_t *mine = (uint32_t *)malloc((size_t)CH * 4), *got = z.nranks > 1 && z.rank == 0 ? (uint32_t *)malloc((size_t)CH * 4) : 0;
  uint64_t *cnt = (uint64_t *)calloc((size_t)z.nranks, 8);
  if (!segs || !mine || !cnt || (z.nranks > 1 && z.rank == 0 && !got)) rc = fail(s, "out of host memory for cribSummary");
  if (!rc && z.rank == 0) for (int i = 0; i < 10; ++i) if (rh_init(&tab[i])) { rc = fail(s, "out of host memory for cribSummary"); break; }
  if (!rc && h10x_shard_segments(s->ctx, segs, z.nSegs + 1)) rc = fail_ctx(s);
  { uint64_t bad = rc ? 1 : 0; if (h10x_shard_allreduce_max_u64(s->ctx, &bad, 1)) rc = fail_ctx(s); else if (bad && !rc) rc = fail(s, "cribSummary failed on another rank"); }
  for (uint32_t g = 0; !rc && g < z.nSegs; ++g) {                                    /* segments in file order; every rank walks the same list (the gathers are collective) */
    const int own = (int)segs[g].rank == z.rank;
    for (uint64_t at = 0; !rc && at < segs[g].entries; at += CH) {
      const uint64_t n = segs[g].entries - at < CH ? segs[g].entries - at : CH;
      if (own && h10x_crib_words(s->ctx, segs[g].localEntryStart + at, n, mine)) { rc = fail_ctx(s); }
      const uint32_t *w = mine;
      if (z.nranks > 1) {
        uint64_t bad = rc ? 1 : 0;
        if (h10x_shard_allreduce_max_u64(s->ctx, &bad, 1)) { rc = fail_ctx(s); break; }
        if (bad) { if (!rc) rc = fail(s, "cribSummary failed on another rank"); break; }
        if (h10x_shard_gather_bytes(s->ctx, mine, own ? n * 4 : 0, got, (uint64_t)CH * 4, cnt)) { rc = fail_ctx(s); break; }
        w = got;
      }
      if (z.rank == 0) {
        RhJob job[10]; pthread_t th[10]; int started[10];
        for (int i = 0; i < 10; ++i) { job[i].h = &tab[i]; job[i].w = w; job[i].n = n; job[i].sel = (i >= 5 ? 0x80000000u : 0u) | ((uint32_t)(i % 5) << 28); job[i].rc = 0;
                                       started[i] = pthread_create(&th[i], 0, rh_job, &job[i]) == 0; }
        for (int i = 0; i < 10; ++i) { if (started[i]) pthread_join(th[i], 0); else rh_job(&job[i]); if (job[i].rc) rc = fail(s, "out of host memory for cribSummary"); }
      }
    }
  }
  if (!rc && z.rank == 0 && out) {
    fprintf(out, "  %d base codes ", (int)counts[10] + 1);                           /* block 0 counts as a base code */
    for (int i = 0; i < 5; ++i) fprintf(out, " %s %llu %d %.1f", cribTypeName[i], (unsigned long long)counts[i], tab[i].n, counts[i] / (double)tab[i].n);
    fprintf(out, "\n  %d cluster codes ", (int)counts[11]);
    for (int i = 0; i < 5; ++i) fprintf(out, " %s %llu %d %.1f", cribTypeName[i], (unsigned long long)counts[5 + i], tab[5 + i].n, counts[5 + i] / (double)tab[5 + i].n);
    fputc('\n', out);
  }
  for (int i = 0; i < 10; ++i) free(tab[i].keys);
  free(segs); free(mine); free(got); free(cnt);
  return rc;
}

/* --sortFQB <in.fqb> <out.fqb>: the record sort between fq2b and --readFQB (README.md:26 shells out to
   `bsort -k 4 -r 120`), on the device. Needs a context only for its stream and allocator: created with the session's
   current parameters if there is none yet. */
int h10x_session_sortFQB(h10x_session *s, const char *inPath, const char *outPath) {
  FILE *f = fopen(inPath, "rb");
  if (!f) return fail(s, "failed to open fqb file %s", inPath);
  fseeko(f, 0, SEEK_END); const off_t bytes = ftello(f); fseeko(f, 0, SEEK_SET);
  if (bytes % 120) { fclose(f); return fail(s, "%s: size %lld is not a multiple of the 120-byte record", inPath, (long long)bytes); }
  const uint64_t n = (uint64_t)bytes / 120;
  uint32_t *in = (uint32_t *)malloc(bytes ? (size_t)bytes : 8), *out = (uint32_t *)malloc(bytes ? (size_t)bytes : 8);
  int rc = 0; FILE *g = 0;
  if (!in || !out) { rc = fail(s, "out of host memory for %lld bytes of records", (long long)bytes); goto done; }
  if (n && fread(in, 120, n, f) != n) { rc = fail(s, "failed to read %s", inPath); goto done; }
  if (!s->ctx && session_init(s)) { rc = -1; goto done; }
  if (h10x_sort_fqb(s->ctx, in, n, out)) { rc = fail_ctx(s); goto done; }
  if (!(g = fopen(outPath, "wb"))) { rc = fail(s, "failed to open output file %s", outPath); goto done; }
  if (n && fwrite(out, 120, n, g) != n) rc = fail(s, "failed to write %s", outPath);
done:
  fclose(f); if (g) fclose(g); free(in); free(out);
  return rc;
}
