/* h10x_host.c — see h10x_host.h. Plain C; links libh10x_hip.so. */
#define _GNU_SOURCE
#include "h10x_host.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <time.h>

#include <sys/types.h>
#define ARRAY_MAGIC 8918274                            /* array.h:56 */
typedef struct { int32_t magic, pad0; uint64_t base; int32_t dim, size, max, pad1; } array_hdr;   /* array.h:41-50 */

struct h10x_session {
  int k, w, r, B, N, chunk, ct, device;                /* params (hash10x.c:25-33) */
  int timing;                                          /* measurement hook: enable hipEvent timers on every new context */
  int clusterLds;                                      /* test knob forwarded to h10x_set_option("cluster_lds_budget") */
  int firstCap;                                        /* test knob forwarded to h10x_set_option("cluster_first_cap") */
  int firstGlobal;                                     /* test knob forwarded to h10x_set_option("cluster_first_global") */
  int bigRanks;                                        /* tuning knob forwarded to h10x_set_option("cluster_big_ranks") */
  int dbgSkip;                                         /* diagnostic knob forwarded to h10x_set_option("cluster_dbg_skip") */
  int threads0, budget0;                               /* tuning knobs forwarded to h10x_set_option("cluster_threads0" / "cluster_budget0") */
  int stamps;                                          /* diagnostic knob forwarded to h10x_set_option("cluster_stamps") */
  int maxSlots;                                        /* test knob forwarded to h10x_set_option("stage_a_max_slots") */
  int16_t *cribChr; uint16_t *cribPos; uint8_t *cribType; uint32_t cribN;   /* host copy of crib[] / cribType[] for the reports */
  h10x_ctx *ctx;
  int ctxK, ctxW, ctxR, ctxB, ctxDev;                   /* parameters the live context was created with */
  /* Array bookkeeping of the reference for the two arrays that are dumped raw into .hash */
  int depthDim, depthMax, blocksDim, blocksMax;
  uint32_t *depthTail; int depthTailFrom;              /* entries [hashNumber, dim) as read from a file (normally zero) */
  char err[1024];
};

static int fail(h10x_session *s, const char *fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(s->err, sizeof s->err, fmt, ap); va_end(ap);
  return -1;
}
static int fail_ctx(h10x_session *s) { snprintf(s->err, sizeof s->err, "%s", h10x_last_error(s->ctx)); return -1; }

h10x_session *h10x_session_new(void) {
  h10x_session *s = (h10x_session *)calloc(1, sizeof *s);
  if (!s) return 0;
  s->k = 21; s->w = 31; s->r = 17; s->B = 28; s->N = 0; s->chunk = 100000; s->ct = 5; s->device = 0;
  s->blocksDim = 1200;                                 /* arrayCreate(1200, ClusterBlock), hash10x.c:1151 */
  return s;
}
void h10x_session_free(h10x_session *s) {
  if (!s) return;
  if (s->ctx) h10x_destroy(s->ctx);
  free(s->depthTail); free(s->cribChr); free(s->cribPos); free(s->cribType); free(s);
}
const char *h10x_session_error(const h10x_session *s) { return s->err; }
h10x_ctx *h10x_session_ctx(h10x_session *s) { return s->ctx; }

static int *param_slot(h10x_session *s, const char *n) {
  if (!strcmp(n, "k")) return &s->k;
  if (!strcmp(n, "w")) return &s->w;
  if (!strcmp(n, "r")) return &s->r;
  if (!strcmp(n, "B")) return &s->B;
  if (!strcmp(n, "N")) return &s->N;
  if (!strcmp(n, "c")) return &s->chunk;
  if (!strcmp(n, "ct")) return &s->ct;
  if (!strcmp(n, "device")) return &s->device;
  if (!strcmp(n, "stage_a_max_slots")) return &s->maxSlots;
  if (!strcmp(n, "timing")) return &s->timing;
  if (!strcmp(n, "cluster_stamps")) return &s->stamps;
  if (!strcmp(n, "cluster_lds_budget")) return &s->clusterLds;
  if (!strcmp(n, "cluster_first_global")) return &s->firstGlobal;
  if (!strcmp(n, "cluster_first_cap")) return &s->firstCap;
  if (!strcmp(n, "cluster_dbg_skip")) return &s->dbgSkip;
  if (!strcmp(n, "cluster_big_ranks")) return &s->bigRanks;
  if (!strcmp(n, "cluster_threads0")) return &s->threads0;
  if (!strcmp(n, "cluster_budget0")) return &s->budget0;
  return 0;
}
int h10x_session_set(h10x_session *s, const char *name, int value) {
  int *p = param_slot(s, name); if (!p) return fail(s, "unknown parameter %s", name);
  *p = value; return 0;
}
int h10x_session_get(const h10x_session *s, const char *name) {
  int *p = param_slot((h10x_session *)s, name); return p ? *p : 0;
}

int h10x_host_array_dim(int dim, int size, int64_t last) {
  while (last >= dim) {                                /* arrayExtend is entered with n == dim when touching in order */
    if (dim * size < (1 << 23)) dim *= 2; else dim += 1024 + ((1 << 23) / size);
  }
  return dim;
}

int64_t h10x_host_check_chunks(const uint32_t *rec, uint64_t total, int N, int chunk, char *err, int errlen) {
  int64_t nReads = 0; uint64_t pos = 0; uint32_t barcode = 0; int64_t curRead = 0;
  while (!N || nReads < N) {
    int64_t thisChunk = (int64_t)chunk - curRead;
    if (thisChunk <= 0) { if (err) snprintf(err, errlen, "chunkSize too small"); return -1; }   /* hash10x.c:206 */
    if (N && nReads + thisChunk > N) thisChunk = N - nReads;
    uint64_t avail = total - pos;
    int64_t nRec = avail < (uint64_t)thisChunk ? (int64_t)avail : thisChunk;
    if (!nRec) break;
    const uint32_t *u = rec + 30 * pos;
    if (!barcode) barcode = u[0];
    for (int64_t i = 0; i < nRec; ++i) {
      if (u[30 * i] == barcode) ++curRead; else { curRead = 1; barcode = u[30 * i]; }
    }
    nReads += nRec; pos += (uint64_t)nRec;
  }
  return nReads;
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

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return 1e3 * t.tv_sec + 1e-6 * t.tv_nsec; }
static int hostprof(void) { static int v = -1; if (v < 0) v = getenv("H10X_HOSTPROF") != 0; return v; }

static int fail_ctx(h10x_session *s);
/* measurement / test knobs follow the context */
static int apply_options(h10x_session *s) {
  h10x_timing_enable(s->ctx, s->timing);
  h10x_set_option(s->ctx, "cluster_stamps", s->stamps);
  h10x_set_option(s->ctx, "cluster_lds_budget", s->clusterLds);
  h10x_set_option(s->ctx, "cluster_first_global", s->firstGlobal);
  h10x_set_option(s->ctx, "cluster_first_cap", s->firstCap);
  h10x_set_option(s->ctx, "cluster_dbg_skip", s->dbgSkip);
  h10x_set_option(s->ctx, "cluster_big_ranks", s->bigRanks);
  h10x_set_option(s->ctx, "cluster_threads0", s->threads0);
  h10x_set_option(s->ctx, "cluster_budget0", s->budget0);
  if (h10x_set_option(s->ctx, "stage_a_max_slots", s->maxSlots)) return fail_ctx(s);
  return 0;
}

/* initialise() (hash10x.c:1099-1118): a fresh context with the currently latched parameters */
static int session_init(h10x_session *s) {
  double t0 = now_ms();
  /* same hasher/table/device as the live context: keep it (its stream and warm memory pool); the next
     read/load call resets every table, which is all initialise() does to the state */
  if (s->ctx && s->ctxK == s->k && s->ctxW == s->w && s->ctxR == s->r && s->ctxB == s->B && s->ctxDev == s->device &&
      s->k > 0 && s->w > 0) {
    s->depthDim = 1 << 20; s->depthMax = 0;
    free(s->depthTail); s->depthTail = 0;
    h10x_timing_reset(s->ctx);                         /* timers are per initialise() */
    return apply_options(s);
  }
  if (s->ctx) { h10x_destroy(s->ctx); s->ctx = 0; }
  double t1 = now_ms();
  h10x_params p; memset(&p, 0, sizeof p);
  p.k = s->k; p.w = s->w; p.B = s->B;
  if (s->k > 0 && s->w > 0) p.factor1 = h10x_factor1_from_seed(s->r);
  if (h10x_create(&s->ctx, &p, s->device, 0, s->err, (int)sizeof s->err)) return -1;
  s->ctxK = s->k; s->ctxW = s->w; s->ctxR = s->r; s->ctxB = s->B; s->ctxDev = s->device;
  if (hostprof()) fprintf(stderr, "hostprof: destroy %.3f ms, create %.3f ms\n", t1 - t0, now_ms() - t1);
  if (apply_options(s)) return -1;
  s->depthDim = 1 << 20; s->depthMax = 0;             /* arrayCreate(1 << 20, U32), hash10x.c:1114 */
  free(s->depthTail); s->depthTail = 0;
  return 0;
}

static void after_readFQB(h10x_session *s) {
  h10x_sizes z; h10x_get_sizes(s->ctx, &z);
  /* hashDepth: touched at indices 1 .. hashNumber-1 in ascending order of first touch (hash10x.c:178) */
  if (z.hashNumber > 1) { s->depthDim = h10x_host_array_dim(1 << 20, 4, (int64_t)z.hashNumber - 1); s->depthMax = (int)z.hashNumber; }
  else { s->depthDim = 1 << 20; s->depthMax = 0; }
  /* clusterBlocks: arrayp(…,1) then one more per barcode (hash10x.c:200,218); main() creates it once */
  s->blocksDim = h10x_host_array_dim(s->blocksDim > 0 ? s->blocksDim : 1200, 32, (int64_t)z.nBlocks - 1);
  s->blocksMax = (int)z.nBlocks;
}

int h10x_session_readFQB_mem(h10x_session *s, const uint32_t *rec, uint64_t n) {
  if (session_init(s)) return -1;
  int64_t use = h10x_host_check_chunks(rec, n, s->N, s->chunk, s->err, (int)sizeof s->err);
  if (use < 0) return -1;
  if (h10x_read_fqb(s->ctx, rec, (uint64_t)use)) return fail_ctx(s);
  after_readFQB(s);
  return 0;
}

int h10x_session_readFQB_dev(h10x_session *s, const uint32_t *devRec, uint64_t n) {
  if (session_init(s)) return -1;
  if (s->N && (uint64_t)s->N < n) n = (uint64_t)s->N;
  if (h10x_read_fqb_device(s->ctx, devRec, n)) return fail_ctx(s);
  after_readFQB(s);
  return 0;
}

int h10x_session_shardReadFQB_mem(h10x_session *s, h10x_comm *comm, const uint32_t *rec, uint64_t n) {
  if (session_init(s)) return -1;
  if (h10x_shard_attach(s->ctx, comm)) return fail(s, "h10x_shard_attach failed");
  if (h10x_shard_read_fqb(s->ctx, rec, n)) return fail_ctx(s);
  return 0;
}
int h10x_session_shardReadFQB_dev(h10x_session *s, h10x_comm *comm, const uint32_t *devRec, uint64_t n) {
  if (session_init(s)) return -1;
  if (h10x_shard_attach(s->ctx, comm)) return fail(s, "h10x_shard_attach failed");
  if (h10x_shard_read_fqb_device(s->ctx, devRec, n)) return fail_ctx(s);
  return 0;
}
int h10x_session_shardGather(h10x_session *s) {
  if (!s->ctx) return fail(s, "no hash state loaded: use readFQB or readHash first");
  if (h10x_shard_gather(s->ctx)) return fail_ctx(s);
  h10x_sizes z;
  if (!h10x_get_sizes(s->ctx, &z) && z.nBlocks) after_readFQB(s);   /* Array dims as one --readFQB of the whole file would leave them */
  return 0;
}

int h10x_session_readFQB(h10x_session *s, const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) return fail(s, "failed to open fqb file %s", path);                      /* hash10x.c:1201 */
  fseek(f, 0, SEEK_END); long long sz = ftell(f); fseek(f, 0, SEEK_SET);
  uint64_t n = (uint64_t)sz / 120;                                                 /* fread(u,120,…) ignores a partial tail */
  if (s->N && (uint64_t)s->N < n) n = (uint64_t)s->N;
  uint32_t *rec = (uint32_t *)malloc(n ? n * 120 : 120);
  if (!rec) { fclose(f); return fail(s, "myalloc failure requesting %lld bytes", sz); }
  if (n && fread(rec, 120, n, f) != n) { fclose(f); free(rec); return fail(s, "file read problem"); }   /* hash10x.c:209 */
  fclose(f);
  int rc = h10x_session_readFQB_mem(s, rec, n);
  free(rec);
  return rc;
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
  h10x_sizes z; h10x_get_sizes(s->ctx, &z);
  s->blocksDim = s->blocksMax = (int)z.nBlocks;                                    /* arrayCreate(n) + arrayMax = n, hash10x.c:961-962 */
  return 0;
}

/* histogramReport (hash10x.c:351-375), same arithmetic (ints and doubles) and text */
static void histogram_report(FILE *f, const char *prefix, const int *a, int n) {
  uint64_t sum = 0, total = 0;
  for (int i = 0; i < n; ++i) { sum += (uint64_t)a[i]; total += (uint64_t)((int64_t)i * a[i]); }
  uint64_t partSum = 0, partTotal = 0, max = 0, massMax = 0;
  int median = 0, massMedian = 0, n99 = 0, nMass99 = 0, mode = 0, massMode = 0;
  int t50 = (int)(sum * 0.5), tMass50 = (int)(total * 0.5), t99 = (int)(sum * 0.99), tMass99 = (int)(total * 0.99);
  for (int i = 0; i < n; ++i) {
    int v = a[i];
    partSum += (uint64_t)v; partTotal += (uint64_t)((int64_t)i * v);
    fprintf(f, "%s_HIST %6d %d %.4f %.4f\n", prefix, i, v, partSum / (double)sum, partTotal / (double)total);
    if ((uint64_t)v > max) { mode = i; max = (uint64_t)v; }
    if ((uint64_t)((int64_t)i * v) > massMax) { massMode = i; massMax = (uint64_t)((int64_t)i * v); }
    if (partSum > (uint64_t)t50 && !median) median = i;
    if (partTotal > (uint64_t)tMass50 && !massMedian) massMedian = i;
    if (partSum > (uint64_t)t99 && !n99) n99 = i;
    if (partTotal > (uint64_t)tMass99 && !nMass99) nMass99 = i;
  }
  fprintf(f, "%s_STATS MEAN %.1f", prefix, total / (double)sum);
  fprintf(f, "  MODE %d  MEDIAN %d  PERCENT99 %d", mode, median, n99);
  fprintf(f, "  MASS_MODE %d  N50 %d  N99 %d\n", massMode, massMedian, nMass99);
}
static int *hist_grow(int *h, int *cap, int need) {
  if (need < *cap) return h;
  int nc = *cap; while (nc <= need) nc *= 2;
  h = (int *)realloc(h, (size_t)nc * sizeof(int)); memset(h + *cap, 0, (size_t)(nc - *cap) * sizeof(int)); *cap = nc;
  return h;
}

int h10x_session_hashStats(h10x_session *s, FILE *f) {         /* hashDepthHist, hash10x.c:377-386 */
  h10x_sizes z;
  if (!s->ctx || h10x_get_sizes(s->ctx, &z) || !s->depthMax) { fprintf(stderr, "  no hash list to print stats for\n"); return 0; }
  uint32_t *depth = (uint32_t *)calloc((size_t)z.hashNumber + 1, 4);
  if (h10x_export(s->ctx, 0, 0, depth, 0, 0)) { free(depth); return fail_ctx(s); }
  int cap = 1024, top = 0; int *h = (int *)calloc((size_t)cap, sizeof(int));
  for (int i = 0; i < s->depthMax; ++i) {                      /* arrayMax(hashDepth) entries, index 0 included */
    int d = (uint32_t)i < z.hashNumber ? (int)depth[i] : 0;
    h = hist_grow(h, &cap, d); ++h[d]; if (d + 1 > top) top = d + 1;
  }
  histogram_report(f, "HASH_COUNT", h, top);
  free(h); free(depth);
  return 0;
}

int h10x_session_codeStats(h10x_session *s, FILE *f) {         /* codeSizeHist, hash10x.c:388-402 */
  h10x_sizes z;
  if (!s->ctx || h10x_get_sizes(s->ctx, &z) || !z.nBlocks) { fprintf(stderr, "  no barcodes to print stats for\n"); return 0; }
  h10x_block *b = (h10x_block *)calloc((size_t)z.nBlocks, sizeof *b);
  if (h10x_export(s->ctx, 0, 0, 0, b, 0)) { free(b); return fail_ctx(s); }
  int capH = 1024, capC = 1024, topH = 0, topC = 0;
  int *hh = (int *)calloc((size_t)capH, sizeof(int)), *hc = (int *)calloc((size_t)capC, sizeof(int));
  for (uint32_t i = 0; i < z.nBlocks; ++i) {
    int nh = (int)b[i].nHash, ns = (int)b[i].nSubCluster;
    hh = hist_grow(hh, &capH, nh); ++hh[nh]; if (nh + 1 > topH) topH = nh + 1;
    hc = hist_grow(hc, &capC, ns); ++hc[ns]; if (ns + 1 > topC) topC = ns + 1;
  }
  histogram_report(f, "CODE_SIZE", hh, topH);
  if (topC > 1) histogram_report(f, "CODE_CLUSTER", hc, topC);
  free(hh); free(hc); free(b);
  return 0;
}

/* writeHashFile (hash10x.c:244-267) + arrayWrite (array.c:213-218); heap-pointer fields are written as 0 */
int h10x_session_writeHash(h10x_session *s, const char *path) {
  if (!s->ctx) return fail(s, "no hash state loaded: use readFQB or readHash first");
  h10x_sizes z; if (h10x_get_sizes(s->ctx, &z)) return fail_ctx(s);
  const uint64_t T = (uint64_t)1 << z.B;
  uint32_t *hashIndex = (uint32_t *)malloc(T * 4);
  uint64_t *hashValue = (uint64_t *)malloc((size_t)z.hashNumber * 8);
  uint32_t *depth = (uint32_t *)calloc((size_t)s->depthDim > z.hashNumber ? (size_t)s->depthDim : z.hashNumber, 4);
  h10x_block *blocks = (h10x_block *)calloc((size_t)s->blocksDim > z.nBlocks ? (size_t)s->blocksDim : z.nBlocks, sizeof(h10x_block));
  h10x_clushash *ch = (h10x_clushash *)malloc(z.nClusHash ? z.nClusHash * 8 : 8);
  int rc = 0; FILE *f = 0;
  if (!hashIndex || !hashValue || !depth || !blocks || !ch) { rc = fail(s, "out of host memory for .hash export"); goto done; }
  if (h10x_export(s->ctx, hashIndex, hashValue, depth, blocks, ch)) { rc = fail_ctx(s); goto done; }
  if (s->depthTail)                                                                 /* bytes beyond max travel unchanged from --readHash */
    for (int i = s->depthTailFrom; i < s->depthDim; ++i) depth[i] = s->depthTail[i - s->depthTailFrom];
  for (uint32_t i = 0; i < z.nBlocks; ++i) blocks[i].clusHash = 0;
  if (!(f = fopen(path, "wb"))) { rc = fail(s, "failed to open hash file %s", path); goto done; }
  {
    uint32_t version = 2; uint16_t chs = 8, cbs = 32; int32_t B = z.B;
    if (fwrite("10XH", 4, 1, f) != 1 || fwrite(&version, 4, 1, f) != 1 || fwrite(&chs, 2, 1, f) != 1 ||
        fwrite(&cbs, 2, 1, f) != 1 || fwrite(&B, 4, 1, f) != 1) { rc = fail(s, "write fail 1"); goto done; }
    if (fwrite(hashIndex, 4, T, f) != T) { rc = fail(s, "write fail 2"); goto done; }
    if (fwrite(&z.hashNumber, 4, 1, f) != 1) { rc = fail(s, "failed to write hashNumber"); goto done; }
    if (fwrite(hashValue, 8, z.hashNumber, f) != z.hashNumber) { rc = fail(s, "failed to write hashValue"); goto done; }
    array_hdr h = {ARRAY_MAGIC, 0, 0, s->depthDim, 4, s->depthMax, 0};
    if (fwrite(&h, 32, 1, f) != 1 || fwrite(depth, 4, (size_t)s->depthDim, f) != (size_t)s->depthDim) { rc = fail(s, "failed to write hashDepth array"); goto done; }
    array_hdr hb = {ARRAY_MAGIC, 0, 0, s->blocksDim, 32, s->blocksMax, 0};
    if (fwrite(&hb, 32, 1, f) != 1 || fwrite(blocks, 32, (size_t)s->blocksDim, f) != (size_t)s->blocksDim) { rc = fail(s, "failed to write clusterBlocks array"); goto done; }
    if (z.nClusHash && fwrite(ch, 8, z.nClusHash, f) != z.nClusHash) { rc = fail(s, "write fail 3"); goto done; }
  }
done:
  if (f) fclose(f);
  free(hashIndex); free(hashValue); free(depth); free(blocks); free(ch);
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

/* ---------------------------------------------------------------------------------------------------------------------
 * crib: --cribBuild, --clusterReport, --cribSummary (hash10x.c:406-521, 870-952, 1017-1061). The genomes are hashed and
 * looked up on the device (h10x_crib_genome / h10x_crib_finish); what is left here is the FASTA reader and the text.
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

static void crib_free(h10x_session *s) { free(s->cribChr); free(s->cribPos); free(s->cribType); s->cribChr = 0; s->cribPos = 0; s->cribType = 0; s->cribN = 0; }

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
    fprintf(out, "  read %d known and %d unknown hashes from %d sequences in crib genome\n", (int)nPresent, (int)nAbsent, (int)nSeq);
    if (out != stdout) printf("  read %d known and %d unknown hashes from %d sequences in crib genome\n", (int)nPresent, (int)nAbsent, (int)nSeq);
  }
  if (h10x_crib_finish(s->ctx)) return fail_ctx(s);
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

/* host copies of crib[] / cribType[] for the text reports, fetched once per crib */
static int crib_fetch(h10x_session *s) {
  h10x_sizes z; if (h10x_get_sizes(s->ctx, &z)) return fail_ctx(s);
  if (s->cribChr && s->cribN == z.hashNumber) return 0;
  crib_free(s);
  s->cribChr = (int16_t *)malloc((size_t)z.hashNumber * 2); s->cribPos = (uint16_t *)malloc((size_t)z.hashNumber * 2); s->cribType = (uint8_t *)malloc(z.hashNumber);
  if (!s->cribChr || !s->cribPos || !s->cribType) { crib_free(s); return fail(s, "out of host memory for the crib"); }
  if (h10x_crib_export(s->ctx, s->cribChr, s->cribPos, s->cribType, 0)) { crib_free(s); return fail_ctx(s); }
  s->cribN = z.hashNumber;
  return 0;
}

/* cribText (hash10x.c:511-521) */
static const char *crib_text(const h10x_session *s, const uint32_t *depth, uint32_t x, char *text) {
  char *t = text;
  t += sprintf(t, "%d", (int)x);
  if (s->cribChr) t += sprintf(t, ":%s", cribTypeName[s->cribType[x]]);
  if (s->cribChr && s->cribType[x] > 0 && s->cribType[x] < 4) t += sprintf(t, "_%d.%d", (int)s->cribChr[x], (int)s->cribPos[x]);
  sprintf(t, "-%d", (int)depth[x]);
  return text;
}

/* codeClusterReport (hash10x.c:870-952): per barcode a CLUSTER_SUMMARY line and one CODE_CLUSTER line per sub-cluster
   with its reads, hashes, crib composition, the chromosome / position span of its first located hash and the hashes
   that disagree with it. With or without a crib, like the reference. */
int h10x_session_clusterReport(h10x_session *s, int codeMin, int codeMax, FILE *out) {
  if (!s->ctx) return fail(s, "no hash state loaded: use readFQB or readHash first");
  h10x_sizes z; if (h10x_get_sizes(s->ctx, &z)) return fail_ctx(s);
  if (!codeMax) codeMax = (int)z.nBlocks;                                            /* hash10x.c:1263-1264: codeMin 0 stays 0 (block 0 is reported, empty) */
  if (codeMin < 0 || codeMax > (int)z.nBlocks) return fail(s, "clusterReport code range %d..%d outside 0..%u", codeMin, codeMax, z.nBlocks);
  uint32_t sizes_dim = 0; const int haveCrib = h10x_crib_sizes(s->ctx, &sizes_dim, 0) == 0;
  if (haveCrib) { if (crib_fetch(s)) return -1; } else crib_free(s);
  int rc = 0;
  h10x_block *blocks = (h10x_block *)calloc((size_t)z.nBlocks + 1, sizeof *blocks);
  h10x_clushash *ch = (h10x_clushash *)malloc(z.nClusHash ? z.nClusHash * 8 : 8);
  uint32_t *depth = (uint32_t *)calloc((size_t)z.hashNumber + 1, 4), *nGood = (uint32_t *)calloc((size_t)z.nBlocks + 1, 4);
  typedef struct { int n, nRead, nt[5]; int16_t chr; uint16_t pMin, pMax; int nBad, bad; } ReportInfo;
  ReportInfo *info = 0; int *readClus = 0, *badLink = 0; size_t capInfo = 0, capRead = 0, capLink = 0;
  if (!blocks || !ch || !depth || !nGood) { rc = fail(s, "out of host memory for clusterReport"); goto done; }
  if (h10x_export(s->ctx, 0, 0, depth, blocks, ch)) { rc = fail_ctx(s); goto done; }
  if (h10x_export_ngood(s->ctx, nGood)) { rc = fail_ctx(s); goto done; }
  {
    uint64_t off = 0, totalGoodHash = 0; double totalPointToMin = 0.0;
    for (int c = 1; c < codeMin; ++c) off += blocks[c].nHash;
    for (int code = codeMin; code < codeMax; ++code) {
      const h10x_block *b = &blocks[code]; const h10x_clushash *e = ch + off; off += b->nHash;
      const size_t nInfo = (size_t)b->nSubCluster + 1;
      if (nInfo > capInfo) { capInfo = nInfo * 2; info = (ReportInfo *)realloc(info, capInfo * sizeof *info); }
      if ((size_t)b->nRead + 1 > capRead) { capRead = ((size_t)b->nRead + 1) * 2; readClus = (int *)realloc(readClus, capRead * sizeof(int)); }
      if ((size_t)b->nHash + 1 > capLink) { capLink = ((size_t)b->nHash + 1) * 2; badLink = (int *)realloc(badLink, capLink * sizeof(int)); }
      memset(info, 0, nInfo * sizeof *info); memset(readClus, 0, ((size_t)b->nRead + 1) * sizeof(int)); memset(badLink, 0, ((size_t)b->nHash + 1) * sizeof(int));
      int nClusHash = 0;
      for (uint32_t i = 0; i < b->nHash; ++i) {
        const int cl = e[i].subCluster; if (!cl) continue;
        ++nClusHash;
        if (e[i].read < b->nRead) readClus[e[i].read] = cl;
        if ((size_t)cl >= nInfo) continue;                                           /* stale label beyond nSubCluster: out of bounds in the reference */
        ReportInfo *r = &info[cl];
        ++r->n;
        const uint32_t bh = e[i].hash;
        if (haveCrib) {
          const int t = s->cribType[bh];
          ++r->nt[t];
          if (t > 0 && t < 4) {
            const int16_t hc = s->cribChr[bh]; const uint16_t hp = s->cribPos[bh];
            if (!r->chr) { r->chr = hc; r->pMin = r->pMax = hp; }
            else if (hc == r->chr) { if (hp < r->pMin) r->pMin = hp; if (hp > r->pMax) r->pMax = hp; }
            else { ++r->nBad; badLink[i] = r->bad; r->bad = (int)i; }
          }
        }
      }
      int nClusRead = 0;
      for (uint32_t i = 0; i < b->nRead; ++i) { const int cl = readClus[i]; if (cl) { if ((size_t)cl < nInfo) ++info[cl].nRead; ++nClusRead; } }
      fprintf(out, "  CLUSTER_SUMMARY %d nRead %d nHash %d nGoodHash %d nClusHash %d nClusRead %d nSubCluster %d\n",
              code, (int)b->nRead, (int)b->nHash, (int)nGood[code], nClusHash, nClusRead, (int)b->nSubCluster);
      for (uint32_t i = 1; i <= b->nSubCluster; ++i) {
        const ReportInfo *r = &info[i]; if (!r->n) continue;
        fprintf(out, "    CODE_CLUSTER %d %d : %d reads %d hashes", code, (int)i, r->nRead, r->n);
        if (haveCrib) {
          fprintf(out, " %d hom, %d htA, %d htB, %d mul, %d err", r->nt[3], r->nt[1], r->nt[2], r->nt[4], r->nt[0]);
          if (r->chr) fprintf(out, "  chr %d pos %d %d", (int)r->chr, (int)r->pMin, (int)r->pMax - (int)r->pMin + 1);
          if (r->nBad) {
            fprintf(out, "  OTHER %d", r->nBad);
            int x = r->bad, j = 10; char text[64];
            while (x && j--) { fprintf(out, " %s", crib_text(s, depth, e[x].hash, text)); x = badLink[x]; }
          }
        }
        fputc('\n', out);
      }
      totalGoodHash += nGood[code];
      totalPointToMin += b->pointToMin;
    }
    if (totalGoodHash) fprintf(out, "  MIN_POINT_DENSITY %.3f\n", totalPointToMin / totalGoodHash);
  }
done:
  free(blocks); free(ch); free(depth); free(nGood); free(info); free(readClus); free(badLink);
  return rc;
}

/* cribSummary (hash10x.c:1017-1061): per crib type, hash entries and distinct hashes in base barcodes and in the barcodes
   --clusterSplit made (clusterParent != 0) */
int h10x_session_cribSummary(h10x_session *s, FILE *out) {
  if (!s->ctx) return fail(s, "no hash state loaded: use readFQB or readHash first");
  uint32_t dim = 0;
  if (h10x_crib_sizes(s->ctx, &dim, 0)) { fprintf(stderr, "cribSummary requires crib\n"); return 0; }
  if (crib_fetch(s)) return -1;
  h10x_sizes z; if (h10x_get_sizes(s->ctx, &z)) return fail_ctx(s);
  h10x_block *blocks = (h10x_block *)calloc((size_t)z.nBlocks + 1, sizeof *blocks);
  h10x_clushash *ch = (h10x_clushash *)malloc(z.nClusHash ? z.nClusHash * 8 : 8);
  uint8_t *seen = (uint8_t *)calloc((size_t)z.hashNumber + 1, 1);                     /* bit 0: in a base barcode, bit 1: in a cluster barcode */
  if (!blocks || !ch || !seen) { free(blocks); free(ch); free(seen); return fail(s, "out of host memory for cribSummary"); }
  if (h10x_export(s->ctx, 0, 0, 0, blocks, ch)) { free(blocks); free(ch); free(seen); return fail_ctx(s); }
  fprintf(stderr, "made hash objects\n");
  unsigned long long countBase[5] = {0}, countCluster[5] = {0}; int distinctBase[5] = {0}, distinctCluster[5] = {0};
  int nBaseCode = 0, nSubClusterCode = 0; uint64_t off = 0;
  for (uint32_t i = 0; i < z.nBlocks; ++i) {
    const h10x_block *b = &blocks[i]; const int isCluster = b->clusterParent != 0;
    if (isCluster) ++nSubClusterCode; else ++nBaseCode;
    if (i == 0) continue;                                                            /* block 0 owns no hashes (hash10x.c:256) */
    for (uint32_t j = 0; j < b->nHash; ++j) {
      const uint32_t h = ch[off + j].hash; const int t = s->cribType[h]; const uint8_t bit = isCluster ? 2 : 1;
      if (isCluster) ++countCluster[t]; else ++countBase[t];
      if (!(seen[h] & bit)) { seen[h] |= bit; if (isCluster) ++distinctCluster[t]; else ++distinctBase[t]; }
    }
    off += b->nHash;
  }
  fprintf(out, "  %d base codes ", nBaseCode);
  for (int i = 0; i < 5; ++i) fprintf(out, " %s %llu %d %.1f", cribTypeName[i], countBase[i], distinctBase[i], countBase[i] / (double)distinctBase[i]);
  fprintf(out, "\n  %d cluster codes ", nSubClusterCode);
  for (int i = 0; i < 5; ++i) fprintf(out, " %s %llu %d %.1f", cribTypeName[i], countCluster[i], distinctCluster[i], countCluster[i] / (double)distinctCluster[i]);
  fputc('\n', out);
  free(blocks); free(ch); free(seen);
  return 0;
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
