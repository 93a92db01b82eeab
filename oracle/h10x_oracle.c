/* h10x_oracle.c — CPU restatement of hash10x's hot path (see h10x_oracle.h).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT: the checker for the HIP path. Plain serial C that follows the
 * reference's algorithms and complexity (serial stage A/B with a sequentially filled probe table,
 * dense O(G^2) clustering) so that it can also serve as the "port" CPU baseline on the GPU box,
 * where /root/reference does not exist. Written from SURVEY.md Appendix C against the reference's
 * behaviour; no reference source text is reproduced.
 *
 * PARITY PINNED by tests/test_oracle_vs_reference.py (oracle/_ref binaries, this container) and
 * tests/test_golden.py (committed fixtures generated from those binaries).
 */
#define _GNU_SOURCE
#include "h10x_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#ifdef _OPENMP
#include <omp.h>
#endif

struct orc_state {
  int k, w, seed, B;
  uint64_t factor1, mask, patternRC[4];
  int shift1;
  uint64_t tableSize, tableMask;
  uint32_t *hashIndex;            /* 2^B, 0 = empty                     (hash10x.c:91)  */
  uint64_t *hashValue;            /* 2^(B-2)                            (hash10x.c:92)  */
  uint32_t hashNumber;            /* next free index; starts at 1       (hash10x.c:1113) */
  uint32_t *depth; int depthDim, depthMax;           /* Array hashDepth (hash10x.c:94)  */
  orc_block *blocks; int blocksDim, blocksMax;       /* Array clusterBlocks (hash10x.c:96) */
  orc_clushash **clus;            /* clusHash of each block, parallel to blocks[] */
  uint64_t *rowOff; uint32_t *rows;                  /* hashCodes as CSR (hash10x.c:317-347) */
  uint8_t *within; int rangeMin, rangeMax;           /* hashWithinRange (hash10x.c:525-539) */
  uint16_t **good; int *nGood; int goodBlocks;       /* goodHashes (hash10x.c:722-766) */
  char err[512];
};

static int fail(orc_state *o, const char *fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(o->err, sizeof o->err, fmt, ap); va_end(ap);
  return -1;
}
const char *orc_last_error(const orc_state *o) { return o->err; }

/* ---------------------------------------------------------------- Array growth (array.c:144-185) */
/* dimension after the reference's uArray(a,i) touches element n of an array whose current dim is dim */
static int grown_dim(int dim, int size, int n) {
  if (n < dim) return dim;
  if (dim * size < (1 << 23)) dim *= 2; else dim += 1024 + ((1 << 23) / size);
  if (n >= dim) dim = n + 1;
  return dim;
}

static uint32_t *depth_at(orc_state *o, int i) {          /* array(hashDepth,i,U32) */
  if (i >= o->depthMax) {
    if (i >= o->depthDim) {
      int nd = grown_dim(o->depthDim, 4, i);
      uint32_t *p = (uint32_t *)calloc((size_t)nd, 4);
      memcpy(p, o->depth, (size_t)o->depthMax * 4);
      free(o->depth); o->depth = p; o->depthDim = nd;
    }
    o->depthMax = i + 1;
  }
  return &o->depth[i];
}

static orc_block *block_at(orc_state *o, int i) {         /* arrayp(clusterBlocks,i,ClusterBlock) */
  if (i >= o->blocksMax) {
    if (i >= o->blocksDim) {
      int nd = grown_dim(o->blocksDim, (int)sizeof(orc_block), i);
      orc_block *p = (orc_block *)calloc((size_t)nd, sizeof(orc_block));
      orc_clushash **c = (orc_clushash **)calloc((size_t)nd, sizeof(*c));
      memcpy(p, o->blocks, (size_t)o->blocksMax * sizeof(orc_block));
      memcpy(c, o->clus, (size_t)o->blocksMax * sizeof(*c));
      free(o->blocks); free(o->clus); o->blocks = p; o->clus = c; o->blocksDim = nd;
    }
    o->blocksMax = i + 1;
  }
  return &o->blocks[i];
}

/* ---------------------------------------------------------------- seqhash (seqhash.c:20-35,58-80) */
uint64_t orc_factor1_from_seed(int seed) {
  srandom((unsigned)seed);                                 /* hash10x.c:1101 */
  uint64_t hi = (uint64_t)random();                        /* seqhash.c:29: (random() << 32) | random() | 1 */
  uint64_t lo = (uint64_t)random();
  return (hi << 32) | lo | 1;
}

static void free_good(orc_state *o) {
  if (o->good) { for (int c = 0; c < o->goodBlocks; ++c) free(o->good[c]); free(o->good); free(o->nGood); }
  o->good = 0; o->nGood = 0; o->goodBlocks = 0;
}

orc_state *orc_create(int k, int w, int seed, int B, char *err, int errlen) {
  /* hash10x.c:1103,1107-1108 and seqhash.c:24-25 */
  if (k <= 0 || w <= 0) { if (err) snprintf(err, errlen, "k %d, w %d must be > 0; run without args for usage", k, w); return 0; }
  if (k >= 32) { if (err) snprintf(err, errlen, "seqhash k %d must be between 1 and 32\n", k); return 0; }
  if (B < 20 || B > 30) { if (err) snprintf(err, errlen, "hashTableBits %d out of range 20-30", B); return 0; }
  orc_state *o = (orc_state *)calloc(1, sizeof *o);
  o->k = k; o->w = w; o->seed = seed; o->B = B;
  o->factor1 = orc_factor1_from_seed(seed);
  o->mask = ((uint64_t)1 << (2 * k)) - 1;
  o->shift1 = 64 - 2 * k;
  for (int i = 0; i < 4; ++i) o->patternRC[i] = (uint64_t)(3 - i) << (2 * (k - 1));
  o->tableSize = (uint64_t)1 << B; o->tableMask = o->tableSize - 1;
  o->hashIndex = (uint32_t *)calloc(o->tableSize, 4);
  o->hashValue = (uint64_t *)calloc(o->tableSize >> 2, 8);
  o->hashNumber = 1;
  o->depthDim = 1 << 20; o->depth = (uint32_t *)calloc((size_t)o->depthDim, 4);   /* hash10x.c:1114 */
  o->blocksDim = 1200; o->blocks = (orc_block *)calloc(1200, sizeof(orc_block));     /* hash10x.c:1151 */
  o->clus = (orc_clushash **)calloc(1200, sizeof(*o->clus));
  return o;
}

static void free_blocks(orc_state *o) {
  if (o->clus) for (int i = 0; i < o->blocksMax; ++i) free(o->clus[i]);
  free(o->clus); free(o->blocks); o->clus = 0; o->blocks = 0; o->blocksMax = o->blocksDim = 0;
}

void orc_destroy(orc_state *o) {
  if (!o) return;
  free_good(o); free_blocks(o);
  free(o->hashIndex); free(o->hashValue); free(o->depth); free(o->rowOff); free(o->rows); free(o->within);
  free(o);
}

static inline uint64_t khash(const orc_state *o, uint64_t x) { return (x * o->factor1) >> o->shift1; }  /* seqhash.c:58-59 */

/* seqhash.c:154-195 — every k-mer of s is hashed on both strands; the smaller value is its
   canonical hash, and it is a mosh iff that value is divisible by w. pos = k-mer start. */
int orc_mosh_sequence(const orc_state *o, const uint8_t *s, int len, uint64_t *hash, int *pos, int cap) {
  if (len < o->k) return 0;
  uint64_t f = 0, r = 0; int n = 0;
  for (int i = 0; i < len; ++i) {
    f = ((f << 2) & o->mask) | s[i];                       /* seqhash.c:74 / :166 */
    r = (r >> 2) | o->patternRC[s[i]];                     /* seqhash.c:75 / :167 */
    if (i + 1 < o->k) continue;
    uint64_t hf = khash(o, f), hr = khash(o, r);
    uint64_t h = hf < hr ? hf : hr;                        /* seqhash.c:67-68 */
    if (h % (uint64_t)o->w == 0) {                         /* seqhash.c:171,189 */
      if (n < cap) { hash[n] = h; if (pos) pos[n] = i + 1 - o->k; }
      ++n;
    }
  }
  return n;
}

void orc_unpack160(const uint32_t *u, uint8_t *out) {      /* hash10x.c:112 */
  for (int i = 0; i < 10; ++i) for (int j = 15; j >= 0; --j) *out++ = (u[i] >> (2 * j)) & 3;
}

/* ---------------------------------------------------------------- hash -> index (hash10x.c:139-152) */
static uint32_t index_find(orc_state *o, uint64_t hash, int add, int *tooSmall) {
  uint64_t slot = hash & o->tableMask;
  uint64_t step = ((hash >> o->B) & o->tableMask) | 1;
  uint32_t ix;
  while ((ix = o->hashIndex[slot]) && o->hashValue[ix] != hash) slot = (slot + step) & o->tableMask;
  if (!ix && add) {
    ix = o->hashIndex[slot] = o->hashNumber++;
    o->hashValue[ix] = hash;
    if (o->hashNumber > (o->tableSize >> 2) - 2) *tooSmall = 1;
  }
  return ix;
}

/* ---------------------------------------------------------------- processBlock (hash10x.c:154-186) */
typedef struct { uint64_t hash; int read; int seq; } tmp_hash;
static int cmp_tmp(const void *a, const void *b) {         /* stable: glibc qsort is a merge sort */
  const tmp_hash *x = (const tmp_hash *)a, *y = (const tmp_hash *)b;
  if (x->hash != y->hash) return x->hash < y->hash ? -1 : 1;
  return x->seq < y->seq ? -1 : x->seq > y->seq;
}
static int cmp_clus_index(const void *a, const void *b) {
  uint32_t x = ((const orc_clushash *)a)->hash, y = ((const orc_clushash *)b)->hash;
  return x < y ? -1 : x > y;
}

static int process_block(orc_state *o, const uint32_t *rec, int code) {
  orc_block *b = &o->blocks[code];
  size_t cap = 4096, n = 0;
  tmp_hash *t = (tmp_hash *)malloc(cap * sizeof *t);
  uint8_t s1[160], s2[160]; uint64_t hs[160];
  for (uint32_t i = 0; i < b->nRead; ++i, rec += 30) {
    orc_unpack160(rec, s1); orc_unpack160(rec + 15, s2);
    for (int part = 0; part < 2; ++part) {                 /* hash10x.c:162-163 */
      int m = part ? orc_mosh_sequence(o, s2, 150, hs, 0, 160) : orc_mosh_sequence(o, s1 + 23, 127, hs, 0, 160);
      for (int j = 0; j < m; ++j) {
        if (n == cap) { cap *= 2; t = (tmp_hash *)realloc(t, cap * sizeof *t); }
        t[n].hash = hs[j]; t[n].read = (int)i; t[n].seq = (int)n; ++n;
      }
    }
  }
  qsort(t, n, sizeof *t, cmp_tmp);
  /* hash10x.c:167-172: the reference reads element 0 of a calloc'd array even when it is empty, so a
     block without moshes still yields one entry {hash 0, read 0} */
  size_t u = 1;
  if (n == 0) { t[0].hash = 0; t[0].read = 0; }
  uint64_t lastHash = t[0].hash;                           /* keep the first (= lowest read) of each run */
  for (size_t j = 1; j < n; ++j)
    if (t[j].hash != lastHash) { lastHash = t[j].hash; t[u++] = t[j]; }
  int tooSmall = 0;
  b->nHash = (uint32_t)u;
  orc_clushash *c = (orc_clushash *)calloc(u, sizeof *c);   /* zeroed: SURVEY F4 */
  for (size_t i = 0; i < u; ++i) {
    uint32_t ix = index_find(o, t[i].hash, 1, &tooSmall);
    if (tooSmall) { free(t); free(c); return fail(o, "hashTableSize is too small"); }
    ++*depth_at(o, (int)ix);
    c[i].hash = ix; c[i].read = (uint16_t)t[i].read;
  }
  qsort(c, u, sizeof *c, cmp_clus_index);
  o->clus[code] = c;
  free(t);
  return 0;
}

/* ---------------------------------------------------------------- fillHashTable (hash10x.c:317-347) */
static void fill_table(orc_state *o) {
  free(o->rowOff); free(o->rows);
  uint32_t U = o->hashNumber;
  o->rowOff = (uint64_t *)calloc((size_t)U + 1, 8);
  for (uint32_t i = 1; i < U; ++i) o->rowOff[i + 1] = o->rowOff[i] + ((int)i < o->depthMax ? o->depth[i] : 0);
  o->rows = (uint32_t *)malloc((o->rowOff[U] + 1) * 4);
  uint64_t *fillp = (uint64_t *)malloc(((size_t)U + 1) * 8);
  memcpy(fillp, o->rowOff, ((size_t)U + 1) * 8);
  for (int code = 1; code < o->blocksMax; ++code) {
    const orc_block *b = &o->blocks[code]; const orc_clushash *c = o->clus[code];
    for (uint32_t j = 0; j < b->nHash; ++j) o->rows[fillp[c[j].hash]++] = (uint32_t)code;
  }
  free(fillp);
}

/* ---------------------------------------------------------------- readFQB (hash10x.c:188-236) */
int orc_read_fqb(orc_state *o, const uint32_t *recs, uint64_t total, int N, int chunkSize) {
  int nReads = 0; uint64_t pos = 0, blockStart = 0;
  uint32_t barcode = 0;
  int cur = 1;
  block_at(o, 1)->nRead = 0;
  while (!N || nReads < N) {
    int thisChunk = chunkSize - (int)o->blocks[cur].nRead;
    if (thisChunk <= 0) return fail(o, "chunkSize too small");
    if (N && nReads + thisChunk > N) thisChunk = N - nReads;
    uint64_t avail = total - pos;
    int nRec = avail < (uint64_t)thisChunk ? (int)avail : thisChunk;
    if (!nRec) break;
    const uint32_t *u = recs + 30 * pos;
    if (!barcode) barcode = u[0];                          /* hash10x.c:212 */
    for (int i = 0; i < nRec; ++i) {
      if (u[30 * i] == barcode) ++o->blocks[cur].nRead;
      else {
        if (process_block(o, recs + 30 * blockStart, cur)) return -1;
        cur = o->blocksMax; block_at(o, cur)->nRead = 1;
        barcode = u[30 * i]; blockStart = pos + (uint64_t)i;
      }
    }
    nReads += nRec; pos += (uint64_t)nRec;
  }
  /* the block open at EOF is never hashed */
  fill_table(o);
  return 0;
}

/* ---------------------------------------------------------------- depth range + good hashes */
typedef struct { uint32_t depth; uint16_t pos; } good_key;
static int cmp_good(const void *a, const void *b) {        /* hash10x.c:726-730, stable => ties by position */
  const good_key *x = (const good_key *)a, *y = (const good_key *)b;
  if (x->depth != y->depth) return x->depth < y->depth ? -1 : 1;
  return x->pos < y->pos ? -1 : x->pos > y->pos;
}

int orc_depth_range(orc_state *o, int min, int max) {
  /* hash10x.c:528-539: flags are only ever set, so ranges accumulate; identical repeat is a no-op */
  if (!(o->within && min == o->rangeMin && max == o->rangeMax)) {
    if (!o->within) o->within = (uint8_t *)calloc(o->hashNumber, 1);
    for (uint32_t i = 0; i < o->hashNumber; ++i) {
      int n = (int)i < o->depthDim ? (int)o->depth[i] : 0;
      if (n >= min && n < max) o->within[i] = 1;
    }
    o->rangeMin = min; o->rangeMax = max;
  }
  /* hash10x.c:738-766 */
  free_good(o);
  o->goodBlocks = o->blocksMax;
  o->good = (uint16_t **)calloc((size_t)o->blocksMax, sizeof *o->good);
  o->nGood = (int *)calloc((size_t)o->blocksMax, sizeof(int));
  for (int c = 0; c < o->blocksMax; ++c) {
    const orc_block *b = &o->blocks[c]; const orc_clushash *ch = o->clus[c];
    if (b->nHash > 65535) { o->nGood[c] = 0; o->good[c] = (uint16_t *)calloc(1, 2); continue; }
    good_key *g = (good_key *)malloc(((size_t)b->nHash + 1) * sizeof *g); int n = 0;
    for (uint32_t i = 0; i < b->nHash; ++i)
      if (o->within[ch[i].hash]) { g[n].depth = o->depth[ch[i].hash]; g[n].pos = (uint16_t)i; ++n; }
    qsort(g, (size_t)n, sizeof *g, cmp_good);
    o->good[c] = (uint16_t *)malloc(((size_t)n + 1) * 2);
    for (int i = 0; i < n; ++i) o->good[c][i] = g[i].pos;
    o->nGood[c] = n;
    free(g);
  }
  return 0;
}

/* ---------------------------------------------------------------- codeClusterFind (hash10x.c:770-835) */
static void cluster_find(orc_state *o, int code, int threshold) {
  orc_block *b = &o->blocks[code]; orc_clushash *ch = o->clus[code];
  int *firstShare = (int *)calloc((size_t)o->blocksMax, sizeof(int));   /* 1 + first good-hash rank sharing each barcode */
  int n = o->nGood[code];
  if (!n) { free(firstShare); return; }
  const uint16_t *g = o->good[code];
  int *cnt = (int *)malloc((size_t)n * sizeof(int));
  int clusterMin[257];
  for (int i = 0; i < n; ++i) ch[g[i]].subCluster = 0;
  b->nSubCluster = 0; b->pointToMin = 0.0;
  for (int i = 1; i < n; ++i) {                            /* i = 0 is never processed (hash10x.c:789) */
    uint32_t x = ch[g[i]].hash;
    memset(cnt, 0, (size_t)n * sizeof(int));
    const uint32_t *row = o->rows + o->rowOff[x]; uint32_t d = o->depth[x];
    for (uint32_t j = 0; j < d; ++j) {
      int cj = (int)row[j];
      if (cj == code) continue;
      if (!firstShare[cj]) firstShare[cj] = i + 1;
      ++cnt[firstShare[cj] - 1];
    }
    int best = 0, bestCnt = 0, tot = 0;
    for (int j = 0; j < i; ++j) { if (cnt[j] > bestCnt) { best = j; bestCnt = cnt[j]; } tot += cnt[j]; }
    if (bestCnt >= threshold) {
      orc_clushash *m = &ch[g[best]];
      if (!m->subCluster) {
        if (++b->nSubCluster > 255) {                      /* hash10x.c:810-816: abandon, partial pointToMin kept */
          b->nSubCluster = 0;
          for (int j = 0; j < i; ++j) ch[g[j]].subCluster = 0;
          break;
        }
        m->subCluster = (uint8_t)b->nSubCluster;
        clusterMin[b->nSubCluster] = best;
      }
      ch[g[i]].subCluster = m->subCluster;
      b->pointToMin += cnt[clusterMin[m->subCluster]] / (double)tot;
    }
  }
  free(firstShare); free(cnt);
}

/* ---------------------------------------------------------------- codeClusterReadMerge (hash10x.c:837-868) */
static void read_merge(orc_state *o, int code) {
  orc_block *b = &o->blocks[code]; orc_clushash *ch = o->clus[code];
  if (!b->nSubCluster) return;
  int ns = (int)b->nSubCluster;
  int *readCluster = (int *)calloc((size_t)b->nRead + 1, sizeof(int));
  int rep[257], dead[257];
  for (int i = 0; i <= ns; ++i) { rep[i] = i; dead[i] = 0; }
  for (uint32_t i = 0; i < b->nHash; ++i) {
    int hc = rep[ch[i].subCluster]; if (!hc) continue;
    int rc = rep[readCluster[ch[i].read]];
    if (hc == rc) continue;
    if (!rc) readCluster[ch[i].read] = hc;
    else {
      int lo = hc < rc ? hc : rc, hi = hc < rc ? rc : hc;
      for (int j = 1; j <= ns; ++j) if (rep[j] == hi) rep[j] = lo;
      dead[hi] = 1;
    }
  }
  int alive[257]; alive[0] = 0;
  for (int j = 1; j <= ns; ++j) alive[j] = alive[j - 1] + 1 - dead[j];
  for (int j = 1; j <= ns; ++j) rep[j] = alive[rep[j]];
  b->nSubCluster = (uint32_t)alive[ns];
  for (uint32_t i = 0; i < b->nHash; ++i) ch[i].subCluster = (uint8_t)rep[ch[i].subCluster];
  free(readCluster);
}

int orc_cluster(orc_state *o, int codeMin, int codeMax, int threshold, int nThreads) {
  if (!codeMin) codeMin = 1;                               /* hash10x.c:1243-1244 */
  if (!codeMax) codeMax = o->blocksMax;
  if (!o->good) return fail(o, "!! you must set hashDepthRange before cluster");
  (void)nThreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nThreads > 0 ? nThreads : 1)
#endif
  for (int code = codeMin; code < codeMax; ++code) { cluster_find(o, code, threshold); read_merge(o, code); }
  return 0;
}

/* ---------------------------------------------------------------- clusterSplitCodes (hash10x.c:956-1013) */
int orc_cluster_split(orc_state *o) {
  int nCodes = o->blocksMax, nSub = 0;
  for (int i = 0; i < nCodes; ++i) nSub += (int)o->blocks[i].nSubCluster;
  int nNew = nCodes + nSub;
  if (nNew <= 0) return fail(o, "clusterSplit: no blocks");
  orc_block *nb = (orc_block *)calloc((size_t)nNew, sizeof *nb);
  orc_clushash **nc = (orc_clushash **)calloc((size_t)nNew, sizeof *nc);
  int keep = 0, ext = nCodes - 1;                          /* new cluster j of a parent lands at ext + j */
  for (int i = 0; i < nCodes; ++i) {
    orc_block *old = &o->blocks[i]; orc_clushash *oc = o->clus[i];
    if (!old->nSubCluster) { nb[keep] = *old; nc[keep] = oc; o->clus[i] = 0; ++keep; continue; }
    int ns = (int)old->nSubCluster;
    int *count = (int *)calloc((size_t)ns + 1, sizeof(int));
    for (uint32_t j = 0; j < old->nHash; ++j) ++count[oc[j].subCluster];
    nc[keep] = (orc_clushash *)calloc((size_t)count[0] + 1, sizeof(orc_clushash));
    for (int j = 1; j <= ns; ++j) {
      nc[ext + j] = (orc_clushash *)calloc((size_t)count[j] + 1, sizeof(orc_clushash));
      nb[ext + j].clusterParent = (uint32_t)i + 1;
    }
    int *readMap = (int *)calloc((size_t)old->nRead + 1, sizeof(int));
    for (uint32_t j = 0; j < old->nHash; ++j) {
      orc_clushash e = oc[j]; int cl = e.subCluster; e.subCluster = 0;
      if (cl) {
        if (!readMap[e.read]) readMap[e.read] = (int)++nb[ext + cl].nRead;
        e.read = (uint16_t)(readMap[e.read] - 1);
        nc[ext + cl][nb[ext + cl].nHash++] = e;
      } else nc[keep][nb[keep].nHash++] = e;
    }
    nb[keep].nRead = old->nRead;
    ++keep; ext += ns;
    free(count); free(readMap); free(oc); o->clus[i] = 0;
  }
  free(o->blocks); free(o->clus);
  o->blocks = nb; o->clus = nc; o->blocksMax = o->blocksDim = nNew;   /* arrayCreate(n): dim = max = n */
  fill_table(o);
  return 0;
}

/* ---------------------------------------------------------------- .hash I/O (hash10x.c:240-315, array.c:213-238) */
typedef struct { int32_t magic, pad0; uint64_t base; int32_t dim, size, max, pad1; } array_hdr;   /* array.h:41-50 */
#define ARRAY_MAGIC 8918274

int orc_write_hash(orc_state *o, const char *path) {
  FILE *f = fopen(path, "wb"); if (!f) return fail(o, "failed to open hash file %s", path);
  uint32_t version = 2; uint16_t chs = 8, cbs = 32; int32_t B = o->B;
  int ok = fwrite("10XH", 4, 1, f) == 1 && fwrite(&version, 4, 1, f) == 1 && fwrite(&chs, 2, 1, f) == 1 &&
           fwrite(&cbs, 2, 1, f) == 1 && fwrite(&B, 4, 1, f) == 1;
  ok = ok && fwrite(o->hashIndex, 4, o->tableSize, f) == o->tableSize;
  ok = ok && fwrite(&o->hashNumber, 4, 1, f) == 1 && fwrite(o->hashValue, 8, o->hashNumber, f) == o->hashNumber;
  array_hdr h = {ARRAY_MAGIC, 0, 0, o->depthDim, 4, o->depthMax, 0};
  ok = ok && fwrite(&h, 32, 1, f) == 1 && fwrite(o->depth, 4, (size_t)o->depthDim, f) == (size_t)o->depthDim;
  array_hdr hb = {ARRAY_MAGIC, 0, 0, o->blocksDim, 32, o->blocksMax, 0};
  ok = ok && fwrite(&hb, 32, 1, f) == 1;
  for (int i = 0; ok && i < o->blocksDim; ++i) {
    orc_block b = o->blocks[i]; b.clusHashPtr = 0;         /* canonical form: SURVEY App. B.1 */
    ok = fwrite(&b, 32, 1, f) == 1;
  }
  for (int i = 1; ok && i < o->blocksMax; ++i)
    if (o->blocks[i].nHash) ok = fwrite(o->clus[i], 8, o->blocks[i].nHash, f) == o->blocks[i].nHash;
  fclose(f);
  return ok ? 0 : fail(o, "write fail");
}

int orc_read_hash(orc_state *o, const char *path) {
  FILE *f = fopen(path, "rb"); if (!f) return fail(o, "failed to open hash file %s", path);
  char name[5] = {0}; uint32_t version; uint16_t chs, cbs; int32_t B;
  if (fread(name, 4, 1, f) != 1 || fread(&version, 4, 1, f) != 1 || fread(&chs, 2, 1, f) != 1 || fread(&cbs, 2, 1, f) != 1)
    { fclose(f); return fail(o, "read fail 0"); }
  if (strcmp(name, "10XH")) { fclose(f); return fail(o, "not a 10X hash file"); }
  if (version > 2) { fclose(f); return fail(o, "hash file version mismatch: file %d > code %d", version, 2); }
  if (chs != 8) { fclose(f); return fail(o, "ClusterHash structure size mismatch: file %d != code %d", chs, 8); }
  if (cbs != 32) { fclose(f); return fail(o, "ClusterBlock structure size mismatch: file %d != code %d", cbs, 32); }
  if (fread(&B, 4, 1, f) != 1) { fclose(f); return fail(o, "read fail 1"); }
  if (B != o->B) { fclose(f); return fail(o, "incompatible hash table size: rerun with -B %d", B); }
  if (fread(o->hashIndex, 4, o->tableSize, f) != o->tableSize) { fclose(f); return fail(o, "read fail 2"); }
  array_hdr h;
  if (version == 1) {                                      /* hash10x.c:286-292 */
    if (fread(&h, 32, 1, f) != 1) { fclose(f); return fail(o, "failed to read hashValue array"); }
    uint64_t *tmp = (uint64_t *)malloc((size_t)h.dim * 8 + 8);
    if (fread(tmp, 8, (size_t)h.dim, f) != (size_t)h.dim) { free(tmp); fclose(f); return fail(o, "failed to read hashValue array"); }
    o->hashNumber = (uint32_t)h.max; memcpy(o->hashValue, tmp, (size_t)h.max * 8); free(tmp);
  } else {
    if (fread(&o->hashNumber, 4, 1, f) != 1) { fclose(f); return fail(o, "failed to read hashNumber"); }
    if (fread(o->hashValue, 8, o->hashNumber, f) != o->hashNumber) { fclose(f); return fail(o, "failed to read hashValue"); }
  }
  if (fread(&h, 32, 1, f) != 1) { fclose(f); return fail(o, "failed to read hashDepth array"); }
  free(o->depth); o->depth = (uint32_t *)calloc((size_t)h.dim + 1, 4); o->depthDim = h.dim; o->depthMax = h.max;
  if (fread(o->depth, 4, (size_t)h.dim, f) != (size_t)h.dim) { fclose(f); return fail(o, "failed to read hashDepth array"); }
  if (fread(&h, 32, 1, f) != 1) { fclose(f); return fail(o, "failed to read clusterBlocks array"); }
  free_blocks(o);
  o->blocks = (orc_block *)calloc((size_t)h.dim + 1, sizeof(orc_block));
  o->clus = (orc_clushash **)calloc((size_t)h.dim + 1, sizeof *o->clus);
  o->blocksDim = h.dim; o->blocksMax = h.max;
  if (fread(o->blocks, 32, (size_t)h.dim, f) != (size_t)h.dim) { fclose(f); return fail(o, "failed to read clusterBlocks array"); }
  for (int i = 1; i < o->blocksMax; ++i) {
    orc_block *b = &o->blocks[i]; b->clusHashPtr = 0;
    o->clus[i] = (orc_clushash *)calloc((size_t)b->nHash + 1, 8);
    if (fread(o->clus[i], 8, b->nHash, f) != b->nHash) { fclose(f); return fail(o, "read fail 3"); }
  }
  fclose(f);
  free(o->within); o->within = 0; free_good(o);
  fill_table(o);
  return 0;
}

/* ---------------------------------------------------------------- accessors */
uint32_t orc_hash_number(const orc_state *o) { return o->hashNumber; }
int orc_table_bits(const orc_state *o) { return o->B; }
const uint32_t *orc_hash_index(const orc_state *o) { return o->hashIndex; }
const uint64_t *orc_hash_value(const orc_state *o) { return o->hashValue; }
const uint32_t *orc_hash_depth(const orc_state *o, int *dim, int *max) { if (dim) *dim = o->depthDim; if (max) *max = o->depthMax; return o->depth; }
const orc_block *orc_blocks(const orc_state *o, int *dim, int *max) { if (dim) *dim = o->blocksDim; if (max) *max = o->blocksMax; return o->blocks; }
const orc_clushash *orc_block_clushash(const orc_state *o, int code) { return (code >= 0 && code < o->blocksMax) ? o->clus[code] : 0; }
const uint32_t *orc_hash_codes(const orc_state *o, uint32_t index) { return o->rows + o->rowOff[index]; }
const uint16_t *orc_good_hashes(const orc_state *o, int code, int *n) {
  if (!o->good || code < 0 || code >= o->goodBlocks) { if (n) *n = 0; return 0; }
  if (n) *n = o->nGood[code];
  return o->good[code];
}
/* Σ_c Σ_{i in good(c)} depth(x_i)  — the gather volume of clustering */
uint64_t orc_sum_good_depth(const orc_state *o, int codeMin, int codeMax, uint64_t *sumGood, uint64_t *sumHash) {
  if (!codeMin) codeMin = 1;
  if (!codeMax) codeMax = o->blocksMax;
  uint64_t s = 0, g = 0, h = 0;
  if (o->good) for (int c = codeMin; c < codeMax; ++c) {
    g += (uint64_t)o->nGood[c]; if (o->nGood[c]) h += o->blocks[c].nHash;
    for (int i = 0; i < o->nGood[c]; ++i) s += o->depth[o->clus[c][o->good[c][i]].hash];
  }
  if (sumGood) *sumGood = g;
  if (sumHash) *sumHash = h;
  return s;
}
