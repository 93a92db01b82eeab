/* h10x_host.h — host side of the hash10x command surface, in C, above the C ABI of include/h10x.h.
 *
 * A session holds what the reference keeps in globals (hash10x.c:25-33, 85-104): the latched
 * parameters, the device context, and the Array bookkeeping (dim/max of hashDepth and
 * clusterBlocks, array.c:144-170) that decides bytes of the .hash file. Each function is one
 * command of the reference's argv loop (hash10x.c:1200-1269); hash10x_main.c is that loop.
 * All compute goes through libh10x_hip.so — there is no CPU path here.
 */
#ifndef H10X_HOST_H
#define H10X_HOST_H
#include <stdint.h>
#include <stdio.h>
#include "../../include/h10x.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct h10x_session h10x_session;

h10x_session *h10x_session_new(void);                 /* defaults of hash10x.c:1131-1137 */
void h10x_session_free(h10x_session *s);
/* -k -w -r -B -N -c -ct (hash10x.c:1174-1179,1239) plus "device" (HIP ordinal); latched until the
   next readFQB/readHash like the reference */
int  h10x_session_set(h10x_session *s, const char *name, int value);
int  h10x_session_get(const h10x_session *s, const char *name);
const char *h10x_session_error(const h10x_session *s);
h10x_ctx *h10x_session_ctx(h10x_session *s);

int  h10x_session_readFQB(h10x_session *s, const char *path);                         /* hash10x.c:1200-1205 */
int  h10x_session_begin(h10x_session *s);                 /* initialise() with the latched parameters (callers that use the C ABI directly) */
int  h10x_session_after_read(h10x_session *s);            /* Array dims as a finished --readFQB leaves them */
int  h10x_session_readFQB_mem(h10x_session *s, const uint32_t *records, uint64_t nRecords);
int  h10x_session_readFQB_dev(h10x_session *s, const uint32_t *devRecords, uint64_t nRecords);  /* records already in HBM */
int  h10x_session_readHash(h10x_session *s, const char *path);                        /* hash10x.c:1206-1211 */
int  h10x_session_writeHash(h10x_session *s, const char *path);                       /* hash10x.c:1212-1215 */
int  h10x_session_hashDepthRange(h10x_session *s, int min, int max);                  /* hash10x.c:1216-1219 */
int  h10x_session_cluster(h10x_session *s, int codeMin, int codeMax);                 /* hash10x.c:1241-1261 */
int  h10x_session_clusterSplit(h10x_session *s);                                      /* hash10x.c:1267 */

/* --hashStats / --codeStats (hash10x.c:351-402): the reference's histogram reports; the histograms are filled on the
   device (h10x_report_histogram), the text is the reference's, including its Array dim: hashDepth is histogrammed over
   arrayMax entries, index 0 included. f = NULL: take part without printing (ranks > 0 of a sharded session). */
int  h10x_session_hashStats(h10x_session *s, FILE *f);
int  h10x_session_codeStats(h10x_session *s, FILE *f);

/* --cribBuild <genome1.fa> <genome2.fa>, --clusterReport <codeMin> <codeMax>, --cribSummary (hash10x.c:470-521, 870-952,
   1017-1061): truth labels from two haplotype FASTAs (hashed and looked up on the device) and the reference's text
   reports. printTables = the --tables flag (CRIB_TABLE lines). The per-cluster figures come from the device
   (h10x_cluster_report); out = NULL: take part without printing (ranks > 0 of a sharded session). */
int  h10x_session_cribBuild(h10x_session *s, const char *fa1, const char *fa2, FILE *out, int printTables);
int  h10x_session_clusterReport(h10x_session *s, int codeMin, int codeMax, FILE *out);
/* after h10x_session_cluster(codeMin, codeMax): the reference's --verbose lines of those blocks (hash10x.c:827-834, 867) to out, its "too many clusters" notes
   (hash10x.c:813) to err; either may be null. Collective when sharded (rank 0 prints). */
int  h10x_session_clusterVerbose(h10x_session *s, int codeMin, int codeMax, FILE *out, FILE *err);
int  h10x_session_cribSummary(h10x_session *s, FILE *out);

/* --sortFQB <in.fqb> <out.fqb> (addition): the record sort the reference leaves to `bsort -k 4 -r 120` (README.md:26),
   on the device: records ordered by their first 4 bytes, stably */
int  h10x_session_sortFQB(h10x_session *s, const char *inPath, const char *outPath);

/* multi-GPU (include/h10x.h "multi-GPU"): one session per rank, each holding a contiguous barcode range of the sorted file
   (cut with h10x_host_partition / _partition_file; -N is applied by the launcher before cutting). Every command of a
   sharded session is collective: all ranks call it with the same arguments; the text commands print on the rank whose
   FILE* is not NULL (rank 0) and only take part on the others. --writeHash, --clusterSplit, --cribBuild and the reports
   work on the shards as they are (no gather): see h10x.h. _file streams this rank's records [first, first + n) of the
   file into HBM and applies the reference's chunk semantics (-c) over the whole file. shardGather turns rank 0 into a
   single-GPU session holding everything (for continuing on one GPU). */
int  h10x_session_shardReadFQB_mem(h10x_session *s, h10x_comm *comm, const uint32_t *shardRecords, uint64_t nRecords);
int  h10x_session_shardReadFQB_dev(h10x_session *s, h10x_comm *comm, const uint32_t *devShardRecords, uint64_t nRecords);
int  h10x_session_shardReadFQB_file(h10x_session *s, h10x_comm *comm, const char *path, uint64_t firstRecord, uint64_t nRecords);
int  h10x_session_shardReadHash(h10x_session *s, h10x_comm *comm, const char *path);      /* --readHash: every rank reads its cut of the blocks */
int  h10x_session_shardGather(h10x_session *s);

/* dimension the reference's Array reaches when elements are first touched in ascending order up to
   lastIndex, starting from initialDim (array.c:144-185) */
int  h10x_host_array_dim(int initialDim, int elemSize, int64_t lastIndex);
/* test hook: what the reference's HASH object (hash.c) counts after hashAdd(HASH_INT(key)) of these keys — the restatement behind --cribSummary's second figures */
int  h10x_host_refhash_count(const int32_t *keys, uint64_t n);
/* readFQB's chunk loop (hash10x.c:202-223) replayed on the barcode column: returns the number of
   records it would consume (honours -N), or -1 with "chunkSize too small" in err */
int64_t h10x_host_check_chunks(const uint32_t *records, uint64_t nRecords, int N, int chunkSize, char *err, int errlen);

/* contiguous barcode-range shards for nParts GPUs: cut[g] = first record of shard g, always on a
   barcode-run boundary, balanced by record count; cut[nParts] = nRecords. Returns 0. */
int  h10x_host_partition(const uint32_t *records, uint64_t nRecords, int nParts, uint64_t *cut);
/* the same cuts for the first nRecords records of a file, reading only the barcode words around each cut */
int  h10x_host_partition_file(const char *path, uint64_t nRecords, int nParts, uint64_t *cut, char *err, int errlen);

/* starts loading the library's device code for `device` on a thread of its own (once per process and device; --readFQB joins it before its first kernels) */
void h10x_host_warm_start(int device);
#ifdef __cplusplus
}
#endif
#endif
