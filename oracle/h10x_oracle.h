/* h10x_oracle.h — CPU restatement of hash10x's mosh-construction + clustering path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; hash10x_amd/ never does.
 *
 * PARITY PINNING: this restatement is checked byte-for-byte (canonical .hash)
 * against the real reference compiled from /root/reference into oracle/_ref/ (oracle/Makefile)
 * by tests/test_oracle_vs_reference.py, and against the committed fixtures in tests/golden/
 * that were generated from that binary (tests/golden/make_golden.py).
 *
 * Every function cites the reference file:line it restates.
 */
#ifndef H10X_ORACLE_H
#define H10X_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* on-disk / in-memory records, identical in layout to the reference's (hash10x.c:35-43, 62-70) */
typedef struct { uint32_t hash; uint16_t read; uint8_t subCluster; uint8_t flags; } orc_clushash;   /* 8 B  */
typedef struct { uint32_t nRead, nHash, nSubCluster, clusterParent; uint64_t clusHashPtr; double pointToMin; } orc_block; /* 32 B */

typedef struct orc_state orc_state;

/* seqhash.c:20-35 with srandom(seed) of hash10x.c:1101 — returns factor1 (glibc random()) */
uint64_t orc_factor1_from_seed(int seed);

/* hash10x.c:1099-1118 initialise(); returns NULL and fills err on a die() condition */
orc_state *orc_create(int k, int w, int seed, int B, char *err, int errlen);
void orc_destroy(orc_state *o);
const char *orc_last_error(const orc_state *o);

/* seqhash.c:154-195: moshes of s[0..len) (base codes 0..3); returns count, fills hash/pos (cap entries) */
int orc_mosh_sequence(const orc_state *o, const uint8_t *s, int len, uint64_t *hash, int *pos, int cap);
/* hash10x.c:108-119: 10 packed words -> 160 base codes */
void orc_unpack160(const uint32_t *u10, uint8_t *out160);

/* hash10x.c:188-236 readFQB over an in-memory file image + hash10x.c:317-347 fillHashTable.
   N = 0 for all records. Returns 0, or -1 with orc_last_error() = the reference's die() text. */
int orc_read_fqb(orc_state *o, const uint32_t *records, uint64_t nRecords, int N, int chunkSize);
/* hash10x.c:528-539 + 738-766 */
int orc_depth_range(orc_state *o, int min, int max);
/* hash10x.c:1241-1261 (+770-868); nThreads > 1 uses OpenMP over barcodes like -DOMP */
int orc_cluster(orc_state *o, int codeMin, int codeMax, int clusterThreshold, int nThreads);
/* hash10x.c:956-1013 */
int orc_cluster_split(orc_state *o);
/* hash10x.c:244-267 / 269-315 (+array.c:213-238); pointer fields written as zero (canonical form) */
int orc_write_hash(orc_state *o, const char *path);
int orc_read_hash(orc_state *o, const char *path);

/* accessors (borrowed pointers, valid until the next mutating call) */
uint32_t        orc_hash_number(const orc_state *o);
int             orc_table_bits(const orc_state *o);
const uint32_t *orc_hash_index(const orc_state *o);           /* 2^B entries */
const uint64_t *orc_hash_value(const orc_state *o);           /* hashNumber entries */
const uint32_t *orc_hash_depth(const orc_state *o, int *dim, int *max);
const orc_block *orc_blocks(const orc_state *o, int *dim, int *max);
const orc_clushash *orc_block_clushash(const orc_state *o, int code);
const uint32_t *orc_hash_codes(const orc_state *o, uint32_t index);   /* hashDepth[index] barcodes, ascending */
const uint16_t *orc_good_hashes(const orc_state *o, int code, int *n);
uint64_t        orc_sum_good_depth(const orc_state *o, int codeMin, int codeMax, uint64_t *sumGood, uint64_t *sumHash);

#ifdef __cplusplus
}
#endif
#endif
