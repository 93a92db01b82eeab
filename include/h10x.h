/* h10x.h — C ABI of libh10x_hip.so: the MI355X (gfx950) implementation of hash10x's
 * mosh-construction + per-barcode clustering path.
 *
 * The reference (richarddurbin/hash10x) has no library/FFI boundary: its seams are global-state C
 * functions called from main()'s argv loop (hash10x.c:1158-1279). Each entry point below replaces
 * one of those seams and cites it. A maintainer of the reference would call these from the same
 * places in hash10x.c (see INTEGRATION.md for the exact stub); our own C host program
 * (hash10x_amd/host/hash10x_main.c) does exactly that behind the reference's command surface.
 *
 * Conventions: every function returns 0 on success, non-zero on failure; h10x_last_error() then
 * returns the message — the reference's own die() text (utils.c:18-29) where the reference would
 * have died (e.g. "hashTableSize is too small"). Plain pointers and sizes only; host pointers are
 * caller-owned; device memory is owned by the context. One host thread per context (the reference
 * is single-threaded at this boundary; its OMP loop lives inside --cluster, as our kernels do).
 * There is NO CPU fallback: without a HIP device h10x_create() fails.
 */
#ifndef H10X_H
#define H10X_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): h10x_warm, h10x_alloc_stats, h10x_pinned_*, h10x_ingest_fqb_async / _wait, H10X_TABLE_CLUSTER_RAW, the exchange counters of
   h10x_counters; option "cluster_dbg_skip" gone. libh10x_host.so and the Python loader refuse a libh10x_hip.so of another version. */
#define H10X_ABI_VERSION 3

typedef struct h10x_ctx h10x_ctx;

/* hasher + table parameters latched by -k -w -r -B (hash10x.c:1131-1134, 1174-1177) */
typedef struct {
  int32_t  k;          /* k-mer length, 1..31                      (seqhash.c:24)            */
  int32_t  w;          /* mosh modulus, >= 1                       (seqhash.c:25) */
  int32_t  B;          /* hash index table bits, 20..30            (hash10x.c:1107-1108)     */
  int32_t  reserved;
  uint64_t factor1;    /* multiplier of hashFunc (seqhash.c:29,58-59): h10x_factor1_from_seed(r) */
} h10x_params;

/* == ClusterBlock (hash10x.c:62-70), 32 bytes; clusHash is a heap pointer in the reference and is
   always written as 0 by this library (canonical form) */
typedef struct {
  uint32_t nRead, nHash, nSubCluster, clusterParent;
  uint64_t clusHash;
  double   pointToMin;
} h10x_block;

/* == ClusterHash (hash10x.c:35-43), 8 bytes */
typedef struct {
  uint32_t hash;        /* hash INDEX (1-based, first-appearance order) */
  uint16_t read;
  uint8_t  subCluster;
  uint8_t  flags;       /* isHet:1 isHom:1 isErr:1 isSure:1 — never set on this path */
} h10x_clushash;

typedef struct {
  int32_t  B;
  uint32_t hashNumber;  /* number of distinct hashes + 1 (index 0 unused)   (hash10x.c:90)  */
  uint32_t nBlocks;     /* arrayMax(clusterBlocks) = barcodes + 1           (hash10x.c:96)  */
  uint32_t reserved;
  uint64_t nClusHash;   /* sum of nHash over blocks                                          */
  uint64_t nRecords;    /* read pairs consumed by the last read_fqb                          */
} h10x_sizes;

/* srandom(seed); (random() << 32) | random() | 1 — glibc, as seqhashCreate draws it after
   initialise() seeds the generator (hash10x.c:1101, seqhash.c:29). Host-side helper. */
uint64_t h10x_factor1_from_seed(int32_t seed);

int  h10x_abi_version(void);
/* "src:<first 16 hex digits of the sha256 over the .hip and .hpp files of csrc (sorted by name) and this header>": what this library
   was built from (measurements record it; tests compare it with the sources of the tree they run in) */
const char *h10x_build_id(void);
/* number of HIP devices visible (0 if none / no driver); never initialises a context */
int  h10x_device_count(void);

/* replaces initialise() (hash10x.c:1099-1118). device = HIP ordinal; stream = hipStream_t to launch
   on, or NULL for the device's default stream. Fails (no fallback) if the device is unusable or the
   parameters would make the reference die. err/errlen receive the message when *ctx stays NULL. */
int  h10x_create(h10x_ctx **ctx, const h10x_params *p, int device, void *stream, char *err, int errlen);
void h10x_destroy(h10x_ctx *ctx);
const char *h10x_last_error(const h10x_ctx *ctx);

/* replaces readFQB() + fillHashTable() (hash10x.c:188-236, 317-347) for a whole sorted .fqb image:
   n_records records of 30 uint32 (fq2b.c:142-160). Barcode blocks are runs of equal word 0; the
   block still open at the end is kept with nHash = 0 (hash10x.c:209). The caller applies
   -N (pass only the first N records). The side effects of the reference's chunked fread loop (hash10x.c:202-223:
   die("chunkSize too small") for a barcode of chunkSize or more pairs, and the all-A barcode swallowing the next run
   when its own run ends at a chunk boundary, hash10x.c:212) are replayed from the barcode run starts when
   h10x_set_option(ctx, "chunk_size", c) was called with c > 0 (the session layer passes -c); 0 = no chunk semantics.
   _device: records already resident in device memory (HBM). */
/* Both return once the last launches are queued (every size the host needs has been read back by then): the next call on the
   context waits for them, and a device fault in them is reported there. */
int  h10x_read_fqb(h10x_ctx *ctx, const uint32_t *host_records, uint64_t n_records);
/* The same in chunks, as the reference's loop reads the file (hash10x.c:202-223: fread of chunkSize records at a time): every call
   appends n_records records to the context's record image ON THE DEVICE (the chunk buffer is the caller's again when the call returns;
   barcode runs may straddle chunks in any way, the runs are found over the whole image); the call with final_chunk != 0 (n_records may
   be 0) hashes the image exactly as h10x_read_fqb_device and releases it. The host never holds more than one chunk. h10x_ingest_reserve
   (optional, before the first chunk) announces the total so that the image is allocated once instead of growing geometrically.
   The chunk semantics of "chunk_size" apply to the whole image as above. A failed call drops the image; h10x_ingest_reserve(ctx, 0)
   gives up an ingest that will not be closed. */
int  h10x_ingest_reserve(h10x_ctx *ctx, uint64_t n_records_total);
int  h10x_ingest_fqb(h10x_ctx *ctx, const uint32_t *host_records, uint64_t n_records, int final_chunk);
/* The same as a pipeline (round 4): the chunk lies in page-locked memory from h10x_pinned_alloc and its upload is only QUEUED — the call returns
   while the DMA runs, so the caller reads the next chunk of the file into another buffer meanwhile (hash10x.c:202-209's fread loop, with the file
   read and the PCIe transfer side by side). The buffer must not be touched until h10x_ingest_wait(ctx, slot) has returned for the slot (0..7)
   the chunk was queued under. The ingest is closed as before, by h10x_ingest_fqb / h10x_shard_ingest_fqb with final_chunk = 1 (which waits for
   every queued upload). */
/* Loads the library's device code on `device` ahead of its first use (HIP loads a code object when the first of its kernels is launched — some tens
   of milliseconds for this library, otherwise spent inside the first command). Thread-safe; meant to be called from a thread of its own while the
   caller opens and reads its input. 0 on success. No counterpart in the reference (a CPU program has no such step). */
int  h10x_warm(int device);
/* Device blocks this process has obtained from hipMalloc so far (count, bytes): the library keeps freed blocks and hands them out again, so a
   repeated command on one context should add nothing here — a measurement hook (tests/test_gpu_parity.py), no counterpart in the reference. */
void h10x_alloc_stats(uint64_t *calls, uint64_t *bytes);
void *h10x_pinned_alloc(size_t bytes);
void  h10x_pinned_free(void *p);
int  h10x_ingest_fqb_async(h10x_ctx *ctx, const uint32_t *pinned_records, uint64_t n_records, int slot);
int  h10x_ingest_wait(h10x_ctx *ctx, int slot);
int  h10x_read_fqb_device(h10x_ctx *ctx, const uint32_t *dev_records, uint64_t n_records);

/* replaces the external record sort between fq2b and hash10x (README.md:26 `bsort -k 4 -r 120 x.fqb`): orders the
   120-byte records by their first 4 bytes (byte 0 most significant), stably, so that equal barcodes are contiguous.
   in / out must not overlap. _device: both buffers in device memory. */
int  h10x_sort_fqb(h10x_ctx *ctx, const uint32_t *host_in, uint64_t n_records, uint32_t *host_out);
int  h10x_sort_fqb_device(h10x_ctx *ctx, const uint32_t *dev_in, uint64_t n_records, uint32_t *dev_out);

/* replaces the state that readHashFile() + fillHashTable() leave behind (hash10x.c:269-315,
   317-347): uploads the tables of a parsed .hash file and rebuilds the hash->barcode lists.
   hashDepth has hashNumber entries, blocks has nBlocks entries (entry 0 unused), clusHash is the
   concatenation of blocks 1..nBlocks-1. */
int  h10x_load_state(h10x_ctx *ctx, const uint32_t *hashIndex, uint32_t hashNumber,
                     const uint64_t *hashValue, const uint32_t *hashDepth,
                     const h10x_block *blocks, uint32_t nBlocks, const h10x_clushash *clusHash);

/* replaces hashWithinRangeBuild() + goodHashesBuild() (hash10x.c:528-539, 738-766); ranges
   accumulate over calls exactly as in the reference. Returns once the work is queued on the context's stream (it has
   no host-side result); the next call on the context — h10x_cluster, an export — waits for it, and a device fault in
   it is reported there. */
int  h10x_depth_range(h10x_ctx *ctx, int32_t min, int32_t max);

/* replaces the --cluster loop: codeClusterFind() + codeClusterReadMerge() for code in
   [codeMin, codeMax) (hash10x.c:1241-1261, 770-868); 0,0 => 1..nBlocks. Fails with the reference's
   "!! you must set hashDepthRange before cluster" if no range was set. */
int  h10x_cluster(h10x_ctx *ctx, int32_t codeMin, int32_t codeMax, int32_t clusterThreshold);

/* replaces clusterSplitCodes() (hash10x.c:956-1013) */
int  h10x_cluster_split(h10x_ctx *ctx);

/* ---- crib: truth labels from two haplotype genomes (csrc/stage_d.hip) ----
   h10x_crib_genome replaces cribAddGenome() (hash10x.c:426-453) for one genome: `codes` holds one byte per base
   (0..3 as readSequence() yields them with dna2indexConv, 'N' -> 0: hash10x.c:432), all sequences concatenated,
   sequence s = codes[seqStart[s] .. seqStart[s+1]); which = 0 for the first .fa, 1 for the second. nPresent / nAbsent
   receive the "known" / "unknown" mosh counts the reference prints. h10x_crib_finish is the classification loop of
   cribBuild() (hash10x.c:476-494): cribType[], the merged crib[], and the four depth histograms (err, het, hom, mul). */
int  h10x_crib_genome(h10x_ctx *ctx, const uint8_t *codes, const uint64_t *seqStart, uint32_t nSeq, int which,
                      uint64_t *nPresent, uint64_t *nAbsent);
int  h10x_crib_finish(h10x_ctx *ctx);
/* histDim = entries per depth histogram; arrayMax[4] = arrayMax() of the reference's aErr, aHet, aHom, aMul */
int  h10x_crib_sizes(h10x_ctx *ctx, uint32_t *histDim, uint32_t arrayMax[4]);
/* copy-out (NULL = skip): chr/pos/type have hashNumber entries (CribInfo.chr, CribInfo.pos, cribType), hist 4 * histDim */
int  h10x_crib_export(h10x_ctx *ctx, int16_t *chr, uint16_t *pos, uint8_t *type, uint32_t *hist);
/* per-block good-hash counts (nGoodHashes[], hash10x.c:723) for --clusterReport: nBlocks entries; fails before --hashDepthRange */
int  h10x_export_ngood(h10x_ctx *ctx, uint32_t *nGood);

/* what writeHashFile() needs (hash10x.c:244-267): sizes, then a copy-out of any subset of the
   tables (NULL = skip). hashIndex: 2^B, hashValue/hashDepth: hashNumber, blocks: nBlocks,
   clusHash: nClusHash entries. */
int  h10x_get_sizes(h10x_ctx *ctx, h10x_sizes *out);
int  h10x_export(h10x_ctx *ctx, uint32_t *hashIndex, uint64_t *hashValue, uint32_t *hashDepth,
                 h10x_block *blocks, h10x_clushash *clusHash);

/* ---- multi-GPU: barcodes sharded over N ranks, hashes owned by value range (csrc/shard.hip) ----
   The reference has no counterpart (its only parallelism is the OMP loop inside --cluster); the invariant is
   that N ranks produce exactly the bytes one rank produces. A communicator is either RCCL (one process per GPU:
   rank 0 calls h10x_comm_unique_id, ships the 128 bytes to the others by any means, every rank calls
   h10x_comm_create_rccl) or in-process (h10x_comm_create_local: N ranks driven by N threads of one process,
   used by the tests and by single-process multi-GPU hosts). */
typedef struct h10x_comm h10x_comm;
int  h10x_comm_unique_id(void *id128);
int  h10x_comm_create_rccl(h10x_comm **comm, int rank, int nranks, const void *id128, int device, char *err, int errlen);
int  h10x_comm_create_local(h10x_comm **comms /* nranks outputs */, int nranks);
/* RCCL communicators for N ranks of ONE process (ncclCommInitAll), rank r on devices[r]: what a single-process host — the C program's
   --gpus N — uses when the box has a GPU per rank; each communicator is then driven by its rank's thread. Distinct devices required. */
int  h10x_comm_create_rccl_all(h10x_comm **comms /* nranks outputs */, int nranks, const int *devices, char *err, int errlen);
/* peer access between every pair of the listed devices (best effort; ranks of an in-process communicator that sit on different
   devices then copy device to device directly instead of through the host) */
int  h10x_device_enable_peers(const int *devices, int n);
/* one process per rank like RCCL, but host-staged over TCP (rank r listens on basePort + r at addr): for exercising the
   multi-process launch path where RCCL cannot run — several ranks sharing one GPU on a test box. Not a production path. */
int  h10x_comm_create_socket(h10x_comm **comm, int rank, int nranks, const char *addr, int basePort, char *err, int errlen);
/* "Virtual ranks": the ranks of an in-process communicator share one device and take turns on it. With serialize on, a rank's thread computes only between
   h10x_comm_turn_begin and h10x_comm_turn_end (its caller brackets every command with them) and hands the device on inside every collective while it waits for the
   others — so its stage timers read as if it had the GPU to itself, and its exchange timers (h10x_exchange_get) hold all the waiting. A measuring device for
   `bench.py --virtual-ranks N` on one-GPU boxes; the results are those of any other N-rank run. -1 on a communicator that is not in-process. */
int  h10x_comm_local_serialize(h10x_comm *comm, int on);
int  h10x_comm_turn_begin(h10x_comm *comm);
int  h10x_comm_turn_end(h10x_comm *comm, int device);
void h10x_comm_destroy(h10x_comm *comm);
int  h10x_comm_rank(const h10x_comm *comm);
int  h10x_comm_size(const h10x_comm *comm);
/* bind a communicator to a context (collective calls below are made by every rank in the same order) */
int  h10x_shard_attach(h10x_ctx *ctx, h10x_comm *comm);
/* --readFQB on this rank's contiguous barcode range of the sorted file (cut with h10x_host_partition): stage A
   locally, then the hash-owner exchange; afterwards every rank holds hashDepth of the whole data set and
   blocks/clusHash of its own barcodes. h10x_depth_range / h10x_cluster / h10x_cluster_split / h10x_crib_* work as
   usual and are collective on a sharded context (global barcode numbers; --hashDepthRange also allgathers the barcode
   lists of the in-range hashes, --clusterSplit numbers the new blocks over all ranks and sends the entries' new block
   numbers back to the hash owners, the crib is computed on every rank against the whole table). hashValue / hashIndex of the whole
   set (only --writeHash and the crib read them) are built by h10x_shard_gather, not here. */
int  h10x_shard_read_fqb(h10x_ctx *ctx, const uint32_t *host_records, uint64_t n_records);
int  h10x_shard_read_fqb_device(h10x_ctx *ctx, const uint32_t *dev_records, uint64_t n_records);
/* h10x_ingest_fqb for this rank's record range of the file: chunks append, the closing call runs the sharded --readFQB (collective) */
int  h10x_shard_ingest_fqb(h10x_ctx *ctx, const uint32_t *host_records, uint64_t n_records, int final_chunk);
/* --readHash onto shards (collective): the replicated tables of the file (hashIndex, hashValue, hashDepth: whole data set) plus THIS
   rank's contiguous cut of the file's blocks — localBlocks[0] unused, localBlocks[1 ..] = blocks codeBase + 1 .. of the file, with
   their ClusterHash records concatenated. The hash owners' barcode lists are rebuilt by an exchange (ownership by index range);
   inconsistent files (entries of a hash != its depth) are refused there. */
int  h10x_shard_load_state(h10x_ctx *ctx, const uint32_t *hashIndex, uint32_t hashNumber, const uint64_t *hashValue,
                           const uint32_t *hashDepth, const h10x_block *localBlocks, uint32_t nLocalBlocks,
                           const h10x_clushash *localClusHash, uint32_t codeBase, uint32_t nBlocksGlobal);
/* collective: every rank builds hashValue / hashIndex of the whole set; rank 0 receives every rank's blocks and
   clusHash (in file order), rebuilds the barcode lists and from then on IS a single-GPU context (for continuing on one
   GPU; --writeHash and the reports work on the shards directly, see below) */
int  h10x_shard_gather(h10x_ctx *ctx);
/* ---- a sharded state without a gather ("each rank pwrites its slices") ----
   After --readFQB every rank owns one contiguous range of the file's barcode blocks; every --clusterSplit appends, for
   each range that held parents, a range of new blocks numbered behind ALL existing blocks (hash10x.c:961-1003). A
   segment = one such range on one rank. h10x_shard_segments lists the segments of all ranks in file order (ascending
   globalBase): blocks [localStart, localStart + count) of rank `rank` are blocks [globalBase, ...) of the data set and
   their `entries` ClusterHash records start at localEntryStart in that rank's clusHash (filled in for the caller's own
   segments) and at globalEntryStart in the file's. An unsharded context reports itself as one segment of rank 0.
   h10x_shard_prepare_export (collective) builds hashValue[] / hashIndex[] of the whole set on every rank;
   h10x_export_slice copies elements [first, first + count) of one table of THIS rank to the host. */
typedef struct {
  uint32_t rank, localStart, count, globalBase;
  uint64_t entries, localEntryStart, globalEntryStart;
} h10x_shard_seg;
typedef struct {
  int32_t  rank, nranks, B; uint32_t hashNumber;
  uint32_t nBlocksGlobal;      /* arrayMax(clusterBlocks) of the whole data set */
  uint32_t nSegs;              /* entries h10x_shard_segments will write */
  uint64_t nEntriesGlobal, nRecordsGlobal;
} h10x_shard_info_t;
enum { H10X_TABLE_HASHINDEX = 0, H10X_TABLE_HASHVALUE = 1, H10X_TABLE_HASHDEPTH = 2, H10X_TABLE_BLOCKS = 3, H10X_TABLE_CLUSHASH = 4,
       H10X_TABLE_NGOOD = 5,
       H10X_TABLE_CLUSTER_RAW = 6 /* per block, after h10x_cluster: u32 clusters before the read merge (bit 31: given up at the 256th, hash10x.c:810-816),
                                     u32 good hashes with a label — the figures of the reference's --verbose line (hash10x.c:827-834) */ };
int  h10x_shard_info(h10x_ctx *ctx, h10x_shard_info_t *out);
int  h10x_shard_segments(h10x_ctx *ctx, h10x_shard_seg *out, uint32_t cap);
int  h10x_shard_prepare_export(h10x_ctx *ctx);
int  h10x_export_slice(h10x_ctx *ctx, int table, uint64_t first, uint64_t count, void *dst);
/* collective plumbing for launchers: barrier, max over ranks of a host double (timing), sums / maxima of small host
   arrays (in place), and a gather of byte strings to rank 0 (recv = the strings in rank order, counts[r] = bytes of
   rank r; recv / cap are read on rank 0 only). On an unsharded context these are the identity. */
int  h10x_shard_barrier(h10x_ctx *ctx);
/* do all ranks of the attached communicator say ok? Works before a state is loaded (h10x_shard_allreduce_* are the identity then):
   a launcher whose rank failed on its own — a short read of its part of a file, no memory — reports it here, so that every rank
   leaves with the same verdict instead of waiting in the next collective. Without a communicator *allOk = ok. */
int  h10x_shard_agree(h10x_ctx *ctx, int ok, int *allOk);
int  h10x_shard_allreduce_max(h10x_ctx *ctx, double *value);
int  h10x_shard_allreduce_sum_u64(h10x_ctx *ctx, uint64_t *values, uint32_t n);
int  h10x_shard_allreduce_max_u64(h10x_ctx *ctx, uint64_t *values, uint32_t n);
int  h10x_shard_gather_bytes(h10x_ctx *ctx, const void *send, uint64_t nbytes, void *recv, uint64_t cap, uint64_t *counts);

/* ---- the text reports, reduced on the device (csrc/stage_e.hip): only these results cross PCIe, never clusHash ----
   h10x_report_max / h10x_report_histogram: what histogramReport() is fed with (hash10x.c:351-402): `which` = 0
   hashDepth[first .. first+count) (hashDepthHist), 1 nHash and 2 nSubCluster of this rank's blocks [first, first+count)
   (codeSizeHist); hist has `bins` entries, values >= bins are not counted.
   h10x_cluster_report: codeClusterReport()'s per-barcode and per-sub-cluster figures (hash10x.c:870-952) for this
   rank's blocks [firstBlock, firstBlock+nBlocks): one h10x_block_rep per block, and for block b its sub-clusters
   1 .. min(nSubCluster, 255) consecutively in `clusters` (in block order). Without a crib the crib fields are 0.
   h10x_crib_summary: cribSummary()'s tallies (hash10x.c:1017-1061) over this rank's blocks 1..: counts[0..4] entries
   per crib type in base blocks, [5..9] in blocks made by --clusterSplit, [10] / [11] the number of such blocks; the two
   bitmaps (bit = hash index, (hashNumber + 31) / 32 words) mark the hashes met in each kind. */
typedef struct { uint32_t nGood, nClusHash, nClusRead, reserved; } h10x_block_rep;
typedef struct {
  uint32_t n, nRead, nt[5];      /* hashes, reads, hashes per crib type (err htA htB hom mul) */
  uint32_t nBad;                 /* located hashes on another chromosome than the first located one ("OTHER") */
  int16_t  chr; uint16_t pMin, pMax, nOtherListed;
  uint32_t other[10];            /* hash indices of the last ten of those, in the reference's print order */
} h10x_cluster_rep;
int  h10x_report_max(h10x_ctx *ctx, int which, uint64_t first, uint64_t count, uint32_t *maxValue);
int  h10x_report_histogram(h10x_ctx *ctx, int which, uint64_t first, uint64_t count, uint32_t bins, uint64_t *hist);
int  h10x_cluster_report(h10x_ctx *ctx, uint32_t firstBlock, uint32_t nBlocks, h10x_block_rep *blocks,
                         h10x_cluster_rep *clusters, uint64_t clusterCap, uint64_t *nClusters);
int  h10x_crib_summary(h10x_ctx *ctx, uint64_t counts[12], uint32_t *seenBase, uint32_t *seenCluster);
/* cribSummary's walk over the blocks (hash10x.c:1030-1046) as one word per ClusterHash record [first, first + count) of THIS rank, in clusHash order: bit 31 = the
   record's block was made by --clusterSplit, bits 28-30 = cribType of its hash, bits 0-27 = the hash index. The second and third figure of every type in the
   reference's summary is hashCount() of a HASH object fed in exactly this order (hash.c) — and that count depends on the order once such an object has doubled
   (see RefHash in host/h10x_host.c) — so the host layer replays the words through a restatement of it. */
int  h10x_crib_words(h10x_ctx *ctx, uint64_t first, uint64_t count, uint32_t *words);

/* ---- device memory plumbing for callers that keep the input resident in HBM (bench, pipelines) ----
   plain hipMalloc / hipMemcpy / hipDeviceSynchronize on `device`; return NULL / non-zero on failure */
void *h10x_device_malloc(int device, uint64_t bytes);
/* free and total bytes of the device's memory as the driver sees them (hipMemGetInfo; blocks parked in the library's cache count as used) */
int   h10x_device_mem_info(int device, uint64_t *freeBytes, uint64_t *totalBytes);
int   h10x_device_free(int device, void *ptr);
int   h10x_device_upload(int device, void *dst, const void *src, uint64_t bytes);
int   h10x_device_synchronize(int device);

/* ---- measurement hooks (not part of the reference surface) ----
   Per-kernel device timings collected with hipEvents on the context's stream when enabled.
   names: "mosh_extract", "sort_by_hash", ..., "good_hashes", "cluster" (whole command), "cluster_kernel" (all device work of
   --cluster), "cluster_main" (the main cluster_kernel launch alone) ... (h10x_timing_name(i)). */
int  h10x_timing_enable(h10x_ctx *ctx, int on);
int  h10x_timing_count(const h10x_ctx *ctx);
const char *h10x_timing_name(const h10x_ctx *ctx, int i);
int  h10x_timing_get(h10x_ctx *ctx, int i, double *total_ms, uint64_t *launches);
int  h10x_timing_reset(h10x_ctx *ctx);
/* The exchanges of a sharded context, per kind of collective (names: h10x_exchange_name(i), i < h10x_exchange_count()): calls; bytes this rank sent to and received
   from OTHER ranks; maxPeerOut = the sum over the calls of the largest share one peer received (grouped point-to-point sends: what one xGMI link carried); ms = from
   each call to its completion on the context's stream, waits for slower ranks included (collected while timing is enabled; a rank's compute is its stage timers less
   these). Cleared by h10x_timing_reset. What `bench.py --scaling strong` and `--virtual-ranks` print so that a scaling curve can be read. */
/* sharded contexts: the part of stage timer i (h10x_timing_name) spent inside exchanges — waiting for other ranks and moving bytes; the stage's own compute is the rest */
int  h10x_timing_wait_get(h10x_ctx *ctx, int i, double *ms);
int  h10x_exchange_count(void);
const char *h10x_exchange_name(int i);
int  h10x_exchange_get(h10x_ctx *ctx, int i, uint64_t *calls, uint64_t *bytesOut, uint64_t *bytesIn, uint64_t *maxPeerOut, double *ms, double *msInStages);
/* the stage timer (index for h10x_timing_name) whose kernels ran on the main stream while exchange kind i was on the context's exchange stream, -1 if it ran on the main
   stream with nothing beside it (option "shard_overlap" 0: always) */
int  h10x_exchange_beside(h10x_ctx *ctx, int i);
/* algorithmic work counters of the last commands: see DESIGN.md */
typedef struct {
  uint64_t pairs;            /* read pairs hashed                                   */
  uint64_t kmers;            /* k-mers hashed (237 per pair at k=21)                */
  uint64_t entries;          /* H = sum nHash                                       */
  uint64_t distinct;         /* U = hashNumber - 1                                  */
  uint64_t clustered_codes;  /* barcodes visited by the last h10x_cluster           */
  uint64_t sum_good;         /* sum of good hashes over those barcodes              */
  uint64_t sum_good_depth;   /* sum over good hashes of depth (gathered row entries)*/
  uint64_t sum_hash_clustered; /* sum nHash over barcodes with good hashes          */
  uint64_t fallback_blocks;  /* barcodes that took the global-memory path in stage A */
  uint64_t cluster_class_counts[4]; /* barcodes clustered in: half-CU LDS with 1024 lanes, half-CU LDS with 512 lanes, full-CU LDS (512 lanes), HBM scratch */
  uint64_t cluster_first_mode;     /* placement of the first[] table: 0 dense in LDS, 1 ranked (bitmap) in LDS, 2 per-workgroup HBM slot, 4 translated (16-bit handles into a table in LDS) */
  uint64_t cluster_overflow_blocks; /* ranked placement: barcodes re-run on the HBM path because too many barcodes were present */
  uint64_t cluster_main[4];        /* work of the main cluster launch alone (timer "cluster_main"): good hashes, gathered list entries, nHash, barcodes */
  uint64_t cluster_phase_ticks[8]; /* diagnostic (option "cluster_stamps"): 100 MHz ticks per phase summed over workgroups:
                                      [0] init (+ bitmap), [1] list loop, [2] barrier, [3] replay, [4] quotient, [5] output */
  uint64_t list_words[2];          /* sharded --hashDepthRange: 32-bit words of in-range barcode lists this rank received, [0] as plain
                                      numbers, [1] as they travelled (delta-coded: option "shard_delta_lists"); 0 0 if sent plain */
  uint64_t index_table_form;       /* single-GPU index build, the look-up table behind the ClusterHash records: 0 = hashIndex[] + hashValue[] (or the library's private table: option
                                      "index_priv_table"), 1 = the wide table, entry = index | hash / w, 2 = the wide table in the probed format (index | hash >> B | probe number:
                                      -B 29 / 30 at k = 21), 3 = the probed table failed and the round-5 pair was built (option "index_probed_table") */
  uint64_t shard_reply_path;       /* sharded --readFQB, how this hash owner answered its entries: 0 nothing to answer, 1 by look-up, 2 by scatter, 3 by scatter after a look-up
                                      that failed or did not fit (option "shard_reply_sort") */
} h10x_counters;
int  h10x_get_counters(h10x_ctx *ctx, h10x_counters *out);
/* test / tuning knobs (none changes a result): "stage_a_max_slots" caps the LDS hash-set slots per barcode in stage A (0 =
   default) so that the global-memory fallback can be exercised on small inputs; "chunk_size" (above); "index_no_pack" 1 = index
   build with separate key / block arrays even where the packed one-word entries fit; "index_probed_table" 0 default = the wide look-up table takes the probed entry format
   (index | hash >> B | probe number, in the reference's own geometry: hashIndex[] is its index column) where index | hash / w does not fit 64 bits, 1 = always, 2 = never (two
   tables, as round 5 built them), 3 = always and with one bit of probe number (exercises the fall-back; h10x_counters.index_table_form says what was built); "index_priv_table" 1 = the entry look-ups of the
   index build through the library's own one-read table also where the reference-shaped 64-bit table fits (default 0: only where it
   does not, i.e. -B 29 / 30 at k = 21), 2 = never, 3 = always and undersized (exercises its fall-back); "cluster_narrow_first" 1 = first[] of the
   cluster kernel at 2 bytes per entry in every block, w >= 2 = 4 bytes down to w list-loop waves (default 0: 4 bytes where that
   costs no wave); "cluster_tr_packed" (translated placement of first[]: -1 / 1 = several lists per wave instruction, the default; 0 = round 4's
   one list per wave instruction), "cluster_tr_class_t" (packed form: -1 / 1 = lists of 65 .. 96 entries run two to a unit of three chunks, the default; 0 = one to a
   unit of two chunks like the lists of 97 .. 128), "cluster_lds_budget", "cluster_first_global", "cluster_first_cap", "cluster_big_ranks", "cluster_threads0",
   "cluster_budget0" (placement and launch-class overrides of the tests), "cluster_stamps" (phase stamps into h10x_counters),
   "shard_reply_sort" (sharded index build: 0 default = a hash owner answers by look-up in a table of its distinct hashes, 1 = by scattering from its sorted order; tests of the
   fall-back, reported in h10x_counters.shard_reply_path: 2 = look up, then scatter all the same, 3 = a look-up table that fails, 4 = one that does not fit),
   "shard_overlap" (1 default = the exchanges whose result a later stage needs — the in-range barcode lists, hashDepth[] of the other owners — run on an exchange stream beside
   the main stream's kernels; 0 = every exchange on the main stream),
   "shard_owner_cut" (0 default = hash owners' value ranges cut at the quantiles of the canonical-hash density, equal shares; 1 = equal value ranges),
   "shard_row_shift", "shard_rows_fake_base" (sharded list offsets beyond 32 bits on small inputs), "shard_delta_lists" (-1 default:
   the in-range barcode lists travel delta-coded where bytes are dear — more than one rank on the host-staged TCP backend, not over xGMI; 0 never; 1 always). Unknown name: -1. */
int  h10x_set_option(h10x_ctx *ctx, const char *name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* H10X_H */
