/* hash10x_main.c — `hash10x-amd`: the reference's command surface (hash10x.c:1122-1305) for the in-scope commands, on one
 * MI355X or, with --gpus N, on N of them (one rank per GPU, ranks = threads of this process over the in-process
 * communicator; barcodes sharded, see csrc/shard.hip).
 *
 * The argv grammar IS the drop-in boundary and is the reference's: tokens are taken strictly left to right, every token
 * starts with '-', a command consumes a fixed number of arguments and is skipped as "unknown" when fewer are left
 * (the reference's ARGMATCH(x, n) test, hash10x.c:1173), parameters (-k -w -r -B -N -c -ct) are latched until the next
 * --readFQB / --readHash, each command is echoed as "COMMAND ..." and followed by a resource line. Fatal conditions
 * print "FATAL ERROR: <the reference's message>" and exit(-1) like die() (utils.c:18-29). The dispatch itself is a
 * table of commands, not the reference's if-chain.
 * Additions: --device <n>, --gpus <n>, --sortFQB; the resource line also carries wall-clock seconds.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <time.h>
#include <pthread.h>
#include <sys/stat.h>
#include <sys/resource.h>
#include "h10x_host.h"

static FILE *outFile;
static int printTables;

static void die(const char *fmt, ...) {
  va_list ap; va_start(ap, fmt);
  fprintf(stderr, "FATAL ERROR: "); vfprintf(stderr, fmt, ap); fprintf(stderr, "\n");
  va_end(ap);
  exit(-1);
}
static void say(const char *fmt, ...) {                      /* outFile, and stdout too when -o names a file */
  va_list ap; va_start(ap, fmt); vfprintf(outFile, fmt, ap); va_end(ap);
  if (outFile != stdout) { va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); }
}

/* utils.c:122-148 format (user/system are CPU seconds from getrusage) + wall seconds */
static double wallNow(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static void timeUpdate(FILE *f, int total) {
  static int first = 1; static struct rusage rOld, rFirst; static double wOld, wFirst;
  struct rusage rNew; getrusage(RUSAGE_SELF, &rNew); double wNew = wallNow();
  if (first) { rFirst = rNew; wFirst = wNew; first = 0; rOld = rNew; wOld = wNew; return; }
  const struct rusage *r0 = total ? &rFirst : &rOld; double w0 = total ? wFirst : wOld;
  long us = (rNew.ru_utime.tv_sec - r0->ru_utime.tv_sec) * 1000000L + (rNew.ru_utime.tv_usec - r0->ru_utime.tv_usec);
  long ss = (rNew.ru_stime.tv_sec - r0->ru_stime.tv_sec) * 1000000L + (rNew.ru_stime.tv_usec - r0->ru_stime.tv_usec);
  fprintf(f, "user\t%ld.%06ld\tsystem\t%ld.%06ld\tmax_RSS\t%ld\twall\t%.6f\n", us / 1000000, us % 1000000, ss / 1000000, ss % 1000000,
          rNew.ru_maxrss - r0->ru_maxrss, wNew - w0);
  rOld = rNew; wOld = wNew;
}

/* ---- the ranks: one session each; rank 0 is also the unsharded session when --gpus is 1 ---- */
enum { MAX_RANKS = 64 };
static struct { int n; h10x_session *s[MAX_RANKS]; h10x_comm *comm[MAX_RANKS]; } team = {1, {0}, {0}};
typedef int (*rank_fn)(h10x_session *s, int rank, void *arg);
typedef struct { rank_fn fn; void *arg; int rank, rc; } RankCall;
static void *rank_thread(void *p) { RankCall *c = (RankCall *)p; c->rc = c->fn(team.s[c->rank], c->rank, c->arg); return 0; }
/* every rank runs fn (a collective command); returns the lowest failing rank + 1, 0 if none failed */
static int on_all_ranks(rank_fn fn, void *arg) {
  if (team.n == 1) return fn(team.s[0], 0, arg) ? 1 : 0;
  pthread_t th[MAX_RANKS]; RankCall call[MAX_RANKS];
  for (int r = 0; r < team.n; ++r) { call[r].fn = fn; call[r].arg = arg; call[r].rank = r; call[r].rc = 0; if (pthread_create(&th[r], 0, rank_thread, &call[r])) die("pthread_create failed"); }
  int bad = 0;
  for (int r = 0; r < team.n; ++r) { pthread_join(th[r], 0); if (call[r].rc && !bad) bad = r + 1; }
  return bad;
}
static void die_of(int bad) { die("%s", h10x_session_error(team.s[bad - 1])); }
static void set_all(const char *name, int v) { for (int r = 0; r < team.n; ++r) h10x_session_set(team.s[r], name, v); }

/* The ranks' communicators. With a GPU per rank the exchange is RCCL over xGMI (north_star: "host code stays in C ... RCCL all-to-all
   over xGMI"): ncclCommInitAll gives this one process a communicator per device, each driven by its rank's thread. With fewer devices
   than ranks (a 1-GPU test box) RCCL cannot run — it refuses two ranks on a device — and the ranks use the in-process communicator
   (device-to-device copies behind a thread barrier). H10X_COMM=local forces the latter, H10X_COMM=rccl insists on the former. */
static const char *teamBackend = "none";
static void set_gpus(int n) {
  if (n < 1 || n > MAX_RANKS) die("--gpus %d: must be 1..%d", n, MAX_RANKS);
  if (n == team.n) return;
  int nDev = h10x_device_count(); if (nDev < 1) nDev = 1;
  const char *copy[] = {"k", "w", "r", "B", "N", "c", "ct", "timing", 0};
  const int dev0 = h10x_session_get(team.s[0], "device");
  /* a state loaded under the old team goes with it: its context is bound to that team's communicator and shard layout (the next
     command after --gpus is a --readFQB / --readHash anyway: the parameters are latched until then, like -k -w -r -B) */
  { h10x_session *fresh = h10x_session_new(); if (!fresh) die("out of memory");
    for (int i = 0; copy[i]; ++i) h10x_session_set(fresh, copy[i], h10x_session_get(team.s[0], copy[i]));
    h10x_session_set(fresh, "device", dev0);
    h10x_session_free(team.s[0]); team.s[0] = fresh; }
  for (int r = 1; r < team.n; ++r) { h10x_session_free(team.s[r]); team.s[r] = 0; }
  for (int r = 0; r < team.n; ++r) if (team.comm[r]) { h10x_comm_destroy(team.comm[r]); team.comm[r] = 0; }
  int devs[MAX_RANKS]; devs[0] = dev0;
  for (int r = 1; r < n; ++r) {
    if (!(team.s[r] = h10x_session_new())) die("out of memory");
    for (int i = 0; copy[i]; ++i) h10x_session_set(team.s[r], copy[i], h10x_session_get(team.s[0], copy[i]));
    devs[r] = (dev0 + r) % nDev;
    h10x_session_set(team.s[r], "device", devs[r]);                                    /* more ranks than devices: they share (tests on a 1-GPU box) */
  }
  team.n = n; teamBackend = "none";
  if (n > 1) {
    const char *want = getenv("H10X_COMM");
    const int canRccl = n <= nDev, wantRccl = want ? !strcmp(want, "rccl") : canRccl;
    if (want && strcmp(want, "rccl") && strcmp(want, "local")) die("H10X_COMM=%s: must be rccl or local", want);
    int useLocal = !wantRccl;
    if (wantRccl) {
      char err[256] = "";
      if (!canRccl) die("H10X_COMM=rccl: %d ranks need %d devices, %d visible (RCCL refuses two ranks on one device)", n, n, nDev);
      if (!h10x_comm_create_rccl_all(team.comm, n, devs, err, (int)sizeof err)) teamBackend = "rccl";
      else if (want) die("%s", err);                       /* RCCL was asked for by name: no silent substitute */
      else {                                                /* the default picked it and it cannot start here: the in-process communicator does the same exchange */
        fprintf(stderr, "hash10x-amd: RCCL did not start (%s): the %d ranks use the in-process communicator (peer copies) instead\n", err, n);
        for (int r = 0; r < n; ++r) team.comm[r] = 0;
        useLocal = 1;
      }
    }
    if (useLocal) {
      h10x_device_enable_peers(devs, n);
      if (h10x_comm_create_local(team.comm, n)) die("h10x_comm_create_local failed");
      teamBackend = "local";
    }
  }
}

/* ---- commands ---- */
static void usage(void) {
  h10x_session *s = team.s[0];
  fprintf(stderr, "Usage: hash10x-amd <commands>\n");
  fprintf(stderr, "Commands can be parameter settings with -x, or operations:\n");
  fprintf(stderr, "Be sure to set relevant parameters before invoking an operation!\n");
  fprintf(stderr, "   -k <kmer size> [%d]\n", h10x_session_get(s, "k"));
  fprintf(stderr, "   -w <window> [%d]\n", h10x_session_get(s, "w"));
  fprintf(stderr, "   -r <random number seed> [%d]\n", h10x_session_get(s, "r"));
  fprintf(stderr, "   -B <hash index table bitcount> [%d]\n", h10x_session_get(s, "B"));
  fprintf(stderr, "   -N <num records to read: 0 for all> [%d]\n", h10x_session_get(s, "N"));
  fprintf(stderr, "   -c <file chunkSize in readPairs> [%d]\n", h10x_session_get(s, "c"));
  fprintf(stderr, "   -ct | --clusterThreshold <clusterThreshold> [%d]\n", h10x_session_get(s, "ct"));
  fprintf(stderr, "   -t | --threads <n> : accepted and ignored (clustering runs on the GPU)\n");
  fprintf(stderr, "   -o | --output <output filename> : '-' for stdout\n");
  fprintf(stderr, "   --device <HIP device ordinal> [0]\n");
  fprintf(stderr, "   --gpus <n> [1]: shard the barcodes of the next --readFQB / --readHash over n GPUs (devices device, device+1, ...)\n");
  fprintf(stderr, "   --sortFQB <fqb from fq2b> <sorted fqb output>: sort records by barcode on the GPU (instead of bsort -k 4 -r 120)\n");
  fprintf(stderr, "   --readFQB <sorted fqb input file name>: must have this or readHash\n");
  fprintf(stderr, "   --readHash <hash input file name>\n");
  fprintf(stderr, "   --writeHash <hash output file name>\n");
  fprintf(stderr, "   --hashDepthRange <min> <max>: set limits for hash counts for cluster\n");
  fprintf(stderr, "   --cluster <codeMin> <codeMax>: cluster reads for range of barcodes (1, 0 for all)\n");
  fprintf(stderr, "   --clusterSplit\n");
  fprintf(stderr, "   --cribBuild <genome1.fa> <genome2.fa>: match to genomic hashes\n");
  fprintf(stderr, "   --clusterReport <codeMin> <codeMax>\n");
  fprintf(stderr, "   --cribSummary\n");
  fprintf(stderr, "   --tables : toggle the CRIB_TABLE lines of cribBuild\n");
  fprintf(stderr, "   --hashStats : distribution of hash counts and summary info\n");
  fprintf(stderr, "   --codeStats : distribution of barcode/cluster sizes and summary info\n");
  fprintf(stderr, "   --help : print this usage message\n");
}
static void say_initialised(void) {
  h10x_session *s = team.s[0];
  fprintf(outFile, "hash10x initialised with k = %d, w = %d, random seed = %d, hashtable bits = %d\n",
          h10x_session_get(s, "k"), h10x_session_get(s, "w"), h10x_session_get(s, "r"), h10x_session_get(s, "B"));
}
static h10x_shard_info_t sizes_now(void) {
  h10x_shard_info_t z; memset(&z, 0, sizeof z);
  if (h10x_session_ctx(team.s[0])) h10x_shard_info(h10x_session_ctx(team.s[0]), &z);
  return z;
}
static void say_filled(const h10x_shard_info_t *z) {
  fprintf(outFile, "  filled hash table: %llu hashes from %u barcodes in %u bins\n", (unsigned long long)z->nEntriesGlobal, z->nBlocksGlobal, z->hashNumber);
}

typedef struct { const char *path; uint64_t cut[MAX_RANKS + 1]; } ReadArg;
static int rank_readFQB(h10x_session *s, int r, void *a) {
  ReadArg *ra = (ReadArg *)a;
  return team.n == 1 ? h10x_session_readFQB(s, ra->path) : h10x_session_shardReadFQB_file(s, team.comm[r], ra->path, ra->cut[r], ra->cut[r + 1] - ra->cut[r]);
}
static void cmd_readFQB(char **a) {
  say_initialised();
  ReadArg ra; ra.path = a[0];
  if (team.n > 1) {
    struct stat sb; char err[256];
    if (stat(a[0], &sb)) die("failed to open fqb file %s", a[0]);
    uint64_t n = (uint64_t)sb.st_size / 120; const int N = h10x_session_get(team.s[0], "N");
    if (N > 0 && (uint64_t)N < n) n = (uint64_t)N;
    if (h10x_host_partition_file(a[0], n, team.n, ra.cut, err, (int)sizeof err)) die("%s", err);
  }
  const int bad = on_all_ranks(rank_readFQB, &ra); if (bad) die_of(bad);
  const h10x_shard_info_t z = sizes_now(); const double nb = (double)z.nBlocksGlobal - 1;
  say("  read %llu read pair records for %u barcodes, mean %.2f read pairs per barcode\n", (unsigned long long)z.nRecordsGlobal, z.nBlocksGlobal - 1, z.nRecordsGlobal / nb);
  say("  created %llu hashes, mean %.2f hashes per read pair, %.2f per barcode\n", (unsigned long long)z.nEntriesGlobal, z.nEntriesGlobal / (double)z.nRecordsGlobal, z.nEntriesGlobal / nb);
  say_filled(&z);
}
static int rank_readHash(h10x_session *s, int r, void *a) { return team.n == 1 ? h10x_session_readHash(s, (const char *)a) : h10x_session_shardReadHash(s, team.comm[r], (const char *)a); }
static void cmd_readHash(char **a) {
  say_initialised();
  const int bad = on_all_ranks(rank_readHash, a[0]); if (bad) die_of(bad);
  const h10x_shard_info_t z = sizes_now();
  say("  read %llu hashes in %u barcode blocks\n", (unsigned long long)z.nEntriesGlobal, z.nBlocksGlobal);
  say_filled(&z);
}
static int rank_writeHash(h10x_session *s, int r, void *a) { (void)r; return h10x_session_writeHash(s, (const char *)a); }
static void cmd_writeHash(char **a) {
  const int bad = on_all_ranks(rank_writeHash, a[0]); if (bad) die_of(bad);
  const h10x_shard_info_t z = sizes_now();
  say("  wrote %llu hash table entries and %u barcode blocks\n", 1ULL << z.B, z.nBlocksGlobal);
}
static int rank_range(h10x_session *s, int r, void *a) { (void)r; const int *v = (const int *)a; return h10x_session_hashDepthRange(s, v[0], v[1]); }
static void cmd_hashDepthRange(char **a) {
  int v[2] = {atoi(a[0]), atoi(a[1])};
  const int bad = on_all_ranks(rank_range, v); if (bad) die_of(bad);
  printf("  made goodHashes arrays for hash range %d to %d\n", v[0], v[1]);
}
/* "!! ..." is the reference's soft error: print and carry on (hash10x.c:1257-1260) */
static int soft_or_die(int bad) {
  if (!bad) return 0;
  const char *e = h10x_session_error(team.s[bad - 1]);
  if (strncmp(e, "!!", 2)) die("%s", e);
  fprintf(outFile, "%s\n", e); if (outFile != stdout) fprintf(stderr, "%s\n", e);
  return 1;
}
static int isVerbose = 0;                                                             /* --verbose (hash10x.c:1180): the per-barcode lines of --cluster */
static int rank_cluster(h10x_session *s, int r, void *a) { (void)r; const int *v = (const int *)a; return h10x_session_cluster(s, v[0], v[1]); }
static int rank_verbose(h10x_session *s, int r, void *a) { const int *v = (const int *)a; return h10x_session_clusterVerbose(s, v[0], v[1], r == 0 && isVerbose ? outFile : 0, r == 0 ? stderr : 0); }
static void cmd_verbose(char **a) { (void)a; isVerbose = 1; }
static void cmd_cluster(char **a) {
  int v[2] = {atoi(a[0]), atoi(a[1])};
  if (soft_or_die(on_all_ranks(rank_cluster, v))) return;
  { const int bad = on_all_ranks(rank_verbose, v); if (bad) die_of(bad); }             /* the "too many clusters" notes go to stderr with or without --verbose */
  say("  clustered codes %d to %d\n", v[0] ? v[0] : 1, v[1] ? v[1] : (int)sizes_now().nBlocksGlobal);
}
static int rank_split(h10x_session *s, int r, void *a) { (void)r; (void)a; return h10x_session_clusterSplit(s); }
static void cmd_clusterSplit(char **a) {                                              /* the lines of clusterSplitCodes (hash10x.c:998-1012): count, a time line on stdout, fillHashTable's line */
  (void)a;
  const h10x_shard_info_t before = sizes_now();
  const int bad = on_all_ranks(rank_split, 0); if (bad) die_of(bad);
  const h10x_shard_info_t after = sizes_now();
  say("  made %d additional new barcodes from clusters in %d original barcodes\n", (int)(after.nBlocksGlobal - before.nBlocksGlobal), (int)before.nBlocksGlobal);
  printf("  cluster timepoint: "); timeUpdate(stdout, 0);
  say_filled(&after);
}
static int rank_crib(h10x_session *s, int r, void *a) { char **f = (char **)a; return h10x_session_cribBuild(s, f[0], f[1], r == 0 ? outFile : 0, printTables); }
static void cmd_cribBuild(char **a) { const int bad = on_all_ranks(rank_crib, a); if (bad) die_of(bad); }
static int rank_report(h10x_session *s, int r, void *a) { const int *v = (const int *)a; return h10x_session_clusterReport(s, v[0], v[1], r == 0 ? outFile : 0); }
static void cmd_clusterReport(char **a) { int v[2] = {atoi(a[0]), atoi(a[1])}; soft_or_die(on_all_ranks(rank_report, v)); }
static int rank_summary(h10x_session *s, int r, void *a) { (void)a; return h10x_session_cribSummary(s, r == 0 ? outFile : 0); }
static void cmd_cribSummary(char **a) { (void)a; const int bad = on_all_ranks(rank_summary, 0); if (bad) die_of(bad); }
static int rank_hashStats(h10x_session *s, int r, void *a) { (void)a; return h10x_session_hashStats(s, r == 0 ? outFile : 0); }
static void cmd_hashStats(char **a) { (void)a; const int bad = on_all_ranks(rank_hashStats, 0); if (bad) die_of(bad); }
static int rank_codeStats(h10x_session *s, int r, void *a) { (void)a; return h10x_session_codeStats(s, r == 0 ? outFile : 0); }
static void cmd_codeStats(char **a) { (void)a; const int bad = on_all_ranks(rank_codeStats, 0); if (bad) die_of(bad); }
static void cmd_sortFQB(char **a) { if (h10x_session_sortFQB(team.s[0], a[0], a[1])) die_of(1); }
static void cmd_output(char **a) {
  if (!strcmp(a[0], "-")) outFile = stdout;
  else if (!(outFile = fopen(a[0], "w"))) { fprintf(stderr, "can't open output file %s\n", a[0]); outFile = stdout; }
}
static void cmd_threads(char **a) { (void)a; fprintf(stderr, "  can't set thread number - clustering runs on the GPU\n"); }
static void cmd_tables(char **a) { (void)a; printTables = !printTables; }                 /* hash10x.c:1198 */
static void cmd_help(char **a) { (void)a; usage(); }
static void cmd_gpus(char **a) { set_gpus(atoi(a[0])); fprintf(outFile, "  %d rank(s), communicator: %s\n", team.n, team.n > 1 ? teamBackend : "none (one GPU)"); }
static void cmd_device(char **a) {
  int nDev = h10x_device_count(); if (nDev < 1) nDev = 1;
  for (int r = 0; r < team.n; ++r) h10x_session_set(team.s[r], "device", (atoi(a[0]) + r) % nDev);
}

typedef struct { const char *name; int nArgs; void (*run)(char **args); const char *param; } Command;
static const Command commands[] = {
  {"-k", 1, 0, "k"}, {"-w", 1, 0, "w"}, {"-r", 1, 0, "r"}, {"-B", 1, 0, "B"}, {"-N", 1, 0, "N"}, {"-c", 1, 0, "c"},
  {"-ct", 1, 0, "ct"}, {"--clusterThreshold", 1, 0, "ct"},
  {"--device", 1, cmd_device, 0}, {"--gpus", 1, cmd_gpus, 0},
  {"-t", 1, cmd_threads, 0}, {"--threads", 1, cmd_threads, 0}, {"-o", 1, cmd_output, 0}, {"--output", 1, cmd_output, 0},
  {"--tables", 0, cmd_tables, 0}, {"--verbose", 0, cmd_verbose, 0},
  {"--readFQB", 1, cmd_readFQB, 0}, {"--readHash", 1, cmd_readHash, 0}, {"--writeHash", 1, cmd_writeHash, 0},
  {"--hashDepthRange", 2, cmd_hashDepthRange, 0}, {"--cluster", 2, cmd_cluster, 0}, {"--clusterSplit", 0, cmd_clusterSplit, 0},
  {"--sortFQB", 2, cmd_sortFQB, 0}, {"--cribBuild", 2, cmd_cribBuild, 0}, {"--clusterReport", 2, cmd_clusterReport, 0},
  {"--cribSummary", 0, cmd_cribSummary, 0}, {"--hashStats", 0, cmd_hashStats, 0}, {"--codeStats", 0, cmd_codeStats, 0},
  {"--help", 0, cmd_help, 0},
  {0, 0, 0, 0}};

int main(int argc, char **argv) {
  --argc; ++argv;
  outFile = stdout;
  timeUpdate(stdout, 0);
  if (!(team.s[0] = h10x_session_new())) die("out of memory");
  if (!argc) usage();

  while (argc) {
    if (**argv != '-') die("option/command %s does not start with '-': run without arguments for usage", *argv);
    fprintf(outFile, "COMMAND %s", *argv);
    for (int i = 1; i < argc && *argv[i] != '-'; ++i) fprintf(outFile, " %s", argv[i]);
    fputc('\n', outFile);
    if (outFile != stdout) {                                  /* hash10x.c:1166-1171 as it stands: the echo on stdout carries the command alone, its arguments go to the -o file a second time */
      printf("COMMAND %s", *argv);
      for (int i = 1; i < argc && *argv[i] != '-'; ++i) fprintf(outFile, " %s", argv[i]);
      putchar('\n');
    }
    if (!strcmp(*argv, "--quit") || !strcmp(*argv, "--exit")) break;
    const Command *c = commands;
    while (c->name && (strcmp(c->name, *argv) || argc < 1 + c->nArgs)) ++c;            /* too few arguments left: not a match, like ARGMATCH */
    if (!c->name) die("unknown option/command %s; run without arguments for usage", *argv);
    if (c->param) set_all(c->param, atoi(argv[1])); else c->run(argv + 1);
    argc -= 1 + c->nArgs; argv += 1 + c->nArgs;
    printf("  "); timeUpdate(stdout, 0); fflush(stdout);
  }
  fprintf(outFile, "total resources used: "); timeUpdate(outFile, 1);
  for (int r = 0; r < team.n; ++r) h10x_session_free(team.s[r]);
  for (int r = 0; r < team.n; ++r) if (team.comm[r]) h10x_comm_destroy(team.comm[r]);
  return 0;
}
