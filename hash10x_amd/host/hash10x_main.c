/* hash10x_main.c — `hash10x-amd`: the reference's command loop (hash10x.c:1122-1305) for the
 * in-scope commands, running on one MI355X through libh10x_host / libh10x_hip.
 *
 * Same argv grammar: tokens are processed strictly left to right, every token starts with '-',
 * parameters (-k -w -r -B -N -c -ct) are latched and take effect at the next --readFQB/--readHash,
 * each command is echoed as "COMMAND ..." and followed by a resource line. Fatal conditions print
 * "FATAL ERROR: <the reference's message>" and exit(-1) like die() (utils.c:18-29).
 * Additions: --device <n>; the resource line also carries wall-clock seconds.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <time.h>
#include <sys/resource.h>
#include "h10x_host.h"

static FILE *outFile;

static void die(const char *fmt, ...) {
  va_list ap; va_start(ap, fmt);
  fprintf(stderr, "FATAL ERROR: "); vfprintf(stderr, fmt, ap); fprintf(stderr, "\n");
  va_end(ap);
  exit(-1);
}

static double wallNow(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* utils.c:122-148 format (user/system are CPU seconds from getrusage) + wall seconds */
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

static void usage(h10x_session *s) {
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

static void say(const char *fmt, ...) {                      /* outFile, and stdout too when -o names a file */
  va_list ap; va_start(ap, fmt); vfprintf(outFile, fmt, ap); va_end(ap);
  if (outFile != stdout) { va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); }
}

int main(int argc, char **argv) {
  --argc; ++argv;
  outFile = stdout;
  int printTables = 0;
  timeUpdate(stdout, 0);
  h10x_session *s = h10x_session_new();
  if (!s) die("out of memory");
  if (!argc) usage(s);

  while (argc) {
    if (**argv != '-') die("option/command %s does not start with '-': run without arguments for usage", *argv);
    fprintf(outFile, "COMMAND %s", *argv);
    for (int i = 1; i < argc && *argv[i] != '-'; ++i) fprintf(outFile, " %s", argv[i]);
    fputc('\n', outFile);
#define ARGMATCH(x, n) (!strcmp(*argv, x) && argc >= n && (argc -= n, argv += n))
    if (ARGMATCH("-k", 2)) h10x_session_set(s, "k", atoi(argv[-1]));
    else if (ARGMATCH("-w", 2)) h10x_session_set(s, "w", atoi(argv[-1]));
    else if (ARGMATCH("-r", 2)) h10x_session_set(s, "r", atoi(argv[-1]));
    else if (ARGMATCH("-B", 2)) h10x_session_set(s, "B", atoi(argv[-1]));
    else if (ARGMATCH("-N", 2)) h10x_session_set(s, "N", atoi(argv[-1]));
    else if (ARGMATCH("-c", 2)) h10x_session_set(s, "c", atoi(argv[-1]));
    else if (ARGMATCH("--device", 2)) h10x_session_set(s, "device", atoi(argv[-1]));
    else if (ARGMATCH("-t", 2) || ARGMATCH("--threads", 2))
      fprintf(stderr, "  can't set thread number - clustering runs on the GPU\n");
    else if (ARGMATCH("-o", 2) || ARGMATCH("--output", 2)) {
      if (!strcmp(argv[-1], "-")) outFile = stdout;
      else if (!(outFile = fopen(argv[-1], "w"))) { fprintf(stderr, "can't open output file %s\n", argv[-1]); outFile = stdout; }
    }
    else if (ARGMATCH("--tables", 1)) printTables = !printTables;                       /* hash10x.c:1198 */
    else if (ARGMATCH("--verbose", 1)) { /* accepted: no per-barcode text on the device path */ }
    else if (ARGMATCH("--readFQB", 2)) {
      fprintf(outFile, "hash10x initialised with k = %d, w = %d, random seed = %d, hashtable bits = %d\n",
              h10x_session_get(s, "k"), h10x_session_get(s, "w"), h10x_session_get(s, "r"), h10x_session_get(s, "B"));
      if (h10x_session_readFQB(s, argv[-1])) die("%s", h10x_session_error(s));
      h10x_sizes z; h10x_get_sizes(h10x_session_ctx(s), &z);
      h10x_counters c; h10x_get_counters(h10x_session_ctx(s), &c);
      double nb = (double)z.nBlocks - 1;
      say("  read %llu read pair records for %u barcodes, mean %.2f read pairs per barcode\n",
          (unsigned long long)z.nRecords, z.nBlocks - 1, z.nRecords / nb);
      say("  created %llu hashes, mean %.2f hashes per read pair, %.2f per barcode\n",
          (unsigned long long)z.nClusHash, z.nClusHash / (double)z.nRecords, z.nClusHash / nb);
      fprintf(outFile, "  filled hash table: %llu hashes from %u barcodes in %u bins\n", (unsigned long long)z.nClusHash, z.nBlocks, z.hashNumber);
    }
    else if (ARGMATCH("--readHash", 2)) {
      fprintf(outFile, "hash10x initialised with k = %d, w = %d, random seed = %d, hashtable bits = %d\n",
              h10x_session_get(s, "k"), h10x_session_get(s, "w"), h10x_session_get(s, "r"), h10x_session_get(s, "B"));
      if (h10x_session_readHash(s, argv[-1])) die("%s", h10x_session_error(s));
      h10x_sizes z; h10x_get_sizes(h10x_session_ctx(s), &z);
      say("  read %llu hashes in %u barcode blocks\n", (unsigned long long)z.nClusHash, z.nBlocks);
      fprintf(outFile, "  filled hash table: %llu hashes from %u barcodes in %u bins\n", (unsigned long long)z.nClusHash, z.nBlocks, z.hashNumber);
    }
    else if (ARGMATCH("--writeHash", 2)) {
      if (h10x_session_writeHash(s, argv[-1])) die("%s", h10x_session_error(s));
      h10x_sizes z; h10x_get_sizes(h10x_session_ctx(s), &z);
      say("  wrote %llu hash table entries and %u barcode blocks\n", 1ULL << z.B, z.nBlocks);
    }
    else if (ARGMATCH("--hashDepthRange", 3)) {
      if (h10x_session_hashDepthRange(s, atoi(argv[-2]), atoi(argv[-1]))) die("%s", h10x_session_error(s));
      printf("  made goodHashes arrays for hash range %d to %d\n", atoi(argv[-2]), atoi(argv[-1]));
    }
    else if (ARGMATCH("-ct", 2) || ARGMATCH("--clusterThreshold", 2)) h10x_session_set(s, "ct", atoi(argv[-1]));
    else if (ARGMATCH("--cluster", 3)) {
      int codeMin = atoi(argv[-2]), codeMax = atoi(argv[-1]);
      if (h10x_session_cluster(s, codeMin, codeMax)) {
        const char *e = h10x_session_error(s);
        if (!strncmp(e, "!!", 2)) {                          /* soft error: the reference prints and carries on (hash10x.c:1257-1260) */
          fprintf(outFile, "%s\n", e); if (outFile != stdout) fprintf(stderr, "%s\n", e);
        } else die("%s", e);
      } else {
        if (!codeMin) codeMin = 1;
        if (!codeMax) { h10x_sizes z; h10x_get_sizes(h10x_session_ctx(s), &z); codeMax = (int)z.nBlocks; }
        say("  clustered codes %d to %d\n", codeMin, codeMax);
      }
    }
    else if (ARGMATCH("--clusterSplit", 1)) { if (h10x_session_clusterSplit(s)) die("%s", h10x_session_error(s)); }
    else if (ARGMATCH("--sortFQB", 3)) { if (h10x_session_sortFQB(s, argv[-2], argv[-1])) die("%s", h10x_session_error(s)); }
    else if (ARGMATCH("--cribBuild", 3)) { if (h10x_session_cribBuild(s, argv[-2], argv[-1], outFile, printTables)) die("%s", h10x_session_error(s)); }
    else if (ARGMATCH("--clusterReport", 3)) {
      if (h10x_session_clusterReport(s, atoi(argv[-2]), atoi(argv[-1]), outFile)) {
        const char *e = h10x_session_error(s);
        if (!strncmp(e, "!!", 2)) { fprintf(outFile, "%s\n", e); if (outFile != stdout) fprintf(stderr, "%s\n", e); } else die("%s", e);
      }
    }
    else if (ARGMATCH("--cribSummary", 1)) { if (h10x_session_cribSummary(s, outFile)) die("%s", h10x_session_error(s)); }
    else if (ARGMATCH("--hashStats", 1)) { if (h10x_session_hashStats(s, outFile)) die("%s", h10x_session_error(s)); }
    else if (ARGMATCH("--codeStats", 1)) { if (h10x_session_codeStats(s, outFile)) die("%s", h10x_session_error(s)); }
    else if (ARGMATCH("--help", 1)) usage(s);
    else if (ARGMATCH("--quit", 1) || ARGMATCH("--exit", 1)) break;
    else die("unknown option/command %s; run without arguments for usage", *argv);

    printf("  "); timeUpdate(stdout, 0); fflush(stdout);
  }
  fprintf(outFile, "total resources used: "); timeUpdate(outFile, 1);
  h10x_session_free(s);
  return 0;
}
