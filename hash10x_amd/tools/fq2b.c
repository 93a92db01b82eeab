/* fq2b.c — `fq2b-amd`: FASTQ(.gz) -> packed binary records, the step before the hash10x path.
 *
 * Same command surface and byte-for-byte the same output as the reference's fq2b (fq2b.c:108-185):
 *     fq2b-amd [-10x <whitelist>] [-checkId] [-o <out.fqb>] <R1.fastq.gz> [<R2.fastq.gz>]
 * Record = read-1 bases (2 bits each, 16 per U32, first base in the top bits; the last, partial word is NOT left
 * justified: fq2b.c:33-42), read-1 quality bits (1 = Q >= 23, 32 per word, fq2b.c:53-63), then the same
 * for read 2. With 151-base reads that is 10 + 5 + 10 + 5 = 30 words = the 120-byte .fqb record hash10x reads.
 * -10x keeps only read pairs whose 16-base barcode is within one mismatch of a whitelist entry and rewrites it to
 * that entry (fq2b.c:67-104). gzip inflation dominates, so this stays on the host; the record SORT that follows
 * (the reference shells out to bsort) is `hash10x-amd --sortFQB`, on the GPU.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <stdint.h>
#include <zlib.h>

static void die(const char *fmt, ...) {
  va_list ap; va_start(ap, fmt);
  fprintf(stderr, "FATAL ERROR: "); vfprintf(stderr, fmt, ap); fprintf(stderr, "\n");
  va_end(ap);
  exit(-1);
}

static uint32_t basePack[256], qualBit[256];
static void pack_init(void) {
  const char *b = "acgtACGT";
  for (int i = 0; i < 8; ++i) basePack[(unsigned char)b[i]] = (uint32_t)(i % 4);      /* everything else (N) packs as A */
  for (int i = '$' + 20; i < 256; ++i) qualBit[i] = 1;
}
/* full words while more than `per` symbols remain, then ONE word with the rest in its low bits */
static void pack_symbols(const char *s, uint32_t *u, int len, int per, int bits, const uint32_t *map) {
  while (len > per) {
    uint32_t w = 0;
    for (int i = 0; i < per; ++i) w = (w << bits) | map[(unsigned char)*s++];
    *u++ = w; len -= per;
  }
  uint32_t w = 0;
  for (int i = 0; i < len; ++i) w = (w << bits) | map[(unsigned char)*s++];
  *u = w;
}

/* ---- 10x whitelist: table[v] = 1 + 4 * position + base that turns v into the whitelist barcode it is one step from
   (position counted from the last base); 0 = no whitelist barcode within one mismatch. Later lines win. */
static uint8_t *wlTable;
static long nBad, nFixed, nFixBase[16];
static uint32_t set_base(uint32_t u, int code) { --code; const int pos = code / 4, b = code % 4; return (u & ~(3u << (2 * pos))) | ((uint32_t)b << (2 * pos)); }
static void read_whitelist(const char *path) {
  FILE *f = fopen(path, "r");
  if (!f) die("failed to open 10x whitelist file %s\n", path);
  if (!(wlTable = (uint8_t *)calloc((size_t)1 << 32, 1))) die("can't allocate barcode table");
  char s[64]; int n = 0;
  while (!feof(f) && fscanf(f, "%63s\n", s) == 1) {
    ++n;
    if (strlen(s) != 16) die("bad barcode line %d in %s: %s", n, path, s);
    uint32_t u; pack_symbols(s, &u, 16, 16, 2, basePack);
    for (int i = 0; i < 16; ++i) {
      const uint8_t restore = (uint8_t)(1 + i * 4 + ((u >> (2 * i)) & 3));
      for (int j = 0; j < 4; ++j) wlTable[set_base(u, 1 + i * 4 + j)] = restore;
    }
  }
  fclose(f);
  fprintf(stderr, "read %d barcodes from file %s\n", n, path);
}
static int fix_barcode(uint32_t *u) {
  const uint8_t code = wlTable[*u];
  if (!code) { ++nBad; return 0; }
  const uint32_t v = set_base(*u, code);
  if (v != *u) { ++nFixed; ++nFixBase[15 - (code - 1) / 4]; *u = v; }
  return 1;
}

/* one FASTQ entry. The id line is read to its newline; the sequence length is taken from the first entry of a file
   and every later sequence and quality line is read as exactly that many characters plus '\n' (fq2b.c:187-221) */
static long nEntry;
static int read_fastq(gzFile f, char *id, char *seq, char *qual, int *len) {
  ++nEntry;
  int n = 0, c = 0;
  while (!gzeof(f) && n < 1023 && (c = gzgetc(f)) != '\n') { if (c < 0) break; id[n++] = (char)c; }
  if (gzeof(f) || c < 0) return 0;
  if (id[0] != '@') die("fastq id line for entry %ld does not start with @", nEntry);
  if (c != '\n') die("fastq entry %ld id line does not end in \\n", nEntry);
  id[n] = 0;
  if (*len) { if (gzread(f, seq, (unsigned)(*len + 1)) != *len + 1) die("bad seq gzread entry %ld", nEntry); }
  else { int m = 0; while (!gzeof(f) && m < 1023 && (c = gzgetc(f)) != '\n') { if (c < 0) break; seq[m++] = (char)c; } seq[m] = (char)c; *len = m; }
  if (seq[*len] != '\n') die("fastq entry %ld seq line does not end in \\n", nEntry);
  seq[*len] = 0;
  if (gzgetc(f) != '+' || gzgetc(f) != '\n') die("bad + fastq line entry %ld", nEntry);
  if (gzread(f, qual, (unsigned)(*len + 1)) != *len + 1) die("bad qual gzread entry %ld", nEntry);
  if (qual[*len] != '\n') die("fastq entry %ld qual line does not end in \\n", nEntry);
  qual[*len] = 0;
  return 1;
}

int main(int argc, char **argv) {
  FILE *out = stdout;
  pack_init();
  --argc; ++argv;
  while (argc > 2 && argv[0][0] == '-') {
    if (!strcmp(*argv, "-10x")) { read_whitelist(argv[1]); argc -= 2; argv += 2; }
    else if (!strcmp(*argv, "-checkId")) { --argc; ++argv; }                          /* the reference's flag is always on */
    else if (!strcmp(*argv, "-o")) { if (!(out = fopen(argv[1], "wb"))) die("failed to open output file %s", argv[1]); argc -= 2; argv += 2; }
    else die("Unknown arg %s for fq2b - run without args for usage", *argv);
  }
  if (argc < 1 || argc > 2)
    die("Usage: fq2b-amd [opts] <fastq.gz> [<fastq.gz>]\n"
        "  Converts fastq to binary with 2 bits per base, converting N to A (!).\n"
        "  Throws out read names; one quality bit per base (Q >= 23).\n"
        "  If two fastq files are given they are interleaved.\n"
        "Opts: -10x <whitelist file>\n"
        "      -checkId  checks whether id lines match in first and second files\n"
        "      -o <outfile> [standard output]\n"
        "  10x option matches first16bp barcode of read 1 to whitelist.\n"
        "  Only outputs an entry if there is a match after correcting for 1 mismatch\n");
  gzFile f1 = gzopen(argv[0], "r"); if (!f1) die("failed to open %s", argv[0]);
  gzFile f2 = 0;
  if (argc == 2) { f2 = gzopen(argv[1], "r"); if (!f2) die("failed to open %s", argv[1]); }
  gzbuffer(f1, 1 << 20); if (f2) gzbuffer(f2, 1 << 20);
  static char id1[1024], id2[1024], s1[1024], s2[1024], q1[1024], q2[1024];
  static uint32_t u1s[256], u2s[256], u1q[128], u2q[128];
  int len1 = 0, len2 = 0, mismatch = 0; long n = 0;
  while (read_fastq(f1, id1, s1, q1, &len1)) {
    pack_symbols(s1, u1s, len1, 16, 2, basePack); pack_symbols(q1, u1q, len1, 32, 1, qualBit);
    if (f2) {
      if (!read_fastq(f2, id2, s2, q2, &len2)) die("second fastq file terminated early at %ld", n);
      if (!mismatch && strcmp(id1, id2)) { fprintf(stderr, "proceeding despite paired read ids not matching, e.g. %s %s\n", id1, id2); mismatch = 1; }
    }
    if (wlTable && !fix_barcode(u1s)) continue;
    if (f2) { pack_symbols(s2, u2s, len2, 16, 2, basePack); pack_symbols(q2, u2q, len2, 32, 1, qualBit); }
    fwrite(u1s, 4, (size_t)(len1 + 15) / 16, out); fwrite(u1q, 4, (size_t)(len1 + 31) / 32, out);
    if (f2) { fwrite(u2s, 4, (size_t)(len2 + 15) / 16, out); fwrite(u2q, 4, (size_t)(len2 + 31) / 32, out); }
    ++n;
  }
  if (f2) fprintf(stderr, "written %ld read pairs %d + %d bp packed in %d word records\n", n, len1, len2,
                  (len1 + 15) / 16 + (len1 + 31) / 32 + (len2 + 15) / 16 + (len2 + 31) / 32);
  else fprintf(stderr, "written %ld reads %d bp packed in %d word records\n", n, len1, (len1 + 15) / 16 + (len1 + 31) / 32);
  if (wlTable) {
    fprintf(stderr, "%ld (%.1f%%) not matching barcodes were dropped\n", nBad, 100.0 * nBad / (double)(nBad + n));
    fprintf(stderr, "%ld (%.1f%%) of those that matched were error corrected\n", nFixed, 100.0 * nFixed / (double)n);
    fprintf(stderr, "by base position:");
    for (int i = 0; i < 16; ++i) fprintf(stderr, " %ld", nFixBase[i]);
    fprintf(stderr, "\n");
  }
  if (out != stdout) fclose(out);
  gzclose(f1); if (f2) gzclose(f2);
  return 0;
}
