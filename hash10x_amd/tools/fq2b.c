/* fq2b.c — `fq2b-amd`: FASTQ(.gz) -> packed binary read records, the producer of hash10x's input.
 *
 * Command surface and output bytes of the reference's fq2b (fq2b.c:108-185):
 *     fq2b-amd [-10x <whitelist>] [-checkId] [-o <out.fqb>] <R1.fastq.gz> [<R2.fastq.gz>]
 * Record layout (what hash10x --readFQB consumes): per read ceil(L/16) words of 2-bit bases, first base in
 * the top bits of a word, then ceil(L/32) words of 1-bit qualities (1 = Phred >= 23); the last word of each run holds its
 * leftover symbols in its LOW bits. 151 + 151 bases => 30 words = 120 bytes.
 *
 * Built differently from the reference:
 *   - input: each file is inflated in 4 MiB blocks and cut into lines with memchr (no per-character gzgetc, no fixed-length
 *     record reads); a read of another length than the file's first one is an error, as the record size is fixed;
 *   - packing: one table maps a character to its symbol, words are filled by a single routine for bases and qualities;
 *   - -10x: the whitelist is an open-addressing set of packed barcodes remembering the LAST line each one appeared on; a
 *     read's barcode is looked up together with its 48 one-substitution neighbours and the candidate from the latest
 *     whitelist line wins. That is exactly what the reference's 4 GiB byte table (later lines overwrite earlier ones,
 *     fq2b.c:71-104) returns, for exact matches too, in 16 bytes per whitelist barcode instead of 4 GiB.
 * gzip inflation dominates the run time, so this step stays on the host; the record sort that follows (the reference
 * shells out to bsort) is `hash10x-amd --sortFQB`, on the GPU.
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

/* ------------------------------------------------------------------------------------------ line source */
enum { BLOCK = 4 << 20 };
typedef struct {
  gzFile gz; const char *path;
  char *buf; size_t have, pos;      /* buf[pos .. have) is unread */
  int eof; long entry;
} Source;

static void source_open(Source *s, const char *path) {
  memset(s, 0, sizeof *s);
  s->path = path;
  if (!(s->gz = gzopen(path, "r"))) die("failed to open %s", path);
  gzbuffer(s->gz, 1 << 20);
  if (!(s->buf = (char *)malloc(BLOCK + 1))) die("out of memory");
}
/* next line without its '\n'; *len receives its length. NULL at end of input. A last line without '\n' is returned too
   (*terminated = 0) so that the caller can complain the way it wants. The pointer is valid until the next call. */
static char *source_line(Source *s, size_t *len, int *terminated) {
  for (;;) {
    char *nl = s->have > s->pos ? (char *)memchr(s->buf + s->pos, '\n', s->have - s->pos) : 0;
    if (nl) { char *line = s->buf + s->pos; *len = (size_t)(nl - line); s->pos += *len + 1; *terminated = 1; return line; }
    if (s->eof) {
      if (s->pos == s->have) return 0;
      char *line = s->buf + s->pos; *len = s->have - s->pos; s->pos = s->have; *terminated = 0; return line;
    }
    if (s->have - s->pos >= BLOCK / 2) die("%s: line longer than %d bytes", s->path, BLOCK / 2);
    memmove(s->buf, s->buf + s->pos, s->have - s->pos); s->have -= s->pos; s->pos = 0;
    const int got = gzread(s->gz, s->buf + s->have, (unsigned)(BLOCK - s->have));
    if (got < 0) die("%s: read error", s->path);
    if (got == 0) s->eof = 1;
    s->have += (size_t)got;
  }
}

typedef struct { const char *id, *seq, *qual; size_t idLen, len; } Entry;
/* the four lines of one FASTQ entry; 0 at a clean end of file. Pointers of one entry stay valid together only if the
   entry does not straddle a refill, so the id is copied out by the caller when it needs it. */
static int source_entry(Source *s, Entry *e, char *idCopy, size_t idCap) {
  size_t n; int t;
  ++s->entry;
  char *l = source_line(s, &n, &t);
  if (!l) return 0;
  if (!n || l[0] != '@') die("fastq id line for entry %ld does not start with @", s->entry);
  if (!t) die("fastq entry %ld id line does not end in \\n", s->entry);
  if (n >= idCap) n = idCap - 1;
  memcpy(idCopy, l, n); idCopy[n] = 0; e->id = idCopy; e->idLen = n;
  /* sequence, '+', quality: make sure all three are in the buffer at once by peeking for three newlines first */
  for (;;) {
    size_t p = s->pos; int found = 0;
    while (found < 3) { char *nl = s->have > p ? (char *)memchr(s->buf + p, '\n', s->have - p) : 0; if (!nl) break; p = (size_t)(nl - s->buf) + 1; ++found; }
    if (found == 3 || s->eof) break;
    if (s->have - s->pos >= BLOCK / 2) die("%s: entry %ld longer than %d bytes", s->path, s->entry, BLOCK / 2);
    memmove(s->buf, s->buf + s->pos, s->have - s->pos); s->have -= s->pos; s->pos = 0;
    const int got = gzread(s->gz, s->buf + s->have, (unsigned)(BLOCK - s->have));
    if (got < 0) die("%s: read error", s->path);
    if (got == 0) s->eof = 1;
    s->have += (size_t)got;
  }
  l = source_line(s, &n, &t);
  if (!l || !t) die("fastq entry %ld seq line does not end in \\n", s->entry);
  e->seq = l; e->len = n;
  l = source_line(s, &n, &t);
  if (!l || !t || n != 1 || l[0] != '+') die("bad + fastq line entry %ld", s->entry);
  l = source_line(s, &n, &t);
  if (!l || !t || n != e->len) die("fastq entry %ld qual line does not end in \\n", s->entry);
  e->qual = l;
  return 1;
}

/* ------------------------------------------------------------------------------------------ packing */
static uint8_t baseSym[256], qualSym[256];
static void tables_init(void) {
  baseSym['c'] = baseSym['C'] = 1; baseSym['g'] = baseSym['G'] = 2; baseSym['t'] = baseSym['T'] = 3;   /* a, A, N, anything else: 0 */
  for (int c = '$' + 20; c < 256; ++c) qualSym[c] = 1;                                                   /* Phred+33 >= 23 */
}
/* n symbols of `bits` bits, perWord to a word; returns the words written. Full words come first; the last word takes
   what is left (1..perWord symbols) right-aligned, which for a multiple of perWord is again a full word. */
static size_t pack_run(const char *s, size_t n, int bits, int perWord, const uint8_t *sym, uint32_t *out) {
  if (!n) return 0;
  const size_t words = (n + (size_t)perWord - 1) / (size_t)perWord;
  size_t i = 0;
  for (size_t w = 0; w < words; ++w) {
    const size_t take = w + 1 < words ? (size_t)perWord : n - i;
    uint32_t v = 0;
    for (size_t j = 0; j < take; ++j) v = (v << bits) | sym[(unsigned char)s[i + j]];
    out[w] = v; i += take;
  }
  return words;
}
static size_t pack_read(const Entry *e, uint32_t *out) {
  size_t w = pack_run(e->seq, e->len, 2, 16, baseSym, out);
  return w + pack_run(e->qual, e->len, 1, 32, qualSym, out + w);
}

/* ------------------------------------------------------------------------------------------ 10x whitelist */
typedef struct { uint32_t key; uint32_t line; } Slot;              /* line 0 = empty */
static Slot *wl; static uint64_t wlMask; static int haveWhitelist;
static inline uint64_t wl_home(uint32_t k) { return ((uint64_t)k * 0x9E3779B97F4A7C15ull) >> 20 & wlMask; }
static void wl_put(uint32_t key, uint32_t line) {
  uint64_t p = wl_home(key);
  while (wl[p].line && wl[p].key != key) p = (p + 1) & wlMask;
  wl[p].key = key; wl[p].line = line;                               /* a repeated barcode keeps its latest line */
}
static uint32_t wl_line(uint32_t key) {
  uint64_t p = wl_home(key);
  while (wl[p].line) { if (wl[p].key == key) return wl[p].line; p = (p + 1) & wlMask; }
  return 0;
}
static void whitelist_load(const char *path) {
  FILE *f = fopen(path, "r");
  if (!f) die("failed to open 10x whitelist file %s\n", path);
  size_t cap = 1 << 16, n = 0; uint32_t *codes = (uint32_t *)malloc(cap * 4);
  char word[64];
  while (fscanf(f, "%63s", word) == 1) {
    if (strlen(word) != 16) die("bad barcode line %d in %s: %s", (int)n + 1, path, word);
    if (n == cap) { cap *= 2; codes = (uint32_t *)realloc(codes, cap * 4); }
    if (!codes) die("out of memory");
    pack_run(word, 16, 2, 16, baseSym, &codes[n++]);
  }
  fclose(f);
  uint64_t slots = 1024; while (slots < 2 * n + 2) slots *= 2;
  wlMask = slots - 1;
  if (!(wl = (Slot *)calloc(slots, sizeof(Slot)))) die("can't allocate barcode table");
  for (size_t i = 0; i < n; ++i) wl_put(codes[i], (uint32_t)i + 1);
  free(codes);
  haveWhitelist = 1;
  fprintf(stderr, "read %d barcodes from file %s\n", (int)n, path);
}
static long nDropped, nCorrected, correctedAt[16];                 /* correctedAt[i]: base i of the barcode, 0 = first */
/* the whitelist barcode at most one substitution away that appears on the latest whitelist line; 0 = none */
static int whitelist_match(uint32_t *barcode) {
  const uint32_t v = *barcode;
  uint32_t bestLine = wl_line(v), best = v; int bestBase = -1;
  for (int i = 0; i < 16; ++i) {
    const int sh = 2 * (15 - i);
    for (uint32_t b = 0; b < 4; ++b) {
      const uint32_t u = (v & ~(3u << sh)) | (b << sh);
      if (u == v) continue;
      const uint32_t ln = wl_line(u);
      if (ln > bestLine) { bestLine = ln; best = u; bestBase = i; }
    }
  }
  if (!bestLine) { ++nDropped; return 0; }
  if (bestBase >= 0) { ++nCorrected; ++correctedAt[bestBase]; *barcode = best; }
  return 1;
}

/* ------------------------------------------------------------------------------------------ main */
static const char usageText[] =
  "Usage: fq2b-amd [opts] <fastq.gz> [<fastq.gz>]\n"
  "  Converts fastq to binary with 2 bits per base, converting N to A (!).\n"
  "  Throws out read names; one quality bit per base (Q >= 23).\n"
  "  If two fastq files are given they are interleaved.\n"
  "Opts: -10x <whitelist file>\n"
  "      -checkId  checks whether id lines match in first and second files\n"
  "      -o <outfile> [standard output]\n"
  "  10x option matches first16bp barcode of read 1 to whitelist.\n"
  "  Only outputs an entry if there is a match after correcting for 1 mismatch\n";

int main(int argc, char **argv) {
  const char *outPath = 0, *wlPath = 0;
  int a = 1;
  tables_init();
  for (; argc - a > 2 && argv[a][0] == '-'; ) {
    if (!strcmp(argv[a], "-10x")) { wlPath = argv[a + 1]; a += 2; }
    else if (!strcmp(argv[a], "-o")) { outPath = argv[a + 1]; a += 2; }
    else if (!strcmp(argv[a], "-checkId")) a += 1;                 /* ids are always compared, as in the reference */
    else die("Unknown arg %s for fq2b - run without args for usage", argv[a]);
  }
  const int nFiles = argc - a;
  if (nFiles < 1 || nFiles > 2) die("%s", usageText);
  if (wlPath) whitelist_load(wlPath);
  FILE *out = stdout;
  if (outPath && !(out = fopen(outPath, "wb"))) die("failed to open output file %s", outPath);
  setvbuf(out, 0, _IOFBF, 1 << 20);

  Source src[2]; Entry e[2]; static char id[2][1024];
  for (int i = 0; i < nFiles; ++i) source_open(&src[i], argv[a + i]);
  size_t readLen[2] = {0, 0}; int lenKnown = 0, idWarned = 0; long written = 0;
  uint32_t *rec = (uint32_t *)malloc(4 * (size_t)BLOCK / 8);       /* a line is at most BLOCK/2 symbols */
  if (!rec) die("out of memory");
  while (source_entry(&src[0], &e[0], id[0], sizeof id[0])) {
    /* read 1 is packed now: its line pointers die when file 1 is refilled, not when file 2 is, but keep the order simple */
    size_t words = pack_read(&e[0], rec);
    const size_t len1 = e[0].len;
    if (nFiles == 2) {
      if (!source_entry(&src[1], &e[1], id[1], sizeof id[1])) die("second fastq file terminated early at %ld", written);
      if (!idWarned && strcmp(id[0], id[1])) { fprintf(stderr, "proceeding despite paired read ids not matching, e.g. %s %s\n", id[0], id[1]); idWarned = 1; }
    }
    if (!lenKnown) { readLen[0] = len1; if (nFiles == 2) readLen[1] = e[1].len; lenKnown = 1; }
    if (len1 != readLen[0]) die("fastq entry %ld seq line does not end in \\n", src[0].entry);          /* fixed-size records */
    if (nFiles == 2 && e[1].len != readLen[1]) die("fastq entry %ld seq line does not end in \\n", src[1].entry);
    if (haveWhitelist && !whitelist_match(&rec[0])) continue;
    if (nFiles == 2) words += pack_read(&e[1], rec + words);
    if (fwrite(rec, 4, words, out) != words) die("write failed");
    ++written;
  }
  const int w1 = (int)((readLen[0] + 15) / 16 + (readLen[0] + 31) / 32), w2 = (int)((readLen[1] + 15) / 16 + (readLen[1] + 31) / 32);
  if (nFiles == 2) fprintf(stderr, "written %ld read pairs %d + %d bp packed in %d word records\n", written, (int)readLen[0], (int)readLen[1], w1 + w2);
  else fprintf(stderr, "written %ld reads %d bp packed in %d word records\n", written, (int)readLen[0], w1);
  if (haveWhitelist) {
    fprintf(stderr, "%ld (%.1f%%) not matching barcodes were dropped\n", nDropped, 100.0 * nDropped / (double)(nDropped + written));
    fprintf(stderr, "%ld (%.1f%%) of those that matched were error corrected\n", nCorrected, 100.0 * nCorrected / (double)written);
    fprintf(stderr, "by base position:");
    for (int i = 0; i < 16; ++i) fprintf(stderr, " %ld", correctedAt[i]);
    fprintf(stderr, "\n");
  }
  if (out != stdout && fclose(out)) die("write failed");
  for (int i = 0; i < nFiles; ++i) { gzclose(src[i].gz); free(src[i].buf); }
  free(rec); free(wl);
  return 0;
}
