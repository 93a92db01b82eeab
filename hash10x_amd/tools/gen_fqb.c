/* gen_fqb.c — deterministic synthetic 10x linked-read generator writing sorted .fqb records.
 *
 * Produces the input format consumed by `hash10x --readFQB` (reference: fq2b.c:142-160 for the
 * record layout, fq2b.c:33-42 for the 2-bit packing incl. the un-justified last word, and
 * README.md:36-53 for the LRSIM yeast recipe this stands in for; SURVEY.md §8d for the model).
 *
 * Model: diploid genome (haplotype A uniform random, haplotype B = A with a SNP every ~snp bp),
 * split into chromosomes of at most 60 Mb; each barcode holds Poisson(mol) molecules of length
 * clip(Exp(mean L = 50 kb), L/10, 5L) from a random haplotype/position; read pairs are spread along the
 * molecules in proportion to molecule length; insert ~ N(350,50) clipped to [160,1000];
 * read 1 = 16-base barcode + 7 random spacer bases + 128 genomic, read 2 = 151 genomic bases from
 * the other end (reverse strand); per-base substitution errors at rate e. Barcodes are distinct
 * random non-zero 16-mers and records are emitted grouped by barcode in ascending word-0 order.
 *
 * Usage: gen_fqb -o out.fqb [-P pairs] [-C barcodes] [-G genome_bp] [-e err] [-s seed]
 *                [-m mean_molecules] [-S snp_spacing] [-L mean_molecule_len] [-fa prefix]   (writes prefix.A.fa / prefix.B.fa)
 * Also callable in-process:  h10x_gen_fqb(params, out_words)  (see bottom; used through ctypes).
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

typedef struct { uint64_t s[4]; } rng_t;
static uint64_t splitmix64(uint64_t *x) {
  uint64_t z = (*x += 0x9e3779b97f4a7c15ULL);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
static void rng_seed(rng_t *r, uint64_t seed) { for (int i = 0; i < 4; ++i) r->s[i] = splitmix64(&seed); }
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_next(rng_t *r) {           /* xoshiro256** */
  uint64_t *s = r->s, result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
  s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
  return result;
}
static inline double rng_unif(rng_t *r) { return (rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline uint64_t rng_below(rng_t *r, uint64_t n) { return (uint64_t)(rng_unif(r) * (double)n); }
/* No libm on the random path: glibc picks FMA / non-FMA variants of log/exp/cos per CPU, whose last-ulp
   differences would make the "seeded" data set depend on the host. Everything below is plain IEEE
   + - * / (compile with -ffp-contract=off), so a seed gives the same bytes everywhere. */
static double det_log(double x) {                           /* ln x for x in (0, 1], ~1e-15 relative */
  int e = 0;
  while (x < 1.0) { x *= 2.0; --e; }
  while (x >= 2.0) { x *= 0.5; ++e; }                        /* x in [1, 2) */
  const double s = (x - 1.0) / (x + 1.0), s2 = s * s;
  double term = s, sum = 0.0;
  for (int k = 1; k < 60; k += 2) { sum += term / (double)k; term *= s2; }
  return 2.0 * sum + (double)e * 0.6931471805599453;
}
static double rng_exp1(rng_t *r) {                          /* Exp(1) */
  double u = 1.0 - rng_unif(r);                             /* (0, 1] */
  return -det_log(u);
}
static double rng_normal(rng_t *r) {                        /* Irwin-Hall: 12 uniforms - 6 */
  double s = 0.0;
  for (int i = 0; i < 12; ++i) s += rng_unif(r);
  return s - 6.0;
}
static int rng_poisson(rng_t *r, double lambda) {           /* arrivals of a unit-rate process before time lambda */
  int k = 0; double t = rng_exp1(r);
  while (t < lambda) { ++k; t += rng_exp1(r); }
  return k;
}

typedef struct {
  uint64_t pairs;        /* P  */
  uint32_t barcodes;     /* C  */
  uint64_t genome;       /* G bases per haplotype */
  double   err;          /* substitution error per base */
  uint64_t seed;
  double   mean_mol;     /* mean molecules per barcode */
  uint32_t snp_spacing;  /* mean distance between SNPs on haplotype B */
  double   mean_len;     /* mean molecule length (bp); min = mean/10, max = 5*mean */
} h10x_gen_params;

typedef struct { uint8_t hap; uint64_t start; uint32_t len; } mol_t;

static int cmp_u32(const void *a, const void *b) {
  uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return x < y ? -1 : x > y;
}

/* pack len bases (codes 0..3) into words exactly like the reference's seqPack (fq2b.c:33-42):
   full words hold 16 bases MSB-first; the final partial word is filled from the low bits. */
static void pack_bases(const uint8_t *s, int len, uint32_t *u) {
  while (len > 16) {
    uint32_t w = 0; for (int i = 0; i < 16; ++i) w = (w << 2) | *s++;
    *u++ = w; len -= 16;
  }
  uint32_t w = 0; for (int i = 0; i < len; ++i) w = (w << 2) | *s++;
  *u = w;
}

static uint8_t *g_hapA = 0, *g_hapB = 0;

static void build_genome(const h10x_gen_params *p, rng_t *r) {
  uint64_t G = p->genome;
  g_hapA = (uint8_t *)malloc(G); g_hapB = (uint8_t *)malloc(G);
  if (!g_hapA || !g_hapB) { fprintf(stderr, "gen_fqb: out of memory for genome\n"); exit(1); }
  for (uint64_t i = 0; i < G; i += 32) {
    uint64_t x = rng_next(r);
    for (int j = 0; j < 32 && i + j < G; ++j) g_hapA[i + j] = (x >> (2 * j)) & 3;
  }
  memcpy(g_hapB, g_hapA, G);
  uint64_t pos = rng_below(r, p->snp_spacing ? p->snp_spacing : 1);
  while (p->snp_spacing && pos < G) {
    g_hapB[pos] = (g_hapA[pos] + 1 + rng_below(r, 3)) & 3;
    pos += 1 + rng_below(r, 2 * (uint64_t)p->snp_spacing);
  }
}

static void write_fasta(const char *path, const uint8_t *hap, uint64_t G) {
  static const char map[4] = {'A', 'C', 'G', 'T'};
  const uint64_t chrMax = 60000000ULL;
  FILE *f = fopen(path, "w"); if (!f) { perror(path); exit(1); }
  char line[61]; int chr = 0;
  for (uint64_t c0 = 0; c0 < G; c0 += chrMax) {
    uint64_t c1 = c0 + chrMax < G ? c0 + chrMax : G;
    fprintf(f, ">chr%d\n", ++chr);
    for (uint64_t i = c0; i < c1; i += 60) {
      int n = (int)(c1 - i < 60 ? c1 - i : 60);
      for (int j = 0; j < n; ++j) line[j] = map[hap[i + j]];
      line[n] = 0; fprintf(f, "%s\n", line);
    }
  }
  fclose(f);
}

/* Generates P records (30 words each) into out (caller allocates 120*P bytes) or, if out == NULL,
   streams them to f. Returns number of records written. */
static uint64_t generate(const h10x_gen_params *p, uint32_t *out, FILE *f) {
  rng_t r; rng_seed(&r, p->seed);
  build_genome(p, &r);
  const uint64_t G = p->genome; const uint32_t C = p->barcodes;

  /* distinct non-zero barcodes, ascending */
  uint32_t *bc = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)C);
  if (C <= 100000) {
    for (;;) {                                                /* redraw everything on a duplicate: fine while C^2 << 2^33 */
      for (uint32_t i = 0; i < C; ++i) { uint32_t b; do b = (uint32_t)rng_next(&r); while (!b); bc[i] = b; }
      qsort(bc, C, sizeof(uint32_t), cmp_u32);
      int dup = 0; for (uint32_t i = 1; i < C; ++i) if (bc[i] == bc[i - 1]) { dup = 1; break; }
      if (!dup) break;
    }
  } else {
    /* many barcodes: duplicates among C random 32-bit words are certain (birthday bound), so keep the distinct ones
       and top up until there are C (sets of up to 100000 barcodes keep the older scheme and therefore their data) */
    uint32_t have = 0;
    while (have < C) {
      for (uint32_t i = have; i < C; ++i) { uint32_t b; do b = (uint32_t)rng_next(&r); while (!b); bc[i] = b; }
      qsort(bc, C, sizeof(uint32_t), cmp_u32);
      uint32_t u = 0;
      for (uint32_t i = 0; i < C; ++i) if (i == 0 || bc[i] != bc[i - 1]) bc[u++] = bc[i];
      have = u;
    }
  }

  /* molecules per barcode */
  uint32_t *molOff = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)C + 1));
  size_t molCap = (size_t)(C * (p->mean_mol + 1.0)) + 16, nMol = 0;
  mol_t *mols = (mol_t *)malloc(sizeof(mol_t) * molCap);
  double *blen = (double *)malloc(sizeof(double) * (size_t)C); double totLen = 0;
  for (uint32_t b = 0; b < C; ++b) {
    molOff[b] = (uint32_t)nMol;
    int m = rng_poisson(&r, p->mean_mol); if (m < 1) m = 1;
    blen[b] = 0;
    for (int j = 0; j < m; ++j) {
      if (nMol == molCap) { molCap *= 2; mols = (mol_t *)realloc(mols, sizeof(mol_t) * molCap); }
      double L = p->mean_len * rng_exp1(&r);
      if (L < p->mean_len * 0.1) L = p->mean_len * 0.1;
      if (L < 400.0) L = 400.0;
      if (L > p->mean_len * 5.0) L = p->mean_len * 5.0;
      if (L > (double)G) L = (double)G;
      mol_t *mm = &mols[nMol++];
      mm->len = (uint32_t)L; mm->hap = (uint8_t)(rng_next(&r) & 1);
      mm->start = rng_below(&r, G - mm->len + 1);
      blen[b] += L;
    }
    totLen += blen[b];
  }
  molOff[C] = (uint32_t)nMol;

  uint64_t nOut = 0; double carry = 0.0;
  uint8_t s1[151], s2[151], frag[1024];
  uint32_t rec[30];
  for (uint32_t b = 0; b < C; ++b) {
    double want = (double)p->pairs * blen[b] / totLen + carry;
    uint64_t nb = (uint64_t)want; carry = want - (double)nb;
    if (b == C - 1) nb = p->pairs - nOut;            /* hit P exactly */
    if (nOut + nb > p->pairs) nb = p->pairs - nOut;
    uint32_t m0 = molOff[b], m1 = molOff[b + 1];
    for (uint64_t k = 0; k < nb; ++k) {
      /* pick molecule ∝ length */
      double x = rng_unif(&r) * blen[b]; uint32_t mi = m0;
      while (mi + 1 < m1 && x >= (double)mols[mi].len) { x -= (double)mols[mi].len; ++mi; }
      const mol_t *mm = &mols[mi];
      int ins = (int)(350.0 + 50.0 * rng_normal(&r));
      if (ins < 160) ins = 160;
      if (ins > 1000) ins = 1000;
      if ((uint32_t)ins > mm->len) ins = (int)mm->len;
      uint64_t fs = mm->start + rng_below(&r, (uint64_t)mm->len - (uint64_t)ins + 1);
      const uint8_t *hap = mm->hap ? g_hapB : g_hapA;
      int flip = (int)(rng_next(&r) & 1);
      if (!flip) memcpy(frag, hap + fs, (size_t)ins);
      else for (int i = 0; i < ins; ++i) frag[i] = 3 - hap[fs + (uint64_t)(ins - 1 - i)];
      /* read 1: barcode(16) + spacer(7) + 128 genomic */
      for (int i = 0; i < 16; ++i) s1[i] = (bc[b] >> (2 * (15 - i))) & 3;
      uint64_t sp = rng_next(&r); for (int i = 0; i < 7; ++i) s1[16 + i] = (sp >> (2 * i)) & 3;
      for (int i = 0; i < 128; ++i) s1[23 + i] = i < ins ? frag[i] : 0;
      /* read 2: reverse complement of the fragment's far end */
      for (int i = 0; i < 151; ++i) s2[i] = i < ins ? 3 - frag[ins - 1 - i] : 0;
      if (p->err > 0.0) {
        for (int i = 23; i < 151; ++i) if (rng_unif(&r) < p->err) s1[i] = (s1[i] + 1 + rng_below(&r, 3)) & 3;
        for (int i = 0; i < 151; ++i)  if (rng_unif(&r) < p->err) s2[i] = (s2[i] + 1 + rng_below(&r, 3)) & 3;
      }
      pack_bases(s1, 151, &rec[0]);
      pack_bases(s2, 151, &rec[15]);
      /* quality bits: unused by hash10x (hash10x.c:161 passes q1=q2=0); fill like high-quality reads */
      for (int i = 10; i < 14; ++i) rec[i] = 0xffffffffu;
      for (int i = 25; i < 29; ++i) rec[i] = 0xffffffffu;
      rec[14] = rec[29] = 0x007fffffu;
      if (out) memcpy(out + 30 * nOut, rec, 120);
      else if (fwrite(rec, 120, 1, f) != 1) { perror("gen_fqb: write"); exit(1); }
      ++nOut;
    }
  }
  free(bc); free(molOff); free(mols); free(blen);
  return nOut;
}

/* ======================================================================================================================
 * Generator v2: the same model with COUNTER-BASED random streams, so that any part of the data set can be generated on
 * its own and in parallel (OpenMP): the genome from the seed alone (chunk by chunk), the barcode list and the molecules
 * of barcode b from stream (seed, b), and the records of barcode b from another stream of their own. A rank of a sharded
 * run generates only the barcodes of its shard (--barcodes lo:hi / h10x_gen2_fill); the result does not depend on the
 * number of threads or on which ranges are asked for. v1 (one xoshiro stream through everything) stays as it is: the
 * digests pinned in tests/golden/manifest.json are of its data.
 * Differences of the model: haplotype B has exactly one SNP per window of snp_spacing bases (position and base from the
 * window's hash) instead of gaps uniform in [1, 2 snp]; B is not stored but derived where a fragment is cut from it.
 * ====================================================================================================================== */
static inline uint64_t mix64(uint64_t z) { z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }
enum { ST_GENOME = 1, ST_SNP = 2, ST_BARCODES = 3, ST_MOLECULES = 4, ST_READS = 5 };
static void stream_seed(rng_t *r, uint64_t seed, uint64_t kind, uint64_t index) {
  uint64_t x = mix64(seed + 0x9e3779b97f4a7c15ULL * kind) ^ mix64(index * 0xd1342543de82ef95ULL + kind);
  rng_seed(r, x);
}
typedef struct {
  h10x_gen_params p;
  uint8_t *hapA;                 /* G bases */
  uint32_t *bc;                  /* C barcodes, ascending */
  uint64_t *recOff;              /* C + 1: first record of each barcode; recOff[C] = pairs */
} h10x_gen2_plan;

enum { G_CHUNK = 1 << 16 };
static void gen2_genome(h10x_gen2_plan *pl) {
  const uint64_t G = pl->p.genome; const int64_t nChunk = (int64_t)((G + G_CHUNK - 1) / G_CHUNK);
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < nChunk; ++c) {
    rng_t r; stream_seed(&r, pl->p.seed, ST_GENOME, (uint64_t)c);
    const uint64_t i0 = (uint64_t)c * G_CHUNK, i1 = i0 + G_CHUNK < G ? i0 + G_CHUNK : G;
    for (uint64_t i = i0; i < i1; i += 32) {
      uint64_t x = rng_next(&r);
      for (int j = 0; j < 32 && i + j < i1; ++j) pl->hapA[i + j] = (x >> (2 * j)) & 3;
    }
  }
}
/* haplotype B differs from A at one position per window of snp_spacing bases */
static inline void gen2_snp_of_window(const h10x_gen2_plan *pl, uint64_t w, uint64_t *pos, unsigned *delta) {
  const uint64_t h = mix64(mix64(pl->p.seed + 0x9e3779b97f4a7c15ULL * ST_SNP) ^ (w * 0xd1342543de82ef95ULL));
  *pos = w * pl->p.snp_spacing + h % pl->p.snp_spacing; *delta = 1 + (unsigned)((h >> 40) % 3);
}
static void gen2_fetch(const h10x_gen2_plan *pl, int hap, uint64_t start, int len, uint8_t *dst) {
  memcpy(dst, pl->hapA + start, (size_t)len);
  if (!hap || !pl->p.snp_spacing) return;
  const uint64_t w0 = start / pl->p.snp_spacing, w1 = (start + (uint64_t)len - 1) / pl->p.snp_spacing;
  for (uint64_t w = w0; w <= w1; ++w) {
    uint64_t pos; unsigned d; gen2_snp_of_window(pl, w, &pos, &d);
    if (pos >= start && pos < start + (uint64_t)len && pos < pl->p.genome) dst[pos - start] = (uint8_t)((dst[pos - start] + d) & 3);
  }
}
static int gen2_molecules(const h10x_gen2_plan *pl, uint32_t b, mol_t *mols, int cap, double *blen) {   /* molecules of barcode b */
  const h10x_gen_params *p = &pl->p; const uint64_t G = p->genome;
  rng_t r; stream_seed(&r, p->seed, ST_MOLECULES, b);
  int m = rng_poisson(&r, p->mean_mol); if (m < 1) m = 1; if (m > cap) m = cap;
  double tot = 0;
  for (int j = 0; j < m; ++j) {
    double L = p->mean_len * rng_exp1(&r);
    if (L < p->mean_len * 0.1) L = p->mean_len * 0.1;
    if (L < 400.0) L = 400.0;
    if (L > p->mean_len * 5.0) L = p->mean_len * 5.0;
    if (L > (double)G) L = (double)G;
    mols[j].len = (uint32_t)L; mols[j].hap = (uint8_t)(rng_next(&r) & 1);
    mols[j].start = rng_below(&r, G - mols[j].len + 1);
    tot += L;
  }
  *blen = tot;
  return m;
}
enum { MOL_CAP = 256 };
void h10x_gen2_free(h10x_gen2_plan *pl) { if (!pl) return; free(pl->hapA); free(pl->bc); free(pl->recOff); free(pl); }
/* everything that does not grow with the read count: genome, barcodes, records per barcode (from the molecule lengths of ALL
   barcodes: the same figures whichever part of the set is filled in later) */
h10x_gen2_plan *h10x_gen2_plan_new(const h10x_gen_params *p) {
  h10x_gen2_plan *pl = (h10x_gen2_plan *)calloc(1, sizeof *pl); if (!pl) return 0;
  pl->p = *p; const uint32_t C = p->barcodes;
  pl->hapA = (uint8_t *)malloc(p->genome ? p->genome : 1); pl->bc = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)C + 1));
  pl->recOff = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)C + 1));
  double *blen = (double *)malloc(sizeof(double) * ((size_t)C + 1));
  if (!pl->hapA || !pl->bc || !pl->recOff || !blen) { free(blen); h10x_gen2_free(pl); return 0; }
  gen2_genome(pl);
  { rng_t r; stream_seed(&r, p->seed, ST_BARCODES, 0);                              /* distinct non-zero words, ascending */
    uint32_t have = 0;
    while (have < C) {
      for (uint32_t i = have; i < C; ++i) { uint32_t w; do w = (uint32_t)rng_next(&r); while (!w); pl->bc[i] = w; }
      qsort(pl->bc, C, sizeof(uint32_t), cmp_u32);
      uint32_t u = 0;
      for (uint32_t i = 0; i < C; ++i) if (i == 0 || pl->bc[i] != pl->bc[i - 1]) pl->bc[u++] = pl->bc[i];
      have = u;
    } }
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < (int64_t)C; ++b) { mol_t mols[MOL_CAP]; gen2_molecules(pl, (uint32_t)b, mols, MOL_CAP, &blen[b]); }
  double totLen = 0; for (uint32_t b = 0; b < C; ++b) totLen += blen[b];              /* in index order: the same sum on every host */
  uint64_t nOut = 0; double carry = 0.0;
  for (uint32_t b = 0; b < C; ++b) {
    double want = (double)p->pairs * blen[b] / totLen + carry;
    uint64_t nb = (uint64_t)want; carry = want - (double)nb;
    if (b == C - 1) nb = p->pairs - nOut;
    if (nOut + nb > p->pairs) nb = p->pairs - nOut;
    pl->recOff[b] = nOut; nOut += nb;
  }
  pl->recOff[C] = nOut;
  free(blen);
  return pl;
}
const uint64_t *h10x_gen2_offsets(const h10x_gen2_plan *pl) { return pl->recOff; }
/* first barcode whose records start at or after `record` (shard cuts: rank r of N takes barcodes [cut(r P / N), cut((r + 1) P / N))) */
uint32_t h10x_gen2_cut(const h10x_gen2_plan *pl, uint64_t record) {
  uint32_t lo = 0, hi = pl->p.barcodes;
  while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (pl->recOff[mid] < record) lo = mid + 1; else hi = mid; }
  return lo;
}
/* the records of barcodes [bLo, bHi) into out (30 words each, recOff[bHi] - recOff[bLo] of them); returns their number */
uint64_t h10x_gen2_fill(const h10x_gen2_plan *pl, uint32_t bLo, uint32_t bHi, uint32_t *out) {
  const h10x_gen_params *p = &pl->p;
  if (bHi > p->barcodes) bHi = p->barcodes;
  if (bLo >= bHi) return 0;
  const uint64_t base = pl->recOff[bLo];
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t bb = (int64_t)bLo; bb < (int64_t)bHi; ++bb) {
    const uint32_t b = (uint32_t)bb;
    mol_t mols[MOL_CAP]; double blen; const int m = gen2_molecules(pl, b, mols, MOL_CAP, &blen);
    rng_t r; stream_seed(&r, p->seed, ST_READS, b);
    uint8_t s1[151], s2[151], frag[1024], tmp[1024]; uint32_t rec[30];
    const uint64_t nb = pl->recOff[b + 1] - pl->recOff[b];
    for (uint64_t k = 0; k < nb; ++k) {
      double x = rng_unif(&r) * blen; int mi = 0;
      while (mi + 1 < m && x >= (double)mols[mi].len) { x -= (double)mols[mi].len; ++mi; }
      const mol_t *mm = &mols[mi];
      int ins = (int)(350.0 + 50.0 * rng_normal(&r));
      if (ins < 160) ins = 160;
      if (ins > 1000) ins = 1000;
      if ((uint32_t)ins > mm->len) ins = (int)mm->len;
      const uint64_t fs = mm->start + rng_below(&r, (uint64_t)mm->len - (uint64_t)ins + 1);
      const int flip = (int)(rng_next(&r) & 1);
      if (!flip) gen2_fetch(pl, mm->hap, fs, ins, frag);
      else { gen2_fetch(pl, mm->hap, fs, ins, tmp); for (int i = 0; i < ins; ++i) frag[i] = 3 - tmp[ins - 1 - i]; }
      for (int i = 0; i < 16; ++i) s1[i] = (pl->bc[b] >> (2 * (15 - i))) & 3;
      const uint64_t sp = rng_next(&r); for (int i = 0; i < 7; ++i) s1[16 + i] = (sp >> (2 * i)) & 3;
      for (int i = 0; i < 128; ++i) s1[23 + i] = i < ins ? frag[i] : 0;
      for (int i = 0; i < 151; ++i) s2[i] = i < ins ? 3 - frag[ins - 1 - i] : 0;
      if (p->err > 0.0) {
        for (int i = 23; i < 151; ++i) if (rng_unif(&r) < p->err) s1[i] = (s1[i] + 1 + rng_below(&r, 3)) & 3;
        for (int i = 0; i < 151; ++i)  if (rng_unif(&r) < p->err) s2[i] = (s2[i] + 1 + rng_below(&r, 3)) & 3;
      }
      pack_bases(s1, 151, &rec[0]);
      pack_bases(s2, 151, &rec[15]);
      for (int i = 10; i < 14; ++i) rec[i] = 0xffffffffu;
      for (int i = 25; i < 29; ++i) rec[i] = 0xffffffffu;
      rec[14] = rec[29] = 0x007fffffu;
      memcpy(out + 30 * (pl->recOff[b] - base + k), rec, 120);
    }
  }
  return pl->recOff[bHi] - base;
}
/* the two haplotypes as FASTA (the truth genomes of --cribBuild) */
static void gen2_write_fasta(const h10x_gen2_plan *pl, const char *path, int hap) {
  const uint64_t G = pl->p.genome;
  uint8_t *h = (uint8_t *)malloc(G ? G : 1); if (!h) { fprintf(stderr, "gen_fqb: out of memory\n"); exit(1); }
  for (uint64_t i = 0; i < G; i += 1 << 20) { const uint64_t n = G - i < (1u << 20) ? G - i : (1u << 20); gen2_fetch(pl, hap, i, (int)n, h + i); }
  write_fasta(path, h, G);
  free(h);
}

/* one haplotype of the truth genome as base codes 0..3 — what readSequence() hands cribAddGenome (hash10x.c:426-453) for the FASTA gen2_write_fasta writes, whose
   sequences are the 60 Mb pieces [k * 60e6, (k + 1) * 60e6) of it: out[genome]. For callers that feed the crib through the C ABI without a file (bench.py). */
void h10x_gen2_haplotype(const h10x_gen2_plan *pl, int hap, uint8_t *out) {
  const int64_t G = (int64_t)pl->p.genome;
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t i = 0; i < G; i += 1 << 20) { const int64_t n = G - i < (1 << 20) ? G - i : (1 << 20); gen2_fetch(pl, hap, (uint64_t)i, (int)n, out + i); }
}

/* in-process entry point (ctypes): fills out[30*pairs]; keeps genome until h10x_gen_free() */
uint64_t h10x_gen_fqb(const h10x_gen_params *p, uint32_t *out) {
  if (g_hapA) { free(g_hapA); free(g_hapB); g_hapA = g_hapB = 0; }
  return generate(p, out, 0);
}
void h10x_gen_free(void) { free(g_hapA); free(g_hapB); g_hapA = g_hapB = 0; }

#ifndef H10X_GEN_NO_MAIN
int main(int argc, char **argv) {
  h10x_gen_params p = {2500000ULL, 10000u, 12000000ULL, 0.005, 1ULL, 10.0, 150u, 50000.0};
  const char *outPath = 0, *fa = 0; int version = 1; long bLo = 0, bHi = -1;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "-o") && i + 1 < argc) outPath = argv[++i];
    else if (!strcmp(argv[i], "-v") && i + 1 < argc) version = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--barcodes") && i + 1 < argc) { if (sscanf(argv[++i], "%ld:%ld", &bLo, &bHi) != 2) { fprintf(stderr, "gen_fqb: --barcodes lo:hi\n"); return 2; } }
    else if (!strcmp(argv[i], "-P") && i + 1 < argc) p.pairs = strtoull(argv[++i], 0, 10);
    else if (!strcmp(argv[i], "-C") && i + 1 < argc) p.barcodes = (uint32_t)strtoul(argv[++i], 0, 10);
    else if (!strcmp(argv[i], "-G") && i + 1 < argc) p.genome = strtoull(argv[++i], 0, 10);
    else if (!strcmp(argv[i], "-e") && i + 1 < argc) p.err = atof(argv[++i]);
    else if (!strcmp(argv[i], "-s") && i + 1 < argc) p.seed = strtoull(argv[++i], 0, 10);
    else if (!strcmp(argv[i], "-m") && i + 1 < argc) p.mean_mol = atof(argv[++i]);
    else if (!strcmp(argv[i], "-S") && i + 1 < argc) p.snp_spacing = (uint32_t)strtoul(argv[++i], 0, 10);
    else if (!strcmp(argv[i], "-L") && i + 1 < argc) p.mean_len = atof(argv[++i]);
    else if (!strcmp(argv[i], "-fa") && i + 1 < argc) fa = argv[++i];
    else { fprintf(stderr, "usage: gen_fqb -o out.fqb [-P pairs] [-C barcodes] [-G genome] [-e err] "
                           "[-s seed] [-m molecules] [-S snp] [-L mean_mol_len] [-fa prefix] [-v 2 [--barcodes lo:hi]]\n"); return 2; }
  }
  if (!outPath) { fprintf(stderr, "gen_fqb: -o required\n"); return 2; }
  if (version == 2) {                                       /* counter-based streams, OpenMP, any barcode range on its own */
    h10x_gen2_plan *pl = h10x_gen2_plan_new(&p);
    if (!pl) { fprintf(stderr, "gen_fqb: out of memory\n"); return 1; }
    if (bHi < 0 || bHi > (long)p.barcodes) bHi = (long)p.barcodes;
    if (bLo < 0) bLo = 0;
    if (bLo > bHi) bLo = bHi;
    const uint64_t n = pl->recOff[bHi] - pl->recOff[bLo];
    uint32_t *out = (uint32_t *)malloc(n ? n * 120 : 120);
    if (!out) { fprintf(stderr, "gen_fqb: out of memory for %llu records\n", (unsigned long long)n); return 1; }
    h10x_gen2_fill(pl, (uint32_t)bLo, (uint32_t)bHi, out);
    FILE *f2 = fopen(outPath, "wb"); if (!f2) { perror(outPath); return 1; }
    if (n && fwrite(out, 120, n, f2) != n) { perror("gen_fqb: write"); return 1; }
    fclose(f2); free(out);
    if (fa) { char path[4096]; snprintf(path, sizeof path, "%s.A.fa", fa); gen2_write_fasta(pl, path, 0); snprintf(path, sizeof path, "%s.B.fa", fa); gen2_write_fasta(pl, path, 1); }
    fprintf(stderr, "gen_fqb v2: wrote %llu read pairs of barcodes %ld:%ld of %u (data set: %llu pairs), genome %llu x2, err %g, seed %llu\n",
            (unsigned long long)n, bLo, bHi, p.barcodes, (unsigned long long)p.pairs, (unsigned long long)p.genome, p.err, (unsigned long long)p.seed);
    h10x_gen2_free(pl);
    return 0;
  }
  FILE *f = fopen(outPath, "wb"); if (!f) { perror(outPath); return 1; }
  static char buf[1 << 22]; setvbuf(f, buf, _IOFBF, sizeof buf);
  uint64_t n = generate(&p, 0, f);
  fclose(f);
  if (fa) {
    char path[4096];
    snprintf(path, sizeof path, "%s.A.fa", fa); write_fasta(path, g_hapA, p.genome);
    snprintf(path, sizeof path, "%s.B.fa", fa); write_fasta(path, g_hapB, p.genome);
  }
  fprintf(stderr, "gen_fqb: wrote %llu read pairs, %u barcodes, genome %llu x2, err %g, seed %llu\n",
          (unsigned long long)n, p.barcodes, (unsigned long long)p.genome, p.err, (unsigned long long)p.seed);
  return 0;
}
#endif
