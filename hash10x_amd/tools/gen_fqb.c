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

/* in-process entry point (ctypes): fills out[30*pairs]; keeps genome until h10x_gen_free() */
uint64_t h10x_gen_fqb(const h10x_gen_params *p, uint32_t *out) {
  if (g_hapA) { free(g_hapA); free(g_hapB); g_hapA = g_hapB = 0; }
  return generate(p, out, 0);
}
void h10x_gen_free(void) { free(g_hapA); free(g_hapB); g_hapA = g_hapB = 0; }

#ifndef H10X_GEN_NO_MAIN
int main(int argc, char **argv) {
  h10x_gen_params p = {2500000ULL, 10000u, 12000000ULL, 0.005, 1ULL, 10.0, 150u, 50000.0};
  const char *outPath = 0, *fa = 0;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "-o") && i + 1 < argc) outPath = argv[++i];
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
                           "[-s seed] [-m molecules] [-S snp] [-L mean_mol_len] [-fa prefix]\n"); return 2; }
  }
  if (!outPath) { fprintf(stderr, "gen_fqb: -o required\n"); return 2; }
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
