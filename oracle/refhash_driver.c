/* refhash_driver.c — TEST INFRASTRUCTURE. Feeds a seeded key sequence through the REFERENCE's HASH object (hash.c, compiled from /root/reference where it lies:
 * oracle/Makefile target _ref/refhash) exactly as cribSummary() does (hash10x.c:1038-1046: hashAdd(h, HASH_INT(key)), then hashCount) and prints the count.
 * tests/test_host_cpu.py compares host/h10x_host.c's restatement (h10x_host_refhash_count) with it; the golden counts travel in tests/golden/manifest.json.
 *   refhash <n> <distinct> <seed>     keys = 1 + (splitmix64 stream mod distinct), n of them */
#include "utils.h"
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
static uint64_t sm(uint64_t *x) { uint64_t z = (*x += 0x9e3779b97f4a7c15ULL); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }
int main(int argc, char **argv) {
  if (argc != 4) { fprintf(stderr, "usage: refhash n distinct seed\n"); return 2; }
  const long n = atol(argv[1]); const uint64_t distinct = strtoull(argv[2], 0, 10); uint64_t s = strtoull(argv[3], 0, 10);
  HASH h = hashCreate(1 << 20);
  for (long i = 0; i < n; ++i) { const int key = 1 + (int)(sm(&s) % distinct); hashAdd(h, HASH_INT(key)); }
  printf("%d\n", hashCount(h));
  return 0;
}
