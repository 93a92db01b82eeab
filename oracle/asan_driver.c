/* asan_driver.c — runs the CPU restatement (h10x_oracle.c) under AddressSanitizer / UBSan on a record file: the whole
 * command chain of the hot path incl. --clusterSplit and a .hash round trip. TEST INFRASTRUCTURE (the CPU code
 * is to be run under -fsanitize=address,undefined). usage: asan_driver <fqb> <B> <lo> <hi> <ct> <out.hash> */
#include <stdio.h>
#include <stdlib.h>
#include "h10x_oracle.h"

int main(int argc, char **argv) {
  if (argc != 7) { fprintf(stderr, "usage: asan_driver <fqb> <B> <lo> <hi> <ct> <out.hash>\n"); return 2; }
  FILE *f = fopen(argv[1], "rb"); if (!f) { perror(argv[1]); return 2; }
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  uint32_t *rec = (uint32_t *)malloc(sz > 0 ? (size_t)sz : 4);
  if (fread(rec, 1, (size_t)sz, f) != (size_t)sz) { fprintf(stderr, "short read\n"); return 2; }
  fclose(f);
  char err[256];
  orc_state *o = orc_create(21, 31, 17, atoi(argv[2]), err, sizeof err);
  if (!o) { fprintf(stderr, "%s\n", err); return 1; }
  int rc = orc_read_fqb(o, rec, (uint64_t)sz / 120, 0, 100000);
  if (!rc) rc = orc_depth_range(o, atoi(argv[3]), atoi(argv[4]));
  if (!rc) rc = orc_cluster(o, 1, 0, atoi(argv[5]), 2);
  if (!rc) rc = orc_write_hash(o, argv[6]);
  if (!rc) rc = orc_read_hash(o, argv[6]);
  if (!rc) rc = orc_depth_range(o, atoi(argv[3]), atoi(argv[4]));
  if (!rc) rc = orc_cluster_split(o);
  if (!rc) rc = orc_depth_range(o, atoi(argv[3]), atoi(argv[4]));
  if (!rc) rc = orc_cluster(o, 1, 0, atoi(argv[5]), 1);
  if (!rc) rc = orc_write_hash(o, argv[6]);
  if (rc) fprintf(stderr, "oracle: %s\n", orc_last_error(o));
  orc_destroy(o); free(rec);
  return rc ? 1 : 0;
}
