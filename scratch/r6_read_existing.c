/* scratch/r6_read_existing.c — read an EXISTING file (written by somebody else) with T threads, 4 MiB pieces handed out in order, as --readFQB's reader pool does;
   by pread, or (first argument "mmap") by memcpy out of a MAP_SHARED mapping. The FIRST read of a file fresh in memory-backed storage is what matters (round 6: pread 15 GB/s
   the first time, 150-250 the second: the pages' first access goes through the LRU lists under a lock).
   gcc -O2 -pthread scratch/r6_read_existing.c -o /tmp/r6_read_existing && /tmp/r6_read_existing [mmap] /dev/shm/x.fqb 16 16 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static int fd; static uint64_t size, piece = 4u << 20; static volatile uint64_t next_piece; static const char *map;
static void *job(void *p) {
  char *buf; if (posix_memalign((void **)&buf, 4096, piece)) return 0; memset(buf, 1, piece);
  for (;;) {
    const uint64_t k = __sync_fetch_and_add(&next_piece, 1), at = k * piece;
    if (at >= size) break;
    const uint64_t len = size - at < piece ? size - at : piece; uint64_t g = 0;
    if (map) { memcpy(buf, map + at, len); g = len; }
    while (g < len) { ssize_t r = pread(fd, buf + g, len - g, (off_t)(at + g)); if (r <= 0) { perror("pread"); exit(1); } g += (uint64_t)r; }
  }
  free(buf); (void)p; return 0;
}
int main(int argc, char **argv) {
  if (argc < 2) return 1;
  const int useMap = !strcmp(argv[1], "mmap"); if (useMap) { ++argv; --argc; }
  fd = open(argv[1], O_RDONLY); if (fd < 0) { perror(argv[1]); return 1; }
  struct stat sb; fstat(fd, &sb); size = (uint64_t)sb.st_size;
  if (useMap) { map = mmap(0, size, PROT_READ, MAP_SHARED, fd, 0); if (map == MAP_FAILED) { perror("mmap"); return 1; } }
  for (int a = 2; a < argc; ++a) {
    const int T = atoi(argv[a]); pthread_t th[64]; next_piece = 0;
    const double t0 = now();
    for (int i = 0; i < T; ++i) pthread_create(&th[i], 0, job, 0);
    for (int i = 0; i < T; ++i) pthread_join(th[i], 0);
    const double dt = now() - t0;
    printf("%s of %.1f GB written by another process, %2d threads: %.3f s = %5.1f GB/s\n", map ? "memcpy out of a mapping" : "pread", size / 1e9, T, dt, size / dt / 1e9); fflush(stdout);
  }
  return 0;
}
