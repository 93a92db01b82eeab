/* scratch/r6_io_rate.c — how fast this box moves a file in and out of memory-backed storage, by method and thread count (round 6):
     writes: pwrite from T threads into one new file (what --writeHash did) vs memcpy from T threads into a MAP_SHARED mapping of the new file;
     reads : pread from T threads (what --readFQB does) vs memcpy out of a MAP_SHARED mapping.
   gcc -O2 -pthread scratch/r6_io_rate.c -o /tmp/r6_io_rate && /tmp/r6_io_rate /dev/shm 4 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
typedef struct { int fd, mode; char *map, *buf; uint64_t a, b, piece; } Job;
static void *job(void *p) {
  Job *j = (Job *)p;
  for (uint64_t at = j->a; at < j->b; at += j->piece) {
    const uint64_t k = j->b - at < j->piece ? j->b - at : j->piece;
    if (j->mode == 0) { if (pwrite(j->fd, j->buf, k, (off_t)at) != (ssize_t)k) { perror("pwrite"); exit(1); } }
    else if (j->mode == 1) memcpy(j->map + at, j->buf, k);
    else if (j->mode == 2) { uint64_t g = 0; while (g < k) { ssize_t r = pread(j->fd, j->buf + g, k - g, (off_t)(at + g)); if (r <= 0) { perror("pread"); exit(1); } g += (uint64_t)r; } }
    else memcpy(j->buf, j->map + at, k);
  }
  return 0;
}
int main(int argc, char **argv) {
  const char *dir = argc > 1 ? argv[1] : "/dev/shm"; const uint64_t n = (uint64_t)((argc > 2 ? atof(argv[2]) : 2.0) * (1 << 30));
  const uint64_t piece = 8u << 20; char path[512]; snprintf(path, sizeof path, "%s/r6_io_rate.bin", dir);
  static const char *const name[6] = {"pwrite, new file", "memcpy into a mapping of a new file", "pread", "memcpy out of a mapping", "pwrite, file fallocate()d first", "memcpy into a mapping, fallocate()d first"};
  const int order[6] = {0, 1, 4, 5, 2, 3};
  for (int oi = 0; oi < 6; ++oi) { const int mode = order[oi];
    for (int T = 1; T <= 32; T *= 2) {
      const int wr = mode < 2 || mode >= 4, mapped = mode == 1 || mode == 3 || mode == 5;
      if (wr) unlink(path);
      const int fd = open(path, wr ? O_RDWR | O_CREAT : O_RDONLY, 0666); if (fd < 0) { perror(path); return 1; }
      if (wr && ftruncate(fd, (off_t)n)) { perror("ftruncate"); return 1; }
      double tf = 0;
      if (mode >= 4) { const double a = now(); if (fallocate(fd, 0, 0, (off_t)n)) { perror("fallocate"); return 1; } tf = now() - a; }
      char *map = 0;
      if (mapped) { map = mmap(0, n, wr ? PROT_READ | PROT_WRITE : PROT_READ, MAP_SHARED, fd, 0); if (map == MAP_FAILED) { perror("mmap"); return 1; } }
      Job jb[32]; pthread_t th[32];
      for (int i = 0; i < T; ++i) { jb[i] = (Job){fd, mode == 4 ? 0 : (mode == 5 ? 1 : mode), map, 0, n / T * i, i + 1 == T ? n : n / T * (i + 1), piece}; if (posix_memalign((void **)&jb[i].buf, 4096, piece)) return 1; memset(jb[i].buf, 0x5A + i, piece); }
      const double t0 = now();
      for (int i = 0; i < T; ++i) pthread_create(&th[i], 0, job, &jb[i]);
      for (int i = 0; i < T; ++i) pthread_join(th[i], 0);
      const double dt = now() - t0;
      printf("%-42s %2d threads: %.3f s = %5.1f GB/s", name[mode], T, dt, n / dt / 1e9);
      if (mode >= 4) printf("   (+ fallocate %.3f s: %.1f GB/s all told)", tf, n / (dt + tf) / 1e9);
      printf("\n"); fflush(stdout);
      if (map) munmap(map, n);
      close(fd);
      for (int i = 0; i < T; ++i) free(jb[i].buf);
    }
  }
  unlink(path);
  return 0;
}
