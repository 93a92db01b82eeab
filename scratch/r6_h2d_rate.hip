// scratch/r6_h2d_rate.hip — what bounds --readFQB's "read + upload" at 16 GB/s (round 6): host-to-device copies from page-locked slabs alone, the same with reader threads
// filling other slabs from a memory-backed file meanwhile, and the readers alone (pread into page-locked vs ordinary memory).
//   hipcc -O2 --offload-arch=gfx950 scratch/r6_h2d_rate.hip -o /tmp/r6_h2d_rate -lpthread && /tmp/r6_h2d_rate /dev/shm
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char **argv) {
  const char *dir = argc > 1 ? argv[1] : "/dev/shm";
  const size_t SLAB = 64u << 20, NS = 6, FILEB = (size_t)4 << 30;
  char path[512]; snprintf(path, sizeof path, "%s/r6_h2d.bin", dir);
  { int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0666); std::vector<char> b(64 << 20, 7); for (size_t at = 0; at < FILEB; at += b.size()) if (pwrite(fd, b.data(), b.size(), (off_t)at) != (ssize_t)b.size()) { perror("pwrite"); return 1; } close(fd); }
  char *slab[NS], *plain[NS]; char *dev; hipStream_t st;
  for (size_t k = 0; k < NS; ++k) { CK(hipHostMalloc((void **)&slab[k], SLAB, hipHostMallocDefault)); memset(slab[k], 1, SLAB); plain[k] = (char *)malloc(SLAB); memset(plain[k], 1, SLAB); }
  CK(hipMalloc((void **)&dev, (size_t)2 << 30)); CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  auto h2d = [&](int reps, const char *what) {
    const double t0 = now();
    for (int r = 0; r < reps; ++r) CK(hipMemcpyAsync(dev + (size_t)(r % 16) * SLAB, slab[r % 2], SLAB, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    const double dt = now() - t0; printf("%-72s %.3f s = %5.1f GB/s\n", what, dt, reps * (double)SLAB / dt / 1e9); fflush(stdout);
  };
  h2d(8, "warm-up"); h2d(128, "H2D, 64 MiB page-locked slabs, nothing else running");
  for (int pinnedDst = 0; pinnedDst < 2; ++pinnedDst) for (int T : {4, 16, 32}) {
    std::atomic<size_t> next{0}; const size_t pieces = FILEB / (4u << 20);
    const int fd = open(path, O_RDONLY); const double t0 = now();
    std::vector<std::thread> th;
    for (int i = 0; i < T; ++i) th.emplace_back([&, i] { for (size_t p; (p = next.fetch_add(1)) < pieces;) { char *dst = (pinnedDst ? slab : plain)[2 + (p / 16) % 4] + (p % 16) * (4u << 20); size_t g = 0; while (g < (4u << 20)) { ssize_t r = pread(fd, dst + g, (4u << 20) - g, (off_t)(p * (4u << 20) + g)); if (r <= 0) exit(2); g += (size_t)r; } } });
    for (auto &t : th) t.join();
    const double dt = now() - t0; close(fd);
    printf("pread, %2d threads, 4 MiB pieces into %-35s %.3f s = %5.1f GB/s\n", T, pinnedDst ? "page-locked slabs" : "ordinary memory", dt, (double)FILEB / dt / 1e9); fflush(stdout);
  }
  {                                                          // both at once: 16 readers into slabs 2..5 while slabs 0, 1 are uploaded over and over
    std::atomic<size_t> next{0}; std::atomic<bool> stop{false}; const size_t pieces = FILEB / (4u << 20);
    const int fd = open(path, O_RDONLY); std::vector<std::thread> th; const double t0 = now();
    for (int i = 0; i < 16; ++i) th.emplace_back([&] { for (size_t p; (p = next.fetch_add(1)) < pieces;) { char *dst = slab[2 + (p / 16) % 4] + (p % 16) * (4u << 20); size_t g = 0; while (g < (4u << 20)) { ssize_t r = pread(fd, dst + g, (4u << 20) - g, (off_t)(p * (4u << 20) + g)); if (r <= 0) exit(2); g += (size_t)r; } } });
    size_t up = 0;
    std::thread u([&] { while (!stop) { CK(hipMemcpyAsync(dev + (up % 16) * SLAB, slab[up % 2], SLAB, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); ++up; } });
    for (auto &t : th) t.join();
    const double dt = now() - t0; stop = true; u.join(); close(fd);
    printf("16 readers into page-locked slabs: %5.1f GB/s  WHILE  uploads of other slabs: %5.1f GB/s   (%.3f s)\n", (double)FILEB / dt / 1e9, up * (double)SLAB / dt / 1e9, dt);
  }
  {                                                          // the ingest's own pattern: a FRESH 24 GB device image filled slab by slab, an event per slab, the previous slab's event waited for
    const size_t IMG = (size_t)24 << 30; char *img; CK(hipMalloc((void **)&img, IMG));
    hipEvent_t ev[NS]; for (size_t k = 0; k < NS; ++k) CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    for (int pattern = 0; pattern < 3; ++pattern) {
      const double t0 = now(); double tCall = 0, tWait = 0;
      const size_t n = IMG / SLAB;
      for (size_t k = 0; k < n; ++k) {
        const double a = now();
        CK(hipMemcpyAsync(img + k * SLAB, slab[k % NS], SLAB, hipMemcpyHostToDevice, st));
        if (pattern >= 1) CK(hipEventRecord(ev[k % NS], st));
        const double b = now(); tCall += b - a;
        if (pattern == 2 && k >= 1) CK(hipEventSynchronize(ev[(k - 1) % NS]));
        tWait += now() - b;
      }
      CK(hipStreamSynchronize(st));
      const double dt = now() - t0;
      printf("24 GB image, %-58s %.3f s = %5.1f GB/s (in the calls %.3f s, in the waits %.3f s)\n", pattern == 0 ? "back-to-back async copies, one sync at the end:" : pattern == 1 ? "+ an event recorded behind every copy:" : "+ the previous copy's event waited for (the ingest's loop):", dt, (double)IMG / dt / 1e9, tCall, tWait); fflush(stdout);
    }
    CK(hipFree(img));
  }
  for (int pinnedSrc = 0; pinnedSrc < 2; ++pinnedSrc) {      // --writeHash's write stream: one pwrite stream into a NEW file, 64 MiB pieces, from ordinary / page-locked memory
    char p2[600]; snprintf(p2, sizeof p2, "%s/r6_h2d_w.bin", dir); unlink(p2);
    const int fd = open(p2, O_RDWR | O_CREAT | O_TRUNC, 0666); const size_t total = (size_t)8 << 30; if (ftruncate(fd, (off_t)total)) return 1;
    const double t0 = now();
    for (size_t at = 0; at < total; at += SLAB) if (pwrite(fd, (pinnedSrc ? slab : plain)[(at / SLAB) % 2], SLAB, (off_t)at) != (ssize_t)SLAB) { perror("pwrite"); return 1; }
    const double dt = now() - t0; close(fd); unlink(p2);
    printf("pwrite of a new 8 GB file, one stream, from %-28s %.3f s = %5.1f GB/s\n", pinnedSrc ? "page-locked memory:" : "ordinary memory:", dt, (double)total / dt / 1e9); fflush(stdout);
  }
  unlink(path);
  return 0;
}
