// comm.hip — RCCL and in-process backends of comm.hpp, and their C ABI.
#include "comm.hpp"
#include <rccl/rccl.h>
#include <condition_variable>
#include <memory>

namespace h10x {

// ------------------------------------------------------------------------------------------ RCCL
struct RcclComm : Comm {
  ncclComm_t nc = nullptr;
  DevBuf<unsigned char> small;                               // staging for host allgathers
  ~RcclComm() override { if (nc) (void)ncclCommDestroy(nc); }
  int chk(Ctx *c, ncclResult_t r, const char *what) { return r == ncclSuccess ? 0 : c->fail("RCCL %s failed: %s", what, ncclGetErrorString(r)); }
  int alltoallv(Ctx *c, const void *dSend, const u64 *sendCnt, const u64 *sendOff, void *dRecv, const u64 *recvCnt, const u64 *recvOff,
                size_t eb, hipStream_t stArg) override {
    hipStream_t const st = stArg ? stArg : c->stream;
    // self part by a device copy, peers by one grouped send/recv round (point-to-point over xGMI, all links at once)
    if (sendCnt[rank]) H10X_HIP(c, hipMemcpyAsync((char *)dRecv + recvOff[rank] * eb, (const char *)dSend + sendOff[rank] * eb, sendCnt[rank] * eb,
                                                   hipMemcpyDeviceToDevice, st));
    H10X_TRY(chk(c, ncclGroupStart(), "GroupStart"));
    ncclResult_t bad = ncclSuccess; const char *what = "";
    for (int p = 0; p < n && bad == ncclSuccess; ++p) {
      if (p == rank) continue;
      if (sendCnt[p] && (bad = ncclSend((const char *)dSend + sendOff[p] * eb, sendCnt[p] * eb, ncclChar, p, nc, st)) != ncclSuccess) { what = "Send"; break; }
      if (recvCnt[p] && (bad = ncclRecv((char *)dRecv + recvOff[p] * eb, recvCnt[p] * eb, ncclChar, p, nc, st)) != ncclSuccess) { what = "Recv"; break; }
    }
    const ncclResult_t end = ncclGroupEnd();                 // always closed, also after a failed call inside the group
    if (bad != ncclSuccess) return chk(c, bad, what);
    return chk(c, end, "GroupEnd");
  }
  int allgatherHost(Ctx *c, const void *send, void *recv, size_t bytes) override {
    if (small.n < bytes * (size_t)(n + 1)) H10X_HIP(c, small.alloc(bytes * (size_t)(n + 1)));
    unsigned char *dIn = small.p, *dOut = small.p + bytes;
    H10X_HIP(c, hipMemcpyAsync(dIn, send, bytes, hipMemcpyHostToDevice, c->stream));
    H10X_TRY(chk(c, ncclAllGather(dIn, dOut, bytes, ncclChar, nc, c->stream), "AllGather"));
    H10X_HIP(c, hipMemcpyAsync(recv, dOut, bytes * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    H10X_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
  }
  int barrier(Ctx *c) override { double v = 0; return allreduceMaxHost(c, &v); }
  int allreduceMaxHost(Ctx *c, double *v) override {
    if (small.n < 16) H10X_HIP(c, small.alloc(64));
    H10X_HIP(c, hipMemcpyAsync(small.p, v, 8, hipMemcpyHostToDevice, c->stream));
    H10X_TRY(chk(c, ncclAllReduce(small.p, small.p + 8, 1, ncclDouble, ncclMax, nc, c->stream), "AllReduce"));
    H10X_HIP(c, hipMemcpyAsync(v, small.p + 8, 8, hipMemcpyDeviceToHost, c->stream));
    H10X_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
  }
};

// ------------------------------------------------------------------------------------------ in-process group
struct LocalGroup {
  int n; std::mutex mu; std::condition_variable cv; int arrived = 0; unsigned long gen = 0;
  // "virtual ranks" (bench.py --virtual-ranks): the ranks of the group share ONE device and take turns on it — a rank computes only while it holds the token, hands it on
  // whenever it waits for the others, and its stage timers then read as if it had the GPU to itself (a turnstile with tickets: std::mutex may not change hands between threads)
  bool serialize = false; std::mutex tokMu; std::condition_variable tokCv; int tokHolder = -1;
  void tokAcquire(int r) { std::unique_lock<std::mutex> lk(tokMu); tokCv.wait(lk, [&] { return tokHolder < 0 || tokHolder == r; }); tokHolder = r; }
  void tokRelease(int r) { std::lock_guard<std::mutex> lk(tokMu); if (tokHolder == r) { tokHolder = -1; tokCv.notify_all(); } }
  int failed = 0, failedPrev = 0;                            // a rank's failure in a collective is seen by every rank at the closing barrier
  std::vector<const void *> sendPtr; std::vector<const u64 *> sendCnt, sendOff; std::vector<std::vector<unsigned char>> host;
  std::vector<double> dbl;
  explicit LocalGroup(int n_) : n(n_), sendPtr(n_), sendCnt(n_), sendOff(n_), host(n_), dbl(n_) {}
  // returns whether any rank reported a failure since the previous rendezvous
  bool wait(bool iFailed = false) {
    std::unique_lock<std::mutex> lk(mu);
    const unsigned long g = gen;
    if (iFailed) failed = 1;
    if (++arrived == n) { arrived = 0; failedPrev = failed; failed = 0; ++gen; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g; });
    return failedPrev != 0;
  }
};
struct LocalComm : Comm {
  std::shared_ptr<LocalGroup> g;
  void yield(Ctx *c) { if (g->serialize) { if (c) (void)hipStreamSynchronize(c->stream); g->tokRelease(rank); } }   // my kernels are done: somebody else's turn
  void resume() { if (g->serialize) g->tokAcquire(rank); }
  int alltoallv(Ctx *c, const void *dSend, const u64 *sendCnt, const u64 *sendOff, void *dRecv, const u64 *recvCnt, const u64 *recvOff,
                size_t eb, hipStream_t stArg) override {
    hipStream_t const st = stArg ? stArg : c->stream;       // (threads of one process rendezvous on the host: an exchange on the side stream is complete when this returns)
    H10X_HIP(c, hipStreamSynchronize(st));                   // my send buffer is complete
    yield(c);
    struct Back { LocalComm *l; ~Back() { l->resume(); } } back{this};
    g->sendPtr[rank] = dSend; g->sendCnt[rank] = sendCnt; g->sendOff[rank] = sendOff;
    g->wait();
    int rc = 0;
    for (int p = 0; p < n && !rc; ++p) {                     // pull my part from every rank (may live on another device)
      const u64 cnt = g->sendCnt[p][rank];
      if (cnt != recvCnt[p]) rc = c->fail("alltoallv: rank %d sends %llu elements to rank %d which expects %llu", p, (u64)cnt, rank, (u64)recvCnt[p]);
      else if (cnt && hipMemcpyAsync((char *)dRecv + recvOff[p] * eb, (const char *)g->sendPtr[p] + g->sendOff[p][rank] * eb, cnt * eb, hipMemcpyDefault, st) != hipSuccess)
        rc = c->fail("alltoallv: device copy from rank %d failed", p);
    }
    if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = c->fail("alltoallv: stream synchronisation failed");
    // nobody reuses a send buffer before everyone has pulled — and everybody leaves with the same verdict (a rank that
    // returned early would leave the others waiting here for ever)
    if (g->wait(rc != 0) && !rc) rc = c->fail("alltoallv failed on another rank");
    return rc;
  }
  int allgatherHost(Ctx *c, const void *send, void *recv, size_t bytes) override {
    yield(c);
    g->host[rank].assign((const unsigned char *)send, (const unsigned char *)send + bytes);
    g->wait();
    for (int p = 0; p < n; ++p) memcpy((char *)recv + (size_t)p * bytes, g->host[p].data(), bytes);
    g->wait();
    resume();
    return 0;
  }
  int barrier(Ctx *c) override { yield(c); g->wait(); resume(); return 0; }
  int allreduceMaxHost(Ctx *c, double *v) override {
    yield(c);
    g->dbl[rank] = *v; g->wait();
    double m = g->dbl[0]; for (int p = 1; p < n; ++p) m = g->dbl[p] > m ? g->dbl[p] : m;
    g->wait(); *v = m; resume(); return 0;
  }
};

// ------------------------------------------------------------------------------------------ host-staged over TCP
// One process per rank like RCCL, but every exchange goes device -> host -> socket -> host -> device. Not a data path for
// production (PCIe + loopback); it exists so that the multi-PROCESS launch path (rendezvous, one context per process,
// bench.py under torch.distributed.run) can run where RCCL cannot: several ranks on ONE GPU (RCCL refuses two ranks per
// device), i.e. on a 1-GPU test box. Full mesh: rank r listens on port base + r, connects to the lower ranks.
}  // namespace h10x
#include <sys/socket.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <arpa/inet.h>
#include <poll.h>
#include <fcntl.h>
#include <unistd.h>
#include <cerrno>
namespace h10x {
struct SockComm : Comm {
  std::vector<int> fd;                                       // fd[p] = connection to rank p (-1 for myself)
  bool slowLinks() const override { return true; }
  ~SockComm() override { for (int f : fd) if (f >= 0) close(f); }
  // send nOut bytes to fdOut while receiving nIn bytes from fdIn (either may be -1 / 0): never blocks on one direction
  static int duplex(int fdOut, const char *out, size_t nOut, int fdIn, char *in, size_t nIn) {
    while (nOut || nIn) {
      pollfd pf[2]; int k = 0, io = -1, ii = -1;
      if (nOut) { pf[k] = pollfd{fdOut, POLLOUT, 0}; io = k++; }
      if (nIn) { pf[k] = pollfd{fdIn, POLLIN, 0}; ii = k++; }
      // ranks that share a GPU serialise their kernels: a peer may be minutes behind at config-3 scale; signals interrupt poll
      static const int timeoutMs = [] { const char *e = getenv("H10X_SOCK_TIMEOUT_MS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1800000; }();
      const int pr = poll(pf, (nfds_t)k, timeoutMs);
      if (pr < 0) { if (errno == EINTR) continue; return -1; }
      if (pr == 0) { errno = ETIMEDOUT; return -1; }
      if (io >= 0 && (pf[io].revents & (POLLOUT | POLLERR | POLLHUP))) { const ssize_t w = send(fdOut, out, nOut > (1u << 20) ? (1u << 20) : nOut, MSG_DONTWAIT | MSG_NOSIGNAL); if (w < 0 && errno != EAGAIN && errno != EWOULDBLOCK) return -1; if (w > 0) { out += w; nOut -= (size_t)w; } }
      if (ii >= 0 && (pf[ii].revents & (POLLIN | POLLERR | POLLHUP))) { const ssize_t r = recv(fdIn, in, nIn, MSG_DONTWAIT); if (r == 0 || (r < 0 && errno != EAGAIN && errno != EWOULDBLOCK)) return -1; if (r > 0) { in += r; nIn -= (size_t)r; } }
    }
    return 0;
  }
  // host bytes: part p of `send` (sendOff / sendCnt in bytes) goes to rank p; n - 1 rounds, in round s I send to rank + s and receive from rank - s
  int exchange(Ctx *c, const char *send, const u64 *sendCnt, const u64 *sendOff, char *recv, const u64 *recvCnt, const u64 *recvOff) {
    if (sendCnt[rank] != recvCnt[rank]) return c->fail("socket alltoallv: self part %llu != %llu", (u64)sendCnt[rank], (u64)recvCnt[rank]);
    if (sendCnt[rank]) memcpy(recv + recvOff[rank], send + sendOff[rank], sendCnt[rank]);
    for (int s = 1; s < n; ++s) {
      const int to = (rank + s) % n, from = (rank - s + n) % n;
      if (duplex(fd[to], send + sendOff[to], sendCnt[to], fd[from], recv + recvOff[from], recvCnt[from])) return c->fail("socket exchange with ranks %d / %d failed: %s", to, from, strerror(errno));
    }
    return 0;
  }
  int alltoallv(Ctx *c, const void *dSend, const u64 *sendCnt, const u64 *sendOff, void *dRecv, const u64 *recvCnt, const u64 *recvOff, size_t eb, hipStream_t stArg) override {
    hipStream_t const st = stArg ? stArg : c->stream;
    std::vector<u64> sc((size_t)n), so((size_t)n), rc((size_t)n), ro((size_t)n); u64 ts = 0, tr = 0;
    for (int p = 0; p < n; ++p) { sc[p] = sendCnt[p] * eb; so[p] = ts; ts += sc[p]; rc[p] = recvCnt[p] * eb; ro[p] = tr; tr += rc[p]; }
    std::vector<char> hs(ts ? ts : 1), hr(tr ? tr : 1);
    for (int p = 0; p < n; ++p) if (sc[p]) H10X_HIP(c, hipMemcpyAsync(hs.data() + so[p], (const char *)dSend + sendOff[p] * eb, sc[p], hipMemcpyDeviceToHost, st));
    H10X_HIP(c, hipStreamSynchronize(st));
    H10X_TRY(exchange(c, hs.data(), sc.data(), so.data(), hr.data(), rc.data(), ro.data()));
    for (int p = 0; p < n; ++p) if (rc[p]) H10X_HIP(c, hipMemcpyAsync((char *)dRecv + recvOff[p] * eb, hr.data() + ro[p], rc[p], hipMemcpyHostToDevice, st));
    H10X_HIP(c, hipStreamSynchronize(st));
    return 0;
  }
  int allgatherHost(Ctx *c, const void *send, void *recv, size_t bytes) override {
    std::vector<u64> sc((size_t)n, bytes), so((size_t)n, 0), ro((size_t)n);
    for (int p = 0; p < n; ++p) ro[p] = (u64)p * bytes;
    return exchange(c, (const char *)send, sc.data(), so.data(), (char *)recv, sc.data(), ro.data());
  }
  int barrier(Ctx *c) override { double v = 0; return allreduceMaxHost(c, &v); }
  int allreduceMaxHost(Ctx *c, double *v) override {
    std::vector<double> all((size_t)n);
    H10X_TRY(allgatherHost(c, v, all.data(), 8));
    double m = all[0]; for (int p = 1; p < n; ++p) m = all[p] > m ? all[p] : m;
    *v = m; return 0;
  }
};
}  // namespace h10x

using namespace h10x;
struct h10x_comm { Comm *impl; };

extern "C" {

int h10x_comm_unique_id(void *id128) {
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return -1;
  static_assert(sizeof(id) == 128, "ncclUniqueId size");
  memcpy(id128, &id, 128);
  return 0;
}

int h10x_comm_create_rccl(h10x_comm **out, int rank, int nranks, const void *id128, int device, char *err, int errlen) {
  *out = nullptr;
  if (hipSetDevice(device) != hipSuccess) { if (err) snprintf(err, (size_t)errlen, "hipSetDevice(%d) failed", device); return -1; }
  ncclUniqueId id; memcpy(&id, id128, 128);
  RcclComm *r = new RcclComm(); r->rank = rank; r->n = nranks;
  const ncclResult_t rc = ncclCommInitRank(&r->nc, nranks, id, rank);
  if (rc != ncclSuccess) { if (err) snprintf(err, (size_t)errlen, "ncclCommInitRank failed: %s", ncclGetErrorString(rc)); r->nc = nullptr; delete r; return -1; }
  *out = new h10x_comm{r};
  return 0;
}

int h10x_comm_create_socket(h10x_comm **out, int rank, int nranks, const char *addr, int basePort, char *err, int errlen) {
  *out = nullptr;
  auto bad = [&](const char *what) { if (err) snprintf(err, (size_t)errlen, "socket communicator, rank %d: %s: %s", rank, what, strerror(errno)); return -1; };
  if (nranks < 1 || rank < 0 || rank >= nranks || basePort < 1 || basePort + nranks - 1 > 65535) { errno = EINVAL; return bad("rank / port range (basePort + nranks - 1 must be <= 65535)"); }
  SockComm *sc = new SockComm(); sc->rank = rank; sc->n = nranks; sc->fd.assign((size_t)nranks, -1);
  sockaddr_in a; memset(&a, 0, sizeof a); a.sin_family = AF_INET;
  if (inet_pton(AF_INET, addr && *addr ? addr : "127.0.0.1", &a.sin_addr) != 1) { delete sc; errno = EINVAL; return bad("address"); }
  int srv = -1; const int one = 1;
  if (rank < nranks - 1) {                                   // the higher ranks connect to me
    srv = socket(AF_INET, SOCK_STREAM, 0);
    if (srv < 0 || setsockopt(srv, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one)) { if (srv >= 0) close(srv); delete sc; return bad("socket"); }
    a.sin_port = htons((uint16_t)(basePort + rank));
    if (srv < 0 || bind(srv, (sockaddr *)&a, sizeof a) || listen(srv, nranks)) { if (srv >= 0) close(srv); delete sc; return bad("listen"); }
  }
  for (int p = 0; p < rank; ++p) {                           // I connect to the lower ranks (they may not be up yet: retry for a while)
    int f = -1;
    for (int attempt = 0; attempt < 3000; ++attempt) {
      f = socket(AF_INET, SOCK_STREAM, 0); a.sin_port = htons((uint16_t)(basePort + p));
      if (f >= 0 && connect(f, (sockaddr *)&a, sizeof a) == 0) break;
      if (f >= 0) close(f);
      f = -1; usleep(100000);
    }
    if (f < 0) { if (srv >= 0) close(srv); delete sc; return bad("connect"); }
    setsockopt(f, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    const int32_t me = rank; if (send(f, &me, 4, MSG_NOSIGNAL) != 4) { close(f); if (srv >= 0) close(srv); delete sc; return bad("hello"); }
    sc->fd[p] = f;
  }
  for (int k = rank + 1; k < nranks; ++k) {
    pollfd pf{srv, POLLIN, 0};
    if (poll(&pf, 1, 300000) <= 0) { close(srv); delete sc; errno = ETIMEDOUT; return bad("accept"); }
    const int f = accept(srv, nullptr, nullptr); int32_t who = -1;
    if (f < 0 || recv(f, &who, 4, MSG_WAITALL) != 4 || who <= rank || who >= nranks || sc->fd[who] >= 0) { if (f >= 0) close(f); close(srv); delete sc; return bad("accept"); }
    setsockopt(f, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    sc->fd[who] = f;
  }
  if (srv >= 0) close(srv);
  *out = new h10x_comm{sc};
  return 0;
}

int h10x_comm_create_rccl_all(h10x_comm **outs, int nranks, const int *devices, char *err, int errlen) {
  for (int i = 0; i < nranks; ++i) outs[i] = nullptr;
  for (int i = 0; i < nranks; ++i) for (int j = 0; j < i; ++j)
    if (devices[i] == devices[j]) { if (err) snprintf(err, (size_t)errlen, "RCCL needs one device per rank: ranks %d and %d share device %d", j, i, devices[i]); return -1; }
  std::vector<ncclComm_t> nc((size_t)nranks, nullptr);
  const ncclResult_t rc = ncclCommInitAll(nc.data(), nranks, devices);
  if (rc != ncclSuccess) { if (err) snprintf(err, (size_t)errlen, "ncclCommInitAll failed: %s", ncclGetErrorString(rc)); return -1; }
  for (int i = 0; i < nranks; ++i) { RcclComm *r = new RcclComm(); r->rank = i; r->n = nranks; r->nc = nc[(size_t)i]; outs[i] = new h10x_comm{r}; }
  return 0;
}

int h10x_device_enable_peers(const int *devices, int n) {
  int prev = 0; (void)hipGetDevice(&prev);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
    if (i == j || devices[i] == devices[j]) continue;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) != hipSuccess || !can) continue;
    if (hipSetDevice(devices[i]) != hipSuccess) continue;
    const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
  }
  (void)hipGetLastError();
  (void)hipSetDevice(prev);
  return 0;
}

int h10x_comm_create_local(h10x_comm **outs, int nranks) {
  auto g = std::make_shared<LocalGroup>(nranks);
  for (int i = 0; i < nranks; ++i) { LocalComm *l = new LocalComm(); l->rank = i; l->n = nranks; l->g = g; outs[i] = new h10x_comm{l}; }
  return 0;
}

int h10x_comm_local_serialize(h10x_comm *c, int on) {
  LocalComm *l = c ? dynamic_cast<LocalComm *>(c->impl) : nullptr;
  if (!l) return -1;
  l->g->serialize = on != 0;
  return 0;
}
int h10x_comm_turn_begin(h10x_comm *c) { LocalComm *l = c ? dynamic_cast<LocalComm *>(c->impl) : nullptr; if (!l) return -1; l->resume(); return 0; }
int h10x_comm_turn_end(h10x_comm *c, int device) {
  LocalComm *l = c ? dynamic_cast<LocalComm *>(c->impl) : nullptr; if (!l) return -1;
  if (l->g->serialize) { if (hipSetDevice(device) == hipSuccess) (void)hipDeviceSynchronize(); l->g->tokRelease(l->rank); }
  return 0;
}

void h10x_comm_destroy(h10x_comm *c) { if (c) { delete c->impl; delete c; } }
int h10x_comm_rank(const h10x_comm *c) { return c->impl->rank; }
int h10x_comm_size(const h10x_comm *c) { return c->impl->n; }

}  // extern "C"

namespace h10x { Comm *comm_impl(h10x_comm *c) { return c ? c->impl : nullptr; } }
