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
                size_t eb) override {
    // self part by a device copy, peers by one grouped send/recv round (point-to-point over xGMI, all links at once)
    if (sendCnt[rank]) H10X_HIP(c, hipMemcpyAsync((char *)dRecv + recvOff[rank] * eb, (const char *)dSend + sendOff[rank] * eb, sendCnt[rank] * eb,
                                                   hipMemcpyDeviceToDevice, c->stream));
    H10X_TRY(chk(c, ncclGroupStart(), "GroupStart"));
    for (int p = 0; p < n; ++p) {
      if (p == rank) continue;
      if (sendCnt[p]) H10X_TRY(chk(c, ncclSend((const char *)dSend + sendOff[p] * eb, sendCnt[p] * eb, ncclChar, p, nc, c->stream), "Send"));
      if (recvCnt[p]) H10X_TRY(chk(c, ncclRecv((char *)dRecv + recvOff[p] * eb, recvCnt[p] * eb, ncclChar, p, nc, c->stream), "Recv"));
    }
    H10X_TRY(chk(c, ncclGroupEnd(), "GroupEnd"));
    return 0;
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
  std::vector<const void *> sendPtr; std::vector<const u64 *> sendCnt, sendOff; std::vector<std::vector<unsigned char>> host;
  std::vector<double> dbl;
  explicit LocalGroup(int n_) : n(n_), sendPtr(n_), sendCnt(n_), sendOff(n_), host(n_), dbl(n_) {}
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    const unsigned long g = gen;
    if (++arrived == n) { arrived = 0; ++gen; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g; });
  }
};
struct LocalComm : Comm {
  std::shared_ptr<LocalGroup> g;
  int alltoallv(Ctx *c, const void *dSend, const u64 *sendCnt, const u64 *sendOff, void *dRecv, const u64 *recvCnt, const u64 *recvOff,
                size_t eb) override {
    H10X_HIP(c, hipStreamSynchronize(c->stream));            // my send buffer is complete
    g->sendPtr[rank] = dSend; g->sendCnt[rank] = sendCnt; g->sendOff[rank] = sendOff;
    g->wait();
    for (int p = 0; p < n; ++p) {                            // pull my part from every rank (may live on another device)
      const u64 cnt = g->sendCnt[p][rank];
      if (cnt != recvCnt[p]) return c->fail("alltoallv: rank %d sends %llu elements to rank %d which expects %llu", p, (u64)cnt, rank, (u64)recvCnt[p]);
      if (cnt) H10X_HIP(c, hipMemcpyAsync((char *)dRecv + recvOff[p] * eb, (const char *)g->sendPtr[p] + g->sendOff[p][rank] * eb, cnt * eb,
                                          hipMemcpyDefault, c->stream));
    }
    H10X_HIP(c, hipStreamSynchronize(c->stream));
    g->wait();                                               // nobody reuses a send buffer before everyone has pulled
    return 0;
  }
  int allgatherHost(Ctx *, const void *send, void *recv, size_t bytes) override {
    g->host[rank].assign((const unsigned char *)send, (const unsigned char *)send + bytes);
    g->wait();
    for (int p = 0; p < n; ++p) memcpy((char *)recv + (size_t)p * bytes, g->host[p].data(), bytes);
    g->wait();
    return 0;
  }
  int barrier(Ctx *) override { g->wait(); return 0; }
  int allreduceMaxHost(Ctx *, double *v) override {
    g->dbl[rank] = *v; g->wait();
    double m = g->dbl[0]; for (int p = 1; p < n; ++p) m = g->dbl[p] > m ? g->dbl[p] : m;
    g->wait(); *v = m; return 0;
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

int h10x_comm_create_local(h10x_comm **outs, int nranks) {
  auto g = std::make_shared<LocalGroup>(nranks);
  for (int i = 0; i < nranks; ++i) { LocalComm *l = new LocalComm(); l->rank = i; l->n = nranks; l->g = g; outs[i] = new h10x_comm{l}; }
  return 0;
}

void h10x_comm_destroy(h10x_comm *c) { if (c) { delete c->impl; delete c; } }
int h10x_comm_rank(const h10x_comm *c) { return c->impl->rank; }
int h10x_comm_size(const h10x_comm *c) { return c->impl->n; }

}  // extern "C"

namespace h10x { Comm *comm_impl(h10x_comm *c) { return c ? c->impl : nullptr; } }
