// comm.hpp — the exchange layer of the sharded (multi-GPU) path. One primitive moves device bytes between
// ranks (alltoallv with arbitrary send offsets: all-to-all, allgather-v and gather-v are special cases) and
// one moves small host records (allgatherHost). Two backends:
//   RcclComm  — one process per GPU, ncclSend/ncclRecv groups over xGMI (what bench.py / a launcher use);
//   LocalComm — N ranks as threads of one process (tests on a 1-GPU box, single-process --gpus N), device-to-
//               device copies behind a thread barrier.
#pragma once
#include "common.hpp"

namespace h10x {

struct Comm {
  int rank = 0, n = 1;
  virtual ~Comm() {}
  // recv[r*bytes .. ) = send of rank r
  virtual int allgatherHost(Ctx *c, const void *send, void *recv, size_t bytes) = 0;
  // for every peer p: send elements [sendOff[p], sendOff[p]+sendCnt[p]) of dSend to p; receive recvCnt[p] elements
  // from p at recvOff[p] of dRecv. Counts/offsets in elements of elemBytes bytes. Completes on `st` (null = c->stream): an exchange whose result is not needed
  // until a later stage goes on the context's exchange stream (Ctx::xFork / xJoin) and runs beside the kernels of the main stream — every rank issues its
  // collectives in the same order whatever the stream, and joins before the next one.
  virtual int alltoallv(Ctx *c, const void *dSend, const u64 *sendCnt, const u64 *sendOff, void *dRecv, const u64 *recvCnt,
                        const u64 *recvOff, size_t elemBytes, hipStream_t st = nullptr) = 0;
  virtual int barrier(Ctx *c) = 0;
  // max over ranks of a host double (timing plumbing)
  virtual int allreduceMaxHost(Ctx *c, double *v) = 0;
  // bytes are dear on this backend (host-staged TCP): the in-range barcode lists then travel delta-coded. Over xGMI (and between ranks of one process) they travel as they
  // are: at 8 ranks the coding and decoding cost every rank 4.6 ms of compute on the 1/10 3 Gb set to save 0.6 ms of link time (bench.py --virtual-ranks)
  virtual bool slowLinks() const { return false; }
};

}  // namespace h10x
