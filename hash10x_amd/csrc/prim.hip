// prim.hip — rocPRIM instantiations (device-wide radix sort, segmented sort, scans, reductions).
#include "prim.hpp"
#include <thread>
#include <rocprim/rocprim.hpp>

namespace h10x {

#define PRIM_TWO_PHASE(c, t, CALL)                                                               \
  do { size_t bytes = 0; void *tmp = nullptr;                                                    \
       hipError_t e = (CALL);                                                                    \
       if (e != hipSuccess) return (c)->fail("rocPRIM size query failed: %s", hipGetErrorName(e)); \
       if (bytes > (t).buf.n) { e = (t).buf.alloc(bytes + (bytes >> 2));                         \
         if (e != hipSuccess) return (c)->fail("out of device memory for %zu bytes of sort/scan scratch", bytes); } \
       tmp = (t).buf.p;                                                                          \
       e = (CALL);                                                                               \
       if (e != hipSuccess) return (c)->fail("rocPRIM call failed: %s", hipGetErrorName(e)); } while (0)

int prim_exclusive_scan_u32(Ctx *c, PrimTemp &t, const u32 *in, u32 *out, size_t n) {
  if (!n) return 0;
  PRIM_TWO_PHASE(c, t, rocprim::exclusive_scan(tmp, bytes, in, out, (u32)0, n, rocprim::plus<u32>(), c->stream));
  return 0;
}
// ord[i] = number of runs of equal keys (compared above their low `cb` bits) that start before position i, for i in [0, n]:
// the head flags are formed inside the scan's loads instead of by a kernel of their own
struct RunHead {
  const u64 *key; size_t n; int cb;
  __device__ u32 operator()(size_t i) const { return (i < n && (i == 0 || (key[i] >> cb) != (key[i - 1] >> cb))) ? 1u : 0u; }
};
int prim_run_ordinals_u64(Ctx *c, PrimTemp &t, const u64 *key, int cb, u32 *ord, size_t n) {
  auto flags = rocprim::make_transform_iterator(rocprim::counting_iterator<size_t>(0), RunHead{key, n, cb});
  PRIM_TWO_PHASE(c, t, rocprim::exclusive_scan(tmp, bytes, flags, ord, (u32)0, n + 1, rocprim::plus<u32>(), c->stream));
  return 0;
}
int prim_exclusive_scan_u32_u64(Ctx *c, PrimTemp &t, const u32 *in, u64 *out, size_t n) {
  if (!n) return 0;
  auto in64 = rocprim::make_transform_iterator(in, [] __device__(u32 v) -> u64 { return (u64)v; });
  PRIM_TWO_PHASE(c, t, rocprim::exclusive_scan(tmp, bytes, in64, out, (u64)0, n, rocprim::plus<u64>(), c->stream));
  return 0;
}
int prim_inclusive_scan_u32(Ctx *c, PrimTemp &t, const u32 *in, u32 *out, size_t n) {
  if (!n) return 0;
  PRIM_TWO_PHASE(c, t, rocprim::inclusive_scan(tmp, bytes, in, out, n, rocprim::plus<u32>(), c->stream));
  return 0;
}
int prim_sort_pairs_u64_u32(Ctx *c, PrimTemp &t, const u64 *kin, u64 *kout, const u32 *vin, u32 *vout, size_t n, int b0, int b1) {
  if (!n) return 0;
  PRIM_TWO_PHASE(c, t, rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, n, (unsigned)b0, (unsigned)b1, c->stream));
  return 0;
}
int prim_sort_pairs_u32_u32(Ctx *c, PrimTemp &t, const u32 *kin, u32 *kout, const u32 *vin, u32 *vout, size_t n, int b0, int b1) {
  if (!n) return 0;
  PRIM_TWO_PHASE(c, t, rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, n, (unsigned)b0, (unsigned)b1, c->stream));
  return 0;
}
// ---- the read-back mailbox (common.hpp)
struct MailArgs { const unsigned char *src[Ctx::MAIL_ITEMS]; u32 off[Ctx::MAIL_ITEMS], n[Ctx::MAIL_ITEMS]; u32 count; };
__global__ void mail_post_kernel(MailArgs a, unsigned char *mail, u32 *flag, u32 seq) {
  for (u32 i = 0; i < a.count; ++i)
    for (u32 b = threadIdx.x; b < a.n[i]; b += blockDim.x) mail[a.off[i] + b] = a.src[i][b];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int Ctx::syncReadbacks() {
  hipError_t e = hipSuccess;
  if (pendingReads.empty() || mailDirect) e = hipStreamSynchronize(stream);     // nothing to post, or an ordinary copy is on its way too
  if (!pendingReads.empty() && e == hipSuccess) {
    MailArgs a; a.count = (u32)pendingReads.size();
    for (u32 i = 0; i < a.count; ++i) { a.src[i] = (const unsigned char *)pendingReads[i].dev; a.off[i] = (u32)pendingReads[i].off; a.n[i] = (u32)pendingReads[i].n; }
    u32 *flag = (u32 *)(mail + MAIL_BYTES);
    const u32 seq = ++mailSeq ? mailSeq : ++mailSeq;                            // (never 0: the mailbox starts zeroed)
    mail_post_kernel<<<1, 64, 0, stream>>>(a, mail, flag, seq);
    e = hipGetLastError();
    for (u64 spins = 0; e == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq; ++spins) {
#if !defined(__HIP_DEVICE_COMPILE__)
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#else
      std::this_thread::yield();
#endif
#endif
      if ((spins & 0xFFF) == 0xFFF) {                                           // every few microseconds: is the stream still alive?
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipErrorNotReady) continue;
        if (q == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
        e = q == hipSuccess ? hipErrorUnknown : q;                              // the stream drained without the post, or died
      }
    }
    if (e == hipSuccess) for (const PendingRead &r : pendingReads) memcpy(r.dst, mail + r.off, r.n);
  }
  pendingReads.clear(); mailUsed = 0; mailDirect = false;
  return e == hipSuccess ? 0 : fail("HIP error %s at a read-back", hipGetErrorString(e));
}

int prim_sort_pairs_u32_v16(Ctx *c, PrimTemp &t, const u32 *kin, u32 *kout, const Val16 *vin, Val16 *vout, size_t n, int b0, int b1) {
  if (!n) return 0;
  PRIM_TWO_PHASE(c, t, rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, n, (unsigned)b0, (unsigned)b1, c->stream));
  return 0;
}
// (the index build's big sort. rocPRIM's tuned default for 64-bit keys on gfx950 is 512 lanes x 12 keys; 1024 x 8 is 8-12 % faster on this chip for 36 M and 730 M keys alike
// — scratch/r5_sort_cfg.hip, ten configurations; the library's onesweep moves 3 TB/s here either way)
using SortKeys64Config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 8>, 8, rocprim::block_radix_rank_algorithm::match>, 1024 * 1024>;
int prim_sort_keys_u64(Ctx *c, PrimTemp &t, const u64 *kin, u64 *kout, size_t n, int b0, int b1) {
  if (!n) return 0;
  PRIM_TWO_PHASE(c, t, rocprim::radix_sort_keys<SortKeys64Config>(tmp, bytes, kin, kout, n, (unsigned)b0, (unsigned)(b1 > 64 ? 64 : b1), c->stream));
  return 0;
}
int prim_seg_sort_keys_u64(Ctx *c, PrimTemp &t, const u64 *kin, u64 *kout, u32 n, u32 nSeg, const u32 *begin, const u32 *end, int b0, int b1) {
  if (!n || !nSeg) return 0;
  PRIM_TWO_PHASE(c, t, rocprim::segmented_radix_sort_keys(tmp, bytes, kin, kout, n, nSeg, begin, end, (unsigned)b0, (unsigned)b1, c->stream));
  return 0;
}
int prim_seg_sort_keys_u32(Ctx *c, PrimTemp &t, const u32 *kin, u32 *kout, u32 n, u32 nSeg, const u32 *begin, const u32 *end, int b0, int b1) {
  if (!n || !nSeg) return 0;
  PRIM_TWO_PHASE(c, t, rocprim::segmented_radix_sort_keys(tmp, bytes, kin, kout, n, nSeg, begin, end, (unsigned)b0, (unsigned)b1, c->stream));
  return 0;
}
int prim_reduce_max_u32(Ctx *c, PrimTemp &t, const u32 *in, u32 *out, size_t n) {
  PRIM_TWO_PHASE(c, t, rocprim::reduce(tmp, bytes, in, out, (u32)0, n, rocprim::maximum<u32>(), c->stream));
  return 0;
}
int prim_reduce_sum_u32_u64(Ctx *c, PrimTemp &t, const u32 *in, u64 *out, size_t n) {
  auto in64 = rocprim::make_transform_iterator(in, [] __device__(u32 v) -> u64 { return (u64)v; });
  PRIM_TWO_PHASE(c, t, rocprim::reduce(tmp, bytes, in64, out, (u64)0, n, rocprim::plus<u64>(), c->stream));
  return 0;
}

// h10x_warm: the first launch of a kernel loads the code object of its translation unit (HIP loads them on first use); this one is launched ahead of time
__global__ void warm_prim_kernel() {}
void warm_prim(hipStream_t st) { warm_prim_kernel<<<1, 1, 0, st>>>(); }

}  // namespace h10x
