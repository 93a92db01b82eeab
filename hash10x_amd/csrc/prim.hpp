// prim.hpp — device-wide sort/scan plumbing (rocPRIM) behind plain functions, so that the heavy
// templates are instantiated once (prim.hip). The hand-written kernels live in stage_*.hip.
#pragma once
#include "common.hpp"

namespace h10x {

// scratch for rocPRIM temporary storage, grown on demand, owned by the caller
struct PrimTemp { DevBuf<char> buf; };

int prim_run_ordinals_u64(Ctx *c, PrimTemp &t, const u64 *key, int cb, u32 *ord /* n + 1 */, size_t n);   // runs of equal key >> cb that start before i
int prim_exclusive_scan_u32(Ctx *c, PrimTemp &t, const u32 *in, u32 *out, size_t n);              // out[i] = sum in[0..i)
int prim_exclusive_scan_u32_u64(Ctx *c, PrimTemp &t, const u32 *in, u64 *out, size_t n);          // 64-bit accumulation
int prim_inclusive_scan_u32(Ctx *c, PrimTemp &t, const u32 *in, u32 *out, size_t n);
// stable LSD radix sorts on key bits [beginBit, endBit)
int prim_sort_pairs_u64_u32(Ctx *c, PrimTemp &t, const u64 *kin, u64 *kout, const u32 *vin, u32 *vout, size_t n, int beginBit, int endBit);
int prim_sort_pairs_u32_u32(Ctx *c, PrimTemp &t, const u32 *kin, u32 *kout, const u32 *vin, u32 *vout, size_t n, int beginBit, int endBit);
int prim_sort_keys_u64(Ctx *c, PrimTemp &t, const u64 *kin, u64 *kout, size_t n, int beginBit, int endBit);
struct Val16 { u64 a; u32 b, c; };                          // a 16-byte payload that rides through a sort (stage_b.hip: a distinct hash with its list's start and end)
int prim_sort_pairs_u32_v16(Ctx *c, PrimTemp &t, const u32 *kin, u32 *kout, const Val16 *vin, Val16 *vout, size_t n, int beginBit, int endBit);
// per-segment sort of 64-bit keys (segments = [begin[i], end[i]) ), keys must be distinct per segment
int prim_seg_sort_keys_u64(Ctx *c, PrimTemp &t, const u64 *kin, u64 *kout, u32 n, u32 nSeg, const u32 *begin, const u32 *end, int beginBit, int endBit);
int prim_seg_sort_keys_u32(Ctx *c, PrimTemp &t, const u32 *kin, u32 *kout, u32 n, u32 nSeg, const u32 *begin, const u32 *end, int beginBit, int endBit);
int prim_reduce_max_u32(Ctx *c, PrimTemp &t, const u32 *in, u32 *out, size_t n);
int prim_reduce_sum_u32_u64(Ctx *c, PrimTemp &t, const u32 *in, u64 *out, size_t n);

}  // namespace h10x
