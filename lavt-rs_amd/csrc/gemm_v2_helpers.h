// Helpers shared by the LDS-DMA GEMM generations (gemm_v2.hip: NT family; gemm_tn_v2.hip: TN family): DMA issue, counted waits, the
// transposing LDS reads, slot swizzles.  Internal linkage (anonymous namespace) in each translation unit.
#pragma once
#include <stdlib.h>

#include "gemm_common.h"

using namespace lavt_gemm;

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__device__ __forceinline__ void dma16(const void* src, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)lds_dst, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int G> __device__ __forceinline__ void wait_groups(int g) {      // leave g groups of G vector-memory ops in flight
    if (g <= 0) wait_vmcnt<0>();
    else if (g == 1) wait_vmcnt<G>();
    else if (g == 2) wait_vmcnt<2 * G>();
    else wait_vmcnt<3 * G>();
}
// ---- transposing LDS reads issued from inline asm --------------------------------------------------------------------
// hipcc puts `s_waitcnt vmcnt(0)` in front of the ds_read_tr builtin while LDS-DMA is in flight (it cannot tell the stages
// apart), which would serialise the pipeline.  Asm reads are invisible to that pass; we wait for them ourselves:
// (cdna_hip_programming.md rule 18).
typedef unsigned long long u64;
// One statement = all transposing reads of a K tile (both k-steps) + the wait, early-clobber outputs: the compiler can neither copy a
// destination before its data has landed nor schedule a consumer above the wait (5.7 form i).  One address VGPR per fragment (the slot
// swizzle of the k-major tiles permutes the fragments' 32-byte slots differently in every lane, so they are not a compile-time stride
// apart); the second 4-row block (HO) and the second k-step (KO) are immediates because the swizzle ignores those row bits.
template <int NF, int HO, int KO>
__device__ __forceinline__ void tr_read_frags(const unsigned (&a)[NF], u64 (&l0)[NF], u64 (&h0)[NF], u64 (&l1)[NF], u64 (&h1)[NF]) {
    static_assert(NF == 2 || NF == 4, "NF");
    if constexpr (NF == 4) {
        asm volatile(
            "ds_read_b64_tr_b16 %0, %16\n\tds_read_b64_tr_b16 %1, %16 offset:%c20\n\t"
            "ds_read_b64_tr_b16 %2, %17\n\tds_read_b64_tr_b16 %3, %17 offset:%c20\n\t"
            "ds_read_b64_tr_b16 %4, %18\n\tds_read_b64_tr_b16 %5, %18 offset:%c20\n\t"
            "ds_read_b64_tr_b16 %6, %19\n\tds_read_b64_tr_b16 %7, %19 offset:%c20\n\t"
            "ds_read_b64_tr_b16 %8, %16 offset:%c21\n\tds_read_b64_tr_b16 %9, %16 offset:%c21+%c20\n\t"
            "ds_read_b64_tr_b16 %10, %17 offset:%c21\n\tds_read_b64_tr_b16 %11, %17 offset:%c21+%c20\n\t"
            "ds_read_b64_tr_b16 %12, %18 offset:%c21\n\tds_read_b64_tr_b16 %13, %18 offset:%c21+%c20\n\t"
            "ds_read_b64_tr_b16 %14, %19 offset:%c21\n\tds_read_b64_tr_b16 %15, %19 offset:%c21+%c20\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(l0[0]), "=&v"(h0[0]), "=&v"(l0[1]), "=&v"(h0[1]), "=&v"(l0[2]), "=&v"(h0[2]), "=&v"(l0[3]), "=&v"(h0[3]),
              "=&v"(l1[0]), "=&v"(h1[0]), "=&v"(l1[1]), "=&v"(h1[1]), "=&v"(l1[2]), "=&v"(h1[2]), "=&v"(l1[3]), "=&v"(h1[3])
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "n"(HO), "n"(KO)
            : "memory");
    } else {
        asm volatile(
            "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:%c10\n\t"
            "ds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %9 offset:%c10\n\t"
            "ds_read_b64_tr_b16 %4, %8 offset:%c11\n\tds_read_b64_tr_b16 %5, %8 offset:%c11+%c10\n\t"
            "ds_read_b64_tr_b16 %6, %9 offset:%c11\n\tds_read_b64_tr_b16 %7, %9 offset:%c11+%c10\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(l0[0]), "=&v"(h0[0]), "=&v"(l0[1]), "=&v"(h0[1]), "=&v"(l1[0]), "=&v"(h1[0]), "=&v"(l1[1]), "=&v"(h1[1])
            : "v"(a[0]), "v"(a[1]), "n"(HO), "n"(KO)
            : "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
}
// 32-byte slot swizzle of a k-major [64][CH x 16 B] tile: the 16 K rows one transposing read touches (rows r, r+1, r+2, r+3 of four 8-row
// blocks) land in different slots.  Uses row bits 0, 1, 3, 4 only, so rows r + 4 and r + 32 share the swizzle of row r.
template <int CH> __device__ __forceinline__ int tn_swz(int kr) {
    if constexpr (CH == 8) {
        // 128-byte rows: two rows span the 64 banks, so row bit 0 already alternates the bank half; the two swizzle bits a 4-slot row has
        // go to row bits 1 and 3.  (With bits 0 and 1 -- the general formula -- rows r, r+8, r+16, r+24 of a transposing read met in the
        // same banks: 46 % of the LDS cycles of the 64x64-tile weight-gradient kernels were bank conflicts, rocprofv3 SQ_LDS_BANK_CONFLICT.)
        return (((kr >> 1) & 1) | ((kr >> 2) & 2)) << 1;
    }
    constexpr int FM = (CH / 2 - 1) < 15 ? (CH / 2 - 1) : 15;
    return (((kr & 3) | ((kr >> 1) & 12)) & FM) << 1;
}
__device__ __forceinline__ bf16x8 frag_from(u64 lo, u64 hi) {
    typedef __attribute__((__vector_size__(2 * sizeof(u64)))) u64 u64x2;
    u64x2 v = {lo, hi};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p;
}

}  // namespace
