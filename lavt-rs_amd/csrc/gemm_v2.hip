// Gather-GEMM, second generation (bf16): LDS-DMA multi-stage pipeline.
//
// Why: at batch 2/GPU the backbone GEMMs of Swin-B are small (M ~ 1.8-2.6 k rows).  Measured on MI355X
// (tools/gemm_bench.py), the register-staged kernel of gemm.hip is either L2-bandwidth bound (64x64 tiles: 32 flop/B
// of L2->LDS traffic) or latency bound (128x128 tiles, one workgroup per CU, one K tile in flight).  This kernel keeps
// the 128x128 tile (64 flop/B) and hides the latency with a 3-deep ring of K tiles filled by
// `global_load_lds_dwordx4` (HBM/L2 -> LDS directly, no VGPR staging), counted `s_waitcnt vmcnt(N)` so that later tiles
// stay in flight across the single raw `s_barrier` per K tile:
//
//     prologue: DMA tile 0 -> stage 0, tile 1 -> stage 1
//     for kt:   s_waitcnt vmcnt(L)   (tile kt landed; tile kt+1 may still fly)        L = DMA instructions / wave / tile
//               s_barrier            (everyone's tile kt landed; everyone done reading tile kt-1)
//               DMA tile kt+2 -> stage (kt+2)%3   (= the stage tile kt-1 occupied)
//               16 ds_read_b128 + 32 MFMA on stage kt%3
//
// The LDS image is lane-linear per DMA instruction (wave-uniform base + lane*16 B), so the XOR swizzle that makes the
// fragment reads conflict-free is applied to the per-lane SOURCE address (each lane fetches the logical chunk that
// belongs at its physical position); row gathers (window maps, 3x3 taps, concat) are per-lane source addresses too.
// Rows / chunks that must read zero (padding, halo, tails) fetch from a caller-provided zero page (`p.zeros`).
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
// One k-step (32 K rows) only: the 16-wave 256x256 tiles have 128 registers per lane and hold one k-step of operands at a time.
template <int NF, int HO, int KOFF>
__device__ __forceinline__ void tr_read_frags_step(const unsigned (&a)[NF], u64 (&l)[NF], u64 (&h)[NF]) {
    static_assert(NF == 4, "NF");
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8 offset:%c13\n\tds_read_b64_tr_b16 %1, %8 offset:%c13+%c12\n\t"
        "ds_read_b64_tr_b16 %2, %9 offset:%c13\n\tds_read_b64_tr_b16 %3, %9 offset:%c13+%c12\n\t"
        "ds_read_b64_tr_b16 %4, %10 offset:%c13\n\tds_read_b64_tr_b16 %5, %10 offset:%c13+%c12\n\t"
        "ds_read_b64_tr_b16 %6, %11 offset:%c13\n\tds_read_b64_tr_b16 %7, %11 offset:%c13+%c12\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(l[0]), "=&v"(h[0]), "=&v"(l[1]), "=&v"(h[1]), "=&v"(l[2]), "=&v"(h[2]), "=&v"(l[3]), "=&v"(h[3])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "n"(HO), "n"(KOFF)
        : "memory");
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

// SIMPLE = no conv taps, no concat source, K % 64 == 0: every lane's DMA source is a fixed pointer that advances by a constant per K tile,
// so the K loop carries ~3 instructions per DMA instead of the general path's address arithmetic (which made small GEMMs issue-bound:
// ~220 VALU/SALU instructions per K tile on the one wave a SIMD holds, measured 0.78 us per K tile at M=2592, N=512).
// MODE 2 = convolution taps with Cin (and the concat split) a multiple of the 64-wide K tile: the tap of a K tile is wave-uniform and walks
// forward with the K loop (no division), the lanes' voxel coordinates are computed once, so a neighbour fetch is three range checks and one
// address add.  MODE 0 (anything else) decodes every K tile from scratch.
typedef __attribute__((__vector_size__(8 * sizeof(int)))) int i32x8;
// fp8 (F8): the operand tiles are the same BYTES as the bf16 ones (rows of 128 B = 128 e4m3 elements, 16 per 16-byte chunk), so the DMA ring,
// the chunk swizzle and the fragment reads are unchanged; a lane feeds one v_mfma_scale_f32_16x16x128_f8f6f4 with the two chunks g and g + 4
// (g = lane / 16) of its row -- which 32 k of the 128 a lane group holds is immaterial as long as A and B agree (unit block scales); these
// two are the chunks the bf16 fragments of k-steps 0 and 1 read, i.e. the bank-conflict-free pattern (chunks 2g, 2g+1 conflict 2-way).
// LNA (LayerNorm-folded A operand, round 3): A holds the RAW rows of a LayerNorm input, B the gamma-folded weight (lavt_ln_fold).  Every workgroup
// streams the whole K = C of its rows, so the row sums / sums of squares are accumulated from the resident A tiles (v_dot2c_f32_bf16) and the
// accumulators become rstd_m (acc - mu_m wsum_n) before the ordinary epilogue, whose `bias` is then biasp = b + W beta: the LayerNorm launch and its
// [M, C] output disappear (norm2 -> fc1 of a Swin block, reference lib/backbone.py:243 + :24-30).
template <int BM, int BN, bool BKM, int STAGES, int WAVES, int MODE, bool DACT = false, bool F8 = false, bool LNA = false, bool GD = false, int LEAN = 0>
__global__ __launch_bounds__(WAVES * 64) void gemm_nt_v2_kernel(const lavt_gemm_nt_t p) {
    constexpr bool SIMPLE = MODE == 1, CONVFAST = MODE == 2;
    static_assert(!LNA || (MODE == 1 && !BKM && !DACT && !F8), "LNA: plain k-contiguous problems");
    using T = typename std::conditional<F8, unsigned char, bf16>::type;
    constexpr int BK = F8 ? 128 : 64, EPC = F8 ? 16 : 8;
    static_assert(!(F8 && (BKM || DACT)), "fp8: k-contiguous operands, plain epilogue");
    constexpr int WAVES_M = WAVES == 16 ? 4 : 2;             // wave grid: WAVES_M (M) x WAVES_N (N); 16 waves: 4 x 4 waves of 64 x 64
    constexpr int WAVES_N = WAVES / WAVES_M;
    constexpr int A_INSTR = BM / (8 * WAVES);                // DMA instructions per wave per K tile for A (8 rows x 8 chunks each)
    constexpr int B_CH = BN / EPC;                           // chunks per k-major B row: [BK][BN] unpadded, 32-byte slots swizzled by tn_swz
    constexpr int B_INSTR = BN / (8 * WAVES);                // (k-contiguous and k-major tiles both hold BN * BK elements)
    constexpr int L = A_INSTR + B_INSTR;
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = BN * 128;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MI = WM / 16, NI = WN / 16;
    static_assert(A_INSTR >= 1 && B_INSTR >= 1 && MI >= 1 && NI >= 1, "tile too small for this many waves");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile_id = xcd_tile_id(blockIdx.x, gridDim.x);
    const int tile_m = tile_id / tiles_n, tile_n = tile_id % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int bz = blockIdx.y;

    const T* A = reinterpret_cast<const T*>(p.A) + (int64_t)bz * p.strideA;
    const T* A2 = reinterpret_cast<const T*>(p.A2);
    const T* B = reinterpret_cast<const T*>(p.B) + (int64_t)bz * p.strideB;
    const T* Z = reinterpret_cast<const T*>(p.zeros);
    const bool conv = p.conv_kc > 0;
    const ConvGeom cg = conv_geom(p);

    // ---- per-lane DMA geometry ------------------------------------------------------------------------------
    // k-contiguous tiles: instruction ii covers rows ii*8 .. ii*8+7; lane -> row ii*8 + lane/8, physical chunk lane%8,
    // logical chunk (lane%8) ^ (row%8) = (lane%8) ^ (lane/8): constant per lane.
    const int cl = (lane & 7) ^ (lane >> 3);
    int a_src[A_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int m = m0 + (wave * A_INSTR + i) * 8 + (lane >> 3);
        int src = -1;
        if (m < p.M) src = p.a_rowmap ? p.a_rowmap[m] : m;
        a_src[i] = src;
    }
    int b_row[B_INSTR], b_col[B_INSTR];          // KC: b_row = n (or -1); KM: b_row = k row in tile (or -1), b_col = n (or -1)
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        if constexpr (!BKM) {
            const int n = n0 + (wave * B_INSTR + i) * 8 + (lane >> 3);
            b_row[i] = n < p.N ? n : -1;
            b_col[i] = 0;
        } else {
            const int q = (wave * B_INSTR + i) * 64 + lane;
            const int kr = q / B_CH, cc = (q - kr * B_CH) ^ tn_swz<B_CH>(kr);
            const int n = n0 + cc * EPC;
            const bool ok = n < p.N;
            b_row[i] = ok ? kr : -1;
            b_col[i] = n;
        }
    }

    // SIMPLE: per-lane running pointers (invalid rows / columns point at the zero page with step 0)
    const T* a_ptr[A_INSTR];
    const T* b_ptr[B_INSTR];
    int64_t b_step[B_INSTR];
    int a_step[A_INSTR];
    if constexpr (SIMPLE) {
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            a_ptr[i] = a_src[i] >= 0 ? A + (int64_t)a_src[i] * p.lda + cl * EPC : Z;
            a_step[i] = a_src[i] >= 0 ? BK : 0;
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            if constexpr (!BKM) {
                b_ptr[i] = b_row[i] >= 0 ? B + (int64_t)b_row[i] * p.ldb + cl * EPC : Z;
                b_step[i] = b_row[i] >= 0 ? BK : 0;
            } else {
                b_ptr[i] = b_row[i] >= 0 ? B + (int64_t)b_row[i] * p.ldb + b_col[i] : Z;
                b_step[i] = b_row[i] >= 0 ? (int64_t)BK * p.ldb : 0;
            }
        }
    }
    auto issue_simple = [&](int stage) {
        char* sbase = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            dma16(a_ptr[i], sbase + (wave * A_INSTR + i) * 1024);
            a_ptr[i] += a_step[i];
        }
        char* sb = sbase + A_BYTES;
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            dma16(b_ptr[i], sb + (wave * B_INSTR + i) * 1024);
            b_ptr[i] += b_step[i];
        }
    };
    // CONVFAST: everything a lane needs per K tile is precomputed -- a bit per tap "this row's neighbour lies inside the volume" and the row's
    // pointers into the two sources at the lane's channel chunk -- so a neighbour fetch is one bit test, one wave-uniform 64-bit offset and two
    // selects.  (The first version re-derived (z, y, x) + three range checks per DMA per K tile: ~130 instructions and 4 exec-mask branches per
    // wave per K tile between the barrier and the first MFMA, with all 16 waves of the 256x256 tile in lockstep: 27 of 146 us by ablation.)
    unsigned a_vmask[A_INSTR];
    const T* a_p1[A_INSTR];
    const T* a_p2[A_INSTR];
    int64_t b_lane_off[B_INSTR];
    int c_kin = 0, c_tap = 0, c_dz = 0, c_dy = 0, c_dx = 0;
    if constexpr (CONVFAST) {
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            unsigned m = 0;
            if (a_src[i] >= 0) {
                int z, y, x;
                conv_coords(cg, a_src[i], z, y, x);
                for (int t = 0; t < cg.taps; ++t) {
                    int dz, dy, dx;
                    conv_tap(cg, t, dz, dy, dx);
                    if (p.conv_flip) { dz = -dz; dy = -dy; dx = -dx; }
                    const bool ok = ((unsigned)(z + dz) < (unsigned)cg.d) & ((unsigned)(y + dy) < (unsigned)cg.h) & ((unsigned)(x + dx) < (unsigned)cg.w);
                    m |= (ok ? 1u : 0u) << t;
                }
            }
            a_vmask[i] = m;
            const int64_t row = a_src[i] >= 0 ? a_src[i] : 0;
            a_p1[i] = A + row * p.lda + cl * EPC;
            a_p2[i] = A2 ? A2 + row * p.lda2 + cl * EPC : a_p1[i];
        }
        if (p.conv_tap_split > 0) c_tap = bz * p.conv_tap_split;          // split reduction over the batch index: this entry's first tap
        conv_tap(cg, c_tap, c_dz, c_dy, c_dx);
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            if constexpr (!BKM) {
                b_ptr[i] = b_row[i] >= 0 ? B + (int64_t)b_row[i] * p.ldb + cl * EPC : Z;
                b_step[i] = b_row[i] >= 0 ? BK : 0;
            } else {
                b_lane_off[i] = b_row[i] >= 0 ? (int64_t)b_row[i] * p.ldb + b_col[i] : -1;
            }
        }
    }
    const bool has_a2 = p.A2 != nullptr;
    const int64_t lda1 = p.lda, lda2 = p.lda2;
    const int a_split = p.a_split, conv_kc = p.conv_kc, flip = p.conv_flip;
    auto issue_conv = [&](int stage) {
        char* sbase = smem + stage * STAGE_BYTES;
        // wave-uniform part (scalar unit): element offset of this K tile's (tap, channel block) relative to a row pointer
        const int delta = ((c_dz * cg.h + c_dy) * cg.w + c_dx) * (flip ? -1 : 1);
        const bool second = has_a2 && c_kin >= a_split;                              // a_split % 64 == 0
        const int64_t off = second ? (int64_t)delta * lda2 + (c_kin - a_split) : (int64_t)delta * lda1 + c_kin;
        const unsigned bit = 1u << c_tap;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const T* row = second ? a_p2[i] : a_p1[i];
            const T* src = (a_vmask[i] & bit) ? row + off : Z;
            dma16(src, sbase + (wave * A_INSTR + i) * 1024);
        }
        char* sb = sbase + A_BYTES;
        if constexpr (!BKM) {
#pragma unroll
            for (int i = 0; i < B_INSTR; ++i) {
                dma16(b_ptr[i], sb + (wave * B_INSTR + i) * 1024);
                b_ptr[i] += b_step[i];
            }
        } else {
            const T* bb = B + (int64_t)c_kin * p.ldb + (int64_t)c_tap * p.b_tap_stride;
#pragma unroll
            for (int i = 0; i < B_INSTR; ++i) dma16(b_lane_off[i] >= 0 ? bb + b_lane_off[i] : Z, sb + (wave * B_INSTR + i) * 1024);
        }
        // advance the cursor to the next K tile
        c_kin += BK;
        if (c_kin >= conv_kc) {
            c_kin = 0; ++c_tap;
            if (++c_dx > (cg.kw >> 1)) { c_dx = -(cg.kw >> 1); if (++c_dy > (cg.kh >> 1)) { c_dy = -(cg.kh >> 1); ++c_dz; } }
        }
    };
    auto issue = [&](int kt, int stage) {
        if constexpr (SIMPLE) { issue_simple(stage); return; }
        if constexpr (CONVFAST) { issue_conv(stage); return; }
        char* sbase = smem + stage * STAGE_BYTES;
        // A
        const int k = kt * BK + cl * EPC;
        int kin = k, dz = 0, dy = 0, dx = 0;
        if (conv) {
            const int tap = k / p.conv_kc;
            kin = k - tap * p.conv_kc;
            conv_tap(cg, tap, dz, dy, dx);
            if (p.conv_flip) { dz = -dz; dy = -dy; dx = -dx; }
        }
        const bool second = (p.A2 != nullptr) && kin >= p.a_split;
        const T* base = second ? A2 : A;
        const int64_t ld = second ? p.lda2 : p.lda;
        const int kk = second ? kin - p.a_split : kin;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            int src = a_src[i];
            if (conv) src = conv_nbr(cg, src, dz, dy, dx);
            const T* g = (src >= 0 && k < p.K) ? base + (int64_t)src * ld + kk : Z;
            dma16(g, sbase + (wave * A_INSTR + i) * 1024);
        }
        // B
        char* sb = sbase + A_BYTES;
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            const T* g = Z;
            if constexpr (!BKM) {
                if (b_row[i] >= 0 && k < p.K) g = B + (int64_t)b_row[i] * p.ldb + k;
            } else {
                const int kb = kt * BK + b_row[i];
                if (b_row[i] >= 0 && kb < p.K) {
                    int64_t off;
                    if (conv) { const int t2 = kb / p.conv_kc; off = (int64_t)(kb - t2 * p.conv_kc) * p.ldb + (int64_t)t2 * p.b_tap_stride; }
                    else off = (int64_t)kb * p.ldb;
                    g = B + off + b_col[i];
                }
            }
            dma16(g, sb + (wave * B_INSTR + i) * 1024);
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ktiles = (p.K + BK - 1) / BK;
    constexpr int LN_SU = LNA ? (BM * 8) / (WAVES * 64) : 1;           // 16-byte chunks of an A tile per thread
    static_assert(!LNA || LN_SU * WAVES * 64 == BM * 8, "LNA: whole chunks per thread");
    float ln_s1[LN_SU], ln_s2[LN_SU];
#pragma unroll
    for (int u = 0; u < LN_SU; ++u) { ln_s1[u] = 0.f; ln_s2[u] = 0.f; }
    typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2_t;
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < ktiles) issue(t, t);
    for (int kt = 0; kt < ktiles; ++kt) {
        wait_groups<L>(min(STAGES - 2, ktiles - 1 - kt));          // tile kt landed; up to STAGES-2 younger tiles stay in flight
        __builtin_amdgcn_s_barrier();
        if (kt + STAGES - 1 < ktiles) issue(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
        if constexpr (LNA) {
            const bf16* tA = reinterpret_cast<const bf16*>(smem + (kt % STAGES) * STAGE_BYTES);
            const bf16x2_t ones2 = {(bf16)1.0f, (bf16)1.0f};
            uint4 sc[LN_SU];
            lds_read16_n<LN_SU, WAVES * 8 * 128>(lds_byte_addr(tA) + (unsigned)tid * 16u, sc);      // chunk tid of rows tid / 8 + 8 WAVES u (any chunk order: sums only)
#pragma unroll
            for (int u = 0; u < LN_SU; ++u) {
                const uint4 c4 = sc[u];
                const unsigned wv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bf16x2_t v2 = __builtin_bit_cast(bf16x2_t, wv[e]);
                    ln_s1[u] = __builtin_amdgcn_fdot2_f32_bf16(v2, ones2, ln_s1[u], false);
                    ln_s2[u] = __builtin_amdgcn_fdot2_f32_bf16(v2, v2, ln_s2[u], false);
                }
            }
        }
        if constexpr (F8) {
            const bf16* cA = reinterpret_cast<const bf16*>(smem + (kt % STAGES) * STAGE_BYTES);          // byte image identical to a bf16 [rows][64] tile
            const bf16* cB = reinterpret_cast<const bf16*>(smem + (kt % STAGES) * STAGE_BYTES + A_BYTES);
            i32x8 fa8[MI], fb8[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = wm * WM + i * 16 + (lane & 15), g = lane >> 4;      // chunks g and g + 4: the bf16 fragments' conflict-free read pattern
                const uint4 lo = *reinterpret_cast<const uint4*>(cA + kc_off<bf16>(row, g)), hi = *reinterpret_cast<const uint4*>(cA + kc_off<bf16>(row, g + 4));
                fa8[i] = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int row = wn * WN + j * 16 + (lane & 15), g = lane >> 4;
                const uint4 lo = *reinterpret_cast<const uint4*>(cB + kc_off<bf16>(row, g)), hi = *reinterpret_cast<const uint4*>(cB + kc_off<bf16>(row, g + 4));
                fb8[j] = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)      // (B, A) operand order as in the bf16 path: the accumulators hold C^T tiles; formats 0 = e4m3; scales 0x7F = 2^0
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb8[j], fa8[i], acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            continue;
        } else {
        const T* cA = reinterpret_cast<const T*>(smem + (kt % STAGES) * STAGE_BYTES);
        const T* cB = reinterpret_cast<const T*>(smem + (kt % STAGES) * STAGE_BYTES + A_BYTES);
        if constexpr (WAVES == 16 && BKM) {
            // one k-step of fragments at a time (64 accumulator + 32 fragment registers of the 128 a lane has with 16 waves per workgroup);
            // the k-contiguous form below is plain ds_reads the compiler schedules itself (measured 137 vs 142 us on the decoder conv)
            unsigned ab[NI];
            const int row_off = 8 * (lane >> 4) + ((lane & 15) >> 2), sw = tn_swz<B_CH>(row_off);
#pragma unroll
            for (int j = 0; j < NI; ++j)
                ab[j] = lds_addr(cB) + (unsigned)(row_off * BN + ((((wn * WN) / 8 + 2 * j) ^ sw) + ((lane & 3) >> 1)) * 8 + (lane & 1) * 4) * 2;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 ga[MI], gb[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) ga[i] = frag_kc<T>(cA, wm * WM + i * 16, ks, lane);
                u64 l[NI], h[NI];
                if (ks == 0) tr_read_frags_step<NI, 4 * BN * 2, 0>(ab, l, h);
                else tr_read_frags_step<NI, 4 * BN * 2, 32 * BN * 2>(ab, l, h);
#pragma unroll
                for (int j = 0; j < NI; ++j) gb[j] = frag_from(l[j], h[j]);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mfma16<T>(gb[j], ga[i], acc[i][j]);
            }
            continue;
        }
        bf16x8 fa[2][MI], fb[2][NI];
        if constexpr (BKM) {
            // lane address of (k-step 0, fragment j): row 8*(lane>>4) + ((lane&15)>>2), swizzled 32-byte slot of column wn*WN + 16*j, 8-byte half
            const int row_off = 8 * (lane >> 4) + ((lane & 15) >> 2), sw = tn_swz<B_CH>(row_off);
            unsigned ab[NI];
#pragma unroll
            for (int j = 0; j < NI; ++j)
                ab[j] = lds_addr(cB) + (unsigned)(row_off * BN + ((((wn * WN) / 8 + 2 * j) ^ sw) + ((lane & 3) >> 1)) * 8 + (lane & 1) * 4) * 2;
            u64 l0[NI], h0[NI], l1[NI], h1[NI];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[ks][i] = frag_kc<T>(cA, wm * WM + i * 16, ks, lane);
            tr_read_frags<NI, 4 * BN * 2, 32 * BN * 2>(ab, l0, h0, l1, h1);
#pragma unroll
            for (int j = 0; j < NI; ++j) { fb[0][j] = frag_from(l0[j], h0[j]); fb[1][j] = frag_from(l1[j], h1[j]); }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[ks][i] = frag_kc<T>(cA, wm * WM + i * 16, ks, lane);
#pragma unroll
                for (int j = 0; j < NI; ++j) fb[ks][j] = frag_kc<T>(cB, wn * WN + j * 16, ks, lane);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = mfma16<T>(fb[ks][j], fa[ks][i], acc[i][j]);
        }
    }
    if constexpr (F8) {
        // dequantisation: the tensors were quantised as q = e4m3(x * 448 / amax); amax <= 0 stands for "scale 1" (uncalibrated first step)
        lavt_gemm_nt_t q = p;
        const float da = p.deq_a ? *p.deq_a : 0.f, db = p.deq_b ? *p.deq_b : 0.f;
        q.alpha = p.alpha * (da > 0.f ? da * (1.f / 448.f) : 1.f) * (db > 0.f ? db * (1.f / 448.f) : 1.f);
        nt_epilogue<bf16, MI, NI>(q, acc, m0 + wm * WM, n0 + wn * WN, lane, bz);
        return;
    } else {
    if constexpr (LNA) {
        __syncthreads();                                   // every wave is done with the ring: the row statistics take its first bytes
        float* ln_mu = reinterpret_cast<float*>(smem);
        float* ln_rs = ln_mu + BM;
        const float invK = 1.0f / (float)p.K;
#pragma unroll
        for (int u = 0; u < LN_SU; ++u) {
            float t1 = ln_s1[u], t2 = ln_s2[u];
            t1 += __shfl_xor(t1, 1, 64); t1 += __shfl_xor(t1, 2, 64); t1 += __shfl_xor(t1, 4, 64);
            t2 += __shfl_xor(t2, 1, 64); t2 += __shfl_xor(t2, 2, 64); t2 += __shfl_xor(t2, 4, 64);
            if ((tid & 7) == 0) {
                const int row = (tid >> 3) + (WAVES * 8) * u;
                const float m1 = t1 * invK, rs = rsqrtf(fmaxf(t2 * invK - m1 * m1, 0.f) + p.ln_eps);
                ln_mu[row] = m1;
                ln_rs[row] = rs;
                if (tile_n == 0 && m0 + row < p.M && p.ln_mean) { p.ln_mean[m0 + row] = m1; p.ln_rstd[m0 + row] = rs; }      // for the LayerNorm backward
            }
        }
        __syncthreads();
        // the NI weight-sum quadruples of this lane, loaded unconditionally and all at once (a per-(i, j) "in range ? load : 0" made hipcc branch around
        // every load with its own vmcnt(0): eight serial L2 round trips, +5 us on the 17 us fc1 GEMM); columns beyond N are never stored
        float4 w4[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn * WN + j * 16 + 4 * (lane >> 4);
            w4[j] = *reinterpret_cast<const float4*>(p.ln_wsum + (n + 3 < p.N ? n : 0));
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int ml = wm * WM + i * 16 + (lane & 15);
            const float mu = ln_mu[ml], rs = ln_rs[ml];
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                acc[i][j][0] = rs * (acc[i][j][0] - mu * w4[j].x); acc[i][j][1] = rs * (acc[i][j][1] - mu * w4[j].y);
                acc[i][j][2] = rs * (acc[i][j][2] - mu * w4[j].z); acc[i][j][3] = rs * (acc[i][j][3] - mu * w4[j].w);
            }
        }
    }
    if constexpr (DACT) { nt_epilogue<T, MI, NI, true, GD>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, bz); return; }
    if (!p.epi_lds || p.mul || p.c_f32 || (p.ldc % 8) || (p.C2 && (p.ldc2 % 8 || p.c_split % 8)) || (p.R && p.ldr % 8) || (p.Cpre && p.ldcpre % 8))
        nt_epilogue<T, MI, NI, false, GD, LEAN>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, bz);           // (GD = LAVT_ACT_GELU_D: its own instantiation of the LayerNorm-folded launch)
    else
        nt_epilogue_lds<BM, BN, MI, NI, GD>(p, acc, reinterpret_cast<bf16*>(smem), m0, n0, wm * WM, wn * WN, tid, lane, bz);
    }
}

static inline int conv_taps_of(const lavt_gemm_nt_t& p) {
    return (p.conv_kd > 0 ? p.conv_kd : 1) * (p.conv_kh > 0 ? p.conv_kh : 3) * (p.conv_kw > 0 ? p.conv_kw : 3);
}
template <int BM, int BN, bool BKM, int STAGES, int WAVES, int MODE, bool DACT = false, bool F8 = false, bool LNA = false, bool GD = false, int LEAN = 0> int launch_nt_v2_(const lavt_gemm_nt_t& p, hipStream_t st) {
    if constexpr (LEAN == 0 && !DACT && !LNA && !F8 && MODE != 0) {      // epilogue instantiations without the features a launch does not use (gemm_common.h)
        if (p.act == 0 && !p.mul && !p.Cpre && !p.C2) {
            if (!p.bias && !p.R && !p.row_scale && !p.c_rowmap) return launch_nt_v2_<BM, BN, BKM, STAGES, WAVES, MODE, DACT, F8, LNA, GD, 2>(p, st);
            return launch_nt_v2_<BM, BN, BKM, STAGES, WAVES, MODE, DACT, F8, LNA, GD, 1>(p, st);
        }
    }
    constexpr size_t lds = STAGES * (size_t)(BM * 128 + BN * 128);
    static bool attr_set = false;
    if (!attr_set && lds > 65536) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_v2_kernel<BM, BN, BKM, STAGES, WAVES, MODE, DACT, F8, LNA, GD, LEAN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            lavt_set_error("lavt_gemm_nt(v2): cannot reserve %zu bytes of LDS", lds);
            return LAVT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), p.batch);
    hipLaunchKernelGGL((gemm_nt_v2_kernel<BM, BN, BKM, STAGES, WAVES, MODE, DACT, F8, LNA, GD, LEAN>), grid, dim3(WAVES * 64), lds, st, p);
    LAVT_CHECK_LAUNCH("lavt_gemm_nt(v2)");
    return LAVT_OK;
}
// LayerNorm-folded A operand (ln_wsum != NULL): plain k-contiguous problems whose workgroups see whole rows (every tile walks all of K)
int launch_nt_v2_lna(const lavt_gemm_nt_t& p, hipStream_t st) {
    if (p.b_kmajor || p.conv_kc > 0 || p.A2 || p.a_rowmap || p.K % 64 || p.N % 4 || p.batch != 1 || p.dact_pre || !p.bias || p.alpha != 1.0f) {
        lavt_set_error("lavt_gemm_nt: ln_wsum (LayerNorm-folded A) needs a plain k-contiguous problem: no taps / concat / row gather / batch, K %% 64 == 0, bias = folded bias, alpha = 1");
        return LAVT_ERR_INVALID;
    }
    const long tiles128 = (long)cdiv(p.M, 128) * cdiv(p.N, 128), tiles64 = (long)cdiv(p.M, 64) * cdiv(p.N, 64);
    // (LAVT_ACT_GELU_D is its own instantiation: with the branch on p.act inside one kernel, the classic form ran 3 us per launch slower)
    if (p.act == LAVT_ACT_GELU_D) {
        if (tiles128 >= 200 && p.N >= 128) return tiles128 >= 600 ? launch_nt_v2_<128, 128, false, 2, 8, 1, false, false, true, true>(p, st) : launch_nt_v2_<128, 128, false, 4, 8, 1, false, false, true, true>(p, st);
        return tiles64 >= 600 ? launch_nt_v2_<64, 64, false, 2, 4, 1, false, false, true, true>(p, st) : launch_nt_v2_<64, 64, false, 4, 4, 1, false, false, true, true>(p, st);
    }
    if (tiles128 >= 200 && p.N >= 128) return tiles128 >= 600 ? launch_nt_v2_<128, 128, false, 2, 8, 1, false, false, true>(p, st) : launch_nt_v2_<128, 128, false, 4, 8, 1, false, false, true>(p, st);
    return tiles64 >= 600 ? launch_nt_v2_<64, 64, false, 2, 4, 1, false, false, true>(p, st) : launch_nt_v2_<64, 64, false, 4, 4, 1, false, false, true>(p, st);
}
// fp8 operands (LAVT_FP8): k-contiguous A and B, 128-element K tiles; 128x128 / 8 waves when that fills the chip, else 64x64 / 4 waves
template <int BM, int BN, int STAGES, int WAVES> int launch_nt_v2_f8(const lavt_gemm_nt_t& p, hipStream_t st) {
    const bool simple = p.conv_kc <= 0 && p.A2 == nullptr && p.K % 128 == 0;
    const bool convfast = p.conv_kc > 0 && p.conv_kc % 128 == 0 && (p.A2 == nullptr || p.a_split % 128 == 0) && conv_taps_of(p) <= 32;
    if (simple) return launch_nt_v2_<BM, BN, false, STAGES, WAVES, 1, false, true>(p, st);
    if (convfast) return launch_nt_v2_<BM, BN, false, STAGES, WAVES, 2, false, true>(p, st);
    return launch_nt_v2_<BM, BN, false, STAGES, WAVES, 0, false, true>(p, st);
}
int launch_nt_v2_fp8(const lavt_gemm_nt_t& p, hipStream_t st) {
    if (p.b_kmajor || p.dact_pre || p.c_f32 || p.lda % 16 || p.ldb % 16 || (p.A2 && (p.lda2 % 16 || p.a_split % 16)) || p.K % 16 || (p.conv_kc > 0 && p.conv_kc % 16) || !p.zeros) {
        lavt_set_error("lavt_gemm_nt(fp8): needs k-contiguous e4m3 operands with 16-byte aligned rows (lda, ldb, K, conv_kc, a_split %% 16 == 0), bf16 C, no dact_pre");
        return LAVT_ERR_INVALID;
    }
    const long tiles128 = (long)cdiv(p.M, 128) * cdiv(p.N, 128) * p.batch, tiles64 = (long)cdiv(p.M, 64) * cdiv(p.N, 64) * p.batch;
    if (tiles128 >= 200 && p.N >= 128) return tiles128 >= 600 ? launch_nt_v2_f8<128, 128, 2, 8>(p, st) : launch_nt_v2_f8<128, 128, 4, 8>(p, st);
    return tiles64 >= 600 ? launch_nt_v2_f8<64, 64, 2, 4>(p, st) : launch_nt_v2_f8<64, 64, 4, 4>(p, st);
}
// Fused activation-gradient epilogue (dact_pre): data gradients only (k-major B, plain K walk); the flag is a template parameter so that no
// other instantiation pays its registers (as a run-time branch in every kernel it cost 12 VGPRs and 0.25 ms per step in round 1).
int launch_nt_v2_dact(const lavt_gemm_nt_t& p, hipStream_t st) {
    if (!p.b_kmajor || p.conv_kc > 0 || p.A2 || p.K % 64 || p.lddact % 8 || p.c_f32 || p.C2 || p.Cpre || (p.R && !p.res_first) || p.act || p.mul) {
        lavt_set_error("lavt_gemm_nt: dact_pre needs a plain k-major data-gradient problem (no taps / concat / residual / activation, K %% 64 == 0)");
        return LAVT_ERR_INVALID;
    }
    const long tiles128 = (long)cdiv(p.M, 128) * cdiv(p.N, 128) * p.batch, tiles64 = (long)cdiv(p.M, 64) * cdiv(p.N, 64) * p.batch;
    // the stored-derivative form (dact = LAVT_ACT_STORED, no residual) is its own instantiation (GD): a multiply, none of the transcendental forms
    if (p.dact == LAVT_ACT_STORED && !p.R && !p.bias && !p.c_rowmap) {
        if (tiles128 >= 200 && p.N >= 128) return tiles128 >= 600 ? launch_nt_v2_<128, 128, true, 2, 8, 1, true, false, false, true>(p, st) : launch_nt_v2_<128, 128, true, 4, 8, 1, true, false, false, true>(p, st);
        return tiles64 >= 600 ? launch_nt_v2_<64, 64, true, 2, 4, 1, true, false, false, true>(p, st) : launch_nt_v2_<64, 64, true, 4, 4, 1, true, false, false, true>(p, st);
    }
    if (tiles128 >= 200 && p.N >= 128) return tiles128 >= 600 ? launch_nt_v2_<128, 128, true, 2, 8, 1, true>(p, st) : launch_nt_v2_<128, 128, true, 4, 8, 1, true>(p, st);
    return tiles64 >= 600 ? launch_nt_v2_<64, 64, true, 2, 4, 1, true>(p, st) : launch_nt_v2_<64, 64, true, 4, 4, 1, true>(p, st);
}
template <int BM, int BN, bool BKM, int STAGES, int WAVES> int launch_nt_v2(const lavt_gemm_nt_t& p, hipStream_t st) {
    static const bool general_only = getenv("LAVT_GEMM_GENERAL") != nullptr;
    const bool simple = p.conv_kc <= 0 && p.A2 == nullptr && p.K % 64 == 0;
    const bool convfast = p.conv_kc > 0 && p.conv_kc % 64 == 0 && (p.A2 == nullptr || p.a_split % 64 == 0) && conv_taps_of(p) <= 32;     // (a bit per tap)
    if (simple && !general_only) return launch_nt_v2_<BM, BN, BKM, STAGES, WAVES, 1>(p, st);
    if (convfast && !general_only) return launch_nt_v2_<BM, BN, BKM, STAGES, WAVES, 2>(p, st);
    return launch_nt_v2_<BM, BN, BKM, STAGES, WAVES, 0>(p, st);
}


// ================================================================================================ TN (weight gradients)
// C[I,J] (+)= alpha * sum_k A[k][i] B[k][j]; both operands k-major: LDS tiles [64 k][cols+16] filled by LDS-DMA through a STAGES-deep
// ring, every fragment read with the transposing LDS read (asm, one statement per operand per K tile).  The operands stream from HBM
// (every K tile is new data), so one tile in flight leaves the full HBM latency exposed per K tile (measured 0.9 us with the 2-stage
// ring); with S stages S-2 further tiles stay in flight across the loop-top wait.
//
// Row maps (window order <-> token order) and the DropPath / language row mask are per-K-row side inputs.  A register load of them
// would have to be the YOUNGEST outstanding vector-memory op when it is needed, and vmcnt retires in order -- waiting for it drains the
// whole ring.  So (MAPS = true) they travel through LDS as well: a 4-byte-per-lane DMA per wave per K tile, issued 2(S-1) tiles ahead
// into an 8-slot ring, read back with ds_read when the tile's row addresses are formed.  Every wave issues the same number of
// vector-memory ops per tile (ND tile DMAs + 1 map DMA), which is what makes the counted waits valid.
__device__ __forceinline__ void dma4(const void* src, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)lds_dst, 4, 0, 0);
}

// CS: how the bias gradient (colsum) is produced -- 0 none, 1 scalar walk of the LDS tile by the first BI threads, 2 on the matrix cores
// CONVP: the problem may be a convolution weight gradient (tap-shifted B rows); the grouped launch never is, and without the tap state (per-DMA
// coordinates, wrap tests behind uniform branches) its map-free K loop is shorter
template <int BI, int BJ, int WAVES, int STAGES, bool MAPS, int CS, bool CONVP = true>
__device__ __forceinline__ void tn_tile(const lavt_gemm_tn_t& p, const int tile_linear, const int bz, const int split_idx, const int kt_per_split,
                                        const int nsplit, char* smem) {
    using T = bf16;
    constexpr int BK = 64, EPC = 8;
    constexpr int WAVES_I = 2, WAVES_J = WAVES / WAVES_I;
    // LDS tiles are k-major [BK][BI] / [BK][BJ] without padding; bank conflicts of the transposing reads are avoided by the tn_swz slot
    // swizzle, applied on the DMA side through the source address each lane fetches (the LDS image of a DMA instruction is lane-linear).
    // (A padded row -- 16 extra elements -- cost a third DMA instruction per operand per K tile whose lanes mostly fetched the zero page.)
    constexpr int A_CH = BI / EPC, B_CH = BJ / EPC;
    constexpr int A_INSTR = BK * A_CH / (64 * WAVES), B_INSTR = BK * B_CH / (64 * WAVES);
    static_assert(A_INSTR * 64 * WAVES == BK * A_CH && B_INSTR * 64 * WAVES == BK * B_CH, "whole DMA instructions per operand tile");
    constexpr int A_BYTES = A_INSTR * WAVES * 1024, B_BYTES = B_INSTR * WAVES * 1024, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int WI = BI / WAVES_I, WJ = BJ / WAVES_J, II = WI / 16, JJ = WJ / 16;
    constexpr int G = A_INSTR + B_INSTR + (MAPS ? 1 : 0);              // vector-memory ops per wave per K tile
    constexpr int AHEAD = 2 * (STAGES - 1);                              // map DMA runs this many tiles ahead of the compute
    constexpr int NSLOT = AHEAD + 1, SLOT_BYTES = 768;                   // map ring: [slot][a_map | a_rowscale | b_map][64] (+ one 256 B spare for waves >= 3)
    static_assert((II == 2 || II == 4) && (JJ == 2 || JJ == 4), "fragment counts");
    static_assert(STAGES >= 2 && STAGES <= 4 && 3 * G <= 63, "pipeline depth");

    char* const maps = smem + STAGES * STAGE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave / WAVES_J, wj = wave % WAVES_J;
    const int tiles_j = (p.J + BJ - 1) / BJ;
    const int tile_i = tile_linear / tiles_j, tile_j = tile_linear % tiles_j;
    const int i0 = tile_i * BI, j0 = tile_j * BJ;
    const int ktiles = (p.K + BK - 1) / BK;
    const int kt_begin = split_idx * kt_per_split, kt_end = min(ktiles, kt_begin + kt_per_split);
    if (kt_begin >= kt_end) return;
    const int n = kt_end - kt_begin;

    const T* A = reinterpret_cast<const T*>(p.A) + (int64_t)bz * p.strideA;
    const T* B = reinterpret_cast<const T*>(p.B) + (int64_t)bz * p.strideB;
    const T* B2 = reinterpret_cast<const T*>(p.B2);
    const T* Z = reinterpret_cast<const T*>(p.zeros);
    const bool conv = CONVP && p.conv_kc > 0;
    const ConvGeom cg = conv_geom(p);

    // per-lane DMA geometry (constant over K tiles): LDS chunk q = (wave*INSTR + i)*64 + lane -> (k row, column chunk)
    int a_kr[A_INSTR], a_col[A_INSTR], b_kr[B_INSTR], b_dz[B_INSTR], b_dy[B_INSTR], b_dx[B_INSTR];
    const T* b_base[B_INSTR];
    int64_t b_ld[B_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int q = (wave * A_INSTR + i) * 64 + lane, kr = q / A_CH, cc = (q - kr * A_CH) ^ tn_swz<A_CH>(kr);
        const bool ok = i0 + cc * EPC < p.I;
        a_kr[i] = ok ? kr : -1;
        a_col[i] = i0 + cc * EPC;
    }
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int q = (wave * B_INSTR + i) * 64 + lane, kr = q / B_CH, cc = (q - kr * B_CH) ^ tn_swz<B_CH>(kr);
        const int jb = j0 + cc * EPC;
        const bool ok = jb < p.J;
        b_kr[i] = ok ? kr : -1;
        int jc = jb, dz = 0, dy = 0, dx = 0;
        if (conv) { const int tap = jb / p.conv_kc; jc = jb - tap * p.conv_kc; conv_tap(cg, tap, dz, dy, dx); }
        const bool second = (p.B2 != nullptr) && jc >= p.b_split;
        b_base[i] = (second ? B2 : B) + (second ? jc - p.b_split : jc);
        b_ld[i] = second ? p.ldb2 : p.ldb;
        b_dz[i] = dz; b_dy[i] = dy; b_dx[i] = dx;
    }

    // ---- side inputs of tile t (relative to kt_begin) -> map ring slot t % NSLOT: wave 0 a_rowmap, 1 a_rowscale, 2 b_rowmap, others spare
    const float inv_rsdiv = 1.0f / (float)(p.a_rowscale_div > 1 ? p.a_rowscale_div : 1);
    const int amask = p.a_rowmap ? -1 : 0, bmask = p.b_rowmap ? -1 : 0;
    const unsigned nors = p.a_rowscale ? 0u : 1u, lda_u = (unsigned)p.lda;
    const int Kdim = p.K;
    auto map_dma = [&](int t) {
        if constexpr (MAPS) {
            int k = (kt_begin + t) * BK + lane;
            k = k < p.K ? k : p.K - 1;
            const void* src = Z;
            if (wave == 0 && p.a_rowmap) src = p.a_rowmap + k;
            if (wave == 1 && p.a_rowscale) src = p.a_rowscale + (p.a_rowscale_div > 1 ? fdiv(k, p.a_rowscale_div, inv_rsdiv) : k);
            if (wave == 2 && p.b_rowmap) src = p.b_rowmap + k;
            dma4(src, wave < 3 ? maps + (t % NSLOT) * SLOT_BYTES + wave * 256 : maps + NSLOT * SLOT_BYTES);
        }
    };
    // !MAPS: the K rows are consecutive, so every lane keeps running source pointers (A, and B without taps) and, for conv taps, the
    // running (z, y, x) of its K row -- advanced by 64 rows per tile with a wrap test instead of three divisions per DMA per tile
    // (the conv weight-gradient K loop was instruction-bound: ~200 VALU instructions per K tile per wave against 8 MFMAs).
    const T* a_run[A_INSTR];
    const T* b_run[B_INSTR];
    int a_k[A_INSTR], b_k[B_INSTR], b_z[B_INSTR], b_y[B_INSTR], b_x[B_INSTR], b_delta[B_INSTR];
    if constexpr (!MAPS) {
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            a_k[i] = a_kr[i] >= 0 ? kt_begin * BK + a_kr[i] : (1 << 30);               // dead lanes sit beyond K for good
            a_run[i] = A + (int64_t)(a_kr[i] >= 0 ? a_k[i] : 0) * p.lda + a_col[i];
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            b_k[i] = b_kr[i] >= 0 ? kt_begin * BK + b_kr[i] : (1 << 30);
            const int k0 = b_kr[i] >= 0 ? b_k[i] : 0;
            b_delta[i] = conv ? (b_dz[i] * cg.h + b_dy[i]) * cg.w + b_dx[i] : 0;
            b_run[i] = b_base[i] + (int64_t)(k0 + b_delta[i]) * b_ld[i];
            b_z[i] = b_y[i] = b_x[i] = 0;
            if (conv) conv_coords(cg, k0, b_z[i], b_y[i], b_x[i]);
        }
    }
    auto issue = [&](int t, int stage) {
        char* sa = smem + stage * STAGE_BYTES;
        char* sb = sa + A_BYTES;
        if constexpr (!MAPS) {
#pragma unroll
            for (int i = 0; i < A_INSTR; ++i) {
                dma16(a_k[i] < p.K ? a_run[i] : Z, sa + (wave * A_INSTR + i) * 1024);
                a_run[i] += (int64_t)BK * p.lda;
                a_k[i] += BK;
            }
#pragma unroll
            for (int i = 0; i < B_INSTR; ++i) {
                bool ok = b_k[i] < p.K;
                if (conv) {
                    const int z = b_z[i] + b_dz[i], y = b_y[i] + b_dy[i], x = b_x[i] + b_dx[i];
                    ok = ok && (unsigned)z < (unsigned)cg.d && (unsigned)y < (unsigned)cg.h && (unsigned)x < (unsigned)cg.w;
                    // next tile: 64 rows further along x, carrying into y and z (and on into the next sample, whose z restarts at 0)
                    b_x[i] += BK;
                    if (b_x[i] >= cg.w) {
                        const int q = fdiv(b_x[i], cg.w, cg.inv_w);
                        b_x[i] -= q * cg.w;
                        b_y[i] += q;
                        if (b_y[i] >= cg.h) {
                            const int q2 = b_y[i] / cg.h;
                            b_y[i] -= q2 * cg.h;
                            b_z[i] = (b_z[i] + q2) % cg.d;
                        }
                    }
                }
                dma16(ok ? b_run[i] : Z, sb + (wave * B_INSTR + i) * 1024);
                b_run[i] += (int64_t)BK * b_ld[i];
                b_k[i] += BK;
            }
            return;
        }
        // mapped rows: branch-free -- the per-problem switches (which maps exist) are lane-uniform masks hoisted out of the loop; the first
        // version tested them per DMA inside the K loop: ~30 scalar branches per K tile, 14 vector instructions per MFMA (PMC)
        const int kbase = (kt_begin + t) * BK;
        const int* m_a = reinterpret_cast<const int*>(maps + (t % NSLOT) * SLOT_BYTES);
        const unsigned* m_rs = reinterpret_cast<const unsigned*>(maps + (t % NSLOT) * SLOT_BYTES + 256);
        const int* m_b = reinterpret_cast<const int*>(maps + (t % NSLOT) * SLOT_BYTES + 512);
        const T* a_ptr[A_INSTR];
        const T* b_ptr[B_INSTR];
        int ma[A_INSTR], mb[B_INSTR];
        unsigned mr[A_INSTR];
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) { ma[i] = m_a[a_kr[i] & 63]; mr[i] = m_rs[a_kr[i] & 63]; }      // all LDS reads first (dead lanes, a_kr = -1,
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) mb[i] = m_b[b_kr[i] & 63];                                      //  read slot 63 and are masked below)
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const int k = kbase + (a_kr[i] & 63);
            const int src = (ma[i] & amask) | (k & ~amask);
            const bool ok = ((a_kr[i] | src) >= 0) & (k < Kdim) & (((mr[i] << 1) | nors) != 0u);
            const unsigned off = ok ? (unsigned)src * lda_u + (unsigned)a_col[i] : 0u;
            a_ptr[i] = (ok ? A : Z) + off;
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            const int k = kbase + (b_kr[i] & 63);
            const int src = (mb[i] & bmask) | (k & ~bmask);
            const bool ok = ((b_kr[i] | src) >= 0) & (k < Kdim);
            const unsigned off = ok ? (unsigned)src * (unsigned)b_ld[i] : 0u;
            b_ptr[i] = (ok ? b_base[i] : Z) + off;
        }
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) dma16(a_ptr[i], sa + (wave * A_INSTR + i) * 1024);
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) dma16(b_ptr[i], sb + (wave * B_INSTR + i) * 1024);
    };

    // fragment read addresses relative to a stage's operand tile: K row of the lane, swizzled 32-byte slot of fragment i, 8-byte half
    unsigned relA[II], relB[JJ];
    {
        const int row_off = 8 * (lane >> 4) + ((lane & 15) >> 2);
        const int sA = tn_swz<A_CH>(row_off), sB = tn_swz<B_CH>(row_off);
#pragma unroll
        for (int i = 0; i < II; ++i) relA[i] = (unsigned)(row_off * BI + ((((wi * WI) / 8 + 2 * i) ^ sA) + ((lane & 3) >> 1)) * 8 + (lane & 1) * 4) * 2;
#pragma unroll
        for (int i = 0; i < JJ; ++i) relB[i] = (unsigned)(row_off * BJ + ((((wj * WJ) / 8 + 2 * i) ^ sB) + ((lane & 3) >> 1)) * 8 + (lane & 1) * 4) * 2;
    }
    f32x4 acc[II][JJ];
#pragma unroll
    for (int i = 0; i < II; ++i)
#pragma unroll
        for (int j = 0; j < JJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradient colsum[i] = sum_k A[k][i] of the tile_j == 0 workgroups.  CS == 2: one more "column" of the contraction -- the A fragments
    // times a fragment of ones, on the matrix cores, by the waves of the first wave column.  The scalar form (CS == 1: one thread per column
    // walking the 64 K rows of the LDS tile) runs on a single wave while the others wait at the barrier, with 46 % of its LDS cycles bank
    // conflicts: where few workgroups carry it (many column tiles: the grouped Swin-block launch, 62.8 -> 54.0 us) they were the tail of the
    // launch; where every second workgroup carries it (J = 128: PWAM's 1x1 convolutions) the scalar walk hides better.  A template
    // parameter: the extra accumulators cost the colsum-free 128x128 conv kernels 40 % when they were unconditional.
    const bool cs_wave = CS == 2 && (p.colsum != nullptr) && tile_j == 0 && wj == 0;
    f32x4 cacc[CS == 2 ? II : 1];
#pragma unroll
    for (int i = 0; i < (CS == 2 ? II : 1); ++i) cacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
    float csum = 0.f;
    const bool do_colsum = CS == 1 && (p.colsum != nullptr) && tile_j == 0 && tid < BI;

    // prologue: side inputs of the first AHEAD tiles, then the first STAGES-1 data tiles (each followed by one map DMA: uniform groups)
    if constexpr (MAPS) {
        for (int t = 0; t < AHEAD; ++t) map_dma(t);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
    }
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < n) { issue(t, t % STAGES); map_dma(t + AHEAD); }

    // One K tile.  `stage` is a compile-time constant in the two-stage ring (the loop below is unrolled by hand over the two stages), so the
    // fragment read addresses of a stage are loop-invariant registers instead of 2 (II + JJ) vector adds per K tile: the 64x64-tile kernels are
    // instruction-issue bound (three workgroups' waves share a SIMD: ~150 instructions per wave per K tile against 8 MFMAs).
    constexpr bool UNROLL2 = STAGES == 2 && BI == 64 && BJ == 64;       // (the 128x128 / 8-wave tiles sit at their 128-register cap: 16 more would spill)
    unsigned fragA[UNROLL2 ? 2 : 1][II], fragB[UNROLL2 ? 2 : 1][JJ];
    if constexpr (UNROLL2) {
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
            const unsigned baseA = lds_addr(smem + sg * STAGE_BYTES), baseB = baseA + A_BYTES;
#pragma unroll
            for (int i = 0; i < II; ++i) fragA[sg][i] = baseA + relA[i];
#pragma unroll
            for (int i = 0; i < JJ; ++i) fragB[sg][i] = baseB + relB[i];
        }
    }
    auto k_tile = [&](int j, auto stage_tag) {
        constexpr int SG = decltype(stage_tag)::value;             // >= 0: the stage of tile j (two-stage ring); -1: j % STAGES
        const int stage = SG >= 0 ? SG : j % STAGES;
        wait_groups<G>(min(STAGES - 2, n - 1 - j));
        __builtin_amdgcn_s_barrier();
        if (j + STAGES - 1 < n) { issue(j + STAGES - 1, SG >= 0 ? (SG ^ 1) : (j + STAGES - 1) % STAGES); map_dma(j + STAGES - 1 + AHEAD); }
        const char* cA = smem + stage * STAGE_BYTES;
        unsigned aA[II], aB[JJ];
        if constexpr (SG >= 0) {
#pragma unroll
            for (int i = 0; i < II; ++i) aA[i] = fragA[SG][i];
#pragma unroll
            for (int i = 0; i < JJ; ++i) aB[i] = fragB[SG][i];
        } else {
            const unsigned baseA = lds_addr(cA), baseB = baseA + A_BYTES;
#pragma unroll
            for (int i = 0; i < II; ++i) aA[i] = baseA + relA[i];
#pragma unroll
            for (int i = 0; i < JJ; ++i) aB[i] = baseB + relB[i];
        }
        u64 al0[II], ah0[II], al1[II], ah1[II], bl0[JJ], bh0[JJ], bl1[JJ], bh1[JJ];
        tr_read_frags<II, 4 * BI * 2, 32 * BI * 2>(aA, al0, ah0, al1, ah1);
        tr_read_frags<JJ, 4 * BJ * 2, 32 * BJ * 2>(aB, bl0, bh0, bl1, bh1);
#pragma unroll
        for (int i = 0; i < II; ++i)
#pragma unroll
            for (int jj = 0; jj < JJ; ++jj) {
                acc[i][jj] = mfma16<T>(frag_from(al0[i], ah0[i]), frag_from(bl0[jj], bh0[jj]), acc[i][jj]);
                acc[i][jj] = mfma16<T>(frag_from(al1[i], ah1[i]), frag_from(bl1[jj], bh1[jj]), acc[i][jj]);
            }
        if constexpr (CS == 2) {
            if (cs_wave) {
#pragma unroll
                for (int i = 0; i < II; ++i) {
                    cacc[i] = mfma16<T>(frag_from(al0[i], ah0[i]), ones, cacc[i]);
                    cacc[i] = mfma16<T>(frag_from(al1[i], ah1[i]), ones, cacc[i]);
                }
            }
        }
        if constexpr (CS == 1) {
            if (do_colsum) {
                const T* col = reinterpret_cast<const T*>(cA) + (tid & 7);
#pragma unroll 8
                for (int k = 0; k < BK; ++k) csum += to_f<T>(col[k * BI + (((tid >> 3) ^ tn_swz<A_CH>(k)) << 3)]);
            }
        }
    };
    if constexpr (UNROLL2) {
        int j = 0;
        for (; j + 1 < n; j += 2) { k_tile(j, std::integral_constant<int, 0>{}); k_tile(j + 1, std::integral_constant<int, 1>{}); }
        if (j < n) k_tile(j, std::integral_constant<int, 0>{});
    } else {
        for (int j = 0; j < n; ++j) k_tile(j, std::integral_constant<int, -1>{});
    }

    // split reduction with a partials buffer: this piece's tile goes to partials[piece][I][J] as plain stores (tn_reduce_pieces adds the pieces
    // into C afterwards); otherwise pieces meet in C through atomics
    const bool to_parts = p.partials != nullptr && nsplit > 1;
    float* const pbase = to_parts ? p.partials + (int64_t)bz * nsplit * ((int64_t)p.I * p.J + p.I) : nullptr;      // one [pieces][I][J] + [pieces][I] block per batch entry
    float* C = to_parts ? pbase + (int64_t)split_idx * p.I * p.J : p.C + (int64_t)bz * p.strideC;
    const int64_t ldc_out = to_parts ? p.J : p.ldc;
    float* colsum_out = to_parts ? pbase + (int64_t)nsplit * p.I * p.J + (int64_t)split_idx * p.I : (p.colsum ? p.colsum + (int64_t)bz * p.strideColsum : nullptr);
    const bool atomic = !to_parts && (nsplit > 1 || p.accumulate);
#pragma unroll
    for (int i = 0; i < II; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ii = i0 + wi * WI + i * 16 + 4 * (lane >> 4) + r;
            if (ii >= p.I) continue;
#pragma unroll
            for (int j = 0; j < JJ; ++j) {
                const int jj = j0 + wj * WJ + j * 16 + (lane & 15);
                if (jj >= p.J) continue;
                int64_t col = jj;
                if (CONVP && p.c_conv_permute) { const int t2 = jj / p.conv_kc; col = (int64_t)(jj - t2 * p.conv_kc) * cg.taps + t2; }
                float* dst = C + (int64_t)ii * ldc_out + col;
                if (atomic) atomicAdd(dst, p.alpha * acc[i][j][r]); else *dst = p.alpha * acc[i][j][r];
            }
        }
    }
    if constexpr (CS == 1) {
        if (do_colsum && i0 + tid < p.I) {
            float* cs = colsum_out + i0 + tid;
            if (atomic || (p.colsum_atomic && !to_parts)) atomicAdd(cs, csum * p.alpha); else *cs = csum * p.alpha;
        }
    }
    if (CS == 2 && cs_wave && (lane & 15) == 0) {          // every column of cacc holds the same sums: lanes of column 0 write rows 4 (lane / 16) + r
#pragma unroll
        for (int i = 0; i < (CS == 2 ? II : 1); ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ii = i0 + wi * WI + i * 16 + 4 * (lane >> 4) + r;
                if (ii >= p.I) continue;
                float* cs = colsum_out + ii;
                if (atomic || (p.colsum_atomic && !to_parts)) atomicAdd(cs, cacc[i][r] * p.alpha); else *cs = cacc[i][r] * p.alpha;      // one tile_j == 0 workgroup per I tile when the reduction is not split
            }
    }
}

template <int BI, int BJ, int WAVES, int STAGES, bool MAPS, int CS, bool CONVP = true>
__global__ __launch_bounds__(WAVES * 64) void gemm_tn_v2_kernel(const lavt_gemm_tn_t p, int kt_per_split) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tn_tile<BI, BJ, WAVES, STAGES, MAPS, CS, CONVP>(p, blockIdx.x, blockIdx.y, blockIdx.z, kt_per_split, gridDim.z, smem);
}

// Several independent weight-gradient problems in ONE launch (the four wgrads of a Swin block): together they fill the chip without
// split-K, so each output element has a single writer and is stored plainly instead of through fp32 atomics, which execute at the memory
// side at ~1.3 TB/s chip-wide (MI355X_MICROARCH.md) -- 44 MB of atomic traffic per stage-2 block with the per-problem split-K launches.
constexpr int TN_GROUP_MAX = 6;          // four weight gradients of a block + side members (the padded-row column sums of the windowed qkv bias gradient)
struct TnGroup {
    lavt_gemm_tn_t p[TN_GROUP_MAX];
    int tile_end[TN_GROUP_MAX];      // running count of workgroups (tiles x K splits) up to and including problem k
    int split[TN_GROUP_MAX];         // K splits of problem k (1 = single writer per output element, plain stores)
    int n;
};
template <int BI, int BJ, int WAVES, int STAGES, bool MAPS, int CS>
__global__ __launch_bounds__(WAVES * 64) void gemm_tn_v2_grouped_kernel(const TnGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int k = 0;
    while (k + 1 < g.n && (int)blockIdx.x >= g.tile_end[k]) ++k;
    const int local = blockIdx.x - (k ? g.tile_end[k - 1] : 0);
    const lavt_gemm_tn_t& p = g.p[k];
    const int ns = g.split[k], ktiles = (p.K + 63) / 64;
    // the splits of a tile sit next to each other (local % ns): neighbours in time share the output tile's cache lines for their atomics
    // a member without row maps / masks takes the map-free K loop also inside a group that has mapped members (the mapped loop carries ~14
    // vector instructions per MFMA: tools/wgrad_group_probe.py -- 42.5 us with maps on two of the four members vs 33.4 us without any)
    if constexpr (MAPS) {
        if (!(p.a_rowmap || p.a_rowscale || p.b_rowmap)) {
            tn_tile<BI, BJ, WAVES, STAGES, false, CS, false>(p, local / ns, 0, local % ns, (ktiles + ns - 1) / ns, ns, smem);
            return;
        }
    }
    tn_tile<BI, BJ, WAVES, STAGES, MAPS, CS, false>(p, local / ns, 0, local % ns, (ktiles + ns - 1) / ns, ns, smem);
}

// second stage of a split reduction through partial tiles: C[i][j] += sum_s parts[s][i][J + j], colsum[i] += sum_s parts[nsplit*I*J + s*I + i].
// 64 outputs x 4 piece lanes per workgroup (coalesced 256-byte reads, four independent loads in flight per thread), one writer per output.
__global__ __launch_bounds__(256) void tn_reduce_pieces(const float* __restrict__ parts, int nsplit, int I, int J, float* __restrict__ C, int64_t ldc,
                                                        float* __restrict__ colsum, int64_t strideC, int64_t strideColsum) {
    __shared__ float red[4][64];
    const int64_t W = (int64_t)I * J, total = W + (colsum ? I : 0);
    parts += (int64_t)blockIdx.y * nsplit * (W + I);          // batch entry
    C += (int64_t)blockIdx.y * strideC;
    if (colsum) colsum += (int64_t)blockIdx.y * strideColsum;
    const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + col;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < total) {
        // element e of piece s: the C part is [s][W], the colsum part [nsplit*W + s*I]
        const float* q = e < W ? parts + e : parts + (int64_t)nsplit * W + (e - W);
        const int64_t st = e < W ? W : I;
        int s = sl;
        for (; s + 12 < nsplit; s += 16) { a0 += q[(int64_t)s * st]; a1 += q[(int64_t)(s + 4) * st]; a2 += q[(int64_t)(s + 8) * st]; a3 += q[(int64_t)(s + 12) * st]; }
        for (; s < nsplit; s += 4) a0 += q[(int64_t)s * st];
    }
    red[sl][col] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0 && e < total) {
        const float t = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
        if (e < W) { const int64_t i = e / J; C[i * ldc + (e - i * J)] += t; }
        else colsum[e - W] += t;
    }
}

template <int BI, int BJ, int WAVES, int STAGES, bool MAPS, int CS, bool CONVP = !MAPS> int launch_tn_v2_(const lavt_gemm_tn_t& p, int split, hipStream_t st) {
    if constexpr (CONVP && !MAPS) {          // (mapped problems are never convolutions: tn_v2_eligible) the map-free kernel without the tap state for plain problems
        if (p.conv_kc <= 0 && !p.c_conv_permute) return launch_tn_v2_<BI, BJ, WAVES, STAGES, MAPS, CS, false>(p, split, st);
    }
    constexpr size_t lds = STAGES * (size_t)(64 * (BI + BJ) * 2) + (MAPS ? (2 * (STAGES - 1) + 1) * 768 + 256 : 0);
    static bool attr_set = false;
    if (!attr_set && lds > 65536) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_v2_kernel<BI, BJ, WAVES, STAGES, MAPS, CS, CONVP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            lavt_set_error("lavt_gemm_tn(v2): cannot reserve %zu bytes of LDS", lds);
            return LAVT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const int ktiles = cdiv(p.K, 64);
    const int per = cdiv(ktiles, split);
    dim3 grid(cdiv(p.I, BI) * cdiv(p.J, BJ), p.batch, cdiv(ktiles, per));
    const int pieces = (int)grid.z;
    // partial tiles instead of atomics when the caller lent scratch for them (plain problems only: no conv column permutation, batch 1)
    const bool parts = pieces > 1 && p.partials && !p.c_conv_permute && p.partials_floats >= (int64_t)p.batch * pieces * ((int64_t)p.I * p.J + p.I);
    if (!parts) {
        lavt_gemm_tn_t q = p;
        q.partials = nullptr;
        hipLaunchKernelGGL((gemm_tn_v2_kernel<BI, BJ, WAVES, STAGES, MAPS, CS, CONVP>), grid, dim3(WAVES * 64), lds, st, q, per);
    } else {
        hipLaunchKernelGGL((gemm_tn_v2_kernel<BI, BJ, WAVES, STAGES, MAPS, CS, CONVP>), grid, dim3(WAVES * 64), lds, st, p, per);
        const int64_t total = (int64_t)p.I * p.J + (p.colsum ? p.I : 0);
        hipLaunchKernelGGL(tn_reduce_pieces, dim3((unsigned)cdiv(total, 64), p.batch), dim3(256), 0, st, p.partials, pieces, p.I, p.J, p.C, p.ldc, p.colsum, p.strideC, p.strideColsum);
    }
    LAVT_CHECK_LAUNCH("lavt_gemm_tn(v2)");
    return LAVT_OK;
}
// two-stage ring only (3 / 4 stages cost a resident workgroup per CU and lost end to end in round 1: those instantiations are gone)
template <int BI, int BJ, int WAVES> int launch_tn_v2(const lavt_gemm_tn_t& p, int split, hipStream_t st) {
    const bool maps = p.a_rowmap || p.a_rowscale || p.b_rowmap;
    // colsum on the matrix cores when the workgroups carrying it are a minority (>= 4 column tiles), the scalar walk otherwise; 128x128: scalar
    const int cs = !p.colsum ? 0 : ((BI == 64 && cdiv(p.J, BJ) >= 4) ? 2 : 1);
    if constexpr (BI == 64) {
        if (cs == 2) return maps ? launch_tn_v2_<BI, BJ, WAVES, 2, true, 2>(p, split, st) : launch_tn_v2_<BI, BJ, WAVES, 2, false, 2>(p, split, st);
    }
    if (cs) return maps ? launch_tn_v2_<BI, BJ, WAVES, 2, true, 1>(p, split, st) : launch_tn_v2_<BI, BJ, WAVES, 2, false, 1>(p, split, st);
    return maps ? launch_tn_v2_<BI, BJ, WAVES, 2, true, 0>(p, split, st) : launch_tn_v2_<BI, BJ, WAVES, 2, false, 0>(p, split, st);
}

// the same for the members of a grouped launch that were cut into pieces through partial tiles: blockIdx.y = member
__global__ __launch_bounds__(256) void tn_reduce_pieces_group(const TnGroup g) {
    const lavt_gemm_tn_t& p = g.p[blockIdx.y];
    const int nsplit = g.split[blockIdx.y];
    if (nsplit <= 1 || p.partials == nullptr) return;
    __shared__ float red[4][64];
    const int64_t W = (int64_t)p.I * p.J, total = W + (p.colsum ? p.I : 0);
    if ((int64_t)blockIdx.x * 64 >= total) return;
    const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + col;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < total) {
        const float* q = e < W ? p.partials + e : p.partials + (int64_t)nsplit * W + (e - W);
        const int64_t st = e < W ? W : p.I;
        int s = sl;
        for (; s + 12 < nsplit; s += 16) { a0 += q[(int64_t)s * st]; a1 += q[(int64_t)(s + 4) * st]; a2 += q[(int64_t)(s + 8) * st]; a3 += q[(int64_t)(s + 12) * st]; }
        for (; s < nsplit; s += 4) a0 += q[(int64_t)s * st];
    }
    red[sl][col] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0 && e < total) {
        const float t = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
        if (e < W) { const int64_t i = e / p.J; p.C[i * p.ldc + (e - i * p.J)] += t; }
        else p.colsum[e - W] += t;
    }
}

}  // namespace

// returns 1 when the problem is not for this kernel (caller falls back to gemm.hip), else a LAVT status
int lavt_gemm_nt_v2(const lavt_gemm_nt_t& p, hipStream_t st) {
    if (p.dtype == LAVT_FP8) return launch_nt_v2_fp8(p, st);
    if (p.dtype != LAVT_BF16 || p.zeros == nullptr) return 1;
    const char* e = getenv("LAVT_GEMM_V2");
    if (e && e[0] == '0') return 1;
    if (p.lda % 8 || p.ldb % 8 || (p.A2 && p.lda2 % 8)) return 1;
    if (p.ln_wsum) return launch_nt_v2_lna(p, st);
    if (p.dact_pre) return launch_nt_v2_dact(p, st);
    // Dispatch measured on MI355X (tools/gemm_bench.py, hipGraph-timed): 128x128 tile with 8 waves (2 per SIMD: one wave's DMA issue and
    // LDS reads hide under the other's MFMAs) and a 2-stage ring (64-80 KiB -> 2 workgroups per CU) once there are >= 200 such tiles;
    // otherwise 64x64 tiles / 4 waves (5 workgroups per CU), 3 stages only for long-K problems with few tiles.
    const char* t = getenv("LAVT_GEMM_TILE");
    const int force = t ? atoi(t) : 0;
    const long tiles128 = (long)cdiv(p.M, 128) * cdiv(p.N, 128) * p.batch;
    const long tiles64 = (long)cdiv(p.M, 64) * cdiv(p.N, 64) * p.batch;
    // long reductions on few tiles (3-D convolutions of SepTPWAM: K = 27 C on 144 tiles of 128x128) also take the 128x128 tile: a launch lasts as
    // long as its serial chain of K tiles, and the larger tile moves half the bytes per K tile and flop
    static const int big_long = getenv("LAVT_GEMM_BIG_LONG") ? atoi(getenv("LAVT_GEMM_BIG_LONG")) : 128;
    const bool big = force ? force == 128 : ((tiles128 >= 200 || (big_long > 0 && p.K >= 64 * 64 && tiles128 >= big_long)) && p.N >= 128);
    const char* sg = getenv("LAVT_GEMM_STAGES");
    // Ring depth.  In isolation (operands L2-resident) 2 stages win everywhere; inside the training step the operands of the small
    // GEMMs arrive cold from HBM / Infinity Cache and a 4-deep ring is worth 0.8 ms per step.  The many-tile long-K problems (decoder
    // convolutions: every CU holds 2 workgroups and streams from L2) stay at 2 stages, which keeps two workgroups per CU resident.
    const long wgs = big ? tiles128 : tiles64;
    const int stages = sg ? atoi(sg) : (wgs >= 600 ? 2 : 4);
    const char* wv = getenv("LAVT_GEMM_WAVES");
    const int waves = wv ? atoi(wv) : 8;
#define GO(BM_, BN_, KM_, ST_, WV_) return launch_nt_v2<BM_, BN_, KM_, ST_, WV_>(p, st)
    // 128x256 tile, 8 waves of 64x64: fewer LDS bytes (DMA fill and fragment reads) per MFMA than 128x128; for the long-K, many-tile problems
    const long tiles256 = (long)cdiv(p.M, 128) * cdiv(p.N, 256) * p.batch;
    const bool wide = force ? force == 256 : (getenv("LAVT_GEMM_WIDE") != nullptr && tiles256 >= 256 && p.N % 256 == 0 && p.K >= 1024);
    if (wide) { if (p.b_kmajor) GO(128, 256, true, 2, 8); else GO(128, 256, false, 2, 8); }
    // 256x256 tile, 16 waves (4 x 4 of 64x64), one workgroup per CU: 128 flop per byte of LDS fill.  The 128x128 tile cannot keep enough bytes
    // in flight per CU to cover the L2 latency at the MFMA rate (160 KiB of LDS); measured 1.02 vs 0.82 PFLOP/s on the decoder conv shape.
    // Only when its tiles fill the 256 CUs well (whole rounds at >= 80 %).
    const long tiles256x = (long)cdiv(p.M, 256) * cdiv(p.N, 256) * p.batch;
    const long rounds = (tiles256x + 255) / 256;
    const bool huge = force ? force == 512 : (p.N % 256 == 0 && p.K >= 1024 && tiles256x >= 128 && tiles256x * 10 >= rounds * 256 * 8);
    if (huge) { if (p.b_kmajor) GO(256, 256, true, 2, 16); else GO(256, 256, false, 2, 16); }
    if (big) {
        if (waves == 8) {
            if (stages == 2) { if (p.b_kmajor) GO(128, 128, true, 2, 8); else GO(128, 128, false, 2, 8); }
            if (stages == 3) { if (p.b_kmajor) GO(128, 128, true, 3, 8); else GO(128, 128, false, 3, 8); }
            if (p.b_kmajor) GO(128, 128, true, 4, 8); else GO(128, 128, false, 4, 8);
        }
        if (stages == 2) { if (p.b_kmajor) GO(128, 128, true, 2, 4); else GO(128, 128, false, 2, 4); }
        if (p.b_kmajor) GO(128, 128, true, 3, 4); else GO(128, 128, false, 3, 4);
    }
    if (stages == 2) { if (p.b_kmajor) GO(64, 64, true, 2, 4); else GO(64, 64, false, 2, 4); }
    if (stages == 3) { if (p.b_kmajor) GO(64, 64, true, 3, 4); else GO(64, 64, false, 3, 4); }
    if (p.b_kmajor) GO(64, 64, true, 4, 4); else GO(64, 64, false, 4, 4);
#undef GO
}

static bool tn_v2_eligible(const lavt_gemm_tn_t& p) {
    if (p.dtype != LAVT_BF16 || p.zeros == nullptr) return false;
    if (p.lda % 8 || p.ldb % 8 || (p.B2 && p.ldb2 % 8)) return false;
    if (p.a_rowscale && !p.a_rowscale_binary) return false;
    if ((p.a_rowmap || p.a_rowscale || p.b_rowmap) && (p.conv_kc > 0 || p.B2)) return false;      // the mapped K loop has no taps / second source
    if ((int64_t)p.K * (p.lda > p.ldb ? p.lda : p.ldb) >= (1LL << 31)) return false;                // 32-bit element offsets
    return true;
}

// returns 1 when the group cannot run as one launch (the caller then issues the problems one by one)
int lavt_gemm_tn_grouped_v2(const lavt_gemm_tn_t* probs, int n, hipStream_t st) {
    const char* e = getenv("LAVT_GEMM_V2");
    if ((e && e[0] == '0') || n < 2 || n > TN_GROUP_MAX) return 1;
    TnGroup g;
    bool maps = false;
    int tiles = 0;
    // Tile configuration of the grouped launch: LAVT_TNG_CFG = "tile,waves,stages" (64,4,2 = the round-1/2 form).
    static int cfg_tile = 64, cfg_waves = 4, cfg_stages = 2;
    static bool cfg_read = false;
    if (!cfg_read) {
        cfg_read = true;
        const char* c = getenv("LAVT_TNG_CFG");
        if (c) sscanf(c, "%d,%d,%d", &cfg_tile, &cfg_waves, &cfg_stages);
    }
    int TB = cfg_tile;
    {   // the large tile only where its tiles still occupy most of the chip
        long t128 = 0;
        for (int i = 0; i < n; ++i) t128 += (long)cdiv(probs[i].I, 128) * cdiv(probs[i].J, 128);
        if (TB == 128 && t128 < 128) TB = 64;
    }
    bool any_colsum = false;
    // Pieces per member.  A member with a partials scratch (lavt_gemm_tn_t.partials: its pieces are stored as plain tiles and added into C by one
    // small second kernel) may be cut as finely as its K allows -- the long-K weight gradients of PWAM (K = 28 800 rows = 450 K tiles on 4
    // output tiles each) then run as ONE launch of ~1000 workgroups instead of four launches of ~230 at one workgroup per CU; a member
    // without one is cut into at most 4 pieces that meet through atomics (only if its C holds zeros: split_k < 0), and a chain of more than
    // 128 K tiles without a scratch keeps the group from forming (it would run serially while the short members supply the tile count).
    static const int chain = getenv("LAVT_TNG_CHAIN") ? atoi(getenv("LAVT_TNG_CHAIN")) : 48;     // 32 / 48 / 64 / 128: video step 23.16 / 22.93 / 22.82 / 22.87 ms, image step level
    static const int piece_tiles = getenv("LAVT_TNG_PIECE") ? atoi(getenv("LAVT_TNG_PIECE")) : 8;
    bool any_parts = false;
    int64_t max_total = 0;
    for (int per_piece = piece_tiles; ; per_piece *= 2) {
        tiles = 0; maps = false; any_colsum = false; any_parts = false; max_total = 0;
        for (int i = 0; i < n; ++i) {
            const lavt_gemm_tn_t& p = probs[i];
            if (!tn_v2_eligible(p) || p.batch != 1 || p.conv_kc > 0 || p.B2 || p.I % 8 || p.J % 8) return 1;
            const int ktiles = cdiv(p.K, 64);
            maps = maps || p.a_rowmap || p.a_rowscale || p.b_rowmap;
            any_colsum = any_colsum || p.colsum != nullptr;
            g.p[i] = p;
            int ns = 1;
            const int want = cdiv(ktiles, per_piece);
            const bool parts = p.partials != nullptr && !p.c_conv_permute && ktiles > chain && want > 1 &&
                               p.partials_floats >= (int64_t)want * ((int64_t)p.I * p.J + p.I);
            if (parts) ns = want;
            else {
                g.p[i].partials = nullptr;
                if (ktiles > 128) return 1;
                ns = (p.split_k < 0 && chain > 0) ? cdiv(ktiles, chain) : 1;
                if (ns > 4) ns = 4;
            }
            const int per = cdiv(ktiles, ns);
            ns = cdiv(ktiles, per);                      // no empty pieces
            if (parts && ns > 1) { any_parts = true; max_total = max_total > (int64_t)p.I * p.J + p.I ? max_total : (int64_t)p.I * p.J + p.I; }
            if (ns <= 1) g.p[i].partials = nullptr;
            g.split[i] = ns;
            tiles += cdiv(p.I, TB) * cdiv(p.J, TB) * ns;
            g.tile_end[i] = tiles;
        }
        if (tiles <= 2048 || !any_parts || per_piece >= 64) break;      // too many workgroups: longer pieces
    }
    for (int i = n; i < TN_GROUP_MAX; ++i) { g.p[i] = probs[0]; g.p[i].partials = nullptr; g.tile_end[i] = tiles; g.split[i] = 1; }
    g.n = n;
    if (tiles < (TB == 128 ? 128 : 256)) return 1;   // too few workgroups to fill the chip
    // (round 2: an XCD-contiguous tile order inside each member -- it cuts the 152 MB of fabric traffic -- and a 3-stage ring were both measured
    // on the step: 10.63 vs 10.64 ms and 10.81 vs 10.62 ms; neither is kept)
    if (TB == 128 || cfg_waves != 4 || cfg_stages != 2) {
#define TNG_GO(BT_, WV_, SG_)                                                                                                                          \
    do {                                                                                                                                               \
        const size_t l = SG_ * (size_t)(64 * (BT_ + BT_) * 2) + (maps ? (2 * (SG_ - 1) + 1) * 768 + 256 : 0);                                          \
        static bool attr = false;                                                                                                                      \
        if (!attr && l > 65536) {                                                                                                                      \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_v2_grouped_kernel<BT_, BT_, WV_, SG_, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l);  \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_v2_grouped_kernel<BT_, BT_, WV_, SG_, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l); \
            attr = true;                                                                                                                               \
        }                                                                                                                                              \
        if (maps) hipLaunchKernelGGL((gemm_tn_v2_grouped_kernel<BT_, BT_, WV_, SG_, true, 2>), dim3(tiles), dim3(WV_ * 64), l, st, g);                 \
        else hipLaunchKernelGGL((gemm_tn_v2_grouped_kernel<BT_, BT_, WV_, SG_, false, 2>), dim3(tiles), dim3(WV_ * 64), l, st, g);                     \
    } while (0)
        bool done = true;
        if (TB == 128 && cfg_waves == 8 && cfg_stages == 2) TNG_GO(128, 8, 2);
        else if (TB == 128 && cfg_waves == 8 && cfg_stages == 3) TNG_GO(128, 8, 3);
        else if (TB == 128 && cfg_waves == 8 && cfg_stages == 4) TNG_GO(128, 8, 4);
        else if (TB == 128 && cfg_waves == 4 && cfg_stages == 2) TNG_GO(128, 4, 2);
        else if (TB == 128 && cfg_waves == 4 && cfg_stages == 3) TNG_GO(128, 4, 3);
        else if (TB == 128 && cfg_waves == 4 && cfg_stages == 4) TNG_GO(128, 4, 4);
        else if (TB == 64 && cfg_waves == 4 && cfg_stages == 4) TNG_GO(64, 4, 4);
        else if (TB == 64 && cfg_waves == 4 && cfg_stages == 3) TNG_GO(64, 4, 3);
        else done = false;
#undef TNG_GO
        if (done) {
            if (any_parts) hipLaunchKernelGGL(tn_reduce_pieces_group, dim3((unsigned)cdiv(max_total, 64), n), dim3(256), 0, st, g);
            LAVT_CHECK_LAUNCH("lavt_gemm_tn_grouped(v2)");
            return LAVT_OK;
        }
    }
    const size_t lds = 2 * (size_t)(64 * (64 + 64) * 2) + (maps ? 3 * 768 + 256 : 0);
    if (any_colsum) {
        if (maps) hipLaunchKernelGGL((gemm_tn_v2_grouped_kernel<64, 64, 4, 2, true, 2>), dim3(tiles), dim3(256), lds, st, g);
        else hipLaunchKernelGGL((gemm_tn_v2_grouped_kernel<64, 64, 4, 2, false, 2>), dim3(tiles), dim3(256), lds, st, g);
    } else {
        if (maps) hipLaunchKernelGGL((gemm_tn_v2_grouped_kernel<64, 64, 4, 2, true, 0>), dim3(tiles), dim3(256), lds, st, g);
        else hipLaunchKernelGGL((gemm_tn_v2_grouped_kernel<64, 64, 4, 2, false, 0>), dim3(tiles), dim3(256), lds, st, g);
    }
    if (any_parts) hipLaunchKernelGGL(tn_reduce_pieces_group, dim3((unsigned)cdiv(max_total, 64), n), dim3(256), 0, st, g);
    LAVT_CHECK_LAUNCH("lavt_gemm_tn_grouped(v2)");
    return LAVT_OK;
}

int lavt_gemm_tn_v2(const lavt_gemm_tn_t& p, hipStream_t st) {
    if (p.dtype != LAVT_BF16 || p.zeros == nullptr) return 1;
    const char* e = getenv("LAVT_GEMM_V2");
    if (e && e[0] == '0') return 1;
    if (!tn_v2_eligible(p)) return 1;                            // (only 0 / constant row masks can be folded into the row fetch)
    // Measured (tools/gemm_bench.py tn): the 64x64 / 4-wave tile wins on every weight-gradient shape of the step, the conv wgrads included
    // (369 vs 230 TF/s for 128x128); the split-K factor trades workgroup count (latency hiding) against fp32 atomic traffic.
    const char* t = getenv("LAVT_GEMM_TILE");
    const int force = t ? atoi(t) : 0;
    const int ktiles = cdiv(p.K, 64);
    const long tiles64 = (long)cdiv(p.I, 64) * cdiv(p.J, 64) * p.batch;
    const long tiles128 = (long)cdiv(p.I, 128) * cdiv(p.J, 128) * p.batch;
    // conv weight gradients (long K, >= 48 tiles of 128x128 -- the Swin-T decoder's 384-channel convolutions have 102): the larger tile halves the
    // L2->LDS bytes per MFMA (measured 303 vs 357 us on Swin-B's; 16.9 -> 16.1 ms per Swin-T step)
    static const bool tn128 = getenv("LAVT_TN_BIG") == nullptr || getenv("LAVT_TN_BIG")[0] != '0';
    static const int tn_big_min = getenv("LAVT_TN_BIG_MIN") ? atoi(getenv("LAVT_TN_BIG_MIN")) : 48;
    const bool big = force ? force == 128 : (tn128 && p.conv_kc > 0 && tiles128 >= tn_big_min && ktiles >= 64);
    const long tiles = big ? tiles128 : tiles64;
    int split = p.split_k;
    { const char* se = getenv("LAVT_TN_SPLIT"); if (se) split = atoi(se); }
    if (split <= 0) {
        static const int target = getenv("LAVT_TN_TARGET") ? atoi(getenv("LAVT_TN_TARGET")) : 768;
        split = big ? (int)((384 + tiles / 2) / tiles) : (int)((target + tiles / 2) / tiles);      // ~3 (64-tile) / ~1.5 (128-tile) workgroups per CU
        const int long_k = (ktiles + 127) / 128;          // no workgroup walks more than ~128 K tiles
        if (split < long_k) split = long_k;
        const int max_split = (ktiles + 7) / 8;           // >= 8 K tiles per workgroup
        if (split > max_split) split = max_split;
        if (split < 1) split = 1;
    }
    if (split > ktiles) split = ktiles;
    if (big) return launch_tn_v2<128, 128, 8>(p, split, st);
    return launch_tn_v2<64, 64, 4>(p, split, st);
}
