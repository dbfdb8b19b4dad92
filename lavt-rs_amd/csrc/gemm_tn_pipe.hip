// Gather-GEMM, TN family, software-pipelined K loop (bf16, round 5): the grouped weight gradients of a Swin block -- the four Linear layers'
// dW = dY^T X over the tokens (reference lib/backbone.py:24-30, 121, 141: the backward autograd derives for fc1 / fc2 / qkv / proj) -- when the
// reduction is short enough for every output tile to have ONE writer (K <= 128 K tiles: stage 2 / 3 of the image models at 2-4 images per GPU,
// the video stages 2 / 3).
//
// Why a second TN kernel.  gemm_tn_v2.hip's grouped launch uses 64x64 output tiles so that ~800 workgroups fill the chip three deep; both operands
// are k-major, so every tile streams its own [K][64] panels: 367 MB of L2 -> LDS fill for the 44 MB of operands of a stage-2 block, a serial chain
// of 29 K tiles of ~1.2 us per workgroup, MFMA busy 12 % (rocprofv3 --pmc, profiles/r04_pmc.json) -- 0.11 of the bf16 peak for three rounds.
// Every rearrangement inside that structure (deeper rings, 128x128 tiles under a 128-register cap, rectangular tiles, stream-K, k-split waves) measured
// slower.  What the decoder's pipelined NT kernel showed in round 4 is that a CU ingests ~54 GB/s whatever the tile, and that
// this rate is only reached when fragment reads, MFMAs and the DMA issue are overlapped BY HAND.  So:
//   * 128x128 output tiles, one workgroup of 8 waves (2 x 4, wave tiles of 64 x 32, up to 256 registers per lane) per CU: half the fill bytes per
//     flop of the 64x64 launch (183 MB for the stage-2 block: 198 tiles x 29 K tiles x 32 KB);
//   * both operand tiles k-major [64 k][128] in a STAGES-deep LDS ring filled by buffer-descriptor LDS-DMA (a lane's offset is one 32-bit
//     register; rows beyond K and columns beyond I / J are offsets beyond the descriptor's range: the hardware writes zeros -- no zero page, no
//     per-lane validity arithmetic on the unmapped path);
//   * every fragment through the transposing LDS read (ds_read_b64_tr_b16, slot-swizzled tiles as in gemm_tn_v2.hip), issued from inline asm one
//     MFMA group ahead into double-buffered registers and waited for with counted `s_waitcnt lgkmcnt(n)`; ONE barrier per K tile in front of the
//     second group, after which the stage just consumed is refilled, two MFMAs per DMA instruction;
//   * row maps (window order <-> token order) arrive through the SCALAR unit: a wave's two DMA instructions per operand cover 8 consecutive K rows,
//     so one s_load_dwordx8 per mapped operand per K tile, issued at the top of the K tile and complete at its `lgkmcnt(0)` -- no vector-memory
//     load whose position in the in-order vmcnt queue would drain the ring (the reason gemm_tn_v2.hip carries its maps through an LDS ring);
//     the DropPath row mask is a 64-bit keep mask over the samples, read once;
//   * bias gradients (column sums of A) on the matrix cores against a fragment of ones, by the first wave column of the tile_j == 0 workgroups;
//   * a LayerNorm backward can ride as extra workgroups (two 256-thread units per 512-thread workgroup) on the CUs the 198 tiles leave idle.
// Accumulators hold C^T fragments (B fragment as the first MFMA operand): a lane owns 4 consecutive j of one row i -- 16-byte stores.
#include "gemm_v2_helpers.h"
#include "ln_bwd_body.h"

namespace {

#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t tnp_rsrc_t;
__device__ __forceinline__ tnp_rsrc_t tnp_buf(const void* base, unsigned bytes) {          // raw (stride 0) addressing, offsets >= bytes read zero
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void tnp_dma16(tnp_rsrc_t rs, void* lds_dst, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_dst, 16, voff, 0, 0, 0);
}
#else
struct tnp_rsrc_t { int unused; };
__host__ __device__ inline tnp_rsrc_t tnp_buf(const void*, unsigned) { return tnp_rsrc_t{0}; }
__host__ __device__ inline void tnp_dma16(tnp_rsrc_t, void*, unsigned) {}
#endif

typedef int tnp_i32x8 __attribute__((ext_vector_type(8)));
typedef tnp_i32x8 __attribute__((aligned(4))) tnp_i32x8_u;          // (row maps are 4-byte aligned: s_load_dwordx8 needs no more)
typedef __attribute__((address_space(4))) const tnp_i32x8_u* tnp_cptr8;
typedef __attribute__((address_space(4))) const int* tnp_cptr1;

// transposing reads of NF fragments of one k-step (two reads per fragment: K rows r and r + 4 of the lane's 8-row block), issued only
template <int NF, int HO, int KOFF> __device__ __forceinline__ void tnp_issue_tr(const unsigned (&a)[NF], u64 (&l)[NF], u64 (&h)[NF]) {
    static_assert(NF == 2 || NF == 4, "NF");
    if constexpr (NF == 4)
        asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%c13\n\tds_read_b64_tr_b16 %1, %8 offset:%c13+%c12\n\t"
                     "ds_read_b64_tr_b16 %2, %9 offset:%c13\n\tds_read_b64_tr_b16 %3, %9 offset:%c13+%c12\n\t"
                     "ds_read_b64_tr_b16 %4, %10 offset:%c13\n\tds_read_b64_tr_b16 %5, %10 offset:%c13+%c12\n\t"
                     "ds_read_b64_tr_b16 %6, %11 offset:%c13\n\tds_read_b64_tr_b16 %7, %11 offset:%c13+%c12"
                     : "=&v"(l[0]), "=&v"(h[0]), "=&v"(l[1]), "=&v"(h[1]), "=&v"(l[2]), "=&v"(h[2]), "=&v"(l[3]), "=&v"(h[3])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "n"(HO), "n"(KOFF) : "memory");
    else
        asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%c7\n\tds_read_b64_tr_b16 %1, %4 offset:%c7+%c6\n\t"
                     "ds_read_b64_tr_b16 %2, %5 offset:%c7\n\tds_read_b64_tr_b16 %3, %5 offset:%c7+%c6"
                     : "=&v"(l[0]), "=&v"(h[0]), "=&v"(l[1]), "=&v"(h[1]) : "v"(a[0]), "v"(a[1]), "n"(HO), "n"(KOFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void tnp_rd1(unsigned a, u64& d) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%c2" : "=&v"(d) : "v"(a), "n"(OFF) : "memory");
}
__device__ __forceinline__ void tnp_tie1(u64& l, u64& h) { asm volatile("" : "+v"(l), "+v"(h)); }
template <int CNT> __device__ __forceinline__ void tnp_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CNT) : "memory"); }
// ties registers to the wait in front of it (consumers cannot be scheduled above this empty statement, which stays behind the wait)
template <int N> __device__ __forceinline__ void tnp_tie(u64 (&l)[N], u64 (&h)[N]) {
    if constexpr (N == 4) asm volatile("" : "+v"(l[0]), "+v"(h[0]), "+v"(l[1]), "+v"(h[1]), "+v"(l[2]), "+v"(h[2]), "+v"(l[3]), "+v"(h[3]));
    else asm volatile("" : "+v"(l[0]), "+v"(h[0]), "+v"(l[1]), "+v"(h[1]));
}
template <int N, typename F> __device__ __forceinline__ void tnp_static_for(F&& f) {
    if constexpr (N > 0) {
        tnp_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

constexpr int TNP_MAX = 6;
enum { TNP_ACCUMULATE = 1, TNP_COLSUM_ATOMIC = 2, TNP_NO_B = 4, TNP_VEC4 = 8, TNP_PART_VEC4 = 16, TNP_WT = 32 };
struct TnpMember {
    const bf16* A; const bf16* B; float* C; float* colsum;
    const int32_t* a_map; const int32_t* b_map; const float* a_rs;
    int64_t ldc;
    int lda, ldb, I, J, K, tiles_j, tile_end, rs_div, rs_n, flags;
    float alpha;
    // K pieces (long reductions on few output tiles: the stage-0 / stage-1 blocks, PWAM's 1x1 convolutions): workgroup (tile, piece) contracts K tiles
    // [piece * kt_per, + kt_per) and stores its tile plainly into part[piece][I][J] (+ [pieces][I] column sums behind them); tnp_reduce_pieces adds them into C
    float* part;
    int pieces, kt_per, tiles;
};
struct TnpGroup { TnpMember m[TNP_MAX]; int n, tiles, dbg; };
struct TnpRider {
    const bf16* dy; const bf16* x; const float* gamma; const float* mean; const float* rstd; bf16* dx; float* partials; const bf16* dres;
    int rows, C, blocks;
};

constexpr unsigned TNP_OOB = 0x80000000u;
// Ablation builds (tools/r05_tnp_ablate.sh compiles this file with -DTNP_ABL=n into libraries OUTSIDE the shipped one; the shipped build has no switch):
// bit 0: no MFMAs; bit 1: no fragment reads; bit 2: no DMA inside the K loop (the ring keeps the prologue's tiles); bit 3: ONE K tile (what a launch costs
// outside its K loop: dispatch, prologue, epilogue)
#ifndef TNP_ABL
#define TNP_ABL 0
#endif

// One 128x128 output tile of member m.  MAPS: the member has a row map or a row mask (per-K-tile offset arithmetic); else the K rows are
// consecutive and a lane's offsets just advance.
template <int STAGES, int MAPS>          // MAPS 0: plain rows; 1: row maps / row mask; 2: + a mapped reduction whose length is not a multiple of 8
__device__ __forceinline__ void tnp_tile(const TnpMember& m, const int local, char* smem) {
    constexpr int BT = 128, BK = 64, CH = BT / 8, L = 4;
    constexpr int TILE_BYTES = BK * BT * 2, STAGE_BYTES = 2 * TILE_BYTES;
    constexpr int HO = 4 * BT * 2, KOFF = 32 * BT * 2;
    static_assert(STAGES >= 2 && STAGES <= 5, "ring depth (5 x 32 KB = the whole 160 KB of a CU; 2 x 32 KB lets two workgroups share a CU)");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 2, wj = wave & 3;                       // 2 (i) x 4 (j) waves: wave tile 64 x 32
    const int l15 = lane & 15, g4 = lane >> 4;
    // piece-major: the workgroups of one piece sweep the same K range together (they share operand panels in time)
    const int piece = m.pieces > 1 ? local / m.tiles : 0, tile = local - piece * m.tiles;
    const int tile_i = tile / m.tiles_j, tile_j = tile - tile_i * m.tiles_j;
    const int i0 = tile_i * BT, j0 = tile_j * BT;
    const int kt_all = (m.K + BK - 1) / BK, kt0 = piece * m.kt_per;
    const int ktiles = (TNP_ABL & 8) ? 1 : min(kt_all, kt0 + m.kt_per) - kt0;          // K tiles of this workgroup: [kt0, kt0 + ktiles)
    const int Kd = min(m.K, (kt0 + ktiles) * BK);                                      // rows behind this workgroup's range read zeros (descriptor range / row test)
    const int k_first = kt0 * BK;
    const int lda2 = m.lda * 2, ldb2 = m.ldb * 2;                 // row strides in bytes
    const bool has_b = !(m.flags & TNP_NO_B);

    // ---- DMA geometry: instruction i of wave w fills LDS chunks q = (2 w + i) * 64 + lane of an operand tile: K row q / 16 = 8 w + 4 i + lane / 16,
    //      slot q % 16 holding the column chunk (q % 16) ^ swizzle(row) -------------------------------------------------------------------
    // descriptor ranges: an unmapped operand ends behind its K rows (tiles beyond K read zeros with no arithmetic); a mapped one is addressed by
    // source-row index into a tensor whose extent the problem does not state -- a 2 GB window, rows beyond K masked in the offset arithmetic
    const int32_t* const map_a = m.a_map;
    const int32_t* const map_b = m.b_map;
    const bool has_rs = m.a_rs != nullptr;
    const tnp_rsrc_t rs_a = tnp_buf(m.A, (MAPS && map_a) ? 0x7fffffffu : (unsigned)Kd * (unsigned)lda2);
    const tnp_rsrc_t rs_b = tnp_buf(has_b ? m.B : m.A, !has_b ? 0u : (MAPS && map_b) ? 0x7fffffffu : (unsigned)Kd * (unsigned)ldb2);
    unsigned a_vo[2], b_vo[2];          // MAPS: column byte offset (or TNP_OOB); else the running byte offset of the lane's chunk
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = (wave * 2 + i) * 64 + lane, kr = q >> 4;
        const int cc = (q & 15) ^ tn_swz<CH>(kr);
        const int ca = i0 + cc * 8, cb = j0 + cc * 8;
        if constexpr (MAPS) {
            a_vo[i] = ca < m.I ? (unsigned)ca * 2u : TNP_OOB;
            b_vo[i] = (has_b && cb < m.J) ? (unsigned)cb * 2u : TNP_OOB;
        } else {
            a_vo[i] = ca < m.I ? (unsigned)(k_first + kr) * (unsigned)lda2 + (unsigned)ca * 2u : TNP_OOB;
            b_vo[i] = (has_b && cb < m.J) ? (unsigned)(k_first + kr) * (unsigned)ldb2 + (unsigned)cb * 2u : TNP_OOB;
        }
    }
    const unsigned a_adv = (unsigned)BK * (unsigned)lda2, b_adv = (unsigned)BK * (unsigned)ldb2;

    // ---- MAPS: side inputs.  Row maps through the scalar unit (8 consecutive entries per wave per operand per K tile); the DropPath / language
    //      row mask as a keep bit per sample (<= 64 samples) with a running (row mod rows-per-sample, sample) pair per DMA instruction ---------
    const int amask = map_a ? -1 : 0, bmask = map_b ? -1 : 0;
    unsigned long long keep = ~0ull;
    int rs_k[2] = {0, 0}, rs_s[2] = {0, 0};
    const int rs_div = m.rs_div;
    if constexpr (MAPS) {
        if (has_rs) {
            const float f = lane < m.rs_n ? m.a_rs[lane] : 1.0f;
            keep = __ballot(f != 0.0f);
#pragma unroll
            for (int i = 0; i < 2; ++i) {          // (sample, row inside the sample) of this lane's first K row
                const int k0 = k_first + wave * 8 + i * 4 + g4;
                rs_s[i] = k0 / rs_div;
                rs_k[i] = k0 - rs_s[i] * rs_div;
            }
        }
    }
    tnp_i32x8 sa = {0, 0, 0, 0, 0, 0, 0, 0}, sb = {0, 0, 0, 0, 0, 0, 0, 0};
    // rows of K tile t for this wave: entries [t * 64 + 8 wave, + 8) of either map (K % 8 == 0 on this path: a wave's eight rows are all inside K
    // or all beyond it; beyond K nothing is read -- entry 0 stands in and the rows are masked by k < K below)
    auto map_fetch = [&](int t) {
        if constexpr (MAPS) {
            int e = k_first + t * BK + wave * 8;
            e = e < Kd ? e : 0;
            if constexpr (MAPS == 2) {
                // (round 5: the last stage's 450 tokens, batch 4's 900) the map ends inside this wave's eight entries: read them one by one, clamped to the
                // last entry -- one wide load would read up to 28 bytes past the caller's map; the rows at and beyond K are masked by k < Kd below
                if (e + 8 > m.K) {
                    const int last = m.K - 1;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int idx = min(e + j, last);
                        if (map_a) sa[j] = *reinterpret_cast<tnp_cptr1>(reinterpret_cast<uintptr_t>(map_a + idx));
                        if (map_b) sb[j] = *reinterpret_cast<tnp_cptr1>(reinterpret_cast<uintptr_t>(map_b + idx));
                    }
                    return;
                }
            }
            if (map_a) sa = *reinterpret_cast<tnp_cptr8>(reinterpret_cast<uintptr_t>(map_a + e));
            if (map_b) sb = *reinterpret_cast<tnp_cptr8>(reinterpret_cast<uintptr_t>(map_b + e));
        }
    };
    auto sel4 = [&](const tnp_i32x8& s, int i) -> int {          // entry 4 i + lane / 16
        const int v0 = i ? s[4] : s[0], v1 = i ? s[5] : s[1], v2 = i ? s[6] : s[2], v3 = i ? s[7] : s[3];
        return g4 == 0 ? v0 : g4 == 1 ? v1 : g4 == 2 ? v2 : v3;
    };
    // DMA instruction idx (0, 1: A rows; 2, 3: B rows) of K tile t into the stage at sbase
    auto issue_one = [&](auto idx_c, int t, char* sbase) {
        constexpr int idx = decltype(idx_c)::value;
        constexpr int i = idx & 1;
        if constexpr (idx < 2) {
            char* dst = sbase + (wave * 2 + i) * 1024;
            if constexpr (MAPS) {
                const int k = k_first + t * BK + wave * 8 + i * 4 + g4;
                const int src = (sel4(sa, i) & amask) | (k & ~amask);
                const bool ok = (src >= 0) & (k < Kd) & (((keep >> rs_s[i]) & 1ull) != 0ull);
                tnp_dma16(rs_a, dst, ok ? (unsigned)src * (unsigned)lda2 + a_vo[i] : TNP_OOB);
            } else {
                tnp_dma16(rs_a, dst, a_vo[i]);
                a_vo[i] += a_adv;
            }
        } else {
            char* dst = sbase + TILE_BYTES + (wave * 2 + i) * 1024;
            if constexpr (MAPS) {
                const int k = k_first + t * BK + wave * 8 + i * 4 + g4;
                const int src = (sel4(sb, i) & bmask) | (k & ~bmask);
                const bool ok = (src >= 0) & (k < Kd);
                tnp_dma16(rs_b, dst, ok ? (unsigned)src * (unsigned)ldb2 + b_vo[i] : TNP_OOB);
            } else {
                tnp_dma16(rs_b, dst, b_vo[i]);
                b_vo[i] += b_adv;
            }
        }
    };
    auto issue_end = [&]() {
        if constexpr (MAPS) {
            if (has_rs) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    rs_k[i] += BK;
                    const bool w = rs_k[i] >= rs_div;
                    rs_k[i] -= w ? rs_div : 0;
                    rs_s[i] += w ? 1 : 0;
                }
            }
        }
    };

    // ---- fragment read addresses (bytes inside a stage): K row 8 (lane / 16) + (lane % 16) / 4 of the k-step, swizzled 32-byte slot of the
    //      fragment's 16 columns, 8-byte half by the lane's low bits (gemm_tn_v2.hip's relA / relB) ---------------------------------------
    const unsigned lds0 = lds_addr(smem);
    unsigned a_tr[4], b_tr[2];
    {
        const int row_off = 8 * g4 + (l15 >> 2), sw = tn_swz<CH>(row_off);
#pragma unroll
        for (int f = 0; f < 4; ++f)
            a_tr[f] = lds0 + (unsigned)(row_off * BT + ((((wi * 64) / 8 + 2 * f) ^ sw) + ((lane & 3) >> 1)) * 8 + (lane & 1) * 4) * 2u;
#pragma unroll
        for (int f = 0; f < 2; ++f)
            b_tr[f] = lds0 + (unsigned)TILE_BYTES + (unsigned)(row_off * BT + ((((wj * 32) / 8 + 2 * f) ^ sw) + ((lane & 3) >> 1)) * 8 + (lane & 1) * 4) * 2u;
    }

    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool cs_wave = (m.colsum != nullptr) && tile_j == 0 && wj == 0;
    f32x4 cacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

    u64 fal[2][4], fah[2][4], fbl[2][2], fbh[2][2];
    if constexpr (TNP_ABL & 2) {
#pragma unroll
        for (int b_ = 0; b_ < 2; ++b_) {
#pragma unroll
            for (int f_ = 0; f_ < 4; ++f_) { fal[b_][f_] = 0x3f803f803f803f80ull; fah[b_][f_] = 0x3f803f803f803f80ull; }
#pragma unroll
            for (int f_ = 0; f_ < 2; ++f_) { fbl[b_][f_] = 0x3f803f803f803f80ull; fbh[b_][f_] = 0x3f803f803f803f80ull; }
        }
    }
    // The 12 transposing reads of a k-step, as a list r = 0 .. 11 in the order their data is needed: B fragments first (every MFMA takes one), then the A
    // fragments in MFMA order.  Read r of (k-step KS, stage offset in aa_ / bb_) into register buffer `dst`.
    unsigned aa_[4], bb_[2];
    auto rd_addr = [&](unsigned soff) {
#pragma unroll
        for (int f = 0; f < 4; ++f) aa_[f] = a_tr[f] + soff;
#pragma unroll
        for (int f = 0; f < 2; ++f) bb_[f] = b_tr[f] + soff;
    };
    auto rd = [&](auto r_c, auto ks_c, auto dst_c) {
        constexpr int r = decltype(r_c)::value, KS = decltype(ks_c)::value, dst = decltype(dst_c)::value;
        if constexpr (!(TNP_ABL & 2)) {
            if constexpr (r < 4) tnp_rd1<KS * KOFF + (r & 1) * HO>(bb_[r >> 1], (r & 1) ? fbh[dst][r >> 1] : fbl[dst][r >> 1]);
            else tnp_rd1<KS * KOFF + (r & 1) * HO>(aa_[(r - 4) >> 1], (r & 1) ? fah[dst][(r - 4) >> 1] : fal[dst][(r - 4) >> 1]);
        }
    };
    // MFMA mi of a k-step (row-major over (A fragment, B fragment)): C^T fragment -- B as the first operand
#define TNP_MFMA_ONE(buf, mi_)                                                                                        \
    do {                                                                                                              \
        constexpr int fi_ = (mi_) / 2, fj_ = (mi_) % 2;                                                               \
        if constexpr (!(TNP_ABL & 1)) acc[fi_][fj_] = mfma16<bf16>(frag_from(fbl[buf][fj_], fbh[buf][fj_]), frag_from(fal[buf][fi_], fah[buf][fi_]), acc[fi_][fj_]); \
    } while (0)
#define TNP_COLSUM(buf)                                                                                               \
    do {                                                                                                              \
        if (cs_wave) {                                                                                                \
            _Pragma("unroll") for (int fi_ = 0; fi_ < 4; ++fi_)                                                       \
                cacc[fi_] = mfma16<bf16>(ones, frag_from(fal[buf][fi_], fah[buf][fi_]), cacc[fi_]);                   \
        }                                                                                                             \
    } while (0)
    // One MFMA group = the 8 MFMAs of a k-step (buffer CUR) with the 12 reads of the NEXT k-step (into buffer 1 - CUR) spread between them, three per
    // MFMA pair: a burst of 12 reads in front of a group (the first form of this kernel) kept the matrix pipe of every wave idle while the LDS unit
    // worked through 96 instructions -- MFMA + reads alone took 24.7 us of the 38 us launch (profiles/r05_a_tnp_ablate_burst_reads.txt).
    // Counted waits: reads return in order, so at the start of a group all but the last three of the CURRENT k-step's reads (A fragments 2.h, 3.l, 3.h)
    // have landed (lgkmcnt(3)) -- enough for MFMAs 0 .. 3 --, and after the first six new reads have been issued lgkmcnt(6) covers the rest.
    // BARRIER (second group of a K tile): every read of the tile must have completed before its stage is refilled: lgkmcnt(0), then the tile-landed wait
    // and the barrier, and the new reads are those of the next tile's first k-step; the stage is refilled one DMA instruction per MFMA pair.
#define TNP_GROUP(CUR, NKS, nsoff, BARRIER, kt_)                                                                      \
    do {                                                                                                              \
        if constexpr (BARRIER) {                                                                                      \
            tnp_wait<0>();                                                                                            \
            tnp_tie<4>(fal[CUR], fah[CUR]); tnp_tie<2>(fbl[CUR], fbh[CUR]);                                           \
            if constexpr (TNP_ABL & 4) wait_vmcnt<0>(); else wait_vmcnt<(STAGES - 2) * L>();                          \
            __builtin_amdgcn_s_barrier();                                                                             \
        } else {                                                                                                      \
            tnp_wait<3>();                                                                                            \
            tnp_tie<2>(fbl[CUR], fbh[CUR]); tnp_tie1(fal[CUR][0], fah[CUR][0]); tnp_tie1(fal[CUR][1], fah[CUR][1]);   \
        }                                                                                                             \
        rd_addr(nsoff);                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        tnp_static_for<4>([&](auto c_) {                                                                              \
            constexpr int ci_ = decltype(c_)::value;                                                                  \
            if constexpr (!(BARRIER) && ci_ == 2) {                                                                   \
                tnp_wait<6>();                                                                                        \
                tnp_tie1(fal[CUR][2], fah[CUR][2]); tnp_tie1(fal[CUR][3], fah[CUR][3]);                               \
            }                                                                                                         \
            rd(std::integral_constant<int, 3 * ci_>{}, std::integral_constant<int, NKS>{}, std::integral_constant<int, 1 - (CUR)>{});      \
            rd(std::integral_constant<int, 3 * ci_ + 1>{}, std::integral_constant<int, NKS>{}, std::integral_constant<int, 1 - (CUR)>{});  \
            rd(std::integral_constant<int, 3 * ci_ + 2>{}, std::integral_constant<int, NKS>{}, std::integral_constant<int, 1 - (CUR)>{});  \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            TNP_MFMA_ONE(CUR, 2 * ci_);                                                                               \
            TNP_MFMA_ONE(CUR, 2 * ci_ + 1);                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            if constexpr (BARRIER) {                                                                                  \
                if constexpr (!(TNP_ABL & 4)) issue_one(c_, (kt_) + STAGES, smem + ((kt_) % STAGES) * STAGE_BYTES);   \
                __builtin_amdgcn_sched_barrier(0);                                                                    \
            }                                                                                                         \
        });                                                                                                           \
        TNP_COLSUM(CUR);                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    } while (0)

    // ---- prologue: the first STAGES tiles -------------------------------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < STAGES; ++t) {
        map_fetch(t);
        tnp_static_for<L>([&](auto c) { issue_one(c, t, smem + t * STAGE_BYTES); });
        issue_end();
    }
    wait_vmcnt<(STAGES - 1) * L>();                                // tile 0 landed
    __builtin_amdgcn_s_barrier();
    rd_addr(0u);
    tnp_static_for<12>([&](auto r) { rd(r, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); });
    for (int kt = 0; kt < ktiles; ++kt) {
        const unsigned so = (unsigned)((kt % STAGES) * STAGE_BYTES);
        const unsigned sn = (unsigned)(((kt + 1) % STAGES) * STAGE_BYTES);
        map_fetch(kt + STAGES);                                    // (scalar loads: complete at the lgkmcnt(0) in front of the barrier)
        __builtin_amdgcn_sched_barrier(0);
        TNP_GROUP(0, 1, so, false, kt);                            // k-step 0; the reads of k-step 1 between its MFMAs
        TNP_GROUP(1, 0, sn, true, kt);                             // k-step 1; barrier, the next tile's first reads and the refill of this tile's stage
        issue_end();
        __builtin_amdgcn_sched_barrier(0);
    }
    tnp_wait<0>();                                                 // (the reads requested for the tile beyond K)
    wait_vmcnt<0>();                                               // the tiles issued beyond K
#undef TNP_GROUP
#undef TNP_COLSUM
#undef TNP_MFMA_ONE

    // ---- epilogue: acc[fi][fj][r] = C[i0 + wi*64 + fi*16 + lane%16][j0 + wj*32 + fj*16 + 4 (lane/16) + r] ---------------------------------
    const float alpha = m.alpha;
    const bool to_part = m.pieces > 1;
    float* const Cp = to_part ? m.part + (int64_t)piece * m.I * m.J : m.C;
    const int64_t ldc = to_part ? m.J : m.ldc;
    const bool accum = !to_part && (m.flags & TNP_ACCUMULATE), vec4 = to_part ? (m.flags & TNP_PART_VEC4) != 0 : (m.flags & TNP_VEC4) != 0;
    const bool wt = !to_part && (m.flags & TNP_WT) != 0;          // final gradients leave the L2 as they are stored; partial tiles are re-read by the reduction at once
    if (has_b) {
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) {
            const int ii = i0 + wi * 64 + fi * 16 + l15;
#pragma unroll
            for (int fj = 0; fj < 2; ++fj) {
                const int jj = j0 + wj * 32 + fj * 16 + 4 * g4;
                if (ii < m.I && jj < m.J) {                        // (J % 4 == 0: the four columns of a lane are inside J together)
                    float* dst = Cp + (int64_t)ii * ldc + jj;
                    float4 v = make_float4(alpha * acc[fi][fj][0], alpha * acc[fi][fj][1], alpha * acc[fi][fj][2], alpha * acc[fi][fj][3]);
                    if (vec4) {
                        if (accum) { const float4 o = *reinterpret_cast<const float4*>(dst); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                        if (wt) st16_out(dst, make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)));      // write-through (gemm_common.h)
                        else *reinterpret_cast<float4*>(dst) = v;
                    } else {
                        if (accum) { v.x += dst[0]; v.y += dst[1]; v.z += dst[2]; v.w += dst[3]; }
                        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
                    }
                }
            }
        }
    }
    if (cs_wave && g4 == 0) {          // every row n of the ones-product holds the same sums: lanes of the first lane group write column i = lane % 16
        const bool cat = !to_part && ((m.flags & TNP_COLSUM_ATOMIC) || accum);
        float* const csb = to_part ? m.part + (int64_t)m.pieces * m.I * m.J + (int64_t)piece * m.I : m.colsum;
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) {
            const int ii = i0 + wi * 64 + fi * 16 + l15;
            if (ii < m.I) {
                float* cs = csb + ii;
                if (cat) atomicAdd(cs, alpha * cacc[fi][0]); else *cs = alpha * cacc[fi][0];
            }
        }
    }
}

// blockIdx < g.tiles: output tile; beyond: LayerNorm rider units (LPR > 0)
template <int STAGES, int LPR>
__global__ __launch_bounds__(512) void gemm_tn_pipe_kernel(const TnpGroup g, const TnpRider ln) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = blockIdx.x;
    if (bid >= g.tiles) {
        if constexpr (LPR > 0) {
            // two 256-thread LayerNorm units per workgroup (ln_bwd_body.h is written for 4 waves); an odd unit count repeats the last unit in the spare
            // half: the same rows, the same values, written twice
            int unit = 2 * (bid - g.tiles) + (int)(threadIdx.x >> 8);
            unit = unit < ln.blocks ? unit : ln.blocks - 1;
            layernorm_bwd_body<bf16, LPR, 1, 4, 0>(ln.dy, ln.x, nullptr, ln.gamma, ln.mean, ln.rstd, ln.dx, nullptr, nullptr, ln.partials, ln.dres, ln.rows, ln.C, nullptr,
                                                   nullptr, unit, ln.blocks, (int)(threadIdx.x & 255), reinterpret_cast<float*>(smem + (threadIdx.x >> 8) * 16384));
        }
        return;
    }
    // (tile = block index: an XCD-contiguous order measured 1 us slower -- 38.3 vs 37.2 us -- the eight L2s then see the members one after the other)
    const int t = (g.dbg & 2) ? xcd_tile_id(bid, g.tiles) : bid;
    int k = 0;
    while (k + 1 < g.n && t >= g.m[k].tile_end) ++k;
    const TnpMember& m = g.m[k];
    const int local = t - (k ? g.m[k - 1].tile_end : 0);
    if ((m.a_map || m.b_map) && (m.K & 7)) tnp_tile<STAGES, 2>(m, local, smem);
    else if (m.a_map || m.b_map || m.a_rs) tnp_tile<STAGES, 1>(m, local, smem);
    else tnp_tile<STAGES, 0>(m, local, smem);
}

// second stage of the K pieces: C[i][j] += sum_s part[s][i][j], colsum[i] += sum_s part[pieces][s][i] (fixed order: run-to-run identical; a member whose bias
// gradient is shared with another member of the launch adds atomically -- two addends into zeros).  blockIdx.y = member; a thread owns 4 consecutive outputs
// (16-byte loads of every piece, four pieces in flight) -- as 64 outputs x 4 piece lanes per workgroup the launch took 8 us for 19 MB of partials.
__global__ __launch_bounds__(256) void tnp_reduce_pieces(const TnpGroup g) {
    const TnpMember& m = g.m[blockIdx.y];
    const int ns = m.pieces;
    if (ns <= 1) return;
    const int64_t W = (int64_t)m.I * m.J, total = W + (m.colsum ? m.I : 0);
    const int64_t e0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e0 >= total) return;
    const bool v4 = (m.flags & TNP_PART_VEC4) && (m.flags & TNP_VEC4) && (m.J % 4 == 0);
    if (v4 && e0 + 3 < W) {
        const float* q = m.part + e0;
        // eight pieces in flight per thread (stage 0 cuts its members into ~20 pieces: four in flight were five serial round trips)
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
        int s = 0;
        for (; s + 7 < ns; s += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(q + (int64_t)(s + u) * W);
#pragma unroll
            for (int u = 0; u < 8; u += 4) {
                a0.x += v[u].x; a0.y += v[u].y; a0.z += v[u].z; a0.w += v[u].w; a1.x += v[u + 1].x; a1.y += v[u + 1].y; a1.z += v[u + 1].z; a1.w += v[u + 1].w;
                a2.x += v[u + 2].x; a2.y += v[u + 2].y; a2.z += v[u + 2].z; a2.w += v[u + 2].w; a3.x += v[u + 3].x; a3.y += v[u + 3].y; a3.z += v[u + 3].z; a3.w += v[u + 3].w;
            }
        }
        if (s < ns) {          // the remaining <= 7 pieces, all requested before the first add (a clamped index re-reads the last piece: its value is not added)
            float4 v[7];
#pragma unroll
            for (int u = 0; u < 7; ++u) v[u] = *reinterpret_cast<const float4*>(q + (int64_t)min(s + u, ns - 1) * W);
#pragma unroll
            for (int u = 0; u < 7; ++u)
                if (s + u < ns) { a0.x += v[u].x; a0.y += v[u].y; a0.z += v[u].z; a0.w += v[u].w; }
        }
        const int64_t i = e0 / m.J;
        float4* c = reinterpret_cast<float4*>(m.C + i * m.ldc + (e0 - i * m.J));
        float4 o = *c;
        o.x += (a0.x + a1.x) + (a2.x + a3.x); o.y += (a0.y + a1.y) + (a2.y + a3.y); o.z += (a0.z + a1.z) + (a2.z + a3.z); o.w += (a0.w + a1.w) + (a2.w + a3.w);
        *c = o;
        return;
    }
    for (int k = 0; k < 4; ++k) {
        const int64_t e = e0 + k;
        if (e >= total) break;
        const float* q = e < W ? m.part + e : m.part + (int64_t)ns * W + (e - W);
        const int64_t st = e < W ? W : m.I;
        float t = 0.f;
        for (int s = 0; s < ns; ++s) t += q[(int64_t)s * st];
        if (e < W) { const int64_t i = e / m.J; m.C[i * m.ldc + (e - i * m.J)] += t; }
        else if (m.flags & TNP_COLSUM_ATOMIC) atomicAdd(m.colsum + (e - W), t);
        else m.colsum[e - W] += t;
    }
}

// the same for members cut into MANY pieces on small outputs (stage 0: ~20 pieces of 16-64 K outputs): 64 outputs x 4 piece lanes per workgroup -- four
// times the workgroups and the piece loop four ways parallel (the 4-outputs-per-thread form above took 53 instead of 44 us for the stage-0 group)
__global__ __launch_bounds__(256) void tnp_reduce_pieces_deep(const TnpGroup g) {
    const TnpMember& m = g.m[blockIdx.y];
    const int ns = m.pieces;
    if (ns <= 1) return;
    __shared__ float red[4][64];
    const int64_t W = (int64_t)m.I * m.J, total = W + (m.colsum ? m.I : 0);
    if ((int64_t)blockIdx.x * 64 >= total) return;
    const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + col;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < total) {
        const float* q = e < W ? m.part + e : m.part + (int64_t)ns * W + (e - W);
        const int64_t st = e < W ? W : m.I;
        int s = sl;
        for (; s + 12 < ns; s += 16) { a0 += q[(int64_t)s * st]; a1 += q[(int64_t)(s + 4) * st]; a2 += q[(int64_t)(s + 8) * st]; a3 += q[(int64_t)(s + 12) * st]; }
        for (; s < ns; s += 4) a0 += q[(int64_t)s * st];
    }
    red[sl][col] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0 && e < total) {
        const float t = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
        if (e < W) { const int64_t i = e / m.J; m.C[i * m.ldc + (e - i * m.J)] += t; }
        else if (m.flags & TNP_COLSUM_ATOMIC) atomicAdd(m.colsum + (e - W), t);
        else m.colsum[e - W] += t;
    }
}

template <int STAGES, int LPR> void tnp_launch(const TnpGroup& g, const TnpRider& r, int rider_wgs, hipStream_t st) {
    constexpr size_t lds = (size_t)STAGES * 2 * 64 * 128 * 2;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_pipe_kernel<STAGES, LPR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((gemm_tn_pipe_kernel<STAGES, LPR>), dim3(g.tiles + rider_wgs), dim3(512), lds, st, g, r);
}

}  // namespace

struct lavt_ln_rider_t { const void* dy; const void* x; const float* gamma; const float* mean; const float* rstd; void* dx; float* partials; const void* dres; int rows, C; };
int lavt_ln_bwd_geometry(int dtype, int rows, int C, int* lpr, int* cpl, int* waves);

// The grouped weight-gradient launch on 128x128 pipelined tiles.  Returns 1 when the group does not qualify (the caller takes gemm_tn_v2.hip's
// launch), LAVT_OK when it was launched (with the LayerNorm rider if `ln` was given), 3 when it was launched WITHOUT the rider it was offered.
int lavt_gemm_tn_grouped_pipe(const lavt_gemm_tn_t* probs, int n, hipStream_t st, const lavt_ln_rider_t* ln) {
    const lavt_tuning_t& tun = lavt_tuning();
    if (tun.tn_pipe == 0 || n < 2 || n > TNP_MAX) return 1;
    TnpGroup g;
    int base_tiles = 0;
    long work = 0;                                                // (output tile, K tile) pairs of the launch
    for (int i = 0; i < n; ++i) {
        const lavt_gemm_tn_t& p = probs[i];
        const bool no_b = p.ldb == 0;                             // (the column-sum-only side member: B = the zero page)
        if (p.dtype != LAVT_BF16 || p.batch != 1 || p.conv_kc > 0 || p.B2 || p.c_conv_permute) return 1;
        if (p.I % 8 || p.J % 4 || p.lda % 8 || p.ldb % 8 || p.K < 8) return 1;
        if ((int64_t)(p.K + 1024) * p.lda * 2 >= (1LL << 30) || (int64_t)(p.K + 1024) * p.ldb * 2 >= (1LL << 30)) return 1;          // 32-bit byte offsets inside a 2 GB descriptor, with room for the tiles issued beyond K
        // a MAPPED operand is addressed by source row: the rows its map may name must stay inside the descriptor as well (unknown extent: the K + 1024 rows
        // bounded above; a padded-window source or a large clip can be several times the reduction length -- beyond 2^31 bytes the hardware returns zeros)
        if (p.a_rowmap && p.a_src_rows > 0 && (p.a_src_rows + 1) * p.lda * 2 + 256 >= (1LL << 31)) return 1;
        if (p.b_rowmap && p.b_src_rows > 0 && (p.b_src_rows + 1) * p.ldb * 2 + 256 >= (1LL << 31)) return 1;
        if (p.a_rowscale && (!p.a_rowscale_binary || p.a_rowscale_div < 64 || (p.K + p.a_rowscale_div - 1) / p.a_rowscale_div > 64)) return 1;
        if (no_b && !p.colsum) return 1;
        if (!no_b && p.J % 8) return 1;
        TnpMember& m = g.m[i];
        m.A = (const bf16*)p.A; m.B = (const bf16*)p.B; m.C = p.C; m.colsum = p.colsum;
        m.a_map = p.a_rowmap; m.b_map = p.b_rowmap; m.a_rs = p.a_rowscale;
        m.ldc = p.ldc; m.lda = (int)p.lda; m.ldb = (int)p.ldb; m.I = p.I; m.J = no_b ? 8 : p.J; m.K = p.K;
        m.tiles_j = no_b ? 1 : cdiv(p.J, 128);
        m.tiles = cdiv(p.I, 128) * m.tiles_j;
        m.rs_div = p.a_rowscale ? p.a_rowscale_div : 1 << 30;
        m.rs_n = p.a_rowscale ? (p.K + p.a_rowscale_div - 1) / p.a_rowscale_div : 0;
        m.flags = (p.accumulate ? TNP_ACCUMULATE : 0) | (p.colsum_atomic ? TNP_COLSUM_ATOMIC : 0) | (no_b ? TNP_NO_B : 0) |
                  ((((uintptr_t)p.C & 15) == 0 && p.ldc % 4 == 0) ? TNP_VEC4 : 0) | (tun.probe[5] == 1 ? 0 : TNP_WT);
        m.alpha = p.alpha;
        m.part = nullptr; m.pieces = 1; m.kt_per = cdiv(p.K, 64);
        base_tiles += m.tiles;
        work += (long)m.tiles * cdiv(p.K, 64);
    }
    // K pieces.  A group whose 128x128 tiles fill the chip (>= tn_pipe_min_tiles: the stage-2 / stage-3 Swin-block groups) runs uncut: one writer per output,
    // chains of <= 128 K tiles.  Otherwise (long reductions on few tiles: stage 0 / 1, PWAM's 1x1 convolutions) every member is cut into pieces of about
    // work / 248 K tiles -- one round of workgroups of equal length -- through its partials scratch; a member without one keeps the group on the 64x64 launch.
    int tiles = 0;
    bool any_pieces = false, drop_rider = false;
    int64_t max_total = 0;
    if (base_tiles >= tun.tn_pipe_min_tiles) {
        // (short reductions stay on the 64x64 launch: at 8 K tiles -- the last stage's 450 tokens, 792 tiles -- a workgroup that holds a CU alone spends more
        // on its ring prologue and its 64 KB epilogue than on its K loop, and nothing else runs on the CU meanwhile: 54 us against 49)
        if (work < (long)base_tiles * tun.tn_pipe_min_ktiles) return 1;
        for (int i = 0; i < n; ++i) {
            if (cdiv(probs[i].K, 64) > 128) return 1;
            tiles += g.m[i].tiles;
            g.m[i].tile_end = tiles;
        }
    } else {
        if (tun.tn_pipe < 2) return 1;                            // LAVT_TN_PIPE=1: the uncut groups only
        // ONE round of workgroups (a workgroup holds a CU: 128 KB of LDS), of equal length: the smallest piece length whose workgroup count stays inside the
        // chip -- and leaves a fifth of it to the LayerNorm riders when the launch carries them (they would otherwise queue behind the tiles: +13 us at stage 0)
        auto piece_len = [&](int budget) {
            int l = (int)((work + budget - 1) / budget);
            if (l < 8) l = 8;
            for (;; ++l) {
                long wgs = 0;
                for (int i = 0; i < n; ++i) wgs += (long)g.m[i].tiles * cdiv(cdiv(probs[i].K, 64), l);
                if (wgs <= budget || l >= 128) break;
            }
            return l;
        };
        int len = piece_len(ln ? 208 : 250);
        if (ln != nullptr) {
            // ... unless giving the riders their fifth costs more than their own launch: with the measured 0.8 us per K tile, a LayerNorm launch of
            // ~4 us + its bytes at ~3 TB/s (Swin-T's stage 2 at 8 images: 117 tiles of 113 K tiles stay uncut under the rider budget -- 96 us -- and take
            // 57 K tiles + the reduction without riders: 63 + 16 us).  The caller then launches the LayerNorm itself (return code 3).
            const int len_free = piece_len(250);
            const double ln_us = 4.0 + 6.0 * (double)ln->rows * ln->C / 3.0e6;
            if (0.8 * len > 0.8 * len_free + ln_us + 1.0) { len = len_free; drop_rider = true; }
        }
        for (int i = 0; i < n; ++i) {
            const lavt_gemm_tn_t& p = probs[i];
            TnpMember& m = g.m[i];
            const int kt = cdiv(p.K, 64);
            int ns = cdiv(kt, len);
            const int per = cdiv(kt, ns);
            ns = cdiv(kt, per);                                   // no empty pieces
            if (per > 128) return 1;
            if (ns > 1) {
                const int64_t need = (int64_t)ns * ((int64_t)p.I * m.J + p.I);
                if (p.partials == nullptr || p.partials_floats < need) return 1;
                m.part = p.partials; m.pieces = ns; m.kt_per = per;
                if ((((uintptr_t)p.partials & 15) == 0) && ((int64_t)p.I * m.J) % 4 == 0 && m.J % 4 == 0) m.flags |= TNP_PART_VEC4;
                any_pieces = true;
                const int64_t tot = (int64_t)p.I * m.J + p.I;
                max_total = max_total > tot ? max_total : tot;
            }
            tiles += m.tiles * ns;
            m.tile_end = tiles;
        }
        if (tiles < 96 || tiles > 256) return 1;
    }
    for (int i = n; i < TNP_MAX; ++i) { g.m[i] = g.m[0]; g.m[i].tile_end = tiles; g.m[i].pieces = 1; }
    g.n = n; g.tiles = tiles; g.dbg = tun.probe[6];
    TnpRider r{};
    int rider_wgs = 0, lpr = 0;
    if (ln != nullptr && !drop_rider) {
        int cpl, waves;
        const int blocks = lavt_ln_bwd_geometry(LAVT_BF16, ln->rows, ln->C, &lpr, &cpl, &waves);
        if (waves == 4 && cpl == 1 && (lpr == 16 || lpr == 32 || lpr == 64) && blocks > 0) {
            r = TnpRider{(const bf16*)ln->dy, (const bf16*)ln->x, ln->gamma, ln->mean, ln->rstd, (bf16*)ln->dx, ln->partials, (const bf16*)ln->dres, ln->rows, ln->C, blocks};
            rider_wgs = (blocks + 1) / 2;
        } else lpr = 0;
    }
    const int stages = tun.tn_pipe_stages == 3 ? 3 : tun.tn_pipe_stages == 5 ? 5 : tun.tn_pipe_stages == 2 ? 2 : 4;
#define TNP_GO(S_)                                                                  \
    do {                                                                            \
        if (lpr == 64) tnp_launch<S_, 64>(g, r, rider_wgs, st);                     \
        else if (lpr == 32) tnp_launch<S_, 32>(g, r, rider_wgs, st);                \
        else if (lpr == 16) tnp_launch<S_, 16>(g, r, rider_wgs, st);                \
        else tnp_launch<S_, 0>(g, r, 0, st);                                        \
    } while (0)
    if (stages == 3) TNP_GO(3); else if (stages == 5) TNP_GO(5); else if (stages == 2) TNP_GO(2); else TNP_GO(4);
#undef TNP_GO
    if (any_pieces) {
        int max_ns = 1;
        for (int i = 0; i < n; ++i) max_ns = max_ns > g.m[i].pieces ? max_ns : g.m[i].pieces;
        if (max_ns > 8) hipLaunchKernelGGL(tnp_reduce_pieces_deep, dim3((unsigned)cdiv(max_total, 64), n), dim3(256), 0, st, g);
        else hipLaunchKernelGGL(tnp_reduce_pieces, dim3((unsigned)cdiv(max_total, 1024), n), dim3(256), 0, st, g);
    }
    LAVT_CHECK_LAUNCH("lavt_gemm_tn_grouped(pipe)");
    return (ln != nullptr && lpr == 0) ? 3 : LAVT_OK;
}
