// Gather-GEMM, second generation (bf16), TN family: weight gradients (split from gemm_v2.hip so that the two families compile in parallel;
// the design notes at the top of gemm_v2.hip apply).
#include "gemm_v2_helpers.h"
#include "ln_bwd_body.h"

namespace {

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
// where a tile's result goes: C[ii * ldc + col] / colsum[ii] with absolute (ii, col); `atomic` adds (split reductions meeting in C, accumulate),
// else plain stores.  A tile-local scratch slot is expressed through the pointers (slot - i0 * BJ - j0 with ldc = BJ).
struct TnOut {
    float* C;
    int64_t ldc;
    float* colsum;
    bool atomic, colsum_atomic;
};
template <int BI, int BJ, int WAVES, int STAGES, bool MAPS, int CS, bool CONVP = true>
__device__ __forceinline__ void tn_tile_range(const lavt_gemm_tn_t& p, const int tile_linear, const int bz, const int kt_begin, const int kt_end, const TnOut o, char* smem) {
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
    // (the side-input pointers as locals: read through `p` -- a reference into the grouped launch's 1.2 KB argument block, member chosen at run time -- the
    // compiler re-fetched them with four s_load per K tile, each behind its own `s_waitcnt lgkmcnt(0)` in the middle of the LDS traffic)
    const int32_t* const map_a = p.a_rowmap;
    const int32_t* const map_b = p.b_rowmap;
    const void* const map_rs = p.a_rowscale;
    const int rs_div = p.a_rowscale_div;
    auto map_dma = [&](int t) {
        if constexpr (MAPS) {
            int k = (kt_begin + t) * BK + lane;
            k = k < Kdim ? k : Kdim - 1;
            const void* src = Z;
            if (wave == 0 && map_a) src = map_a + k;
            if (wave == 1 && map_rs) src = reinterpret_cast<const unsigned*>(map_rs) + (rs_div > 1 ? fdiv(k, rs_div, inv_rsdiv) : k);
            if (wave == 2 && map_b) src = map_b + k;
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

    float* const C = o.C;
    const int64_t ldc_out = o.ldc;
    float* const colsum_out = o.colsum;
    const bool atomic = o.atomic;
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
            if (atomic || o.colsum_atomic) atomicAdd(cs, csum * p.alpha); else *cs = csum * p.alpha;
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
                if (atomic || o.colsum_atomic) atomicAdd(cs, cacc[i][r] * p.alpha); else *cs = cacc[i][r] * p.alpha;      // one tile_j == 0 workgroup per I tile when the reduction is not split
            }
    }
}

// one K piece of a tile, the pieces of a tile meeting in C through atomics or -- with a partials buffer -- stored as plain tiles partials[piece][I][J]
// (+ [piece][I] column sums) that tn_reduce_pieces adds into C afterwards
template <int BI, int BJ, int WAVES, int STAGES, bool MAPS, int CS, bool CONVP = true>
__device__ __forceinline__ void tn_tile(const lavt_gemm_tn_t& p, const int tile_linear, const int bz, const int split_idx, const int kt_per_split,
                                        const int nsplit, char* smem) {
    const int ktiles = (p.K + 63) / 64;
    const int kt_begin = split_idx * kt_per_split, kt_end = min(ktiles, kt_begin + kt_per_split);
    const bool to_parts = p.partials != nullptr && nsplit > 1;
    float* const pbase = to_parts ? p.partials + (int64_t)bz * nsplit * ((int64_t)p.I * p.J + p.I) : nullptr;      // one [pieces][I][J] + [pieces][I] block per batch entry
    TnOut o;
    o.C = to_parts ? pbase + (int64_t)split_idx * p.I * p.J : p.C + (int64_t)bz * p.strideC;
    o.ldc = to_parts ? p.J : p.ldc;
    o.colsum = to_parts ? pbase + (int64_t)nsplit * p.I * p.J + (int64_t)split_idx * p.I : (p.colsum ? p.colsum + (int64_t)bz * p.strideColsum : nullptr);
    o.atomic = !to_parts && (nsplit > 1 || p.accumulate);
    o.colsum_atomic = p.colsum_atomic && !to_parts;
    tn_tile_range<BI, BJ, WAVES, STAGES, MAPS, CS, CONVP>(p, tile_linear, bz, kt_begin, kt_end, o, smem);
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
// (128x128 / 8 waves: capped at 128 registers -- second __launch_bounds__ value = waves per SIMD -- so that two workgroups share a CU)
template <int BI, int BJ, int WAVES, int STAGES, bool MAPS, int CS>
__device__ __forceinline__ void grouped_tile(const TnGroup& g, const int bid, char* smem) {
    int k = 0;
    while (k + 1 < g.n && bid >= g.tile_end[k]) ++k;
    const int local = bid - (k ? g.tile_end[k - 1] : 0);
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
template <int BI, int BJ, int WAVES, int STAGES, bool MAPS, int CS>
__global__ __launch_bounds__(WAVES * 64, (BI == 128 && WAVES == 8) ? 4 : 1) void gemm_tn_v2_grouped_kernel(const TnGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    grouped_tile<BI, BJ, WAVES, STAGES, MAPS, CS>(g, blockIdx.x, smem);
}
// The same launch carrying a LayerNorm backward as RIDER workgroups (round 4).  norm1's backward of a Swin block needs the qkv data gradient, which is
// complete when the block's grouped weight-gradient launch is enqueued, and the launch needs nothing from it: as 450 extra 256-thread workgroups of
// the launch it costs no launch of its own on the critical chain (7.7 us x 24 per Swin-B step).  The rider's LDS scratch is the (idle) tile ring.
struct LnRider {
    const bf16* dy; const bf16* x; const float* gamma; const float* mean; const float* rstd; bf16* dx; float* partials; const bf16* dres;
    int rows, C, blocks;
};
template <bool MAPS, int CS, int LPR, int CPL>
__global__ __launch_bounds__(256) void gemm_tn_v2_grouped_ln_kernel(const TnGroup g, const int tiles, const LnRider ln) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x >= tiles) {
        layernorm_bwd_body<bf16, LPR, CPL, 4, 0>(ln.dy, ln.x, nullptr, ln.gamma, ln.mean, ln.rstd, ln.dx, nullptr, nullptr, ln.partials, ln.dres, ln.rows, ln.C, nullptr,
                                                 nullptr, (int)blockIdx.x - tiles, ln.blocks, threadIdx.x, reinterpret_cast<float*>(smem));
        return;
    }
    grouped_tile<64, 64, 4, 2, MAPS, CS>(g, blockIdx.x, smem);
}

// ---- stream-K form of the grouped launch (round 4) ------------------------------------------------------------------------------------
// The 64x64-tile launch is bound by what a CU can ingest: 792 workgroups x 29 K tiles x 16 KB = 367 MB of L2 -> LDS fill for the stage-2 Swin block
// (9.7 TB/s over its 38 us; round-3 review).  128x128 tiles halve the bytes per flop but give 192 tiles for 256 CUs, each a serial chain of 29 K
// tiles (measured 40 us).  Here the (member, tile, K tile) iterations form ONE list that `nw` persistent workgroups cut into equal runs: every
// workgroup moves the same number of bytes.  A run that covers a tile's whole K range stores it directly; a run that starts or ends inside a
// tile stores a partial tile into its own scratch slot ([nw][2] slots: a run has at most one partial head and one partial tail) and
// tn_streamk_fixup adds the partials of a split tile in a fixed order (no atomics: run-to-run identical).
struct TnSk {
    int it_end[TN_GROUP_MAX];        // running count of K-tile iterations up to and including member k
    int kt[TN_GROUP_MAX];            // K tiles of member k
    int tiles_end[TN_GROUP_MAX];     // running count of output tiles
    int total, nw;
    float* slots;                    // [nw][2][BI * BJ + BI]
};
template <int BI, int BJ, int WAVES, int STAGES, bool MAPS, int CS>
__global__ __launch_bounds__(WAVES * 64, (BI == 128 && WAVES == 8) ? 4 : 1) void gemm_tn_v2_streamk_kernel(const TnGroup g, const TnSk sk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = blockIdx.x;
    const int lo = (int)((int64_t)w * sk.total / sk.nw), hi = (int)((int64_t)(w + 1) * sk.total / sk.nw);
    int seg = 0;
    for (int it = lo; it < hi; ++seg) {
        int m = 0;
        while (m + 1 < g.n && it >= sk.it_end[m]) ++m;
        const int ktm = sk.kt[m], local = it - (m ? sk.it_end[m - 1] : 0);
        const int tile = local / ktm, kb = local - tile * ktm, ke = min(ktm, kb + (hi - it));
        const lavt_gemm_tn_t& p = g.p[m];
        TnOut o;
        if (kb == 0 && ke == ktm) {
            o.C = p.C; o.ldc = p.ldc; o.colsum = p.colsum; o.atomic = p.accumulate != 0; o.colsum_atomic = p.colsum_atomic != 0;
        } else {
            const int tiles_j = (p.J + BJ - 1) / BJ, i0 = (tile / tiles_j) * BI, j0 = (tile % tiles_j) * BJ;
            float* slot = sk.slots + ((int64_t)w * 2 + (seg ? 1 : 0)) * (BI * BJ + BI);
            o.C = slot - (int64_t)i0 * BJ - j0; o.ldc = BJ; o.colsum = slot + BI * BJ - i0; o.atomic = false; o.colsum_atomic = false;
        }
        if (seg) __syncthreads();                                  // the previous run's last fragment reads are done before the ring is refilled
        bool done = false;
        if constexpr (MAPS) {
            if (!(p.a_rowmap || p.a_rowscale || p.b_rowmap)) { tn_tile_range<BI, BJ, WAVES, STAGES, false, CS, false>(p, tile, 0, kb, ke, o, smem); done = true; }
        }
        if (!done) tn_tile_range<BI, BJ, WAVES, STAGES, MAPS, CS, false>(p, tile, 0, kb, ke, o, smem);
        it += ke - kb;
    }
}
// C tile (+)= the partial tiles of its runs, in run order.  blockIdx.x = output tile over all members, blockIdx.y = quarter of the tile.
template <int BI, int BJ>
__global__ __launch_bounds__(256) void tn_streamk_fixup(const TnGroup g, const TnSk sk) {
    int m = 0;
    while (m + 1 < g.n && (int)blockIdx.x >= sk.tiles_end[m]) ++m;
    const lavt_gemm_tn_t& p = g.p[m];
    const int tile = blockIdx.x - (m ? sk.tiles_end[m - 1] : 0), ktm = sk.kt[m];
    const int it0 = (m ? sk.it_end[m - 1] : 0) + tile * ktm, it1 = it0 + ktm;
    auto lo_of = [&](int w) { return (int)((int64_t)w * sk.total / sk.nw); };
    int w0 = (int)((int64_t)it0 * sk.nw / sk.total);
    while (w0 + 1 < sk.nw && lo_of(w0 + 1) <= it0) ++w0;
    while (w0 > 0 && lo_of(w0) > it0) --w0;
    int w1 = w0;
    while (w1 + 1 < sk.nw && lo_of(w1 + 1) < it1) ++w1;
    if (w0 == w1) return;                                          // one run covers the tile's whole K range: stored directly
    const int tiles_j = (p.J + BJ - 1) / BJ, i0 = (tile / tiles_j) * BI, j0 = (tile % tiles_j) * BJ;
    constexpr int SLOT = BI * BJ + BI, QUARTER = BI * BJ / 4;
    for (int e = blockIdx.y * QUARTER + threadIdx.x; e < (blockIdx.y + 1) * QUARTER; e += 256) {
        const int i = e / BJ, j = e - i * BJ;
        if (i0 + i >= p.I || j0 + j >= p.J) continue;
        float s = 0.f;
        for (int w = w0; w <= w1; ++w) {
            const int l = lo_of(w);
            s += sk.slots[((int64_t)w * 2 + ((l >= it0 && l < it1) ? 0 : 1)) * SLOT + e];
        }
        float* dst = p.C + (int64_t)(i0 + i) * p.ldc + j0 + j;
        if (p.accumulate) *dst += s; else *dst = s;
    }
    if (blockIdx.y == 0 && p.colsum != nullptr && j0 == 0 && threadIdx.x < BI && i0 + threadIdx.x < p.I) {
        float s = 0.f;
        for (int w = w0; w <= w1; ++w) {
            const int l = lo_of(w);
            s += sk.slots[((int64_t)w * 2 + ((l >= it0 && l < it1) ? 0 : 1)) * SLOT + BI * BJ + threadIdx.x];
        }
        float* dst = p.colsum + i0 + threadIdx.x;
        if (p.colsum_atomic) atomicAdd(dst, s); else if (p.accumulate) *dst += s; else *dst = s;
    }
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
        else if (p.colsum_atomic) atomicAdd(p.colsum + (e - W), t);      // another member of this launch (another blockIdx.y) adds into the same vector: two addends into zeros, order-independent
        else p.colsum[e - W] += t;
    }
}

}  // namespace

static bool tn_v2_eligible(const lavt_gemm_tn_t& p) {
    if (p.dtype != LAVT_BF16 || p.zeros == nullptr) return false;
    if (p.lda % 8 || p.ldb % 8 || (p.B2 && p.ldb2 % 8)) return false;
    if (p.a_rowscale && !p.a_rowscale_binary) return false;
    if ((p.a_rowmap || p.a_rowscale || p.b_rowmap) && (p.conv_kc > 0 || p.B2)) return false;      // the mapped K loop has no taps / second source
    if ((int64_t)p.K * (p.lda > p.ldb ? p.lda : p.ldb) >= (1LL << 31)) return false;                // 32-bit element offsets
    return true;
}

// returns 1 when the group cannot run as one launch (the caller then issues the problems one by one)
struct lavt_ln_rider_t { const void* dy; const void* x; const float* gamma; const float* mean; const float* rstd; void* dx; float* partials; const void* dres; int rows, C; };
int lavt_ln_bwd_geometry(int dtype, int rows, int C, int* lpr, int* cpl, int* waves);
// ln != NULL: a LayerNorm backward to run as rider workgroups of the launch; returns 3 when the group was launched WITHOUT it (the caller launches it)
int lavt_gemm_tn_grouped_pipe(const lavt_gemm_tn_t* probs, int n, hipStream_t st, const lavt_ln_rider_t* ln);
int lavt_gemm_tn_grouped_v2(const lavt_gemm_tn_t* probs, int n, hipStream_t st, const lavt_ln_rider_t* ln) {
    const lavt_tuning_t& tun = lavt_tuning();
    if (tun.gemm_v2_off || n < 2 || n > TN_GROUP_MAX) return 1;
    {   // short reductions on enough 128x128 tiles (the Swin-block groups of stages 2 / 3): the software-pipelined launch of gemm_tn_pipe.hip
        const int rc = lavt_gemm_tn_grouped_pipe(probs, n, st, ln);
        if (rc != 1) return rc;
    }
    TnGroup g;
    bool maps = false;
    int tiles = 0;
    // Tile configuration of the grouped launch: LAVT_TNG_CFG = "tile,waves,stages" (64,4,2 = the round-1/2 form).
    const int cfg_tile = tun.tng_tile, cfg_waves = tun.tng_waves, cfg_stages = tun.tng_stages;
    int TB = cfg_tile;
    {   // the large tile only where its tiles still occupy most of the chip
        long t128 = 0;
        for (int i = 0; i < n; ++i) t128 += (long)cdiv(probs[i].I, 128) * cdiv(probs[i].J, 128);
        if (TB == 128 && t128 < 128) TB = 64;
    }
    // (rectangular 64 x 128 / 128 x 64 tiles, 4 waves -- 24 KB per K tile for 1.05 MFLOP -- measured level with the square tile in round 4: 39.0 / 41.2 vs
    // 39.0 us per stage-2 launch, 8.66 / 8.74 vs 8.56 ms per step; the instantiations are gone)
    const int TBI = TB, TBJ = TB;
    bool any_colsum = false;
    // Pieces per member.  A member with a partials scratch (lavt_gemm_tn_t.partials: its pieces are stored as plain tiles and added into C by one
    // small second kernel) may be cut as finely as its K allows -- the long-K weight gradients of PWAM (K = 28 800 rows = 450 K tiles on 4
    // output tiles each) then run as ONE launch of ~1000 workgroups instead of four launches of ~230 at one workgroup per CU; a member
    // without one is cut into at most 4 pieces that meet through atomics (only if its C holds zeros: split_k < 0), and a chain of more than
    // 128 K tiles without a scratch keeps the group from forming (it would run serially while the short members supply the tile count).
    const int chain = tun.tng_chain, piece_tiles = tun.tng_piece;
    bool any_parts = false;
    int64_t max_total = 0;
    // ... and only where cutting pays: a group whose uncut 64x64 tiles already give every CU two workgroups (>= 512: the Swin-block launch at
    // 4 images per GPU, K = 3600 = 57 K tiles) runs faster uncut -- 64.9 vs 88.4 us (tools/wgrad_sk_time.py) -- unless a chain exceeds 128 K tiles
    long uncut = 0;
    for (int i = 0; i < n; ++i) uncut += (long)cdiv(probs[i].I, TBI) * cdiv(probs[i].J, TBJ);
    const bool cut_pays = uncut < 512;
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
            const bool parts = p.partials != nullptr && !p.c_conv_permute && ktiles > chain && (cut_pays || ktiles > 128) && want > 1 &&
                               p.partials_floats >= (int64_t)want * ((int64_t)p.I * p.J + p.I);
            if (parts) ns = want;
            else {
                g.p[i].partials = nullptr;
                if (ktiles > 128) return 1;
                // (a group whose uncut tiles already give every CU two workgroups is not cut through atomics either: at 4 images per GPU -- K = 3600 = 57 K tiles --
                // two atomic pieces per tile cost 12.60 vs 12.34 ms per step, and a plainly stored gradient needs no zero fill: engine.TrainStep)
                ns = (p.split_k < 0 && chain > 0 && cut_pays) ? cdiv(ktiles, chain) : 1;
                if (ns > 4) ns = 4;
            }
            const int per = cdiv(ktiles, ns);
            ns = cdiv(ktiles, per);                      // no empty pieces
            if (parts && ns > 1) { any_parts = true; max_total = max_total > (int64_t)p.I * p.J + p.I ? max_total : (int64_t)p.I * p.J + p.I; }
            if (ns <= 1) g.p[i].partials = nullptr;
            g.split[i] = ns;
            tiles += cdiv(p.I, TBI) * cdiv(p.J, TBJ) * ns;
            g.tile_end[i] = tiles;
        }
        if (tiles <= 2048 || !any_parts || per_piece >= 64) break;      // too many workgroups: longer pieces
    }
    for (int i = n; i < TN_GROUP_MAX; ++i) { g.p[i] = probs[0]; g.p[i].partials = nullptr; g.tile_end[i] = tiles; g.split[i] = 1; }
    g.n = n;
    if (tiles < (TB == 64 ? 256 : 128)) return 1;   // too few workgroups to fill the chip
    // (round 2: an XCD-contiguous tile order inside each member -- it cuts the 152 MB of fabric traffic -- and a 3-stage ring were both measured
    // on the step: 10.63 vs 10.64 ms and 10.81 vs 10.62 ms; neither is kept)
    if (TB != 64 || cfg_waves != 4 || cfg_stages != 2) {
#define TNG_GO(BT_, WV_, SG_) TNG_GO2(BT_, BT_, WV_, SG_)
#define TNG_GO2(BT_, BU_, WV_, SG_)                                                                                                                          \
    do {                                                                                                                                               \
        const size_t l = SG_ * (size_t)(64 * (BT_ + BU_) * 2) + (maps ? (2 * (SG_ - 1) + 1) * 768 + 256 : 0);                                          \
        static bool attr = false;                                                                                                                      \
        if (!attr && l > 65536) {                                                                                                                      \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_v2_grouped_kernel<BT_, BU_, WV_, SG_, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l);  \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_v2_grouped_kernel<BT_, BU_, WV_, SG_, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l); \
            attr = true;                                                                                                                               \
        }                                                                                                                                              \
        if (maps) hipLaunchKernelGGL((gemm_tn_v2_grouped_kernel<BT_, BU_, WV_, SG_, true, 2>), dim3(tiles), dim3(WV_ * 64), l, st, g);                 \
        else hipLaunchKernelGGL((gemm_tn_v2_grouped_kernel<BT_, BU_, WV_, SG_, false, 2>), dim3(tiles), dim3(WV_ * 64), l, st, g);                     \
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
#undef TNG_GO2
        if (done) {
            if (any_parts) hipLaunchKernelGGL(tn_reduce_pieces_group, dim3((unsigned)cdiv(max_total, 64), n), dim3(256), 0, st, g);
            LAVT_CHECK_LAUNCH("lavt_gemm_tn_grouped(v2)");
            return ln ? 3 : LAVT_OK;
        }
    }
    const size_t lds = 2 * (size_t)(64 * (64 + 64) * 2) + (maps ? 3 * 768 + 256 : 0);
    if (ln != nullptr && any_colsum && lavt_tuning().probe[2] == 0) {
        int lpr, cpl, waves;
        const int blocks = lavt_ln_bwd_geometry(LAVT_BF16, ln->rows, ln->C, &lpr, &cpl, &waves);
        LnRider r{(const bf16*)ln->dy, (const bf16*)ln->x, ln->gamma, ln->mean, ln->rstd, (bf16*)ln->dx, ln->partials, (const bf16*)ln->dres, ln->rows, ln->C, blocks};
        bool rode = true;
#define TNG_LN(LPR_, CPL_)                                                                                                                            \
    do {                                                                                                                                               \
        if (maps) hipLaunchKernelGGL((gemm_tn_v2_grouped_ln_kernel<true, 2, LPR_, CPL_>), dim3(tiles + blocks), dim3(256), lds, st, g, tiles, r);      \
        else hipLaunchKernelGGL((gemm_tn_v2_grouped_ln_kernel<false, 2, LPR_, CPL_>), dim3(tiles + blocks), dim3(256), lds, st, g, tiles, r);          \
    } while (0)
        if (waves != 4 || (size_t)3 * lpr * cpl * 8 * 4 > lds) rode = false;
        else if (lpr == 16 && cpl == 1) TNG_LN(16, 1);
        else if (lpr == 32 && cpl == 1) TNG_LN(32, 1);
        else if (lpr == 64 && cpl == 1) TNG_LN(64, 1);
        else rode = false;          // (C = 1024: two chunks per lane need 164 registers -- a fourth workgroup per CU no longer fits; the two stage-3 blocks launch it on its own)
#undef TNG_LN
        if (rode) {
            if (any_parts) hipLaunchKernelGGL(tn_reduce_pieces_group, dim3((unsigned)cdiv(max_total, 64), n), dim3(256), 0, st, g);
            LAVT_CHECK_LAUNCH("lavt_gemm_tn_grouped_ln(v2)");
            return LAVT_OK;
        }
    }
    if (any_colsum) {
        if (maps) hipLaunchKernelGGL((gemm_tn_v2_grouped_kernel<64, 64, 4, 2, true, 2>), dim3(tiles), dim3(256), lds, st, g);
        else hipLaunchKernelGGL((gemm_tn_v2_grouped_kernel<64, 64, 4, 2, false, 2>), dim3(tiles), dim3(256), lds, st, g);
    } else {
        if (maps) hipLaunchKernelGGL((gemm_tn_v2_grouped_kernel<64, 64, 4, 2, true, 0>), dim3(tiles), dim3(256), lds, st, g);
        else hipLaunchKernelGGL((gemm_tn_v2_grouped_kernel<64, 64, 4, 2, false, 0>), dim3(tiles), dim3(256), lds, st, g);
    }
    if (any_parts) hipLaunchKernelGGL(tn_reduce_pieces_group, dim3((unsigned)cdiv(max_total, 64), n), dim3(256), 0, st, g);
    LAVT_CHECK_LAUNCH("lavt_gemm_tn_grouped(v2)");
    return ln ? 3 : LAVT_OK;
}

// ---- stream-K grouped launch: host side.  sk_plan fills g / sk and returns the scratch floats needed (0 = the group does not qualify).
static int64_t sk_plan(const lavt_gemm_tn_t* probs, int n, TnGroup& g, TnSk& sk, bool& maps, bool& any_colsum) {
    constexpr int TB = 128;
    const lavt_tuning_t& tun = lavt_tuning();
    if (tun.gemm_v2_off || !tun.tn_streamk || n < 2 || n > TN_GROUP_MAX) return 0;
    maps = false; any_colsum = false;
    int its = 0, tiles = 0;
    for (int i = 0; i < n; ++i) {
        const lavt_gemm_tn_t& p = probs[i];
        if (!tn_v2_eligible(p) || p.batch != 1 || p.conv_kc > 0 || p.B2 || p.I % 8 || p.J % 8 || p.c_conv_permute) return 0;
        const int kt = cdiv(p.K, 64), t = cdiv(p.I, TB) * cdiv(p.J, TB);
        maps = maps || p.a_rowmap || p.a_rowscale || p.b_rowmap;
        any_colsum = any_colsum || p.colsum != nullptr;
        g.p[i] = p;
        g.p[i].partials = nullptr;
        g.split[i] = 1;
        its += t * kt; tiles += t;
        sk.it_end[i] = its; sk.kt[i] = kt; sk.tiles_end[i] = tiles;
        g.tile_end[i] = tiles;
    }
    for (int i = n; i < TN_GROUP_MAX; ++i) { g.p[i] = probs[0]; g.p[i].partials = nullptr; g.tile_end[i] = tiles; g.split[i] = 1; sk.it_end[i] = its; sk.kt[i] = 1; sk.tiles_end[i] = tiles; }
    g.n = n;
    // worth it when the 128x128 tiles alone cannot fill the chip in whole rounds but there is enough work for every run to amortise its ring
    // prologue (>= 6 K tiles per run) -- the Swin-block launches of stages 1-3; many-tile members (stage 0: K = 28 800) keep the piece form
    int nw = tun.probe[0] > 0 ? tun.probe[0] : 512;
    if (its / nw < 6) nw = its / 6;
    if (nw < 128 || tiles < 32) return 0;
    sk.total = its; sk.nw = nw;
    return (int64_t)nw * 2 * (TB * TB + TB);
}
int64_t lavt_gemm_tn_grouped_sk_ws_v2(const lavt_gemm_tn_t* probs, int n) {
    TnGroup g; TnSk sk; bool maps, cs;
    return sk_plan(probs, n, g, sk, maps, cs);
}
int lavt_gemm_tn_grouped_sk_v2(const lavt_gemm_tn_t* probs, int n, float* scratch, int64_t scratch_floats, hipStream_t st) {
    TnGroup g; TnSk sk; bool maps, any_colsum;
    const int64_t need = sk_plan(probs, n, g, sk, maps, any_colsum);
    if (need == 0 || scratch == nullptr || scratch_floats < need) return 1;
    sk.slots = scratch;
    const size_t lds = 2 * (size_t)(64 * (128 + 128) * 2) + (maps ? 3 * 768 + 256 : 0);
#define SK_GO(MAPS_, CS_)                                                                                                                              \
    do {                                                                                                                                               \
        static bool attr = false;                                                                                                                      \
        if (!attr) {                                                                                                                                   \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_v2_streamk_kernel<128, 128, 8, 2, MAPS_, CS_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr = true;                                                                                                                               \
        }                                                                                                                                              \
        hipLaunchKernelGGL((gemm_tn_v2_streamk_kernel<128, 128, 8, 2, MAPS_, CS_>), dim3(sk.nw), dim3(512), lds, st, g, sk);                           \
    } while (0)
    if (any_colsum) { if (maps) SK_GO(true, 1); else SK_GO(false, 1); }          // (scalar column sums: the matrix-core form costs 16 accumulator registers the 128-register tile does not have)
    else { if (maps) SK_GO(true, 0); else SK_GO(false, 0); }
#undef SK_GO
    hipLaunchKernelGGL((tn_streamk_fixup<128, 128>), dim3(sk.tiles_end[n - 1], 4), dim3(256), 0, st, g, sk);
    LAVT_CHECK_LAUNCH("lavt_gemm_tn_grouped_sk");
    return LAVT_OK;
}

int lavt_gemm_tn_v2(const lavt_gemm_tn_t& p, hipStream_t st) {
    if (p.dtype != LAVT_BF16 || p.zeros == nullptr) return 1;
    const lavt_tuning_t& tun = lavt_tuning();
    if (tun.gemm_v2_off) return 1;
    if (!tn_v2_eligible(p)) return 1;                            // (only 0 / constant row masks can be folded into the row fetch)
    // Measured (tools/gemm_bench.py tn): the 64x64 / 4-wave tile wins on every weight-gradient shape of the step, the conv wgrads included
    // (369 vs 230 TF/s for 128x128); the split-K factor trades workgroup count (latency hiding) against fp32 atomic traffic.
    const int force = tun.gemm_tile;
    const int ktiles = cdiv(p.K, 64);
    const long tiles64 = (long)cdiv(p.I, 64) * cdiv(p.J, 64) * p.batch;
    const long tiles128 = (long)cdiv(p.I, 128) * cdiv(p.J, 128) * p.batch;
    // conv weight gradients (long K, >= 48 tiles of 128x128 -- the Swin-T decoder's 384-channel convolutions have 102): the larger tile halves the
    // L2->LDS bytes per MFMA (measured 303 vs 357 us on Swin-B's; 16.9 -> 16.1 ms per Swin-T step)
    const bool tn128 = tun.tn_big;
    const int tn_big_min = tun.tn_big_min;
    const bool big = force ? force == 128 : (tn128 && p.conv_kc > 0 && tiles128 >= tn_big_min && ktiles >= 64);
    const long tiles = big ? tiles128 : tiles64;
    int split = p.split_k;
    if (tun.tn_split) split = tun.tn_split;
    if (split <= 0) {
        const int target = tun.tn_target;
        split = big ? (int)((384 + tiles / 2) / tiles) : (int)((target + tiles / 2) / tiles);      // ~3 (64-tile) / ~1.5 (128-tile) workgroups per CU
        const int long_k = (ktiles + 127) / 128;          // no workgroup walks more than ~128 K tiles
        if (split < long_k) split = long_k;
        const int min_kt = p.batch > 1 ? (tun.probe[4] > 0 ? tun.probe[4] : 2) : 8;          // (as lavt_gemm_tn_pieces: batched problems down to 2 K tiles per piece)
        const int max_split = (ktiles + min_kt - 1) / min_kt;           // >= 8 K tiles per workgroup (2 for batched problems)
        if (split > max_split) split = max_split;
        if (split < 1) split = 1;
    }
    if (split > ktiles) split = ktiles;
    if (big) return launch_tn_v2<128, 128, 8>(p, split, st);
    return launch_tn_v2<64, 64, 4>(p, split, st);
}
