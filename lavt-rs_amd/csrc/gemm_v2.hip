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
#include "gemm_v2_helpers.h"

namespace {
// workgroups of the 128x128 tile from which the 2-stage ring (two workgroups per CU) replaces the 4-stage one (LAVT_PROBE slot 7 >= 100 overrides: experiments)
static inline long s2_min128() { const int v = lavt_tuning().probe[7]; return v >= 100 ? v : 257; }


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
    constexpr int WAVES_M = 2;                               // wave grid: WAVES_M (M) x WAVES_N (N)
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

    // side inputs of the epilogue, requested before the first tile (gemm_common.h, NtSide): where their registers are free -- the 64x64 / 4-wave tiles and
    // the activation-gradient kernels (8-wave 128x128 tiles with residual only sit at 116 of the 128 registers that let two workgroups share a CU)
    // (the general activation-gradient kernel on 8-wave tiles -- three side inputs of 8 fragments: 48 registers on top of 110 -- is left out: hipcc kept the
    // kernel at 128 registers and moved the struct to scratch, 208 bytes per lane; the stored-derivative form, GD, carries one input and fits)
    constexpr bool SIDE_PRE = std::is_same<T, bf16>::value && NI % 2 == 0 && MI * NI <= 8 && !LNA && !F8 && ((BM == 64 && WAVES == 4) || (DACT && GD)) && LEAN < 2;
    NtSide<(MI * NI <= 8 ? MI : 1), NI> side;
    side.have = false;
    if constexpr (SIDE_PRE) {
        if ((p.epi_wide & 2) && nt_takes_wide<T, MI, NI>(p, n0 + wn * WN) && (DACT || p.R || p.mul) && (!p.epi_lds || p.mul || (p.ldc % 8)))
            nt_side_load<MI, NI, DACT, GD, LEAN>(p, side, m0 + wm * WM, n0 + wn * WN, lane);
    }
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
    if constexpr (DACT) { nt_epilogue<T, MI, NI, true, GD>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, bz, SIDE_PRE ? &side : nullptr); return; }
    if (!p.epi_lds || p.mul || p.c_f32 || (p.ldc % 8) || (p.C2 && (p.ldc2 % 8 || p.c_split % 8)) || (p.R && p.ldr % 8) || (p.Cpre && p.ldcpre % 8))
        nt_epilogue<T, MI, NI, false, GD, LEAN>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, bz, SIDE_PRE ? &side : nullptr);           // (GD = LAVT_ACT_GELU_D: its own instantiation of the LayerNorm-folded launch)
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
    if (p.act == LAVT_ACT_GELU_D && !p.mul && !p.C2 && !p.R && !p.row_scale && !p.c_rowmap) {
        if (tiles128 >= 200 && p.N >= 128) return tiles128 >= s2_min128() ? launch_nt_v2_<128, 128, false, 2, 8, 1, false, false, true, true>(p, st) : launch_nt_v2_<128, 128, false, 4, 8, 1, false, false, true, true>(p, st);
        return tiles64 >= 512 ? launch_nt_v2_<64, 64, false, 2, 4, 1, false, false, true, true>(p, st) : launch_nt_v2_<64, 64, false, 4, 4, 1, false, false, true, true>(p, st);
    }
    if (tiles128 >= 200 && p.N >= 128) return tiles128 >= s2_min128() ? launch_nt_v2_<128, 128, false, 2, 8, 1, false, false, true>(p, st) : launch_nt_v2_<128, 128, false, 4, 8, 1, false, false, true>(p, st);
    return tiles64 >= 512 ? launch_nt_v2_<64, 64, false, 2, 4, 1, false, false, true>(p, st) : launch_nt_v2_<64, 64, false, 4, 4, 1, false, false, true>(p, st);
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
    if (tiles128 >= 200 && p.N >= 128) return tiles128 >= s2_min128() ? launch_nt_v2_f8<128, 128, 2, 8>(p, st) : launch_nt_v2_f8<128, 128, 4, 8>(p, st);
    return tiles64 >= 512 ? launch_nt_v2_f8<64, 64, 2, 4>(p, st) : launch_nt_v2_f8<64, 64, 4, 4>(p, st);
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
    if (p.dact == LAVT_ACT_STORED && !p.R && !p.bias && !p.c_rowmap && !p.act && !p.mul && !p.Cpre && !p.C2) {      // (its wide epilogue has none of these: any of them takes the general instantiation)
        if (tiles128 >= 200 && p.N >= 128) return tiles128 >= s2_min128() ? launch_nt_v2_<128, 128, true, 2, 8, 1, true, false, false, true>(p, st) : launch_nt_v2_<128, 128, true, 4, 8, 1, true, false, false, true>(p, st);
        return tiles64 >= 512 ? launch_nt_v2_<64, 64, true, 2, 4, 1, true, false, false, true>(p, st) : launch_nt_v2_<64, 64, true, 4, 4, 1, true, false, false, true>(p, st);
    }
    if (tiles128 >= 200 && p.N >= 128) return tiles128 >= s2_min128() ? launch_nt_v2_<128, 128, true, 2, 8, 1, true>(p, st) : launch_nt_v2_<128, 128, true, 4, 8, 1, true>(p, st);
    return tiles64 >= 512 ? launch_nt_v2_<64, 64, true, 2, 4, 1, true>(p, st) : launch_nt_v2_<64, 64, true, 4, 4, 1, true>(p, st);
}
template <int BM, int BN, bool BKM, int STAGES, int WAVES> int launch_nt_v2(const lavt_gemm_nt_t& p, hipStream_t st) {
    const bool general_only = lavt_tuning().gemm_general;
    const bool simple = p.conv_kc <= 0 && p.A2 == nullptr && p.K % 64 == 0;
    const bool convfast = p.conv_kc > 0 && p.conv_kc % 64 == 0 && (p.A2 == nullptr || p.a_split % 64 == 0) && conv_taps_of(p) <= 32;     // (a bit per tap)
    if (simple && !general_only) return launch_nt_v2_<BM, BN, BKM, STAGES, WAVES, 1>(p, st);
    if (convfast && !general_only) return launch_nt_v2_<BM, BN, BKM, STAGES, WAVES, 2>(p, st);
    return launch_nt_v2_<BM, BN, BKM, STAGES, WAVES, 0>(p, st);
}


}  // namespace

// returns 1 when the problem is not for this kernel (caller falls back to gemm.hip), else a LAVT status
int lavt_gemm_nt_v2(const lavt_gemm_nt_t& p, hipStream_t st) {
    if (p.dtype == LAVT_FP8) return launch_nt_v2_fp8(p, st);
    if (p.dtype != LAVT_BF16 || p.zeros == nullptr) return 1;
    const lavt_tuning_t& tun = lavt_tuning();
    if (tun.gemm_v2_off) return 1;
    if (p.lda % 8 || p.ldb % 8 || (p.A2 && p.lda2 % 8)) return 1;
    if (p.ln_wsum) return launch_nt_v2_lna(p, st);
    if (p.dact_pre) return launch_nt_v2_dact(p, st);
    // Dispatch measured on MI355X (tools/gemm_bench.py, hipGraph-timed): 128x128 tile with 8 waves (2 per SIMD: one wave's DMA issue and
    // LDS reads hide under the other's MFMAs) and a 2-stage ring (64-80 KiB -> 2 workgroups per CU) once there are >= 200 such tiles;
    // otherwise 64x64 tiles / 4 waves (5 workgroups per CU), 3 stages only for long-K problems with few tiles.
    const int force = tun.gemm_tile;
    const long tiles128 = (long)cdiv(p.M, 128) * cdiv(p.N, 128) * p.batch;
    const long tiles64 = (long)cdiv(p.M, 64) * cdiv(p.N, 64) * p.batch;
    // long reductions on few tiles (3-D convolutions of SepTPWAM: K = 27 C on 144 tiles of 128x128) also take the 128x128 tile: a launch lasts as
    // long as its serial chain of K tiles, and the larger tile moves half the bytes per K tile and flop
    const int big_long = tun.gemm_big_long;
    const bool big = force ? force == 128 : ((tiles128 >= 200 || (big_long > 0 && p.K >= 64 * 64 && tiles128 >= big_long)) && p.N >= 128);

    // Ring depth.  In isolation (operands L2-resident) 2 stages win everywhere; inside the training step the operands of the small
    // GEMMs arrive cold from HBM / Infinity Cache and a 4-deep ring is worth 0.8 ms per step.  The many-tile long-K problems (decoder
    // convolutions: every CU holds 2 workgroups and streams from L2) stay at 2 stages, which keeps two workgroups per CU resident.
    // Round 5 (second session): for the 128x128 / 8-wave tile the 4-deep ring is 128 KB -- ONE workgroup per CU -- so a launch of 257-600 such workgroups ran in
    // two or three rounds of single workgroups; from 257 workgroups up (more workgroups than CUs) the 2-deep ring (64 KB, two co-resident workgroups whose
    // barrier stalls overlap) wins: batch 4 11.70 -> 11.39 ms, Video-Swin-B 19.55 -> 18.77, Swin-T batch 8 10.89 -> 10.73, headline 7.61 -> 7.58
    // (profiles/r05_zz_ring_depth_threshold_sweep.txt, ..._rule_ab.txt).  Up to 256 workgroups (the M = 1800 GEMMs of batch 2) the 4-deep ring stays.
    const long wgs = big ? tiles128 : tiles64;
    const int stages = tun.gemm_stages ? tun.gemm_stages : (wgs >= (big ? s2_min128() : 512) ? 2 : 4);
    const int waves = tun.gemm_waves;
#define GO(BM_, BN_, KM_, ST_, WV_) return launch_nt_v2<BM_, BN_, KM_, ST_, WV_>(p, st)
    // 128x256 tile, 8 waves of 64x64: fewer LDS bytes (DMA fill and fragment reads) per MFMA than 128x128; for the long-K, many-tile problems
    const long tiles256 = (long)cdiv(p.M, 128) * cdiv(p.N, 256) * p.batch;
    const bool wide = force ? force == 256 : (tun.gemm_wide && tiles256 >= 256 && p.N % 256 == 0 && p.K >= 1024);
    if (wide) { if (p.b_kmajor) GO(128, 256, true, 2, 8); else GO(128, 256, false, 2, 8); }
    // 256x256 tile (gemm_nt_pipe.hip: 8 waves of 128 x 64), one workgroup per CU: 128 flop per byte of LDS fill -- every CU ingests at ~50 GB/s whatever
    // the tile, so the 128x128 tile is fill-bound at half the rate (measured 1.05-1.1 vs 0.6-0.85 PFLOP/s on the decoder conv shapes).
    // Only when its tiles fill the 256 CUs well (whole rounds at >= 80 %).
    const long tiles256x = (long)cdiv(p.M, 256) * cdiv(p.N, 256) * p.batch;
    const long rounds = (tiles256x + 255) / 256;
    const bool huge = force ? force == 512 : (p.N % 256 == 0 && p.K >= 1024 && tiles256x >= 128 && tiles256x * 10 >= rounds * 256 * 8);
    (void)huge;                                     // (the 256x256 tile and the long-K 128x128 problems are taken by gemm_nt_pipe.hip before this dispatcher: gemm.hip)
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
