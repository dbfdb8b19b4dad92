// Gather-GEMM kernels for gfx950 (MI355X): every Linear / 1x1 conv / 3x3 conv of the LAVT hot path,
// forward, data gradient (NT family) and weight gradient (TN family).  See include/lavt_hip.h.
//
// Structure (both families): 256-thread workgroup = 4 waves in a 2x2 arrangement over a BMxBN output tile,
// each wave owning (BM/2)x(BN/2) as 16x16 MFMA accumulators.  Operand tiles are staged
// global -> registers -> LDS with one barrier per K tile (loads for tile t+1 are issued before the MFMAs of
// tile t and written to the other LDS buffer after them), so HBM latency hides under the matrix work.
//   bf16: v_mfma_f32_16x16x32_bf16, K tile 64 (128-byte LDS rows, XOR-swizzled 16-byte chunks -> conflict-free
//         ds_read_b128 fragment reads); operands whose reduction index is the slow memory index ("k-major":
//         x @ W data gradients and all weight gradients) are staged untransposed and read with the hardware
//         transposing LDS read ds_read_b64_tr_b16.
//   fp32: v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain) -- the parity path.
// Row gather on the A side (window partition / shift / zero padding, 3x3 taps with zero halo, concat of two
// sources) and row scatter + residual on the C side are folded into the tile loads / stores: no im2col,
// no permute/roll/pad/cat copies ever touch HBM.
#include <stdlib.h>

#include "gemm_common.h"

using namespace lavt_gemm;

namespace {

// ================================================================================================
//                                           NT family
// ================================================================================================
template <typename T, int BM, int BN, bool BKM>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const lavt_gemm_nt_t p) {
    constexpr int BK = Cfg<T>::BK, EPC = Cfg<T>::EPC, KSTEPS = BK / Cfg<T>::KSTEP;
    constexpr int CPR = BK / EPC;               // 16-byte chunks per k-row
    constexpr int A_RPP = 256 / CPR, A_PASSES = BM / A_RPP;
    constexpr int BKC_PASSES = BN / A_RPP;
    constexpr int BKM_CPR = BN / EPC, BKM_RPP = 256 / BKM_CPR, BKM_PASSES = BK / BKM_RPP;
    constexpr int B_PASSES = BKM ? BKM_PASSES : BKC_PASSES;
    constexpr int A_TILE = BM * Cfg<T>::KC_LD;
    constexpr int B_LD = BN + KM_PAD;
    constexpr int B_TILE = BKM ? BK * B_LD : BN * Cfg<T>::KC_LD;
    constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 16, NI = WN / 16;
    static_assert(A_PASSES >= 1 && B_PASSES >= 1, "tile too small for 256 threads");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    constexpr int STAGE = A_TILE + B_TILE;      // stage s: A at s*STAGE, B at s*STAGE + A_TILE (plain offsets keep the LDS address space)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile_id = xcd_tile_id(blockIdx.x, gridDim.x);
    const int tile_m = tile_id / tiles_n, tile_n = tile_id % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int bz = blockIdx.y;

    const T* A = reinterpret_cast<const T*>(p.A) + (int64_t)bz * p.strideA;
    const T* A2 = reinterpret_cast<const T*>(p.A2);
    const T* B = reinterpret_cast<const T*>(p.B) + (int64_t)bz * p.strideB;
    const bool conv = p.conv_kc > 0;
    const ConvGeom cg = conv_geom(p);

    // ---- per-thread A rows -------------------------------------------------------------------
    const int a_chunk = tid % CPR, a_row0 = tid / CPR;
    int a_src[A_PASSES];          // source row (plain / mapped), -1 = zeros
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
        const int m = m0 + a_row0 + i * A_RPP;
        int src = -1;
        if (m < p.M) src = p.a_rowmap ? p.a_rowmap[m] : m;
        a_src[i] = src;
    }
    const int b_chunk = BKM ? tid % BKM_CPR : a_chunk;
    const int b_row0 = BKM ? tid / BKM_CPR : a_row0;

    uint4 ra[A_PASSES], rb[B_PASSES];
    uint32_t va = 0, vb = 0;       // validity bits of the staged chunks: loads are UNCONDITIONAL (clamped address) so that they
                                   // stay in flight across the MFMAs; invalid chunks are zeroed when written to LDS

    auto load_tiles = [&](int kt) {
        // A
        const int k = kt * BK + a_chunk * EPC;
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
        va = 0;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            int src = a_src[i];
            if (conv) src = conv_nbr(cg, src, dz, dy, dx);
            const bool ok = src >= 0 && k < p.K;
            ra[i] = ldg16(ok ? base + (int64_t)src * ld + kk : A);
            va |= (ok ? 1u : 0u) << i;
        }
        // B
        vb = 0;
        if constexpr (!BKM) {
#pragma unroll
            for (int i = 0; i < B_PASSES; ++i) {
                const int n = n0 + b_row0 + i * A_RPP;
                const bool ok = n < p.N && k < p.K;
                rb[i] = ldg16(ok ? B + (int64_t)n * p.ldb + k : B);
                vb |= (ok ? 1u : 0u) << i;
            }
        } else {
            const int n = n0 + b_chunk * EPC;
#pragma unroll
            for (int i = 0; i < B_PASSES; ++i) {
                const int kb = kt * BK + b_row0 + i * BKM_RPP;
                int64_t off;
                if (conv) { const int t2 = kb / p.conv_kc; off = (int64_t)(kb - t2 * p.conv_kc) * p.ldb + (int64_t)t2 * p.b_tap_stride; }
                else off = (int64_t)kb * p.ldb;
                const bool ok = kb < p.K && n < p.N;
                rb[i] = ldg16(ok ? B + off + n : B);
                vb |= (ok ? 1u : 0u) << i;
            }
        }
    };
    auto store_tiles = [&](int buf) {
        T* dA = smem + buf * STAGE;
        T* dB = dA + A_TILE;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i)
            *reinterpret_cast<uint4*>(dA + kc_off<T>(a_row0 + i * A_RPP, a_chunk)) = ((va >> i) & 1u) ? ra[i] : zero16();
        if constexpr (!BKM) {
#pragma unroll
            for (int i = 0; i < B_PASSES; ++i)
                *reinterpret_cast<uint4*>(dB + kc_off<T>(b_row0 + i * A_RPP, b_chunk)) = ((vb >> i) & 1u) ? rb[i] : zero16();
        } else {
#pragma unroll
            for (int i = 0; i < B_PASSES; ++i)
                *reinterpret_cast<uint4*>(dB + (b_row0 + i * BKM_RPP) * B_LD + b_chunk * EPC) = ((vb >> i) & 1u) ? rb[i] : zero16();
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ktiles = (p.K + BK - 1) / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ktiles) load_tiles(kt + 1);
        __builtin_amdgcn_sched_barrier(0);           // global loads of tile kt+1 are issued before, and consumed after, the MFMAs
        typename FragT<T>::type fa[KSTEPS][MI], fb[KSTEPS][NI];
        const T* cA = smem + cur * STAGE;
        const T* cB = cA + A_TILE;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[ks][i] = frag_kc<T>(cA, wm * WM + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                if constexpr (BKM) fb[ks][j] = frag_km<T>(cB, B_LD, wn * WN + j * 16, ks, lane);
                else fb[ks][j] = frag_kc<T>(cB, wn * WN + j * 16, ks, lane);
            }
        }
        // swapped operands: accumulator rows = n (4 consecutive per lane), columns = m
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = mfma16<T>(fb[ks][j], fa[ks][i], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < ktiles) store_tiles(cur ^ 1);
        __syncthreads();
    }

    if constexpr (std::is_same<T, bf16>::value) {
        if (p.epi_lds && !(p.mul || p.c_f32 || (p.ldc % 8) || (p.C2 && (p.ldc2 % 8 || p.c_split % 8)) || (p.R && p.ldr % 8) || (p.Cpre && p.ldcpre % 8))) {
            nt_epilogue_lds<BM, BN, MI, NI>(p, acc, reinterpret_cast<bf16*>(smem), m0, n0, wm * WM, wn * WN, tid, lane, bz);
            return;
        }
    }
    nt_epilogue<T, MI, NI>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, bz);
}

template <typename T, int BM, int BN, bool BKM> int launch_nt(const lavt_gemm_nt_t& p, hipStream_t st) {
    constexpr int BK = Cfg<T>::BK;
    constexpr int A_TILE = BM * Cfg<T>::KC_LD;
    constexpr int B_TILE = BKM ? BK * (BN + KM_PAD) : BN * Cfg<T>::KC_LD;
    const size_t lds = 2 * (size_t)(A_TILE + B_TILE) * sizeof(T);
    dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), p.batch);
    static bool attr_set = false;      // > 64 KiB of dynamic LDS must be requested once per kernel
    if (!attr_set && lds > 65536) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<T, BM, BN, BKM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            lavt_set_error("lavt_gemm_nt: cannot reserve %zu bytes of LDS", lds);
            return LAVT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_nt_kernel<T, BM, BN, BKM>), grid, dim3(256), lds, st, p);
    LAVT_CHECK_LAUNCH("lavt_gemm_nt");
    return LAVT_OK;
}
// LAVT_GEMM_TILE=128|64 forces a tile configuration (tests exercise both); unset = shape heuristic.
static int forced_tile() {
    return lavt_tuning().gemm_tile;
}
template <typename T> int dispatch_nt(const lavt_gemm_nt_t& p, hipStream_t st) {
    const long tiles128 = (long)cdiv(p.M, 128) * cdiv(p.N, 128) * p.batch;
    const int force = forced_tile();
    // measured on MI355X (tools/gemm_bench.py): 64x64 tiles win on every backbone GEMM of the batch-2 step (they are latency-bound:
    // more workgroups per CU hide it better); 128x128 pays only once there are >= 3 full waves of tiles (decoder convs)
    const bool big = force ? force == 128 : (tiles128 >= 768 && p.N >= 128);
    if (p.b_kmajor) return big ? launch_nt<T, 128, 128, true>(p, st) : launch_nt<T, 64, 64, true>(p, st);
    return big ? launch_nt<T, 128, 128, false>(p, st) : launch_nt<T, 64, 64, false>(p, st);
}

// ================================================================================================
//                                           TN family
// ================================================================================================
template <typename T, int BI, int BJ>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const lavt_gemm_tn_t p, int kt_per_split) {
    constexpr int BK = Cfg<T>::BK, EPC = Cfg<T>::EPC, KSTEPS = BK / Cfg<T>::KSTEP;
    constexpr int A_LD = BI + KM_PAD, B_LD = BJ + KM_PAD;
    constexpr int A_CPR = BI / EPC, A_RPP = 256 / A_CPR, A_PASSES = BK / A_RPP;
    constexpr int B_CPR = BJ / EPC, B_RPP = 256 / B_CPR, B_PASSES = BK / B_RPP;
    constexpr int A_TILE = BK * A_LD, B_TILE = BK * B_LD;
    constexpr int WI = BI / 2, WJ = BJ / 2, II = WI / 16, JJ = WJ / 16;
    static_assert(A_PASSES >= 1 && B_PASSES >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    constexpr int STAGE = A_TILE + B_TILE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int tiles_j = (p.J + BJ - 1) / BJ;
    const int tile_i = blockIdx.x / tiles_j, tile_j = blockIdx.x % tiles_j;
    const int i0 = tile_i * BI, j0 = tile_j * BJ;
    const int bz = blockIdx.y;
    const int ktiles = (p.K + BK - 1) / BK;
    const int kt_begin = blockIdx.z * kt_per_split;
    const int kt_end = min(ktiles, kt_begin + kt_per_split);
    if (kt_begin >= kt_end) return;

    const T* A = reinterpret_cast<const T*>(p.A) + (int64_t)bz * p.strideA;
    const T* B = reinterpret_cast<const T*>(p.B) + (int64_t)bz * p.strideB;
    const T* B2 = reinterpret_cast<const T*>(p.B2);
    const bool conv = p.conv_kc > 0;

    const int a_chunk = tid % A_CPR, a_row0 = tid / A_CPR;
    const int b_chunk = tid % B_CPR, b_row0 = tid / B_CPR;
    const int ia = i0 + a_chunk * EPC;          // first column of this thread's A chunk
    const int jb = j0 + b_chunk * EPC;
    // column-dependent B source (conv tap, concat split) is fixed per thread
    const ConvGeom cg = conv_geom(p);
    int jc = jb, dz = 0, dy = 0, dx = 0;
    if (conv) { const int tap = jb / p.conv_kc; jc = jb - tap * p.conv_kc; conv_tap(cg, tap, dz, dy, dx); }
    const bool b_second = (p.B2 != nullptr) && jc >= p.b_split;
    const T* Bsrc = b_second ? B2 : B;
    const int64_t ldb = b_second ? p.ldb2 : p.ldb;
    const int jcc = b_second ? jc - p.b_split : jc;

    uint4 ra[A_PASSES], rb[B_PASSES];
    uint32_t va = 0, vb = 0;
    float asc[A_PASSES];
    auto load_tiles = [&](int kt) {
        va = 0; vb = 0;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const int k = kt * BK + a_row0 + i * A_RPP;
            const bool in = k < p.K;
            const int kc = in ? k : 0;
            const int src = p.a_rowmap ? p.a_rowmap[kc] : kc;
            const bool ok = in && src >= 0 && ia < p.I;
            ra[i] = ldg16(ok ? A + (int64_t)src * p.lda + ia : A);
            va |= (ok ? 1u : 0u) << i;
            asc[i] = p.a_rowscale ? p.a_rowscale[p.a_rowscale_div > 1 ? kc / p.a_rowscale_div : kc] : 1.f;
            if (p.a_rowscale_binary && asc[i] != 0.f) asc[i] = 1.f;
        }
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int k = kt * BK + b_row0 + i * B_RPP;
            const bool in = k < p.K;
            const int kc = in ? k : 0;
            int src = p.b_rowmap ? p.b_rowmap[kc] : kc;
            if (conv) src = conv_nbr(cg, src, dz, dy, dx);
            const bool ok = in && src >= 0 && jb < p.J;
            rb[i] = ldg16(ok ? Bsrc + (int64_t)src * ldb + jcc : Bsrc);
            vb |= (ok ? 1u : 0u) << i;
        }
    };
    auto store_tiles = [&](int buf) {
        T* dA = smem + buf * STAGE;
        T* dB = dA + A_TILE;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            uint4 v = ((va >> i) & 1u) ? ra[i] : zero16();
            if (p.a_rowscale) {
                float f[EPC];
                chunk_to_f<T>(v, f);
#pragma unroll
                for (int e = 0; e < EPC; ++e) f[e] *= asc[i];
                v = f_to_chunk<T>(f);
            }
            *reinterpret_cast<uint4*>(dA + (a_row0 + i * A_RPP) * A_LD + a_chunk * EPC) = v;
        }
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i)
            *reinterpret_cast<uint4*>(dB + (b_row0 + i * B_RPP) * B_LD + b_chunk * EPC) = ((vb >> i) & 1u) ? rb[i] : zero16();
    };

    f32x4 acc[II][JJ];
#pragma unroll
    for (int i = 0; i < II; ++i)
#pragma unroll
        for (int j = 0; j < JJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float csum = 0.f;
    const bool do_colsum = (p.colsum != nullptr) && tile_j == 0 && tid < BI;

    load_tiles(kt_begin);
    store_tiles(0);
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        if (kt + 1 < kt_end) load_tiles(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        typename FragT<T>::type fa[KSTEPS][II], fb[KSTEPS][JJ];
        const T* cA = smem + cur * STAGE;
        const T* cB = cA + A_TILE;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
            for (int i = 0; i < II; ++i) fa[ks][i] = frag_km<T>(cA, A_LD, wi * WI + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < JJ; ++j) fb[ks][j] = frag_km<T>(cB, B_LD, wj * WJ + j * 16, ks, lane);
        }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int i = 0; i < II; ++i)
#pragma unroll
                for (int j = 0; j < JJ; ++j) acc[i][j] = mfma16<T>(fa[ks][i], fb[ks][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        if (do_colsum) {
            const T* col = cA + tid;
#pragma unroll 8
            for (int k = 0; k < BK; ++k) csum += to_f<T>(col[k * A_LD]);
        }
        if (kt + 1 < kt_end) store_tiles(cur ^ 1);
        __syncthreads();
    }

    float* C = p.C + (int64_t)bz * p.strideC;
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
                if (p.c_conv_permute) { const int t2 = jj / p.conv_kc; col = (int64_t)(jj - t2 * p.conv_kc) * cg.taps + t2; }
                atomicAdd(C + (int64_t)ii * p.ldc + col, p.alpha * acc[i][j][r]);
            }
        }
    }
    if (do_colsum && i0 + tid < p.I) atomicAdd(p.colsum + (int64_t)bz * p.strideColsum + i0 + tid, csum);
}

template <typename T, int BI, int BJ> int launch_tn(const lavt_gemm_tn_t& p, hipStream_t st) {
    constexpr int BK = Cfg<T>::BK;
    const size_t lds = 2 * (size_t)(BK * (BI + KM_PAD) + BK * (BJ + KM_PAD)) * sizeof(T);
    const int tiles = cdiv(p.I, BI) * cdiv(p.J, BJ);
    const int ktiles = cdiv(p.K, BK);
    int split = p.split_k;
    if (split <= 0) {
        split = lavt_tuning().tn_split;
        if (split <= 0) split = (int)(768 / ((long)tiles * p.batch));
        if (split < 1) split = 1;
        const int max_split = (ktiles + 3) / 4;        // at least 4 K tiles per workgroup
        if (split > max_split) split = max_split;
        if (split < 1) split = 1;
    }
    if (split > ktiles) split = ktiles;
    const int per = cdiv(ktiles, split);
    split = cdiv(ktiles, per);
    dim3 grid(tiles, p.batch, split);
    static bool attr_set = false;
    if (!attr_set && lds > 65536) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<T, BI, BJ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            lavt_set_error("lavt_gemm_tn: cannot reserve %zu bytes of LDS", lds);
            return LAVT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_tn_kernel<T, BI, BJ>), grid, dim3(256), lds, st, p, per);
    LAVT_CHECK_LAUNCH("lavt_gemm_tn");
    return LAVT_OK;
}
template <typename T> int dispatch_tn(const lavt_gemm_tn_t& p, hipStream_t st) {
    const int force = forced_tile();
    const bool big = force ? force == 128 : (p.I >= 256 && p.J >= 1024 && p.K >= 8192);
    return big ? launch_tn<T, 128, 128>(p, st) : launch_tn<T, 64, 64>(p, st);
}

}  // namespace

int lavt_gemm_nt_v2(const lavt_gemm_nt_t& p, hipStream_t st);
int lavt_gemm_nt_pipe_tile(const lavt_gemm_nt_t& p, int* stages_out);          // gemm_nt_pipe.hip: software-pipelined K loop
int lavt_gemm_nt_pipe(const lavt_gemm_nt_t& p, int tile, int stages, hipStream_t st);
int lavt_gemm_tn_v2(const lavt_gemm_tn_t& p, hipStream_t st);

extern "C" int lavt_gemm_nt(const lavt_gemm_nt_t* pp, void* stream) {
    LAVT_CHECK_ARG(pp != nullptr, "lavt_gemm_nt: null params");
    lavt_gemm_nt_t p = *pp;
    LAVT_CHECK_ARG(p.dtype == LAVT_F32 || p.dtype == LAVT_BF16 || p.dtype == LAVT_FP8, "lavt_gemm_nt: bad dtype %d", p.dtype);
    const int epc = p.dtype == LAVT_F32 ? 4 : (p.dtype == LAVT_FP8 ? 16 : 8);
    LAVT_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0 && p.batch >= 1, "lavt_gemm_nt: bad shape M=%d N=%d K=%d batch=%d", p.M, p.N, p.K, p.batch);
    LAVT_CHECK_ARG(p.K % epc == 0, "lavt_gemm_nt: K=%d must be a multiple of %d", p.K, epc);
    LAVT_CHECK_ARG(p.A && p.B && p.C, "lavt_gemm_nt: null operand");
    LAVT_CHECK_ARG(p.lda % epc == 0 && p.ldb % epc == 0, "lavt_gemm_nt: lda/ldb must be multiples of %d", epc);
    LAVT_CHECK_ARG(p.ldc % 4 == 0, "lavt_gemm_nt: ldc must be a multiple of 4");
    LAVT_CHECK_ARG(!p.b_kmajor || p.N % epc == 0, "lavt_gemm_nt: k-major B needs N %% %d == 0", epc);
    LAVT_CHECK_ARG(!p.A2 || (p.a_split % epc == 0 && p.lda2 % epc == 0), "lavt_gemm_nt: bad a_split");
    LAVT_CHECK_ARG(!p.C2 || p.c_split % 4 == 0, "lavt_gemm_nt: bad c_split");
    if (p.conv_kc > 0) {
        const int taps = (p.conv_kd > 0 ? p.conv_kd : 1) * (p.conv_kh > 0 ? p.conv_kh : 3) * (p.conv_kw > 0 ? p.conv_kw : 3);
        const int vox = (p.conv_d > 0 ? p.conv_d : 1) * p.conv_h * p.conv_w;
        if (p.conv_kc_split > 0) {
            int st_ = 0;
            LAVT_CHECK_ARG(p.dtype == LAVT_BF16 && p.conv_tap_split == 0 && p.conv_kc_split % 64 == 0 && p.batch * p.conv_kc_split == p.conv_kc && p.K == taps * p.conv_kc_split &&
                           p.strideB == 0 && taps <= 32 && (!p.A2 || p.a_split % 64 == 0) && p.zeros && lavt_gemm_nt_pipe_tile(p, &st_) == 128,
                           "lavt_gemm_nt: conv_kc_split needs batch * conv_kc_split == conv_kc, K == taps * conv_kc_split, strideB == 0 and the pipelined tap-walking kernel (bf16, conv_kc %% 64 == 0)");
        } else if (p.conv_tap_split > 0)
            LAVT_CHECK_ARG(p.dtype == LAVT_BF16 && p.K == p.conv_tap_split * p.conv_kc && p.batch * p.conv_tap_split == taps && p.conv_kc % 64 == 0 && taps <= 32 &&
                           (!p.A2 || p.a_split % 64 == 0) && p.zeros, "lavt_gemm_nt: conv_tap_split needs batch * conv_tap_split == taps and the tap-walking path (conv_kc %% 64 == 0)");
        else
        LAVT_CHECK_ARG(p.K == taps * p.conv_kc && p.conv_kc % epc == 0 && p.conv_h > 0 && p.conv_w > 0, "lavt_gemm_nt: bad conv geometry");
        LAVT_CHECK_ARG(p.M % vox == 0, "lavt_gemm_nt: conv rows %d not a multiple of D*H*W", p.M);
    }
    LAVT_CHECK_ARG((!p.R || p.ldr % 4 == 0) && (!p.Cpre || p.ldcpre % 4 == 0) && (!p.mul || p.ldmul % 4 == 0), "lavt_gemm_nt: ldr/ldcpre/ldmul must be multiples of 4");
    LAVT_CHECK_ARG(p.act != LAVT_ACT_GELU_D || (p.ln_wsum && p.Cpre && !p.mul && !p.C2 && !p.R && !p.row_scale && !p.c_rowmap),
                   "lavt_gemm_nt: LAVT_ACT_GELU_D exists in the LayerNorm-folded launch only (ln_wsum), needs Cpre and takes no multiplier / split / residual / row scale / row map");
    LAVT_CHECK_ARG(!p.res_first || (p.dact_pre && p.R), "lavt_gemm_nt: res_first orders the residual before the fused activation gradient (needs dact_pre and R)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    { p.epi_lds = lavt_tuning().gemm_epi_lds ? 1 : 0;
      const bool wide_off = lavt_tuning().gemm_epi_narrow;          // LAVT_GEMM_EPI=narrow: the 8-byte store form
      p.epi_wide = (!wide_off && p.dtype != LAVT_F32 && !p.c_f32 && p.ldc % 8 == 0 && (!p.C2 || (p.ldc2 % 8 == 0 && p.c_split % 8 == 0)) && (!p.R || p.ldr % 4 == 0) &&
                    (!p.Cpre || p.ldcpre % 8 == 0) && (!p.dact_pre || p.lddact % 4 == 0) && (!p.bias || (p.strideBias % 4 == 0 && ((uintptr_t)p.bias % 16) == 0)) &&
                    ((uintptr_t)p.C % 16) == 0 && (!p.C2 || ((uintptr_t)p.C2 % 16) == 0) && (!p.Cpre || ((uintptr_t)p.Cpre % 16) == 0) && (p.strideC % 8 == 0)) ? 1 : 0;
      // bit 1: the epilogue's side inputs are requested before the K loop (gemm_common.h NtSide; LAVT_SIDE_PRE=0: at the head of the epilogue, the round-2 form)
      if (p.epi_wide && !lavt_tuning().side_pre_off && p.batch == 1) p.epi_wide |= 2; }
    if (p.colstats) {
        int rpb = 0;
        LAVT_CHECK_ARG(lavt_gemm_nt_colstats_plan(&p, &rpb) > 0, "lavt_gemm_nt: colstats only where lavt_gemm_nt_colstats_plan accepts the problem (pipelined bf16 tiles, plain epilogue)");
    }
    {
        int pipe_stages = 0;
        const int pipe_tile = lavt_gemm_nt_pipe_tile(p, &pipe_stages);          // the 256x256 tile and the long-K 128x128 problems (gemm_nt_pipe.hip)
        if (pipe_tile) return lavt_gemm_nt_pipe(p, pipe_tile, pipe_stages, st);
    }
    const int rc2 = lavt_gemm_nt_v2(p, st);          // bf16 LDS-DMA pipeline (gemm_v2.hip); 1 = not applicable
    if (rc2 != 1) return rc2;
    LAVT_CHECK_ARG(p.ln_wsum == nullptr, "lavt_gemm_nt: ln_wsum (LayerNorm-folded A operand) exists on the bf16 LDS-DMA path only");
    LAVT_CHECK_ARG(p.dact_pre == nullptr, "lavt_gemm_nt: dact_pre (fused activation gradient) exists on the bf16 LDS-DMA path only");
    LAVT_CHECK_ARG(p.dtype != LAVT_FP8, "lavt_gemm_nt: fp8 operands exist on the LDS-DMA path only");
    return p.dtype == LAVT_F32 ? dispatch_nt<float>(p, st) : dispatch_nt<bf16>(p, st);
}

extern "C" int lavt_gemm_tn(const lavt_gemm_tn_t* pp, void* stream) {
    LAVT_CHECK_ARG(pp != nullptr, "lavt_gemm_tn: null params");
    lavt_gemm_tn_t p = *pp;
    LAVT_CHECK_ARG(p.dtype == LAVT_F32 || p.dtype == LAVT_BF16, "lavt_gemm_tn: bad dtype %d", p.dtype);
    const int epc = p.dtype == LAVT_F32 ? 4 : 8;
    LAVT_CHECK_ARG(p.I > 0 && p.J > 0 && p.K > 0 && p.batch >= 1, "lavt_gemm_tn: bad shape");
    LAVT_CHECK_ARG(p.I % epc == 0 && p.J % epc == 0, "lavt_gemm_tn: I=%d, J=%d must be multiples of %d", p.I, p.J, epc);
    LAVT_CHECK_ARG(p.A && p.B && p.C, "lavt_gemm_tn: null operand");
    LAVT_CHECK_ARG(p.lda % epc == 0 && p.ldb % epc == 0, "lavt_gemm_tn: lda/ldb must be multiples of %d", epc);
    LAVT_CHECK_ARG(!p.B2 || (p.b_split % epc == 0 && p.ldb2 % epc == 0), "lavt_gemm_tn: bad b_split");
    if (p.conv_kc > 0)
        LAVT_CHECK_ARG(p.J == (p.conv_kd > 0 ? p.conv_kd : 1) * (p.conv_kh > 0 ? p.conv_kh : 3) * (p.conv_kw > 0 ? p.conv_kw : 3) * p.conv_kc &&
                       p.conv_kc % epc == 0 && p.conv_h > 0 && p.conv_w > 0, "lavt_gemm_tn: bad conv geometry");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int rc2 = lavt_gemm_tn_v2(p, st);          // bf16 LDS-DMA kernel (gemm_v2.hip); 1 = not applicable
    if (rc2 != 1) return rc2;
    LAVT_CHECK_ARG(!p.colsum_atomic, "lavt_gemm_tn: colsum_atomic needs the bf16 LDS-DMA kernel (bf16 operands, 16-byte aligned rows, zeros page)");
    return p.dtype == LAVT_F32 ? dispatch_tn<float>(p, st) : dispatch_tn<bf16>(p, st);
}

// upper bound of the K pieces of lavt_gemm_tn (>= 8 K tiles of 64 rows per piece, unless LAVT_TN_SPLIT forces a count)
extern "C" int lavt_gemm_tn_pieces(const lavt_gemm_tn_t* p) {
    if (!p || p->K <= 0) return 1;
    const int ktiles = cdiv(p->K, 64);
    const int se = lavt_tuning().tn_split;
    // (batched problems -- the per-sample word-side reductions of the fused PWAM node: 2-4 output tiles -- are cut down to 2 K tiles per piece, LAVT_PROBE[4]
    // overrides: a piece is a serial chain of ~1 us per K tile on the 2-stage ring, and the launch sits on the critical chain)
    const int min_kt = p->batch > 1 ? (lavt_tuning().probe[4] > 0 ? lavt_tuning().probe[4] : 2) : 8;
    int n = cdiv(ktiles, min_kt);
    if (se > n) n = se;
    if (n > ktiles) n = ktiles;
    return n < 1 ? 1 : n;
}

struct lavt_ln_rider_t { const void* dy; const void* x; const float* gamma; const float* mean; const float* rstd; void* dx; float* partials; const void* dres; int rows, C; };
int lavt_gemm_tn_grouped_v2(const lavt_gemm_tn_t* probs, int n, hipStream_t st, const lavt_ln_rider_t* ln);
int lavt_layernorm_bwd_partial_impl(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx, float* ws,
                                    int64_t ws_floats, const void* dres, int rows, int C, void* stream);
int lavt_ln_bwd_geometry(int dtype, int rows, int C, int* lpr, int* cpl, int* waves);
// n independent weight-gradient problems issued together: one grouped launch without split-K when they qualify (bf16, plain / row-mapped
// operands, >= 256 output tiles in total), else one lavt_gemm_tn call each.  Results are identical either way up to fp32 summation order.
int64_t lavt_gemm_tn_grouped_sk_ws_v2(const lavt_gemm_tn_t* probs, int n);
int lavt_gemm_tn_grouped_sk_v2(const lavt_gemm_tn_t* probs, int n, float* scratch, int64_t scratch_floats, hipStream_t st);
// The stream-K form of the same launch (csrc/gemm_tn_v2.hip): 128x128 tiles, the K-tile iterations of all members dealt in equal runs to persistent
// workgroups, split tiles through `scratch`.  _ws: floats of scratch the group wants, 0 = the group does not qualify (use lavt_gemm_tn_grouped).
extern "C" int64_t lavt_gemm_tn_grouped_sk_ws(const lavt_gemm_tn_t* probs, int n) {
    if (probs == nullptr || n < 1) return 0;
    for (int i = 0; i < n; ++i)
        if (!(probs[i].A && probs[i].B && probs[i].C && probs[i].I > 0 && probs[i].J > 0 && probs[i].K > 0)) return 0;
    return lavt_gemm_tn_grouped_sk_ws_v2(probs, n);
}
extern "C" int lavt_gemm_tn_grouped_sk(const lavt_gemm_tn_t* probs, int n, float* scratch, int64_t scratch_floats, void* stream) {
    LAVT_CHECK_ARG(probs != nullptr && n >= 1 && scratch != nullptr, "lavt_gemm_tn_grouped_sk: bad arguments");
    const int rc = lavt_gemm_tn_grouped_sk_v2(probs, n, scratch, scratch_floats, reinterpret_cast<hipStream_t>(stream));
    LAVT_CHECK_ARG(rc != 1, "lavt_gemm_tn_grouped_sk: the group does not qualify or the scratch is too small (ask lavt_gemm_tn_grouped_sk_ws first)");
    return rc;
}
// lavt_gemm_tn_grouped + ONE LayerNorm backward (the partial-sum form of lavt_layernorm_bwd_partial, bf16, no gather) that rides as extra workgroups of the
// grouped launch when the group runs on the 64x64 launch with column sums; in every other case the two are issued one after the other -- the result is
// the same either way.  ws / ws_floats: the LayerNorm's partial-sum scratch (lavt_layernorm_bwd_blocks(...) * 2 * C floats).
extern "C" int lavt_gemm_tn_grouped_ln(const lavt_gemm_tn_t* probs, int n, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                       void* dx, float* ws, int64_t ws_floats, const void* dres, int rows, int C, void* stream) {
    LAVT_CHECK_ARG(probs != nullptr && n >= 1 && dy && x && gamma && mean && rstd && dx && ws && rows > 0 && C > 0 && C % 8 == 0, "lavt_gemm_tn_grouped_ln: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    bool ok = true;
    for (int i = 0; i < n; ++i) ok = ok && probs[i].A && probs[i].B && probs[i].C && probs[i].I > 0 && probs[i].J > 0 && probs[i].K > 0;
    LAVT_CHECK_ARG(ok, "lavt_gemm_tn_grouped_ln: null operand / bad shape");
    int lpr, cpl, waves;
    const int blocks = lavt_ln_bwd_geometry(LAVT_BF16, rows, C, &lpr, &cpl, &waves);
    LAVT_CHECK_ARG(ws_floats >= (int64_t)blocks * 2 * C, "lavt_gemm_tn_grouped_ln: LayerNorm scratch of %ld floats needed", (long)blocks * 2 * C);
    lavt_ln_rider_t ln{dy, x, gamma, mean, rstd, dx, ws, dres, rows, C};
    int rc = lavt_gemm_tn_grouped_v2(probs, n, st, &ln);
    if (rc == LAVT_OK) return rc;                                  // both launched, the LayerNorm as riders
    if (rc == 1) {                                                 // the group does not form: one launch per problem
        for (int i = 0; i < n; ++i) {
            const int r = lavt_gemm_tn(&probs[i], stream);
            if (r != LAVT_OK) return r;
        }
    } else if (rc != 3) return rc;
    return lavt_layernorm_bwd_partial_impl(LAVT_BF16, dy, x, gamma, mean, rstd, dx, ws, ws_floats, dres, rows, C, stream);
}
extern "C" int lavt_gemm_tn_grouped(const lavt_gemm_tn_t* probs, int n, void* stream) {
    LAVT_CHECK_ARG(probs != nullptr && n >= 1, "lavt_gemm_tn_grouped: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    bool ok = true;
    for (int i = 0; i < n; ++i) ok = ok && probs[i].A && probs[i].B && probs[i].C && probs[i].I > 0 && probs[i].J > 0 && probs[i].K > 0;
    LAVT_CHECK_ARG(ok, "lavt_gemm_tn_grouped: null operand / bad shape");
    const int rc = lavt_gemm_tn_grouped_v2(probs, n, st, nullptr);
    if (rc != 1) return rc;
    for (int i = 0; i < n; ++i) {
        const int r = lavt_gemm_tn(&probs[i], stream);
        if (r != LAVT_OK) return r;
    }
    return LAVT_OK;
}
