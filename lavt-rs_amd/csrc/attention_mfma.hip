// bf16 MFMA formulation of the shifted-window attention core for gfx950 (v_mfma_f32_16x16x32_bf16).
// Reference arithmetic: WindowAttention.forward, lib/backbone.py:123-140 (+ shift mask :634-652).
//
// Forward: ONE WAVE per (window, head), no workgroup barrier in the loop.  Scores are computed transposed,
//   S^T = K Q^T, so that each lane owns one query row i (the MFMA column) and 4*NT keys j in registers: the
//   softmax row reductions are in-register + two wave shuffles (lane ^ 16, lane ^ 32), and the un-normalised
//   P^T accumulator is already the B operand of O^T = V^T P^T (no LDS round trip for P).  K fragments come
//   straight from HBM/L2 (they are k-contiguous); V fragments need the transposed layout and are produced once
//   per window with ds_read_b64_tr_b16 from a small LDS image; both stay in registers for all query tiles.
// Backward: one workgroup per (head, chunk of windows).  Per window: Q, K, V, dO in LDS; phase 1 recomputes
//   P = exp(S - lse) and dS = P (dP - delta) tile by tile (un-transposed, so the relative-position-bias gradient
//   of a thread always hits the same (i, j) and is summed over the chunk's windows IN REGISTERS, one atomic per
//   element per workgroup at the end instead of one per window); P and dS go to LDS as bf16; phase 2 forms
//   dV = P^T dO, dK = dS^T Q, dQ = dS K with MFMA, reading the k-major operands with the transposing LDS read.
#include "common.h"

namespace {

constexpr int HD = 32;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ bf16x8 join4(bf16x4 lo, bf16x4 hi) {
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
// A/B fragment whose 8 k-values are contiguous in LDS: row `row`, elements k0..k0+7
__device__ __forceinline__ bf16x8 lds_row8(const bf16* s, int ld, int row, int k0) {
    return *reinterpret_cast<const bf16x8*>(s + row * ld + k0);
}
// fragment of a k-major LDS tile s[k][ld]: element jj <-> k = kbase + 8*(lane>>4) + jj, column col0 + (lane & 15)
__device__ __forceinline__ bf16x8 lds_kmajor8(const bf16* s, int ld, int kbase, int col0, int lane) {
    const bf16* p = s + (kbase + 8 * (lane >> 4) + ((lane & 15) >> 2)) * ld + col0 + 4 * (lane & 3);
    return join4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p), __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 4 * ld)));
}
__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (bf16)0.f;
    return z;
}
__device__ __forceinline__ bf16x8 ldg8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }

// ================================================================================================ forward
constexpr int V_LD = 36;        // bf16 elements per LDS row of V (72 B: 8-byte aligned rows, spreads banks)

template <int NT>
__global__ __launch_bounds__(256) void wattn_fwd_mfma(const bf16* __restrict__ qkv, const float* __restrict__ bias, int bias_ld,
                                                      const int8_t* __restrict__ region, int nw_img, bf16* __restrict__ out,
                                                      float* __restrict__ lse, int nwin, int N, int heads, float scale) {
    constexpr int NP = NT * 16, KS = NT / 2;
    constexpr int WAVE_LDS = NP * V_LD * 2 + NP;            // V image + region ids (bytes)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    bf16* Vs = reinterpret_cast<bf16*>(smem_raw + wave * ((WAVE_LDS + 15) / 16 * 16));
    uint8_t* Rs = reinterpret_cast<uint8_t*>(Vs + NP * V_LD);
    const int pair = blockIdx.x * 4 + wave;
    const bool live = pair < nwin * heads;
    const int w = live ? pair / heads : 0, h = live ? pair % heads : 0;
    const int C = heads * HD;
    const bf16* base = qkv + (int64_t)w * N * 3 * C + h * HD;          // + row * 3C (+C for k, +2C for v)

    // ---- stage V (and the region ids) -----------------------------------------------------------------
    for (int e = lane; e < NP * 8; e += 64) {
        const int row = e >> 3, c = e & 7;
        uint2 v = make_uint2(0, 0);
        if (live && row < N) v = *reinterpret_cast<const uint2*>(base + (int64_t)row * 3 * C + 2 * C + c * 4);
        *reinterpret_cast<uint2*>(Vs + row * V_LD + c * 4) = v;
    }
    for (int e = lane; e < NP; e += 64) Rs[e] = (live && region && e < N) ? (uint8_t)region[(int64_t)(w % nw_img) * N + e] : 0;
    __syncthreads();
    if (!live) return;

    // ---- K fragments (A operand of S^T = K Q^T) and V^T fragments (A operand of O^T = V^T P^T) -------------
    bf16x8 kf[NT], vf[2][KS];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int j = 16 * t + c16;
        kf[t] = j < N ? ldg8(base + (int64_t)j * 3 * C + C + 8 * g) : zero8();
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // k-slot (g, jj) <-> key j = 32 ks + 16 (jj >> 2) + 4 g + (jj & 3): matches the accumulator rows of tiles 2ks, 2ks+1
            const bf16* p = Vs + (32 * ks + 4 * g + (c16 >> 2)) * V_LD + 16 * u + 4 * (c16 & 3);
            vf[u][ks] = join4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p),
                              __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 16 * V_LD)));
        }

    const int QT = (N + 15) / 16;
    const float* bh = bias + (int64_t)h * N * bias_ld;
    for (int it = 0; it < QT; ++it) {
        const int i = 16 * it + c16;
        const bool vi = i < N;
        const bf16x8 qf = vi ? ldg8(base + (int64_t)i * 3 * C + 8 * g) : zero8();
        const int rid_i = Rs[vi ? i : 0];
        f32x4 s[NT];
        float mx = -1e30f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t], qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            const int j0 = 16 * t + 4 * g;
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (vi) b = *reinterpret_cast<const float4*>(bh + (int64_t)i * bias_ld + j0);     // columns >= N hold -1e30
            const uint32_t rj = *reinterpret_cast<const uint32_t*>(Rs + j0);
            const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = s[t][r] * scale + bb[r];
                if ((int)((rj >> (8 * r)) & 0xFF) != rid_i) v += -100.0f;
                if (j0 + r >= N) v = -1e30f;
                s[t][r] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float p = __expf(s[t][r] - mx); s[t][r] = p; sum += p; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        if (vi && g == 0) lse[((int64_t)w * heads + h) * N + i] = mx + __logf(sum);
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 pf;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) pf[jj] = (bf16)s[2 * ks + (jj >> 2)][jj & 3];
            o[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[0][ks], pf, o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[1][ks], pf, o[1], 0, 0, 0);
        }
        if (vi) {
            const float inv = 1.f / sum;
            bf16* dst = out + ((int64_t)w * N + i) * C + h * HD + 4 * g;
#pragma unroll
            for (int u = 0; u < 2; ++u)
                *reinterpret_cast<uint2*>(dst + 16 * u) = make_uint2(pack_bf16x2(o[u][0] * inv, o[u][1] * inv), pack_bf16x2(o[u][2] * inv, o[u][3] * inv));
        }
    }
}

// ================================================================================================ backward
constexpr int R_LD = 40;        // bf16 elements per LDS row of Q / K / V / dO (80 B rows, 16-byte aligned chunks)

template <int NT>
__global__ __launch_bounds__(256) void wattn_bwd_mfma(const bf16* __restrict__ qkv, const float* __restrict__ bias, int bias_ld,
                                                      const int8_t* __restrict__ region, int nw_img, const bf16* __restrict__ out,
                                                      const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                      bf16* __restrict__ dqkv, float* __restrict__ dbias, int nwin, int N, int heads,
                                                      float scale, int win_per_block) {
    constexpr int NP = NT * 16, P_LD = NP + 8, KS = NP / 32;
    constexpr int KK = (NT + 3) / 4;                       // query tiles per wave (NT >= number of query tiles)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16* Qs = reinterpret_cast<bf16*>(smem_raw);
    bf16* Ks = Qs + NP * R_LD;
    bf16* Vs = Ks + NP * R_LD;
    bf16* Os = Vs + NP * R_LD;                             // dO
    bf16* Ps = Os + NP * R_LD;                             // [NP][P_LD]
    bf16* Ss = Ps + NP * P_LD;                             // dS
    float* dl = reinterpret_cast<float*>(Ss + NP * P_LD);  // delta_i
    float* ls = dl + NP;                                   // lse_i
    uint8_t* Rs = reinterpret_cast<uint8_t*>(ls + NP);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int h = blockIdx.x % heads, chunk = blockIdx.x / heads;
    const int C = heads * HD;
    const int QT = (N + 15) / 16;
    const float* bh = bias + (int64_t)h * N * bias_ld;

    f32x4 db[KK][NT];
#pragma unroll
    for (int k = 0; k < KK; ++k)
#pragma unroll
        for (int t = 0; t < NT; ++t) db[k][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // P / dS rows beyond the last query tile are read by the k-loops of dV / dK: keep them zero for the whole kernel
    for (int e = tid; e < NP * P_LD; e += 256) { Ps[e] = (bf16)0.f; Ss[e] = (bf16)0.f; }

    const int w_end = min(nwin, (chunk + 1) * win_per_block);
    for (int w = chunk * win_per_block; w < w_end; ++w) {
        __syncthreads();
        // ---- stage Q, K, V, dO (rows >= N are zero), delta_i = sum_d dO*O, lse_i, region ids ----------------------
        const bf16* base = qkv + (int64_t)w * N * 3 * C + h * HD;
        for (int e = tid; e < NP * 4; e += 256) {
            const int row = e >> 2, c = e & 3;
            uint4 q = make_uint4(0, 0, 0, 0), k = q, v = q, d = q, o = q;
            if (row < N) {
                const bf16* r = base + (int64_t)row * 3 * C + c * 8;
                q = *reinterpret_cast<const uint4*>(r);
                k = *reinterpret_cast<const uint4*>(r + C);
                v = *reinterpret_cast<const uint4*>(r + 2 * C);
                const int64_t oo = ((int64_t)w * N + row) * C + h * HD + c * 8;
                d = *reinterpret_cast<const uint4*>(dout + oo);
                o = *reinterpret_cast<const uint4*>(out + oo);
            }
            *reinterpret_cast<uint4*>(Qs + row * R_LD + c * 8) = q;
            *reinterpret_cast<uint4*>(Ks + row * R_LD + c * 8) = k;
            *reinterpret_cast<uint4*>(Vs + row * R_LD + c * 8) = v;
            *reinterpret_cast<uint4*>(Os + row * R_LD + c * 8) = d;
            float fd[8], fo[8], part = 0.f;
            chunk_to_f<bf16>(d, fd);
            chunk_to_f<bf16>(o, fo);
#pragma unroll
            for (int x = 0; x < 8; ++x) part += fd[x] * fo[x];
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            if (c == 0) { dl[row] = part; ls[row] = row < N ? lse[((int64_t)w * heads + h) * N + row] : 0.f; }
        }
        for (int e = tid; e < NP; e += 256) Rs[e] = (region && e < N) ? (uint8_t)region[(int64_t)(w % nw_img) * N + e] : 0;
        __syncthreads();

        // ---- phase 1: P and dS tiles (rows i = 4g+r of the query tile, column j = lane & 15) ------------------------
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            const int it = wave + 4 * k;
            if (it < QT) {
                const bf16x8 qf = lds_row8(Qs, R_LD, 16 * it + c16, 8 * g);
                const bf16x8 of = lds_row8(Os, R_LD, 16 * it + c16, 8 * g);
                float li[4], di[4];
                int ri[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int i = 16 * it + 4 * g + r; li[r] = ls[i]; di[r] = dl[i]; ri[r] = Rs[i]; }
                // all bias values of this query tile are requested up front: one L2 round trip per tile instead of one per key tile
                float bb[NT][4];
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * it + 4 * g + r, j = 16 * t + c16;
                        bb[t][r] = (i < N && j < N) ? bh[(int64_t)i * bias_ld + j] : 0.f;
                    }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int j = 16 * t + c16;
                    const bf16x8 kfr = lds_row8(Ks, R_LD, j, 8 * g);
                    const bf16x8 vfr = lds_row8(Vs, R_LD, j, 8 * g);
                    const f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kfr, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(of, vfr, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    const int rj = Rs[j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * it + 4 * g + r;
                        float p = 0.f, ds = 0.f;
                        if (i < N && j < N) {
                            float a = s[r] * scale + bb[t][r];
                            if (rj != ri[r]) a += -100.0f;
                            p = __expf(a - li[r]);
                            ds = p * (dp[r] - di[r]);
                        }
                        db[k][t][r] += ds;
                        Ps[i * P_LD + j] = (bf16)p;
                        Ss[i * P_LD + j] = (bf16)ds;
                    }
                }
            }
        }
        __syncthreads();

        // ---- phase 2: dV = P^T dO, dK = scale dS^T Q (NT x 2 tiles each), dQ = scale dS K (QT x 2 tiles) -------------
        const int n_kv = NT * 2, n_all = 2 * n_kv + QT * 2;
        for (int idx = wave; idx < n_all; idx += 4) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            if (idx < 2 * n_kv) {
                const bool is_k = idx >= n_kv;
                const int id = is_k ? idx - n_kv : idx;
                const int jt = id >> 1, u = id & 1;
                const bf16* Am = is_k ? Ss : Ps;           // A[row j][k = i]  = M[i][j]  (k-major)
                const bf16* Bm = is_k ? Qs : Os;           // B[k = i][col d]  (k-major)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_kmajor8(Am, P_LD, 32 * ks, 16 * jt, lane),
                                                                  lds_kmajor8(Bm, R_LD, 32 * ks, 16 * u, lane), acc, 0, 0, 0);
                const float f = is_k ? scale : 1.f;
                bf16* dst = dqkv + (int64_t)w * N * 3 * C + (is_k ? C : 2 * C) + h * HD + 16 * u + c16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * jt + 4 * g + r;
                    if (j < N) dst[(int64_t)j * 3 * C] = (bf16)(acc[r] * f);
                }
            } else {
                const int id = idx - 2 * n_kv;
                const int it = id >> 1, u = id & 1;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row8(Ss, P_LD, 16 * it + c16, 32 * ks + 8 * g),
                                                                  lds_kmajor8(Ks, R_LD, 32 * ks, 16 * u, lane), acc, 0, 0, 0);
                bf16* dst = dqkv + (int64_t)w * N * 3 * C + h * HD + 16 * u + c16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * it + 4 * g + r;
                    if (i < N) dst[(int64_t)i * 3 * C] = (bf16)(acc[r] * scale);
                }
            }
        }
    }
    // ---- relative-position-bias gradient: one atomic per (i, j) per workgroup -----------------------------------
    float* dbh = dbias + (int64_t)h * N * bias_ld;
#pragma unroll
    for (int k = 0; k < KK; ++k) {
        const int it = wave + 4 * k;
        if (it < QT) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * it + 4 * g + r, j = 16 * t + c16;
                    if (i < N && j < N) atomicAdd(dbh + (int64_t)i * bias_ld + j, db[k][t][r]);
                }
        }
    }
}

template <int NT> size_t bwd_lds_bytes() {
    constexpr int NP = NT * 16, P_LD = NP + 8;
    return (size_t)4 * NP * R_LD * 2 + (size_t)2 * NP * P_LD * 2 + (size_t)2 * NP * 4 + NP;
}

}  // namespace

int lavt_window_attn_fwd_mfma(const void* qkv, const float* bias, int bias_ld, const int8_t* region, int nw_img, void* out, float* lse,
                              int nwin, int N, int heads, float scale, hipStream_t st) {
    const int NT = N <= 64 ? 4 : 10;
    if (N > 160 || bias_ld < NT * 16 || bias_ld % 4) { lavt_set_error("lavt_window_attn_fwd(mfma): N=%d bias_ld=%d unsupported", N, bias_ld); return LAVT_ERR_INVALID; }
    const int blocks = cdiv((long)nwin * heads, 4);
    const size_t wave_lds = (size_t)((NT * 16 * V_LD * 2 + NT * 16 + 15) / 16 * 16);
    if (NT == 4) hipLaunchKernelGGL(wattn_fwd_mfma<4>, dim3(blocks), dim3(256), 4 * wave_lds, st, (const bf16*)qkv, bias, bias_ld, region, nw_img, (bf16*)out, lse, nwin, N, heads, scale);
    else hipLaunchKernelGGL(wattn_fwd_mfma<10>, dim3(blocks), dim3(256), 4 * wave_lds, st, (const bf16*)qkv, bias, bias_ld, region, nw_img, (bf16*)out, lse, nwin, N, heads, scale);
    LAVT_CHECK_LAUNCH("lavt_window_attn_fwd(mfma)");
    return LAVT_OK;
}

int lavt_window_attn_bwd_mfma(const void* qkv, const float* bias, int bias_ld, const int8_t* region, int nw_img, const void* out,
                              const void* dout, const float* lse, void* dqkv, float* dbias, int nwin, int N, int heads, float scale,
                              hipStream_t st) {
    const int NT = N <= 64 ? 4 : 10;
    if (N > 160 || bias_ld < N) { lavt_set_error("lavt_window_attn_bwd(mfma): N=%d bias_ld=%d unsupported", N, bias_ld); return LAVT_ERR_INVALID; }
    // ~2 workgroups per CU; each sums its windows' bias gradient in registers before touching memory
    int wpb = (int)(((long)nwin * heads + 511) / 512);
    if (wpb < 1) wpb = 1;
    const int chunks = cdiv(nwin, wpb);
    dim3 grid(chunks * heads);
    if (NT == 4) {
        hipLaunchKernelGGL(wattn_bwd_mfma<4>, grid, dim3(256), bwd_lds_bytes<4>(), st, (const bf16*)qkv, bias, bias_ld, region, nw_img, (const bf16*)out,
                           (const bf16*)dout, lse, (bf16*)dqkv, dbias, nwin, N, heads, scale, wpb);
    } else {
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wattn_bwd_mfma<10>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds_bytes<10>()) != hipSuccess) {
                lavt_set_error("lavt_window_attn_bwd(mfma): cannot reserve %zu bytes of LDS", bwd_lds_bytes<10>());
                return LAVT_ERR_LAUNCH;
            }
            attr_set = true;
        }
        hipLaunchKernelGGL(wattn_bwd_mfma<10>, grid, dim3(256), bwd_lds_bytes<10>(), st, (const bf16*)qkv, bias, bias_ld, region, nw_img, (const bf16*)out,
                           (const bf16*)dout, lse, (bf16*)dqkv, dbias, nwin, N, heads, scale, wpb);
    }
    LAVT_CHECK_LAUNCH("lavt_window_attn_bwd(mfma)");
    return LAVT_OK;
}
