// Shifted-window attention core (scores + relative-position bias + shift mask + softmax + PV) and its
// backward, one workgroup per (window, head).  Reference: WindowAttention.forward, lib/backbone.py:123-140.
//
// This file holds the exact-fp32 VALU formulation used for LAVT_F32 (parity path) and, until the MFMA
// formulation in attention_mfma.hip takes a shape, for LAVT_BF16 as well (bf16 storage, fp32 math).
// K and V of the window live in LDS for the whole workgroup; each wave owns query rows i = wave, wave+4, ...
// and spreads the key index j over its 64 lanes, so the softmax row reductions are wave shuffles.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int HD = 32;          // head_dim is 32 in every stage of every Swin variant
constexpr int KV_LD = HD + 1;   // +1 float: lanes index rows -> conflict-free LDS reads

template <typename T>
__device__ __forceinline__ void load_kv(const T* qkv, int64_t row0, int C, int h, int N, float* Ks, float* Vs, float* Qs, float scale) {
    // each thread copies (row, d) pairs; global reads are 32 contiguous elements per row
    for (int e = threadIdx.x; e < N * HD; e += blockDim.x) {
        const int j = e / HD, d = e % HD;
        const T* r = qkv + (row0 + j) * (int64_t)(3 * C) + h * HD + d;
        if (Qs) Qs[j * KV_LD + d] = to_f<T>(r[0]) * scale;
        Ks[j * KV_LD + d] = to_f<T>(r[C]);
        Vs[j * KV_LD + d] = to_f<T>(r[2 * C]);
    }
}

// ------------------------------------------------------------------------------------------------ forward
template <typename T, int NJ>
__global__ __launch_bounds__(256) void window_attn_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ bias,
                                                              const int8_t* __restrict__ region, int nw_img, T* __restrict__ out,
                                                              float* __restrict__ lse, int N, int heads, float scale, int bias_ld) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Ks = reinterpret_cast<float*>(smem_raw);
    float* Vs = Ks + N * KV_LD;
    float* Qs = Vs + N * KV_LD;
    float* Pw = Qs + N * KV_LD;                 // [4][NJ*64]
    const int w = blockIdx.x / heads, h = blockIdx.x % heads;
    const int C = heads * HD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)w * N;
    load_kv<T>(qkv, row0, C, h, N, Ks, Vs, Qs, scale);
    __syncthreads();
    const int8_t* reg = region ? region + (int64_t)(w % nw_img) * N : nullptr;
    const float* bh = bias + (int64_t)h * N * bias_ld;
    float* P = Pw + wave * NJ * 64;

    int rid_j[NJ];
#pragma unroll
    for (int t = 0; t < NJ; ++t) { const int j = lane + 64 * t; rid_j[t] = (reg && j < N) ? reg[j] : 0; }

    for (int i = wave; i < N; i += 4) {
        const float* q = Qs + i * KV_LD;
        const int rid_i = reg ? reg[i] : 0;
        float s[NJ], mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NJ; ++t) {
            const int j = lane + 64 * t;
            float a = -INFINITY;
            if (j < N) {
                a = 0.f;
                const float* k = Ks + j * KV_LD;
#pragma unroll
                for (int d = 0; d < HD; ++d) a = fmaf(q[d], k[d], a);
                a += bh[(int64_t)i * bias_ld + j];
                if (rid_j[t] != rid_i) a += -100.0f;
            }
            s[t] = a;
            mx = fmaxf(mx, a);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NJ; ++t) { s[t] = (lane + 64 * t < N) ? __expf(s[t] - mx) : 0.f; sum += s[t]; }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int t = 0; t < NJ; ++t) P[lane + 64 * t] = s[t] * inv;
        if (lane == 0) lse[((int64_t)w * heads + h) * N + i] = mx + __logf(sum);
        __builtin_amdgcn_wave_barrier();
        // PV: lanes 0..31 take even j, lanes 32..63 odd j, for column d = lane & 31
        const int d = lane & 31, par = lane >> 5;
        float o = 0.f;
        for (int j = par; j < N; j += 2) o = fmaf(P[j], Vs[j * KV_LD + d], o);
        o += __shfl_xor(o, 32, 64);
        if (lane < 32) out[(row0 + i) * C + h * HD + d] = from_f<T>(o);
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------ backward
// relative position bias: table[(2wd-1)(2wh-1)(2ww-1)][heads] <-> dense[heads][N][ld]   (wd = 1: the 2-D Swin table).
// Token i of an N-token window has the coordinates of token i of the FULL (wd,wh,ww) window -- this reproduces the
// reference's `relative_position_index[:N, :N]` slice for clipped video windows (lib/video_swin_transformer.py:150).
__device__ __forceinline__ int relpos_index(int i, int j, int wd, int wh, int ww) {
    const int di = i / (wh * ww), hi = (i / ww) % wh, wi = i % ww;
    const int dj = j / (wh * ww), hj = (j / ww) % wh, wj = j % ww;
    return ((di - dj + wd - 1) * (2 * wh - 1) + (hi - hj + wh - 1)) * (2 * ww - 1) + (wi - wj + ww - 1);
}
template <typename T, int NJ>
__global__ __launch_bounds__(256) void window_attn_bwd_kernel(const T* __restrict__ qkv, const float* __restrict__ bias,
                                                              const int8_t* __restrict__ region, int nw_img,
                                                              const T* __restrict__ out, const T* __restrict__ dout,
                                                              const float* __restrict__ lse, T* __restrict__ dqkv,
                                                              float* __restrict__ dbias, int N, int heads, float scale, int bias_ld,
                                                              float* __restrict__ dtable, int wd, int wh, int ww) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Ks = reinterpret_cast<float*>(smem_raw);
    float* Vs = Ks + N * KV_LD;
    float* Qs = Vs + N * KV_LD;                 // scaled q
    float* dKs = Qs + N * KV_LD;                // cross-wave accumulators
    float* dVs = dKs + N * KV_LD;
    float* Sw = dVs + N * KV_LD;                // [4][NJ*64] dS rows
    float* Dw = Sw + 4 * NJ * 64;               // [4][HD] dO row
    const int w = blockIdx.x / heads, h = blockIdx.x % heads;
    const int C = heads * HD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)w * N;
    load_kv<T>(qkv, row0, C, h, N, Ks, Vs, Qs, scale);
    for (int e = threadIdx.x; e < N * KV_LD; e += blockDim.x) { dKs[e] = 0.f; dVs[e] = 0.f; }
    __syncthreads();
    const int8_t* reg = region ? region + (int64_t)(w % nw_img) * N : nullptr;
    const float* bh = bias + (int64_t)h * N * bias_ld;
    float* dbh = dbias ? dbias + (int64_t)h * N * bias_ld : nullptr;
    float* dS = Sw + wave * NJ * 64;
    float* dOr = Dw + wave * HD;

    int rid_j[NJ];
    float dk[NJ][HD], dv[NJ][HD];
#pragma unroll
    for (int t = 0; t < NJ; ++t) {
        const int j = lane + 64 * t;
        rid_j[t] = (reg && j < N) ? reg[j] : 0;
#pragma unroll
        for (int d = 0; d < HD; ++d) { dk[t][d] = 0.f; dv[t][d] = 0.f; }
    }

    for (int i = wave; i < N; i += 4) {
        const float* q = Qs + i * KV_LD;
        const int rid_i = reg ? reg[i] : 0;
        const float l = lse[((int64_t)w * heads + h) * N + i];
        // dO row and delta = sum_d dO*O
        float dl = 0.f;
        if (lane < HD) {
            const float g = to_f<T>(dout[(row0 + i) * C + h * HD + lane]);
            dOr[lane] = g;
            dl = g * to_f<T>(out[(row0 + i) * C + h * HD + lane]);
        }
        dl = wave_sum(dl);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < NJ; ++t) {
            const int j = lane + 64 * t;
            float ds = 0.f;
            if (j < N) {
                float a = 0.f, dp = 0.f;
                const float* k = Ks + j * KV_LD;
                const float* v = Vs + j * KV_LD;
#pragma unroll
                for (int d = 0; d < HD; ++d) { a = fmaf(q[d], k[d], a); dp = fmaf(dOr[d], v[d], dp); }
                a += bh[(int64_t)i * bias_ld + j];
                if (rid_j[t] != rid_i) a += -100.0f;
                const float pj = __expf(a - l);
                ds = pj * (dp - dl);
                if (dtable) atomicAdd(dtable + (int64_t)relpos_index(i, j, wd, wh, ww) * heads + h, ds);      // parity path: plain global atomics
                else atomicAdd(dbh + (int64_t)i * bias_ld + j, ds);
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    dv[t][d] = fmaf(pj, dOr[d], dv[t][d]);
                    dk[t][d] = fmaf(ds, q[d], dk[t][d]);          // q already carries `scale`
                }
            }
            dS[j] = ds;
        }
        __builtin_amdgcn_wave_barrier();
        // dQ[i][d] = scale * sum_j dS[j] K[j][d]
        const int d = lane & 31, par = lane >> 5;
        float o = 0.f;
        for (int j = par; j < N; j += 2) o = fmaf(dS[j], Ks[j * KV_LD + d], o);
        o += __shfl_xor(o, 32, 64);
        if (lane < 32) dqkv[(row0 + i) * (int64_t)(3 * C) + h * HD + d] = from_f<T>(o * scale);
        __builtin_amdgcn_wave_barrier();
    }
    // cross-wave reduction of dK, dV through LDS, then store
#pragma unroll
    for (int t = 0; t < NJ; ++t) {
        const int j = lane + 64 * t;
        if (j < N) {
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                atomicAdd(dKs + j * KV_LD + d, dk[t][d]);
                atomicAdd(dVs + j * KV_LD + d, dv[t][d]);
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < N * HD; e += blockDim.x) {
        const int j = e / HD, d = e % HD;
        T* r = dqkv + (row0 + j) * (int64_t)(3 * C) + h * HD + d;
        r[C] = from_f<T>(dKs[j * KV_LD + d]);
        r[2 * C] = from_f<T>(dVs[j * KV_LD + d]);
    }
}

__global__ void relpos_expand_kernel(const float* __restrict__ table, float* __restrict__ dense, int wd, int wh, int ww, int N, int heads, int ld) {
    const int64_t total = (int64_t)heads * N * ld;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int j = e % ld, i = (e / ld) % N, h = e / ((int64_t)N * ld);
        dense[e] = j < N ? table[relpos_index(i, j, wd, wh, ww) * heads + h] : -1e30f;       // padding columns can never win a softmax
    }
}
// one wave per (table row, head): lanes stride over the query tokens i, the matching key j follows from the offsets -- deterministic
__global__ void relpos_reduce_kernel(const float* __restrict__ ddense, float* __restrict__ dtable, int wd, int wh, int ww, int N, int heads, int ld) {
    const int R = (2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1);
    const int e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (e >= R * heads) return;
    const int h = e % heads, idx = e / heads;
    const int dw = idx % (2 * ww - 1) - (ww - 1), dh = (idx / (2 * ww - 1)) % (2 * wh - 1) - (wh - 1), dd = idx / ((2 * ww - 1) * (2 * wh - 1)) - (wd - 1);
    float s = 0.f;
    for (int i = lane; i < N; i += 64) {
        const int zi = i / (wh * ww), yi = (i / ww) % wh, xi = i % ww;
        const int zj = zi - dd, yj = yi - dh, xj = xi - dw;
        if (zj < 0 || zj >= wd || yj < 0 || yj >= wh || xj < 0 || xj >= ww) continue;
        const int j = (zj * wh + yj) * ww + xj;
        if (j < N) s += ddense[((int64_t)h * N + i) * ld + j];
    }
    s = wave_sum(s);
    if (lane == 0) dtable[idx * heads + h] += s;
}

// ---- row softmax of attention scores for the composed (GEMM + softmax + GEMM) path used for windows too large for the
// fused kernels (Video-Swin: N = 392 / 1152): p = softmax_j(s[i][j] + bias[i][j] + mask(region_i, region_j)), one wave per row.
// One wave per row, the row held in registers as 16-byte chunks (lane owns chunks lane, lane + 64, ...: one pass over s, 16-byte
// loads and stores; ld % (16 / sizeof(T)) == 0 and ld <= SM_KMAX * 64 chunks).
constexpr int SM_KMAX = 6;
template <typename T>
__global__ __launch_bounds__(256) void attn_softmax_fwd_kernel(const T* __restrict__ s, const float* __restrict__ bias, int bias_ld,
                                                               const int8_t* __restrict__ region, int nw_img, T* __restrict__ p,
                                                               int64_t rows, int rpw, int N, int ld, int heads) {
    // rows are ordered (window, head, token): block b = row / rpw is (window b / heads, head b % heads); bias is [heads][N][bias_ld]
    constexpr int EPC = Chunk<T>::N;
    const int lane = threadIdx.x & 63;
    const int nch = ld / EPC;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const int i = (int)(row % rpw);
        const int64_t blk = row / rpw;
        const int64_t w = blk / heads;
        const T* sr = s + row * ld;
        T* pr = p + row * ld;
        if (i >= N) {                       // padding row of a window whose token count was rounded up
            float z[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) z[e] = 0.f;
            for (int c = lane; c < nch; c += 64) *reinterpret_cast<uint4*>(pr + c * EPC) = f_to_chunk<T>(z);
            continue;
        }
        const float* br = bias + ((blk % heads) * (int64_t)N + i) * bias_ld;
        const int8_t* reg = region ? region + (w % nw_img) * N : nullptr;
        // vector side loads need 16-byte aligned bias rows and EPC-byte aligned region rows (N % EPC == 0 covers both row strides)
        const bool vec = (bias_ld % 4 == 0) && (N % EPC == 0) && ((reinterpret_cast<uintptr_t>(bias) & 15) == 0) && (!region || (reinterpret_cast<uintptr_t>(region) & 7) == 0);
        const int ri = reg ? reg[i] : 0;
        float v[SM_KMAX][EPC];
        float mx = -1e30f;
#pragma unroll
        for (int k = 0; k < SM_KMAX; ++k) {
            const int c = lane + 64 * k;
            if (c < nch) {
                chunk_to_f<T>(*reinterpret_cast<const uint4*>(sr + c * EPC), v[k]);
                if (vec && c * EPC + EPC <= N) {
                    // whole chunk inside the row: bias as 16-byte loads, the chunk's region ids as one 4 / 8-byte load (the per-element
                    // form below issues 2 scalar loads per element: 16 of them per 16-byte data load, measured 149 vs 52 us for the backward)
                    float b[EPC];
                    const float4 b0 = *reinterpret_cast<const float4*>(br + c * EPC);
                    b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w;
                    if constexpr (EPC == 8) { const float4 b1 = *reinterpret_cast<const float4*>(br + c * EPC + 4); b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w; }
                    unsigned long long r8 = 0;
                    if (reg) {
                        if constexpr (EPC == 8) r8 = *reinterpret_cast<const unsigned long long*>(reg + c * EPC);
                        else r8 = *reinterpret_cast<const unsigned*>(reg + c * EPC);
                    }
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        float x = v[k][e] + b[e];
                        if (reg && (int)(int8_t)((r8 >> (8 * e)) & 0xFF) != ri) x += -100.0f;
                        v[k][e] = x;
                        mx = fmaxf(mx, x);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const int j = c * EPC + e;
                        float x = -1e30f;
                        if (j < N) {
                            x = v[k][e] + br[j];
                            if (reg && reg[j] != ri) x += -100.0f;
                        }
                        v[k][e] = x;
                        mx = fmaxf(mx, x);
                    }
                }
            }
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < SM_KMAX; ++k)
            if (lane + 64 * k < nch)
#pragma unroll
                for (int e = 0; e < EPC; ++e) { const float ex = (lane + 64 * k) * EPC + e < N ? __expf(v[k][e] - mx) : 0.f; v[k][e] = ex; sum += ex; }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int k = 0; k < SM_KMAX; ++k) {
            const int c = lane + 64 * k;
            if (c < nch) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[k][e] *= inv;
                *reinterpret_cast<uint4*>(pr + c * EPC) = f_to_chunk<T>(v[k]);
            }
        }
    }
}
// ds = p * (dp - sum_j p dp), in place on dp; padding columns of the row are written as 0
template <typename T>
__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(const T* __restrict__ p, T* __restrict__ dp, int64_t rows, int N, int ld) {
    constexpr int EPC = Chunk<T>::N;
    const int lane = threadIdx.x & 63;
    const int nch = ld / EPC;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const T* pr = p + row * ld;
        T* dr = dp + row * ld;
        float fp[SM_KMAX][EPC], fd[SM_KMAX][EPC];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < SM_KMAX; ++k) {
            const int c = lane + 64 * k;
            if (c < nch) {
                chunk_to_f<T>(*reinterpret_cast<const uint4*>(pr + c * EPC), fp[k]);
                chunk_to_f<T>(*reinterpret_cast<const uint4*>(dr + c * EPC), fd[k]);
#pragma unroll
                for (int e = 0; e < EPC; ++e) if (c * EPC + e < N) dot += fp[k][e] * fd[k][e];
            }
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int k = 0; k < SM_KMAX; ++k) {
            const int c = lane + 64 * k;
            if (c < nch) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) fd[k][e] = c * EPC + e < N ? fp[k][e] * (fd[k][e] - dot) : 0.f;
                *reinterpret_cast<uint4*>(dr + c * EPC) = f_to_chunk<T>(fd[k]);
            }
        }
    }
}

template <typename T>
int launch_fwd(const void* qkv, const float* bias, int bias_ld, const int8_t* region, int nw_img, void* out, float* lse, int nwin, int N,
               int heads, float scale, hipStream_t st) {
    const int NJ = (N + 63) / 64;
    const size_t lds = (size_t)(3 * N * KV_LD + 4 * NJ * 64) * sizeof(float);
    dim3 grid(nwin * heads);
#define L(NJ_)                                                                                                              \
    hipLaunchKernelGGL((window_attn_fwd_kernel<T, NJ_>), grid, dim3(256), lds, st, (const T*)qkv, bias, region, nw_img, \
                       (T*)out, lse, N, heads, scale, bias_ld)
    if (NJ == 1) L(1); else if (NJ == 2) L(2); else if (NJ == 3) L(3); else { lavt_set_error("lavt_window_attn_fwd: N=%d > 192 not supported by this kernel", N); return LAVT_ERR_INVALID; }
#undef L
    LAVT_CHECK_LAUNCH("lavt_window_attn_fwd");
    return LAVT_OK;
}
template <typename T>
int launch_bwd(const void* qkv, const float* bias, int bias_ld, const int8_t* region, int nw_img, const void* out, const void* dout,
               const float* lse, void* dqkv, float* dbias, float* dtable, int wd, int wh, int ww, int nwin, int N, int heads, float scale, hipStream_t st) {
    const int NJ = (N + 63) / 64;
    const size_t lds = (size_t)(5 * N * KV_LD + 4 * NJ * 64 + 4 * HD) * sizeof(float);
    dim3 grid(nwin * heads);
#define L(NJ_)                                                                                                              \
    hipLaunchKernelGGL((window_attn_bwd_kernel<T, NJ_>), grid, dim3(256), lds, st, (const T*)qkv, bias, region, nw_img, \
                       (const T*)out, (const T*)dout, lse, (T*)dqkv, dbias, N, heads, scale, bias_ld, dtable, wd, wh, ww)
    if (NJ == 1) L(1); else if (NJ == 2) L(2); else if (NJ == 3) L(3); else { lavt_set_error("lavt_window_attn_bwd: N=%d > 192 not supported by this kernel", N); return LAVT_ERR_INVALID; }
#undef L
    LAVT_CHECK_LAUNCH("lavt_window_attn_bwd");
    return LAVT_OK;
}

}  // namespace

int lavt_window_attn_fwd_mfma(const void* qkv, const float* table, const int8_t* region, int nw_img, void* out, float* lse,
                              int wd, int wh, int ww, int nwin, int N, int heads, float scale, hipStream_t st);
int lavt_window_attn_bwd_mfma(const void* qkv, const float* table, const int8_t* region, int nw_img, const void* out, const void* dout,
                              const float* lse, void* dqkv, float* dtable, int bias_ld, float* ws, float* parts, int wd, int wh, int ww, int nwin, int N,
                              int heads, float scale, hipStream_t st, const lavt_dtable_job_t* prev, lavt_dtable_job_t* mine);
int lavt_attn_dtable_run_mfma(const lavt_dtable_job_t* jb, hipStream_t st);
int lavt_attn_dtable_finish_multi_impl(const int64_t* desc, int n, int max_R, int max_heads, int total_heads, hipStream_t st);
int lavt_window_attn_bwd_pieces_mfma(int nwin, int N, int heads);
int64_t lavt_window_attn_bwd_ws_mfma(int nwin, int N, int heads, int bias_ld, int wd, int wh, int ww);
// LAVT_ATTN_SIMPLE=1 forces the VALU formulation for bf16 too (A/B tests of the MFMA kernels)
static bool use_mfma(int dtype, int N, int bias_ld) {
    return dtype == LAVT_BF16 && N <= 400 && bias_ld >= (N <= 64 ? 64 : N <= 160 ? 160 : 416) && !lavt_tuning().attn_simple;
}

extern "C" int lavt_window_attn_fwd(int dtype, const void* qkv, const float* bias, int bias_ld, const int8_t* region, int nw_img, void* out,
                                    float* lse, const float* table, int wd, int wh, int ww, int nwin, int N, int heads, int head_dim,
                                    float scale, void* stream) {
    LAVT_CHECK_ARG(head_dim == HD, "lavt_window_attn_fwd: head_dim %d != 32", head_dim);
    LAVT_CHECK_ARG(qkv && out && lse && nwin > 0 && N > 0 && heads > 0 && bias_ld >= N, "lavt_window_attn_fwd: bad arguments");
    LAVT_CHECK_ARG(!region || nw_img > 0, "lavt_window_attn_fwd: region needs nw_img");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (use_mfma(dtype, N, bias_ld) && table && wd > 0 && wh > 0 && ww > 0 && N <= wd * wh * ww)
        return lavt_window_attn_fwd_mfma(qkv, table, region, nw_img, out, lse, wd, wh, ww, nwin, N, heads, scale, st);
    LAVT_CHECK_ARG(bias != nullptr, "lavt_window_attn_fwd: the exact-fp32 kernel needs the dense bias (lavt_relpos_expand)");
    if (dtype == LAVT_F32) return launch_fwd<float>(qkv, bias, bias_ld, region, nw_img, out, lse, nwin, N, heads, scale, st);
    if (dtype == LAVT_BF16) return launch_fwd<bf16>(qkv, bias, bias_ld, region, nw_img, out, lse, nwin, N, heads, scale, st);
    lavt_set_error("lavt_window_attn_fwd: bad dtype %d", dtype);
    return LAVT_ERR_INVALID;
}

extern "C" int lavt_window_attn_bwd(int dtype, const void* qkv, const float* bias, int bias_ld, const int8_t* region, int nw_img,
                                    const void* out, const void* dout, const float* lse, void* dqkv, const float* table, float* dtable,
                                    float* ws, int64_t ws_floats, float* parts, int wd, int wh, int ww, int nwin, int N, int heads, int head_dim,
                                    float scale, void* stream) {
    LAVT_CHECK_ARG(head_dim == HD, "lavt_window_attn_bwd: head_dim %d != 32", head_dim);
    LAVT_CHECK_ARG(qkv && out && dout && lse && dqkv && table && (dtable || parts) && nwin > 0 && N > 0 && heads > 0, "lavt_window_attn_bwd: bad arguments");
    LAVT_CHECK_ARG(wd > 0 && wh > 0 && ww > 0 && N <= wd * wh * ww, "lavt_window_attn_bwd: window shape (wd, wh, ww) must cover N tokens");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // the bf16 MFMA kernel takes the bias values from an LDS copy of the table and needs one fp32 [N][bias_ld] slab per (window, head)
    const bool fast = use_mfma(dtype, N, bias_ld) && ws && ws_floats >= lavt_window_attn_bwd_ws_mfma(nwin, N, heads, bias_ld, wd, wh, ww);
    if (fast) return lavt_window_attn_bwd_mfma(qkv, table, region, nw_img, out, dout, lse, dqkv, dtable, bias_ld, ws, parts, wd, wh, ww, nwin, N, heads, scale, st, nullptr, nullptr);
    LAVT_CHECK_ARG(parts == nullptr && dtable, "lavt_window_attn_bwd: the deferred table-gradient form exists on the bf16 MFMA path only");
    LAVT_CHECK_ARG(bias && bias_ld >= N, "lavt_window_attn_bwd: the exact-fp32 kernel needs the dense bias (lavt_relpos_expand)");
    if (dtype == LAVT_F32) return launch_bwd<float>(qkv, bias, bias_ld, region, nw_img, out, dout, lse, dqkv, nullptr, dtable, wd, wh, ww, nwin, N, heads, scale, st);
    if (dtype == LAVT_BF16) return launch_bwd<bf16>(qkv, bias, bias_ld, region, nw_img, out, dout, lse, dqkv, nullptr, dtable, wd, wh, ww, nwin, N, heads, scale, st);
    lavt_set_error("lavt_window_attn_bwd: bad dtype %d", dtype);
    return LAVT_ERR_INVALID;
}
// The bf16 MFMA backward in its deferred-table form with the binning CHAINED: this launch's binning is not launched but described in *mine, and the
// binning `prev` describes (an earlier launch's, NULL for none) runs as extra workgroups of this launch.  lavt_attn_dtable_run launches a job alone.
extern "C" int lavt_window_attn_bwd_chained(int dtype, const void* qkv, int bias_ld, const int8_t* region, int nw_img, const void* out, const void* dout,
                                            const float* lse, void* dqkv, const float* table, float* ws, int64_t ws_floats, float* parts, int wd, int wh, int ww,
                                            int nwin, int N, int heads, int head_dim, float scale, const lavt_dtable_job_t* prev, lavt_dtable_job_t* mine,
                                            void* stream) {
    LAVT_CHECK_ARG(head_dim == HD, "lavt_window_attn_bwd_chained: head_dim %d != 32", head_dim);
    LAVT_CHECK_ARG(qkv && out && dout && lse && dqkv && table && parts && mine && nwin > 0 && N > 0 && heads > 0, "lavt_window_attn_bwd_chained: bad arguments");
    LAVT_CHECK_ARG(wd > 0 && wh > 0 && ww > 0 && N <= wd * wh * ww, "lavt_window_attn_bwd_chained: window shape (wd, wh, ww) must cover N tokens");
    LAVT_CHECK_ARG(use_mfma(dtype, N, bias_ld) && ws && ws_floats >= lavt_window_attn_bwd_ws_mfma(nwin, N, heads, bias_ld, wd, wh, ww),
                   "lavt_window_attn_bwd_chained: bf16 MFMA path only (N <= 400, scratch of lavt_window_attn_bwd_ws floats)");
    return lavt_window_attn_bwd_mfma(qkv, table, region, nw_img, out, dout, lse, dqkv, nullptr, bias_ld, ws, parts, wd, wh, ww, nwin, N, heads, scale,
                                     reinterpret_cast<hipStream_t>(stream), prev, mine);
}
extern "C" int lavt_attn_dtable_run(const lavt_dtable_job_t* job, void* stream) {
    LAVT_CHECK_ARG(job && job->slab && job->part && job->gx > 0 && job->gz > 0 && job->heads > 0, "lavt_attn_dtable_run: bad job");
    return lavt_attn_dtable_run_mfma(job, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int64_t lavt_window_attn_bwd_ws(int dtype, int nwin, int N, int heads, int bias_ld, int wd, int wh, int ww) {
    return use_mfma(dtype, N, bias_ld) ? lavt_window_attn_bwd_ws_mfma(nwin, N, heads, bias_ld, wd, wh, ww) : 0;
}
// deferred table gradient: number of per-workgroup histograms ([pieces][heads][R] floats in `parts`); 0 when this (dtype, N) has no deferred form
extern "C" int lavt_window_attn_bwd_pieces(int dtype, int nwin, int N, int heads, int bias_ld) {
    return use_mfma(dtype, N, bias_ld) ? lavt_window_attn_bwd_pieces_mfma(nwin, N, heads) : 0;
}
extern "C" int lavt_attn_dtable_finish_multi(const int64_t* desc, int n, int max_R, int max_heads, void* stream) {
    LAVT_CHECK_ARG(desc && n > 0 && max_R > 0 && max_heads > 0, "lavt_attn_dtable_finish_multi: bad arguments");
    return lavt_attn_dtable_finish_multi_impl(desc, n, max_R, max_heads, 0, reinterpret_cast<hipStream_t>(stream));
}
/* the same with the launch sized for the (layer, head) pairs that exist: total_heads = sum over the layers of their head counts (n <= 64) */
extern "C" int lavt_attn_dtable_finish_multi_compact(const int64_t* desc, int n, int max_R, int total_heads, void* stream) {
    LAVT_CHECK_ARG(desc && n > 0 && n <= 64 && max_R > 0 && total_heads > 0, "lavt_attn_dtable_finish_multi_compact: bad arguments (n <= 64)");
    return lavt_attn_dtable_finish_multi_impl(desc, n, max_R, 0, total_heads, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int lavt_attn_uses_table(int dtype, int N) { return use_mfma(dtype, N, N <= 64 ? 64 : N <= 160 ? 160 : 416) ? 1 : 0; }

extern "C" int lavt_relpos_expand(const float* table, float* dense, int wd, int wh, int ww, int N, int heads, int ld, void* stream) {
    LAVT_CHECK_ARG(table && dense && wd > 0 && wh > 0 && ww > 0 && N > 0 && N <= wd * wh * ww && heads > 0 && ld >= N, "lavt_relpos_expand: bad arguments");
    const int64_t total = (int64_t)heads * N * ld;
    int blocks = cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(relpos_expand_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), table, dense, wd, wh, ww, N, heads, ld);
    LAVT_CHECK_LAUNCH("lavt_relpos_expand");
    return LAVT_OK;
}
extern "C" int lavt_relpos_reduce(const float* ddense, float* dtable, int wd, int wh, int ww, int N, int heads, int ld, void* stream) {
    LAVT_CHECK_ARG(ddense && dtable && wd > 0 && wh > 0 && ww > 0 && N > 0 && N <= wd * wh * ww && heads > 0 && ld >= N, "lavt_relpos_reduce: bad arguments");
    const int total = (2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1) * heads;
    hipLaunchKernelGGL(relpos_reduce_kernel, dim3(cdiv(total, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), ddense, dtable, wd, wh, ww, N, heads, ld);
    LAVT_CHECK_LAUNCH("lavt_relpos_reduce");
    return LAVT_OK;
}
extern "C" int lavt_attn_softmax_fwd(int dtype, const void* s, const float* bias, int bias_ld, const int8_t* region, int nw_img, void* p,
                                     int64_t rows, int rpw, int N, int ld, int heads, void* stream) {
    LAVT_CHECK_ARG(s && bias && p && rows > 0 && N > 0 && rpw >= N && ld >= N && bias_ld >= N && heads >= 1 && (!region || nw_img > 0), "lavt_attn_softmax_fwd: bad arguments");
    LAVT_CHECK_ARG(ld % (dtype == LAVT_F32 ? 4 : 8) == 0 && ld / (dtype == LAVT_F32 ? 4 : 8) <= SM_KMAX * 64, "lavt_attn_softmax_fwd: ld=%d must be a multiple of the 16-byte chunk and at most %d chunks", ld, SM_KMAX * 64);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int blocks = cdiv(rows, 4);
    if (blocks > 8192) blocks = 8192;
    if (dtype == LAVT_F32) hipLaunchKernelGGL(attn_softmax_fwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)s, bias, bias_ld, region, nw_img, (float*)p, rows, rpw, N, ld, heads);
    else if (dtype == LAVT_BF16) hipLaunchKernelGGL(attn_softmax_fwd_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)s, bias, bias_ld, region, nw_img, (bf16*)p, rows, rpw, N, ld, heads);
    else { lavt_set_error("lavt_attn_softmax_fwd: bad dtype %d", dtype); return LAVT_ERR_INVALID; }
    LAVT_CHECK_LAUNCH("lavt_attn_softmax_fwd");
    return LAVT_OK;
}
// Dense bias gradient of the composed path: out[h][i][j] (fp32, [heads][N][ld]) = sum over windows of ds[w][h][i][j] (ds [nwin][heads][rpw][ld],
// rows i >= N are padding).  One thread per 16-byte chunk of an output row, windows walked 4 at a time (independent loads in flight).
template <typename T>
__global__ __launch_bounds__(256) void attn_dbias_sum_kernel(const T* __restrict__ ds, float* __restrict__ out, int nwin, int heads, int N, int rpw, int ld) {
    constexpr int EPC = Chunk<T>::N;
    const int nch = ld / EPC;
    const int64_t total = (int64_t)heads * N * nch;
    const int64_t wstride = (int64_t)heads * rpw * ld;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % nch);
        const int64_t hi = t / nch;
        const int i = (int)(hi % N), h = (int)(hi / N);
        const T* src = ds + ((int64_t)h * rpw + i) * ld + c * EPC;
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        int w = 0;
        for (; w + 3 < nwin; w += 4) {
            float f[4][EPC];
#pragma unroll
            for (int u = 0; u < 4; ++u) chunk_to_f<T>(*reinterpret_cast<const uint4*>(src + (w + u) * wstride), f[u]);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] += (f[0][e] + f[1][e]) + (f[2][e] + f[3][e]);
        }
        for (; w < nwin; ++w) {
            float f[EPC];
            chunk_to_f<T>(*reinterpret_cast<const uint4*>(src + w * wstride), f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] += f[e];
        }
        float* dst = out + ((int64_t)h * N + i) * ld + c * EPC;
#pragma unroll
        for (int e = 0; e < EPC; ++e) dst[e] = acc[e];
    }
}

extern "C" int lavt_attn_dbias_sum(int dtype, const void* ds, float* out, int nwin, int heads, int N, int rpw, int ld, void* stream) {
    LAVT_CHECK_ARG(ds && out && nwin > 0 && heads > 0 && N > 0 && rpw >= N && ld >= N && ld % (dtype == LAVT_F32 ? 4 : 8) == 0, "lavt_attn_dbias_sum: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)heads * N * (ld / (dtype == LAVT_F32 ? 4 : 8));
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (dtype == LAVT_F32) hipLaunchKernelGGL(attn_dbias_sum_kernel<float>, dim3((int)blocks), dim3(256), 0, st, (const float*)ds, out, nwin, heads, N, rpw, ld);
    else if (dtype == LAVT_BF16) hipLaunchKernelGGL(attn_dbias_sum_kernel<bf16>, dim3((int)blocks), dim3(256), 0, st, (const bf16*)ds, out, nwin, heads, N, rpw, ld);
    else { lavt_set_error("lavt_attn_dbias_sum: bad dtype %d", dtype); return LAVT_ERR_INVALID; }
    LAVT_CHECK_LAUNCH("lavt_attn_dbias_sum");
    return LAVT_OK;
}
extern "C" int lavt_attn_softmax_bwd(int dtype, const void* p, void* dp, int64_t rows, int N, int ld, void* stream) {
    LAVT_CHECK_ARG(p && dp && rows > 0 && N > 0 && ld >= N, "lavt_attn_softmax_bwd: bad arguments");
    LAVT_CHECK_ARG(ld % (dtype == LAVT_F32 ? 4 : 8) == 0 && ld / (dtype == LAVT_F32 ? 4 : 8) <= SM_KMAX * 64, "lavt_attn_softmax_bwd: ld=%d must be a multiple of the 16-byte chunk and at most %d chunks", ld, SM_KMAX * 64);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int blocks = cdiv(rows, 4);
    if (blocks > 8192) blocks = 8192;
    if (dtype == LAVT_F32) hipLaunchKernelGGL(attn_softmax_bwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)p, (float*)dp, rows, N, ld);
    else if (dtype == LAVT_BF16) hipLaunchKernelGGL(attn_softmax_bwd_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)p, (bf16*)dp, rows, N, ld);
    else { lavt_set_error("lavt_attn_softmax_bwd: bad dtype %d", dtype); return LAVT_ERR_INVALID; }
    LAVT_CHECK_LAUNCH("lavt_attn_softmax_bwd");
    return LAVT_OK;
}
