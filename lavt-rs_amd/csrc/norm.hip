// LayerNorm (optionally fused with the PatchMerging 2x2 gather) and the per-channel statistics /
// normalisation kernels behind InstanceNorm1d (PWAM) and BatchNorm2d(+ReLU) (decoder).  All HBM-bound:
// 16-byte accesses, one wave per LayerNorm row (row kept in registers, fp32 math), wave-shuffle reductions.
#include <string.h>
#include "common.h"
#include "fp8_pack.h"
#include "ln_bwd_body.h"
#include "dtable_body.h"

namespace {

constexpr int LN_MAXE = 32;   // floats of one row held per lane: covers C <= 2048

// EPC consecutive fp32 constants (16-byte aligned: EPC is 4 or 8 and the column offset a multiple of it) as float4 loads
template <int EPC> __device__ __forceinline__ void ldc(const float* p, float* out) {
#pragma unroll
    for (int q = 0; q < EPC / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
        out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w;
    }
}

// ---------------------------------------------------------------------------------------------- LayerNorm
template <typename T, int LPR>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const int32_t* __restrict__ gather,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                            int rows, int C, float eps) {
    constexpr int EPC = Chunk<T>::N, MAXC = LPR == 64 ? LN_MAXE / EPC : 1, RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, lir = lane % LPR;
    const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool live = row < rows;
    const int nchunk = C / EPC;
    float v[MAXC * EPC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lir + LPR * c;
        float f[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) f[e] = 0.f;
        if (live && ch < nchunk) {
            bool ok;
            const T* src = ln_src<T>(x, gather, row, C, ch * EPC, ok);
            if (ok) chunk_to_f<T>(*reinterpret_cast<const uint4*>(src), f);
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e) { v[c * EPC + e] = f[e]; s += f[e]; }
    }
    const float mu = group_sum<LPR>(s) / C;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (lir + LPR * c < nchunk)
#pragma unroll
            for (int e = 0; e < EPC; ++e) { const float d = v[c * EPC + e] - mu; q += d * d; }
    const float rs = rsqrtf(group_sum<LPR>(q) / C + eps);
    if (!live) return;
    if (lir == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lir + LPR * c;
        if (ch < nchunk) {
            float f[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) f[e] = (v[c * EPC + e] - mu) * rs * gamma[ch * EPC + e] + beta[ch * EPC + e];
            *reinterpret_cast<uint4*>(y + row * C + ch * EPC) = f_to_chunk<T>(f);
        }
    }
}

template <typename T, int LPR, int CPL, int WAVES, int FLAGS = 3>
__global__ __launch_bounds__(WAVES * 64) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const int32_t* __restrict__ gather, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            T* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ partials, const T* __restrict__ dres, int rows, int C,
                                                            T* __restrict__ xn_out_, const float* __restrict__ beta) {
    __shared__ float red[(WAVES - 1) * (LPR * CPL * Chunk<T>::N)];
    layernorm_bwd_body<T, LPR, CPL, WAVES, FLAGS>(dy, x, gather, gamma, mean, rstd, dx, dgamma, dbeta, partials, dres, rows, C, xn_out_, beta, blockIdx.x, gridDim.x,
                                                  threadIdx.x, red);
}

// The same launch carrying the table-gradient binning of an attention backward as RIDER workgroups (round 6): blocks >= ln_blocks run dtable_block on the dS
// slabs of the attention launch issued a few launches earlier (still in the Infinity Cache).  As riders of the NEXT attention-backward launch (round 4) the
// binning blocks took that launch's second workgroup slot per CU at its 55 KB of LDS -- the launch grew from 16.7-20.7 us to 24.5; a LayerNorm backward is
// 256-thread workgroups without LDS pressure, eight to a CU, and waits on HBM for most of its 8 us.
struct DtableRider {
    const bf16* slab; float* part;
    int slab_ld, wd, wh, ww, nwin, N, heads, rows_per_block, win_per_group, gx, gz;
};
template <typename T, int LPR, int CPL, int FLAGS>
__global__ __launch_bounds__(256) void layernorm_bwd_dtable_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dx,
                                                                   float* __restrict__ partials, const T* __restrict__ dres, int rows, int C,
                                                                   T* __restrict__ xn_out_, const float* __restrict__ beta, const int ln_blocks, const DtableRider job) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    __shared__ float red[3 * (LPR * CPL * Chunk<T>::N)];
    if ((int)blockIdx.x >= ln_blocks) {
        const int u = (int)blockIdx.x - ln_blocks;
        const int bx = u % job.gx, hh = (u / job.gx) % job.heads, bz = u / (job.gx * job.heads);
        dtable_block(job.slab, job.part, job.slab_ld, job.wd, job.wh, job.ww, job.nwin, job.N, job.heads, job.rows_per_block, job.win_per_group, bx, hh, bz, job.gx,
                     threadIdx.x, dsm);
        return;
    }
    layernorm_bwd_body<T, LPR, CPL, 4, FLAGS>(dy, x, nullptr, gamma, mean, rstd, dx, nullptr, nullptr, partials, dres, rows, C, xn_out_, beta, blockIdx.x, ln_blocks,
                                              threadIdx.x, red);
}

// Second stage of the two-stage reductions: partials [nblk][W] (W = groups*2*C: per group first the C "sum-1" values, then the C
// "sum-2" values) -> o1[g*C+c] += sum_k partials[k][(2g)*C+c], o2[g*C+c] += sum_k partials[k][(2g+1)*C+c].
// 32 columns x 8 row-slices per workgroup: coalesced 128-B reads, 8-way parallel walk over nblk, LDS combine.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partials, int nblk, int W, int C,
                                                              float* __restrict__ o1, float* __restrict__ o2) {
    // blockIdx.y splits the nblk partial rows when there are many of them (the pieces then meet in the output through atomics)
    __shared__ float red[8][33];
    const int col = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int w = blockIdx.x * 32 + col;
    const int per = (nblk + gridDim.y - 1) / gridDim.y, k0 = blockIdx.y * per, k1 = min(nblk, k0 + per);
    float a = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (w < W) {
        // (four rows in flight per thread and up to 16 row pieces: one row at a time in 4 pieces, the 900 x 1024 partial sums of a 512-channel BatchNorm
        // backward -- 3.7 MB -- took 10.6 us)
        int k = k0 + slice;
        for (; k + 24 < k1; k += 32) {
            a += partials[(int64_t)k * W + w]; a1 += partials[(int64_t)(k + 8) * W + w];
            a2 += partials[(int64_t)(k + 16) * W + w]; a3 += partials[(int64_t)(k + 24) * W + w];
        }
        for (; k < k1; k += 8) a += partials[(int64_t)k * W + w];
    }
    red[slice][col] = (a + a1) + (a2 + a3);
    __syncthreads();
    if (slice == 0 && w < W && k0 < k1) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][col];
        const int g2 = w / C, c = w - g2 * C;          // g2 = 2*g + which
        float* dst = (g2 & 1) ? o2 : o1;
        if (gridDim.y > 1) atomicAdd(dst + (int64_t)(g2 >> 1) * C + c, t);
        else dst[(int64_t)(g2 >> 1) * C + c] += t;
    }
}
// Several partial sets in one launch (deferred LayerNorm weight / bias gradients): grid = (column blocks of 32, row slices, sets).
// desc[s] = {partials, nblk, C, o1, o2}; partials [nblk][2][C]; the row slices of a set meet through atomics.
// COMPACT: blockIdx.x runs over the column blocks of ALL sets (the host passes their total): the (128 column blocks, 8 slices, sets) grid of the first
// version started 57 344 workgroups for Swin-B's 56 sets, two thirds of which found no columns (34 us for ~80 MB of partial sums).
// (round 5) a workgroup covers 128 columns (a float4 per lane, 32 lanes) x 8 row slices, four rows in flight per thread: as 32 scalar columns per
// workgroup the 75 MB of Swin-B's partial sums were read at 2.2 TB/s (33.6 us, the last launch of every backward pass).
constexpr int RPM_COLS = 128;
template <bool COMPACT>
__global__ __launch_bounds__(256) void reduce_partials_multi_kernel(const int64_t* __restrict__ desc, int nsets) {
    int set = blockIdx.z, cb = blockIdx.x;
    if constexpr (COMPACT) {
        // (set, column block) of this workgroup: lane s of every wave reads the width of set s (nsets <= 64), an inclusive prefix sum over the lanes, and
        // the first lane whose sum exceeds blockIdx.x names the set -- one round trip (as a serial scan over the descriptors every workgroup paid
        // up to 56 dependent loads: slower than the empty workgroups it was meant to save)
        const int ln = threadIdx.x & 63;
        int nb = ln < nsets ? (2 * (int)desc[(int64_t)ln * 5 + 2] + RPM_COLS - 1) / RPM_COLS : 0;
        int pre = nb;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(pre, o, 64);
            if (ln >= o) pre += up;
        }
        const unsigned long long hit = __ballot(pre > (int)blockIdx.x);
        set = hit ? __ffsll((long long)hit) - 1 : nsets - 1;
        const int before = __shfl(pre - nb, set, 64);
        cb = (int)blockIdx.x - before;
    }
    const int64_t* d = desc + (int64_t)set * 5;
    const float* partials = reinterpret_cast<const float*>(d[0]);
    const int nblk = (int)d[1], C = (int)d[2], W = 2 * C;
    float* o1 = reinterpret_cast<float*>(d[3]);
    float* o2 = reinterpret_cast<float*>(d[4]);
    if (cb * RPM_COLS >= W) return;
    __shared__ float4 red[8][33];
    const int col = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int w = cb * RPM_COLS + 4 * col;
    const int per = (nblk + gridDim.y - 1) / gridDim.y, k0 = blockIdx.y * per, k1 = min(nblk, k0 + per);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a, e = a;
    if (w + 3 < W && (W & 3) == 0) {
        int k = k0 + slice;
        for (; k + 24 < k1; k += 32) {
            const float4 v0 = *reinterpret_cast<const float4*>(partials + (int64_t)k * W + w), v1 = *reinterpret_cast<const float4*>(partials + (int64_t)(k + 8) * W + w);
            const float4 v2 = *reinterpret_cast<const float4*>(partials + (int64_t)(k + 16) * W + w), v3 = *reinterpret_cast<const float4*>(partials + (int64_t)(k + 24) * W + w);
            a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w; b.x += v1.x; b.y += v1.y; b.z += v1.z; b.w += v1.w;
            c.x += v2.x; c.y += v2.y; c.z += v2.z; c.w += v2.w; e.x += v3.x; e.y += v3.y; e.z += v3.z; e.w += v3.w;
        }
        for (; k < k1; k += 8) {
            const float4 v0 = *reinterpret_cast<const float4*>(partials + (int64_t)k * W + w);
            a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
        }
    } else if (w < W) {          // (a width that is not a multiple of 4: the classifier head's 2-column bias set)
        for (int k = k0 + slice; k < k1; k += 8) {
            a.x += partials[(int64_t)k * W + w];
            if (w + 1 < W) a.y += partials[(int64_t)k * W + w + 1];
            if (w + 2 < W) a.z += partials[(int64_t)k * W + w + 2];
            if (w + 3 < W) a.w += partials[(int64_t)k * W + w + 3];
        }
    }
    red[slice][col] = make_float4((a.x + b.x) + (c.x + e.x), (a.y + b.y) + (c.y + e.y), (a.z + b.z) + (c.z + e.z), (a.w + b.w) + (c.w + e.w));
    __syncthreads();
    if (slice == 0 && w < W && k0 < k1) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float4 v = red[k][col]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (w + i < W) atomicAdd((w + i >= C ? o2 : o1) + (w + i >= C ? w + i - C : w + i), tv[i]);
    }
}
static inline dim3 reduce_partials_grid(int nblk, int W) { return dim3(cdiv(W, 32), nblk >= 768 ? 16 : nblk >= 384 ? 8 : nblk >= 128 ? 4 : (nblk >= 48 ? 2 : 1)); }

// ---------------------------------------------------------------------------------------------- column statistics
// grid: (row blocks, groups).  Thread t owns chunk column t % cpr and walks rows t / cpr, + 256/cpr, ...
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void colstats_kernel(const T* __restrict__ a, const T* __restrict__ xin, const T* __restrict__ yout,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const T* __restrict__ mul, int relu, float* __restrict__ o1,
                                                       float* __restrict__ o2, float* __restrict__ partials, int rows, int C, int rows_per_block, int zero_out,
                                                       const float* __restrict__ gamma = nullptr, const float* __restrict__ beta = nullptr) {
    // !BWD: a = x; o1 += sum (x-K), o2 += sum (x-K)^2, K = x[g][0][:].     BWD: a = dy; g = dy (*mul) (masked y>0); o1 += sum g, o2 += sum g*xhat
    constexpr int EPC = Chunk<T>::N;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = reinterpret_cast<float*>(smem_raw);       // [256][2*EPC]
    const int cpr = C / EPC;
    const int g = blockIdx.y;
    const int r_begin = blockIdx.x * rows_per_block, r_end = min(rows, r_begin + rows_per_block);
    float s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    // two-stage form: the outputs are produced by the reduce kernel that FOLLOWS this launch (it adds: several row slices may meet there),
    // so the first row block clears them here and the caller needs no zero-fill launch of its own
    if (partials && zero_out && blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += 256) { o1[(int64_t)g * C + c] = 0.f; o2[(int64_t)g * C + c] = 0.f; }
    // chunk columns are walked in slabs of 256 when cpr > 256 is impossible here (C <= 2048 -> cpr <= 512)
    for (int cbase = 0; cbase < cpr; cbase += 256) {
        const int span = min(256, cpr - cbase);
        const int tc = threadIdx.x % span, tr = threadIdx.x / span, rstep = 256 / span;
        if (tr < rstep) {
            const int col = (cbase + tc) * EPC;
            float mu[EPC], rs[EPC], sc[EPC], sh[EPC];
            // BWD with ReLU behind an affine normalisation and no multiplier: the mask y > 0 is recomputed from x with the forward's own expression
            // (norm_apply_kernel: x * sc + sh) instead of reading the output map -- one of three input streams less
            const bool remask = BWD && relu && !mul && gamma;
            if (BWD) {
                ldc<EPC>(mean + (int64_t)g * C + col, mu);
                ldc<EPC>(rstd + (int64_t)g * C + col, rs);
                if (remask) {
                    float ga[EPC], be[EPC];
                    ldc<EPC>(gamma + col, ga); ldc<EPC>(beta + col, be);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) { sc[e] = rs[e] * ga[e]; sh[e] = be[e] - mu[e] * rs[e] * ga[e]; }
                }
            } else {
                // shifted sums: accumulate (x - K) and (x - K)^2 with K = the group's first row, so that the variance does not
                // come out of E[x^2] - E[x]^2 of large-mean data (colstats_center_kernel undoes the shift)
                chunk_to_f<T>(*reinterpret_cast<const uint4*>(a + (int64_t)g * rows * C + col), mu);
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll 4
            for (int r = r_begin + tr; r < r_end; r += rstep) {
                const int64_t off = ((int64_t)g * rows + r) * C + col;
                float f[EPC];
                chunk_to_f<T>(*reinterpret_cast<const uint4*>(a + off), f);
                if (!BWD) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) { const float d = f[e] - mu[e]; s1[e] += d; s2[e] += d * d; }
                } else {
                    float fx[EPC];
                    chunk_to_f<T>(*reinterpret_cast<const uint4*>(xin + off), fx);
                    if (mul) {
                        float fm[EPC];
                        chunk_to_f<T>(*reinterpret_cast<const uint4*>(mul + off), fm);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) f[e] *= fm[e];
                    }
                    if (remask) {
#pragma unroll
                        for (int e = 0; e < EPC; ++e) if (!(fx[e] * sc[e] + sh[e] > 0.f)) f[e] = 0.f;
                    } else if (relu) {
                        float fy[EPC];
                        chunk_to_f<T>(*reinterpret_cast<const uint4*>(yout + off), fy);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) if (!(fy[e] > 0.f)) f[e] = 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < EPC; ++e) { s1[e] += f[e]; s2[e] += f[e] * (fx[e] - mu[e]) * rs[e]; }
                }
            }
        }
        // reduce the rstep row-lanes that share a chunk column
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPC; ++e) { red[threadIdx.x * 2 * EPC + e] = s1[e]; red[threadIdx.x * 2 * EPC + EPC + e] = s2[e]; }
        __syncthreads();
        if (threadIdx.x < span) {
            float t1[EPC], t2[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) { t1[e] = 0.f; t2[e] = 0.f; }
            for (int k = 0; k < rstep; ++k)
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    t1[e] += red[(k * span + threadIdx.x) * 2 * EPC + e];
                    t2[e] += red[(k * span + threadIdx.x) * 2 * EPC + EPC + e];
                }
            const int col = (cbase + threadIdx.x) * EPC;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                if (partials) {
                    float* pp = partials + (((int64_t)blockIdx.x * gridDim.y + g) * 2) * C + col + e;
                    pp[0] = t1[e]; pp[C] = t2[e];
                } else { atomicAdd(o1 + (int64_t)g * C + col + e, t1[e]); atomicAdd(o2 + (int64_t)g * C + col + e, t2[e]); }
            }
        }
        __syncthreads();
    }
}

// shifted sums (s1 = sum (x-K), s2 = sum (x-K)^2 over the group's `rows` rows) -> sum x and the centred second moment M2 = sum (x - mean)^2
template <typename T>
__global__ void colstats_center_kernel(const T* __restrict__ x, float* __restrict__ o1, float* __restrict__ o2, int rows, int C, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int g = i / C, c = i - g * C;
    const float k = to_f<T>(x[(int64_t)g * rows * C + c]);
    const float s1 = o1[i], s2 = o2[i];
    o1[i] = s1 + (float)rows * k;
    o2[i] = fmaxf(s2 - s1 * s1 / (float)rows, 0.f);
}

// Second stage of the forward statistics when nothing sits between the sums and their use (InstanceNorm; BatchNorm on one rank): the partial
// rows are summed, re-centred about the pivot row and turned into mean / rstd (and the running estimates) in ONE kernel instead of
// reduce_partials + colstats_center + stats_finalize.  Workgroup = 16 columns x {shifted sum, shifted square sum} x 8 row slices.
template <typename T>
__global__ __launch_bounds__(1024) void colstats_finish_kernel(const float* __restrict__ partials, int nblk, int groups, int C, const T* __restrict__ x,
                                                               int rows, float eps, float* __restrict__ mean, float* __restrict__ rstd,
                                                               float* __restrict__ sum_out, float* __restrict__ m2_out,
                                                               float* __restrict__ running_mean, float* __restrict__ running_var, float momentum) {
    // 32 row slices (was 8: 112 dependent-latency loads per thread on the 900 row blocks of the decoder maps, 22 us per launch)
    __shared__ float red[32][32];
    const int c16 = threadIdx.x & 15, which = (threadIdx.x >> 4) & 1, slice = threadIdx.x >> 5;
    const int c = blockIdx.x * 16 + c16, g = blockIdx.y;
    const int64_t W = (int64_t)groups * 2 * C;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < C) {
        const float* q = partials + ((int64_t)g * 2 + which) * C + c;
        int k = slice;
        for (; k + 96 < nblk; k += 128) { a0 += q[(int64_t)k * W]; a1 += q[(int64_t)(k + 32) * W]; a2 += q[(int64_t)(k + 64) * W]; a3 += q[(int64_t)(k + 96) * W]; }
        for (; k < nblk; k += 32) a0 += q[(int64_t)k * W];
    }
    red[slice][threadIdx.x & 31] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (slice == 0 && which == 0 && c < C) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) { s1 += red[k][c16]; s2 += red[k][16 + c16]; }
        const float kpiv = to_f<T>(x[(int64_t)g * rows * C + c]);
        const float n = (float)rows;
        const float sum = s1 + n * kpiv, m2 = fmaxf(s2 - s1 * s1 / n, 0.f);
        const float mu = sum / n, var = m2 / n;
        const int64_t i = (int64_t)g * C + c;
        if (mean) { mean[i] = mu; rstd[i] = rsqrtf(var + eps); }
        if (sum_out) { sum_out[i] = sum; m2_out[i] = m2; }
        if (running_mean) running_mean[i] = (1.f - momentum) * running_mean[i] + momentum * mu;
        if (running_var) running_var[i] = (1.f - momentum) * running_var[i] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
    }
}

// Second stage for block statistics that a GEMM epilogue stored (lavt_gemm_nt_t.colstats: per block of rows the column sums and the second moments
// about the block's own mean): parallel-variance combination in ONE pass over the (small) table, about a pivot p = the mean of block 0:
//   M2 = sum_b [ M2_b + n_b (mean_b - p)^2 ] - N (mean - p)^2      (the correction term is ~1/rows_per_block of M2: no cancellation to speak of)
// Workgroup = 32 columns x 32 block slices.
__global__ __launch_bounds__(1024) void colstats_finish_blocks_kernel(const float* __restrict__ partials, int nblk, int rpb, int rows, int C, float eps,
                                                                      float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ sum_out,
                                                                      float* __restrict__ m2_out, float* __restrict__ running_mean,
                                                                      float* __restrict__ running_var, float momentum) {
    __shared__ float red[2][32][33];
    const int c32 = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + c32;
    const int64_t W = 2 * (int64_t)C;
    float a = 0.f, b = 0.f, piv = 0.f;
    if (c < C) {
        piv = partials[c] / (float)min(rows, rpb);
        for (int k = slice; k < nblk; k += 32) {
            const int nb = min(max(rows - k * rpb, 0), rpb);
            if (nb > 0) {
                const float sb = partials[(int64_t)k * W + c], qb = partials[(int64_t)k * W + C + c];
                const float d = sb / (float)nb - piv;
                a += sb;
                b += qb + (float)nb * d * d;
            }
        }
    }
    red[0][slice][c32] = a;
    red[1][slice][c32] = b;
    __syncthreads();
    if (slice == 0 && c < C) {
        float sum = 0.f, q = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) { sum += red[0][k][c32]; q += red[1][k][c32]; }
        const float n = (float)rows, mu = sum / n;
        const float m2 = fmaxf(q - n * (mu - piv) * (mu - piv), 0.f), var = m2 / n;
        if (mean) { mean[c] = mu; rstd[c] = rsqrtf(var + eps); }
        if (sum_out) { sum_out[c] = sum; m2_out[c] = m2; }
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
    }
}

__global__ void stats_finalize_kernel(const float* sum, const float* m2, float count, float eps, float* mean, float* rstd,
                                      float* running_mean, float* running_var, float momentum, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float mu = sum[i] / count;
    const float var = m2[i] / count;
    mean[i] = mu;
    rstd[i] = rsqrtf(var + eps);
    if (running_mean) running_mean[i] = (1.f - momentum) * running_mean[i] + momentum * mu;
    if (running_var) running_var[i] = (1.f - momentum) * running_var[i] + momentum * var * (count / fmaxf(count - 1.f, 1.f));
}

// Apply kernels: a thread owns one 16-byte channel chunk of a group and walks rows, so the per-channel constants (mean, rstd, gamma, beta,
// backward sums) sit in registers instead of being re-fetched for every element (the element-indexed form ran at ~0.9 TB/s on the
// 28 800 x 512 decoder maps).  grid = (row blocks, groups); columns beyond 256 chunks are walked in slabs.
// Q8 (bf16 only; BASELINE.json configs[4]): the kernel also writes the e4m3 twin of its output -- q = e4m3(bf16(y) * 448 / *amax_prev), the bytes
// lavt_fp8_quantize would produce from y -- and records |max| of y into *amax_cur (delayed scaling): the consuming convolution's quantiser launch disappears.
template <typename T, bool Q8 = false>
__global__ __launch_bounds__(256) void norm_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const T* __restrict__ mul, int relu, T* __restrict__ y, int rows, int C, int rows_per_block,
                                                         unsigned char* __restrict__ q = nullptr, const float* __restrict__ amax_prev = nullptr,
                                                         float* __restrict__ amax_cur = nullptr) {
    constexpr int EPC = Chunk<T>::N;
    float q_s = 1.f, q_m = 0.f;
    if constexpr (Q8) q_s = q8_scale(amax_prev);
    const int cpr = C / EPC, g = blockIdx.y;
    const int r_begin = blockIdx.x * rows_per_block, r_end = min(rows, r_begin + rows_per_block);
    for (int cbase = 0; cbase < cpr; cbase += 256) {
        const int span = min(256, cpr - cbase);
        const int tc = threadIdx.x % span, tr = threadIdx.x / span, rstep = 256 / span;
        if (tr >= rstep) continue;
        const int col = (cbase + tc) * EPC;
        // per-channel constants folded to one multiply-add: y = x * sc + sh  (the 4 x EPC scalar loads of the first version were most of the
        // kernel's instructions at 4 rows per thread: 2 TB/s on the 28 800 x 512 maps)
        float sc[EPC], sh[EPC];
        {
            float mu[EPC], rs[EPC], ga[EPC], be[EPC];
            ldc<EPC>(mean + (int64_t)g * C + col, mu);
            ldc<EPC>(rstd + (int64_t)g * C + col, rs);
            if (gamma) { ldc<EPC>(gamma + col, ga); ldc<EPC>(beta + col, be); }
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float gg = gamma ? ga[e] : 1.f, bb = gamma ? be[e] : 0.f;
                sc[e] = rs[e] * gg; sh[e] = bb - mu[e] * rs[e] * gg;
            }
        }
#pragma unroll 4
        for (int r = r_begin + tr; r < r_end; r += rstep) {
            const int64_t off = ((int64_t)g * rows + r) * C + col;
            float f[EPC], fm[EPC];
            chunk_to_f<T>(*reinterpret_cast<const uint4*>(x + off), f);
            if (mul) chunk_to_f<T>(*reinterpret_cast<const uint4*>(mul + off), fm);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                float v = f[e] * sc[e] + sh[e];
                if (mul) v *= fm[e];
                if (relu) v = fmaxf(v, 0.f);
                f[e] = v;
            }
            const uint4 out = f_to_chunk<T>(f);
            *reinterpret_cast<uint4*>(y + off) = out;
            if constexpr (Q8) {
                chunk_to_f<T>(out, f);
                *reinterpret_cast<uint2*>(q + off) = q8_chunk8(f, q_s, q_m);
            }
        }
    }
    if constexpr (Q8) q8_block_amax(q_m, amax_cur);
}

// AMAX (bf16; configs[4]): |max| of the stored dx is recorded into *amax (zeroed by the caller's bookkeeping once per step): the |max| pass in front of a
// CURRENT-scaling quantisation of this gradient disappears.
template <typename T, bool AMAX = false>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ yout,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const T* __restrict__ mul, int relu, const float* __restrict__ s1,
                                                             const float* __restrict__ s2, float inv_count, T* __restrict__ dx,
                                                             T* __restrict__ dmul, int rows, int C, int rows_per_block, float* __restrict__ amax = nullptr) {
    constexpr int EPC = Chunk<T>::N;
    float q_m = 0.f;
    const int cpr = C / EPC, g = blockIdx.y;
    const int r_begin = blockIdx.x * rows_per_block, r_end = min(rows, r_begin + rows_per_block);
    for (int cbase = 0; cbase < cpr; cbase += 256) {
        const int span = min(256, cpr - cbase);
        const int tc = threadIdx.x % span, tr = threadIdx.x / span, rstep = 256 / span;
        if (tr >= rstep) continue;
        const int col = (cbase + tc) * EPC;
        float mu[EPC], rs[EPC], ga[EPC], be[EPC], a1[EPC], a2[EPC];
        ldc<EPC>(mean + (int64_t)g * C + col, mu);
        ldc<EPC>(rstd + (int64_t)g * C + col, rs);
        ldc<EPC>(s1 + (int64_t)g * C + col, a1);
        ldc<EPC>(s2 + (int64_t)g * C + col, a2);
        if (gamma) { ldc<EPC>(gamma + col, ga); ldc<EPC>(beta + col, be); }
        const bool remask = relu && !mul && gamma;          // the ReLU mask from x with the forward's expression (x * sc + sh > 0): the output map is not read
        float sc[EPC], sh[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            a1[e] *= inv_count; a2[e] *= inv_count;
            if (!gamma) { ga[e] = 1.f; be[e] = 0.f; }
            sc[e] = rs[e] * ga[e]; sh[e] = be[e] - mu[e] * rs[e] * ga[e];
        }
#pragma unroll 2
        for (int r = r_begin + tr; r < r_end; r += rstep) {
            const int64_t off = ((int64_t)g * rows + r) * C + col;
            float fg[EPC], fx[EPC], fm[EPC], fy[EPC], fdm[EPC];
            chunk_to_f<T>(*reinterpret_cast<const uint4*>(dy + off), fg);
            chunk_to_f<T>(*reinterpret_cast<const uint4*>(x + off), fx);
            if (mul) chunk_to_f<T>(*reinterpret_cast<const uint4*>(mul + off), fm);
            if (relu && !remask) chunk_to_f<T>(*reinterpret_cast<const uint4*>(yout + off), fy);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float xh = (fx[e] - mu[e]) * rs[e];
                float gg = fg[e];
                if (mul) { fdm[e] = gg * (xh * ga[e] + be[e]); gg *= fm[e]; }
                if (remask) { if (!(fx[e] * sc[e] + sh[e] > 0.f)) gg = 0.f; }
                else if (relu && !(fy[e] > 0.f)) gg = 0.f;
                fg[e] = ga[e] * rs[e] * (gg - a1[e] - xh * a2[e]);
            }
            const uint4 out = f_to_chunk<T>(fg);
            *reinterpret_cast<uint4*>(dx + off) = out;
            if constexpr (AMAX) {
                chunk_to_f<T>(out, fg);
#pragma unroll
                for (int e = 0; e < EPC; ++e) q_m = fmaxf(q_m, fabsf(fg[e]));
            }
            if (mul && dmul) *reinterpret_cast<uint4*>(dmul + off) = f_to_chunk<T>(fdm);
        }
    }
    if constexpr (AMAX) q8_block_amax(q_m, amax);
}

// rows per workgroup for the apply kernels: ~2k workgroups in total, at least one full pass of the 256 threads over their rows
static int apply_rows_per_block(int rows, int groups, int C, int epc) {
    const int cpr = C / epc, span = cpr < 256 ? cpr : 256, rstep = 256 / span;
    int rpb = cdiv((long)rows * groups, 1024);       // ~1k workgroups, >= 8 rows per thread: the per-channel constants are amortised
    if (rpb < 8 * rstep) rpb = 8 * rstep;
    return rpb;
}

inline int ew_grid(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

}  // namespace

#define DISPATCH_T(dtype, NAME, ...)                                   \
    if (dtype == LAVT_F32) { using T = float; __VA_ARGS__; }           \
    else if (dtype == LAVT_BF16) { using T = bf16; __VA_ARGS__; }      \
    else { lavt_set_error(NAME ": bad dtype %d", dtype); return LAVT_ERR_INVALID; }

extern "C" int lavt_layernorm_fwd(int dtype, const void* x, const int32_t* gather, const float* gamma, const float* beta, void* y,
                                  float* mean, float* rstd, int rows, int C, float eps, void* stream) {
    const int epc = dtype == LAVT_F32 ? 4 : 8;
    LAVT_CHECK_ARG(x && gamma && beta && y && mean && rstd && rows > 0, "lavt_layernorm_fwd: bad arguments");
    LAVT_CHECK_ARG(C > 0 && C <= 2048 && C % epc == 0 && (!gather || (C / 4) % epc == 0), "lavt_layernorm_fwd: unsupported C=%d", C);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nchunk = C / epc;
#define LN_FWD(LPR_) hipLaunchKernelGGL((layernorm_fwd_kernel<T, LPR_>), dim3(cdiv(rows, 4 * (64 / LPR_))), dim3(256), 0, st, (const T*)x, gather, gamma, beta, (T*)y, mean, rstd, rows, C, eps)
    DISPATCH_T(dtype, "lavt_layernorm_fwd", if (nchunk <= 16) LN_FWD(16); else if (nchunk <= 32) LN_FWD(32); else LN_FWD(64));
#undef LN_FWD
    LAVT_CHECK_LAUNCH("lavt_layernorm_fwd");
    return LAVT_OK;
}

static int ln_bwd_geometry(int dtype, int rows, int C, int* lpr_out, int* cpl_out, int* waves_out) {
    const int epc = dtype == LAVT_F32 ? 4 : 8;
    const int nchunk = C / epc;
    const int lpr = nchunk <= 16 ? 16 : nchunk <= 32 ? 32 : 64;
    const int cpl = cdiv(nchunk, lpr);                       // 1 (C <= 512 bf16 / 256 fp32), 2, or up to 4 / 8
    // One row pass per wave, 4-wave workgroups: the kernel needs 95-99 VGPRs, i.e. 5 waves per SIMD = five such workgroups per CU (1280 on the
    // chip), so up to 1280 blocks run as ONE round.  (8-wave workgroups: two per CU -- 576 blocks (Video-Swin stage 2, 4608 rows) or 900 (stage 0)
    // were two rounds, 16.9 us for 19 MB; 4 waves x 2 serial rows: the second row's loads started after the first row's reductions.)
    // More rows than one round holds: two (or more, grid-stride) serial rows per wave.
    int waves = 4;
    const int rpb = 4 * (64 / lpr);
    int blocks = cdiv(rows, rpb);
    if (blocks > 1280) blocks = cdiv(rows, 2 * rpb);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    const int old_geom = lavt_tuning().ln_bwd_waves;
    if (old_geom == 8 && cpl <= 2 && rows > 256) { waves = 8; blocks = cdiv(rows, 8 * (64 / lpr)); if (blocks > 1024) blocks = 1024; }
    *lpr_out = lpr; *cpl_out = cpl;
    if (waves_out) *waves_out = waves;
    return blocks;
}
extern "C" int lavt_layernorm_bwd_blocks(int dtype, int rows, int C) { int a, b; return ln_bwd_geometry(dtype, rows, C, &a, &b, nullptr); }
// geometry of the plain (no gather, no xn output) partial-sum form, for the grouped weight-gradient launch that runs it in rider workgroups
int lavt_ln_bwd_geometry(int dtype, int rows, int C, int* lpr, int* cpl, int* waves) { return ln_bwd_geometry(dtype, rows, C, lpr, cpl, waves); }
int lavt_layernorm_bwd_partial_impl(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx, float* ws,
                                    int64_t ws_floats, const void* dres, int rows, int C, void* stream);

static int layernorm_bwd_impl(int dtype, const void* dy, const void* x, const int32_t* gather, const float* gamma,
                              const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta, float* ws, int64_t ws_floats,
                              const void* dres, int rows, int C, void* stream, bool partial_only, void* xn_out = nullptr, const float* beta = nullptr) {
    const int epc = dtype == LAVT_F32 ? 4 : 8;
    LAVT_CHECK_ARG(dy && x && gamma && mean && rstd && dx && (partial_only || (dgamma && dbeta)) && rows > 0, "lavt_layernorm_bwd: bad arguments");
    LAVT_CHECK_ARG(C > 0 && C <= 2048 && C % epc == 0 && (!gather || (C / 4) % epc == 0), "lavt_layernorm_bwd: unsupported C=%d", C);
    LAVT_CHECK_ARG(!(gather && dres), "lavt_layernorm_bwd: dres is not supported together with gather");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int lpr, cpl, waves;
    const int blocks = ln_bwd_geometry(dtype, rows, C, &lpr, &cpl, &waves);
    float* partials = (ws && ws_floats >= (int64_t)blocks * 2 * C) ? ws : nullptr;
    LAVT_CHECK_ARG(!partial_only || partials, "lavt_layernorm_bwd_partial: scratch of %ld floats needed", (long)blocks * 2 * C);
#define LN_BWD_F(LPR_, CPL_, WV_, F_) hipLaunchKernelGGL((layernorm_bwd_kernel<T, LPR_, CPL_, WV_, F_>), dim3(blocks), dim3(WV_ * 64), 0, st, (const T*)dy, (const T*)x, gather, gamma, mean, rstd, (T*)dx, dgamma, dbeta, partials, (const T*)dres, rows, C, (T*)xn_out, beta)
#define LN_BWD(LPR_, CPL_, WV_) do { if (gather) LN_BWD_F(LPR_, CPL_, WV_, 1); else if (xn_out) LN_BWD_F(LPR_, CPL_, WV_, 2); else LN_BWD_F(LPR_, CPL_, WV_, 0); } while (0)
    DISPATCH_T(dtype, "lavt_layernorm_bwd",
               if (waves == 8) { if (lpr == 16) LN_BWD(16, 1, 8); else if (lpr == 32) LN_BWD(32, 1, 8); else if (cpl == 1) LN_BWD(64, 1, 8); else LN_BWD(64, 2, 8); }
               else if (lpr == 16) LN_BWD(16, 1, 4); else if (lpr == 32) LN_BWD(32, 1, 4);
               else if (cpl == 1) LN_BWD(64, 1, 4); else if (cpl == 2) LN_BWD(64, 2, 4); else if (cpl <= 4) LN_BWD(64, 4, 4); else LN_BWD(64, 8, 4));
#undef LN_BWD
#undef LN_BWD_F
    if (partials && !partial_only) hipLaunchKernelGGL(reduce_partials_kernel, reduce_partials_grid(blocks, 2 * C), dim3(256), 0, st, partials, blocks, 2 * C, C, dgamma, dbeta);
    LAVT_CHECK_LAUNCH("lavt_layernorm_bwd");
    return LAVT_OK;
}
extern "C" int lavt_layernorm_bwd(int dtype, const void* dy, const void* x, const int32_t* gather, const float* gamma,
                                  const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta, float* ws, int64_t ws_floats,
                                  const void* dres, int rows, int C, void* stream) {
    return layernorm_bwd_impl(dtype, dy, x, gather, gamma, mean, rstd, dx, dgamma, dbeta, ws, ws_floats, dres, rows, C, stream, false);
}
extern "C" int lavt_layernorm_bwd_partial(int dtype, const void* dy, const void* x, const int32_t* gather, const float* gamma, const float* mean,
                                          const float* rstd, void* dx, float* ws, int64_t ws_floats, const void* dres, int rows, int C, void* stream) {
    return layernorm_bwd_impl(dtype, dy, x, gather, gamma, mean, rstd, dx, nullptr, nullptr, ws, ws_floats, dres, rows, C, stream, true);
}
int lavt_layernorm_bwd_partial_impl(int dtype, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx, float* ws,
                                    int64_t ws_floats, const void* dres, int rows, int C, void* stream) {
    return layernorm_bwd_impl(dtype, dy, x, nullptr, gamma, mean, rstd, dx, nullptr, nullptr, ws, ws_floats, dres, rows, C, stream, true);
}
// the same + the LayerNorm output xn = xhat * gamma + beta written on the way (no gather form): for a forward that folded the norm into the consumer's
// GEMM (lavt_gemm_nt.ln_wsum) and therefore never materialised it -- the consumer's weight gradient reads it
extern "C" int lavt_layernorm_bwd_partial_xn(int dtype, const void* dy, const void* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                                             void* dx, void* xn, float* ws, int64_t ws_floats, const void* dres, int rows, int C, void* stream) {
    LAVT_CHECK_ARG(xn && beta, "lavt_layernorm_bwd_partial_xn: xn and beta required");
    return layernorm_bwd_impl(dtype, dy, x, nullptr, gamma, mean, rstd, dx, nullptr, nullptr, ws, ws_floats, dres, rows, C, stream, true, xn, beta);
}
// lavt_layernorm_bwd_partial_xn + the binning job of an earlier attention-backward launch as rider workgroups.  Returns 1 WITHOUT launching anything when this
// LayerNorm's geometry has no rider form (fp32, 8-wave or wide-row variants): the caller then issues the two launches on their own.
extern "C" int lavt_layernorm_bwd_partial_xn_dtable(int dtype, const void* dy, const void* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                                                    void* dx, void* xn, float* ws, int64_t ws_floats, const void* dres, int rows, int C, const lavt_dtable_job_t* job, void* stream) {
    LAVT_CHECK_ARG(dy && x && gamma && mean && rstd && dx && ws && job && rows > 0 && ((xn != nullptr) == (beta != nullptr)), "lavt_layernorm_bwd_partial_xn_dtable: bad arguments");
    int lpr, cpl, waves;
    const int blocks = ln_bwd_geometry(dtype, rows, C, &lpr, &cpl, &waves);
    if (dtype != LAVT_BF16 || waves != 4 || cpl > 2 || C % 8 || ws_floats < (int64_t)blocks * 2 * C) return 1;
    static_assert(sizeof(DtableRider) == sizeof(lavt_dtable_job_t), "lavt_dtable_job_t mirrors DtableRider");
    DtableRider r;
    memcpy(&r, job, sizeof(r));
    const int R = (2 * r.wd - 1) * (2 * r.wh - 1) * (2 * r.ww - 1);
    const size_t lds = (size_t)(R + r.N) * 4 + 16;
    const int riders = r.gx * r.heads * r.gz;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    using T = bf16;
#define LN_BWD_DT_F(LPR_, CPL_, F_) hipLaunchKernelGGL((layernorm_bwd_dtable_kernel<T, LPR_, CPL_, F_>), dim3(blocks + riders), dim3(256), lds, st, (const T*)dy, (const T*)x, gamma, mean, rstd, (T*)dx, ws, (const T*)dres, rows, C, (T*)xn, beta, blocks, r)
#define LN_BWD_DT(LPR_, CPL_) do { if (xn) LN_BWD_DT_F(LPR_, CPL_, 2); else LN_BWD_DT_F(LPR_, CPL_, 0); } while (0)          /* (xn == NULL: the plain LayerNorm backward) */
    if (lpr == 16) LN_BWD_DT(16, 1); else if (lpr == 32) LN_BWD_DT(32, 1); else if (cpl == 1) LN_BWD_DT(64, 1); else LN_BWD_DT(64, 2);
#undef LN_BWD_DT
#undef LN_BWD_DT_F
    LAVT_CHECK_LAUNCH("lavt_layernorm_bwd_partial_xn_dtable");
    return LAVT_OK;
}
extern "C" int lavt_layernorm_bwd_xn(int dtype, const void* dy, const void* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                                     void* dx, void* xn, float* dgamma, float* dbeta, float* ws, int64_t ws_floats, const void* dres, int rows, int C, void* stream) {
    LAVT_CHECK_ARG(xn && beta, "lavt_layernorm_bwd_xn: xn and beta required");
    return layernorm_bwd_impl(dtype, dy, x, nullptr, gamma, mean, rstd, dx, dgamma, dbeta, ws, ws_floats, dres, rows, C, stream, false, xn, beta);
}
// column blocks of a set of width C in the compact grid of lavt_reduce_partials_multi (the caller passes their sum over the sets)
extern "C" int lavt_reduce_partials_column_blocks(int C) { return (2 * C + RPM_COLS - 1) / RPM_COLS; }
extern "C" int lavt_reduce_partials_multi(const int64_t* desc, int n, int total_column_blocks, void* stream) {
    LAVT_CHECK_ARG(desc && n > 0, "lavt_reduce_partials_multi: bad arguments");
    if (total_column_blocks > 0 && n <= 64)          // sum over the sets of lavt_reduce_partials_column_blocks(C): one workgroup column per 128 columns that exist
        hipLaunchKernelGGL(reduce_partials_multi_kernel<true>, dim3(total_column_blocks, 8, 1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), desc, n);
    else          // (the caller does not know the widths: widest supported set C = 2048 -> 32 column blocks per set)
        hipLaunchKernelGGL(reduce_partials_multi_kernel<false>, dim3(4096 / RPM_COLS, 8, n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), desc, n);
    LAVT_CHECK_LAUNCH("lavt_reduce_partials_multi");
    return LAVT_OK;
}

// Row blocks of the column-statistics kernels: a thread walks t rows of its chunk column (t = 8, or 4 / 2 when that is what it takes to reach
// ~256 workgroups; the loads of a thread's rows are independent, the loop is unrolled).  The first version used 64-row blocks whatever the
// width: at C = 1024 (2 row lanes per column) a thread walked 29 rows serially in 8 workgroups -- 25 us for 0.9 MB.
static int stats_launch_geometry(int rows, int groups, int C, int epc, int* rows_per_block) {
    const int cpr = C / epc, span = cpr < 256 ? cpr : 256, rstep = 256 / span;
    if (groups < 1) groups = 1;
    int t = 8;
    while (t > 2 && (long)cdiv(rows, t * rstep) * groups < 256) t >>= 1;
    int rpb = t * rstep;
    const int cap = 1024 / groups > 0 ? 1024 / groups : 1;
    if (cdiv(rows, rpb) > cap) rpb = cdiv(cdiv(rows, cap), rstep) * rstep;
    *rows_per_block = rpb;
    return cdiv(rows, rpb);
}

extern "C" int lavt_colstats(int dtype, const void* x, float* sum, float* sumsq, float* ws, int64_t ws_floats, int groups, int rows, int C, void* stream) {
    const int epc = dtype == LAVT_F32 ? 4 : 8;
    LAVT_CHECK_ARG(x && sum && sumsq && groups > 0 && rows > 0 && C > 0 && C % epc == 0, "lavt_colstats: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rpb;
    const int blocks = stats_launch_geometry(rows, groups, C, epc, &rpb);
    float* partials = (ws && ws_floats >= (int64_t)blocks * groups * 2 * C) ? ws : nullptr;
    DISPATCH_T(dtype, "lavt_colstats",
               hipLaunchKernelGGL((colstats_kernel<T, false>), dim3(blocks, groups), dim3(256), 256 * 2 * Chunk<T>::N * sizeof(float), st,
                                  (const T*)x, (const T*)nullptr, (const T*)nullptr, (const float*)nullptr, (const float*)nullptr, (const T*)nullptr, 0, sum, sumsq, partials, rows, C, rpb, 1));
    if (partials) {
        // reduce + re-centre fused (the finish kernel of lavt_colstats_meanrstd with the sum / M2 outputs instead of mean / rstd): 2 launches, not 3
        DISPATCH_T(dtype, "lavt_colstats",
                   hipLaunchKernelGGL(colstats_finish_kernel<T>, dim3(cdiv(C, 16), groups), dim3(1024), 0, st, partials, blocks, groups, C, (const T*)x, rows, 0.f,
                                      (float*)nullptr, (float*)nullptr, sum, sumsq, (float*)nullptr, (float*)nullptr, 0.f));
    } else {
        DISPATCH_T(dtype, "lavt_colstats",
                   hipLaunchKernelGGL(colstats_center_kernel<T>, dim3(cdiv(groups * C, 256)), dim3(256), 0, st, (const T*)x, sum, sumsq, rows, C, groups * C));
    }
    LAVT_CHECK_LAUNCH("lavt_colstats");
    return LAVT_OK;
}

extern "C" int lavt_colstats_finish_blocks(const float* partials, int nblk, int rows_per_block, int rows, int C, float eps, float* mean, float* rstd,
                                           float* sum_out, float* m2_out, float* running_mean, float* running_var, float momentum, void* stream) {
    LAVT_CHECK_ARG(partials && nblk > 0 && rows_per_block > 0 && rows > 0 && C > 0 && (int64_t)nblk * rows_per_block >= rows, "lavt_colstats_finish_blocks: bad arguments");
    LAVT_CHECK_ARG((mean != nullptr) == (rstd != nullptr) && (sum_out != nullptr) == (m2_out != nullptr) && (mean || sum_out), "lavt_colstats_finish_blocks: mean / rstd and sum / M2 come in pairs");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(colstats_finish_blocks_kernel, dim3(cdiv(C, 32)), dim3(1024), 0, st, partials, nblk, rows_per_block, rows, C, eps, mean, rstd, sum_out, m2_out,
                       running_mean, running_var, momentum);
    LAVT_CHECK_LAUNCH("lavt_colstats_finish_blocks");
    return LAVT_OK;
}

extern "C" int lavt_colstats_meanrstd(int dtype, const void* x, float* mean, float* rstd, float* ws, int64_t ws_floats, int groups, int rows, int C,
                                      float eps, float* running_mean, float* running_var, float momentum, void* stream) {
    const int epc = dtype == LAVT_F32 ? 4 : 8;
    LAVT_CHECK_ARG(x && mean && rstd && ws && groups > 0 && rows > 0 && C > 0 && C % epc == 0, "lavt_colstats_meanrstd: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rpb;
    const int blocks = stats_launch_geometry(rows, groups, C, epc, &rpb);
    LAVT_CHECK_ARG(ws_floats >= (int64_t)blocks * groups * 2 * C, "lavt_colstats_meanrstd: scratch of %ld floats needed", (long)blocks * groups * 2 * C);
    DISPATCH_T(dtype, "lavt_colstats_meanrstd",
               hipLaunchKernelGGL((colstats_kernel<T, false>), dim3(blocks, groups), dim3(256), 256 * 2 * Chunk<T>::N * sizeof(float), st,
                                  (const T*)x, (const T*)nullptr, (const T*)nullptr, (const float*)nullptr, (const float*)nullptr, (const T*)nullptr, 0, mean, rstd, ws, rows, C, rpb, 0);
               hipLaunchKernelGGL(colstats_finish_kernel<T>, dim3(cdiv(C, 16), groups), dim3(1024), 0, st, ws, blocks, groups, C, (const T*)x, rows, eps, mean, rstd,
                                  (float*)nullptr, (float*)nullptr, running_mean, running_var, momentum));
    LAVT_CHECK_LAUNCH("lavt_colstats_meanrstd");
    return LAVT_OK;
}

// SyncBatchNorm forward, after the all-gather: every rank's (sum, centred M2) over `rows` rows each -> statistics of the global batch (parallel
// variance combination, Chan et al.) -> mean / rstd (+ running estimates), ONE launch instead of ~9 element-wise torch kernels + lavt_stats_finalize
__global__ void syncbn_combine_kernel(const float* __restrict__ allst, int world, float rows, float eps, float* __restrict__ mean, float* __restrict__ rstd,
                                      float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float tot = 0.f;
    for (int r = 0; r < world; ++r) tot += allst[((int64_t)r * 2) * C + c];
    const float n = rows * (float)world, gmean = tot / n;
    float m2 = 0.f;
    for (int r = 0; r < world; ++r) {
        const float d = allst[((int64_t)r * 2) * C + c] / rows - gmean;
        m2 += allst[((int64_t)r * 2 + 1) * C + c] + rows * d * d;
    }
    const float var = m2 / n;
    mean[c] = gmean;
    rstd[c] = rsqrtf(var + eps);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * gmean;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
}
extern "C" int lavt_syncbn_combine(const float* allst, int world, float rows_per_rank, float eps, float* mean, float* rstd, float* running_mean,
                                   float* running_var, float momentum, int C, void* stream) {
    LAVT_CHECK_ARG(allst && mean && rstd && world > 0 && rows_per_rank > 0 && C > 0, "lavt_syncbn_combine: bad arguments");
    hipLaunchKernelGGL(syncbn_combine_kernel, dim3(cdiv(C, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), allst, world, rows_per_rank, eps, mean, rstd,
                       running_mean, running_var, momentum, C);
    LAVT_CHECK_LAUNCH("lavt_syncbn_combine");
    return LAVT_OK;
}

extern "C" int lavt_stats_finalize(const float* sum, const float* sumsq, float count, float eps, float* mean, float* rstd,
                                   float* running_mean, float* running_var, float momentum, int n, void* stream) {
    LAVT_CHECK_ARG(sum && sumsq && mean && rstd && n > 0 && count > 0, "lavt_stats_finalize: bad arguments");
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(cdiv(n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), sum, sumsq, count, eps, mean, rstd, running_mean, running_var, momentum, n);
    LAVT_CHECK_LAUNCH("lavt_stats_finalize");
    return LAVT_OK;
}

extern "C" int lavt_norm_apply(int dtype, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                               const void* mul, int relu, void* y, int groups, int rows, int C, void* stream) {
    const int epc = dtype == LAVT_F32 ? 4 : 8;
    LAVT_CHECK_ARG(x && mean && rstd && y && groups > 0 && rows > 0 && C % epc == 0 && (!gamma == !beta), "lavt_norm_apply: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int rpb = apply_rows_per_block(rows, groups, C, epc);
    DISPATCH_T(dtype, "lavt_norm_apply",
               hipLaunchKernelGGL(norm_apply_kernel<T>, dim3(cdiv(rows, rpb), groups), dim3(256), 0, st, (const T*)x, mean, rstd, gamma, beta, (const T*)mul, relu, (T*)y, rows, C, rpb));
    LAVT_CHECK_LAUNCH("lavt_norm_apply");
    return LAVT_OK;
}

extern "C" int lavt_norm_bwd_stats(int dtype, const void* dy, const void* x, const void* y, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, const void* mul, int relu, float* s1, float* s2,
                                   float* ws, int64_t ws_floats, int groups, int rows, int C, void* stream) {
    const int zero_out = 1;          // with scratch (two-stage form) s1 / s2 are cleared by this call; without it they must arrive zeroed (atomics)
    const int epc = dtype == LAVT_F32 ? 4 : 8;
    LAVT_CHECK_ARG(dy && x && mean && rstd && s1 && s2 && (!relu || y) && groups > 0 && rows > 0 && C % epc == 0 && (gamma == nullptr) == (beta == nullptr), "lavt_norm_bwd_stats: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rpb;
    const int blocks = stats_launch_geometry(rows, groups, C, epc, &rpb);
    float* partials = (ws && ws_floats >= (int64_t)blocks * groups * 2 * C) ? ws : nullptr;
    DISPATCH_T(dtype, "lavt_norm_bwd_stats",
               hipLaunchKernelGGL((colstats_kernel<T, true>), dim3(blocks, groups), dim3(256), 256 * 2 * Chunk<T>::N * sizeof(float), st,
                                  (const T*)dy, (const T*)x, (const T*)y, mean, rstd, (const T*)mul, relu, s1, s2, partials, rows, C, rpb, zero_out, gamma, beta));
    if (partials) hipLaunchKernelGGL(reduce_partials_kernel, reduce_partials_grid(blocks, groups * 2 * C), dim3(256), 0, st, partials, blocks, groups * 2 * C, C, s1, s2);
    LAVT_CHECK_LAUNCH("lavt_norm_bwd_stats");
    return LAVT_OK;
}

extern "C" int lavt_norm_bwd_apply(int dtype, const void* dy, const void* x, const void* y, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, const void* mul, int relu, const float* s1, const float* s2,
                                   float count, void* dx, void* dmul, int groups, int rows, int C, void* stream) {
    const int epc = dtype == LAVT_F32 ? 4 : 8;
    LAVT_CHECK_ARG(dy && x && mean && rstd && s1 && s2 && dx && (!relu || y) && count > 0 && C % epc == 0, "lavt_norm_bwd_apply: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int rpb = apply_rows_per_block(rows, groups, C, epc);
    DISPATCH_T(dtype, "lavt_norm_bwd_apply",
               hipLaunchKernelGGL(norm_bwd_apply_kernel<T>, dim3(cdiv(rows, rpb), groups), dim3(256), 0, st, (const T*)dy, (const T*)x, (const T*)y, mean, rstd, gamma, beta,
                                  (const T*)mul, relu, s1, s2, 1.f / count, (T*)dx, (T*)dmul, rows, C, rpb));
    LAVT_CHECK_LAUNCH("lavt_norm_bwd_apply");
    return LAVT_OK;
}

/* lavt_norm_apply with an e4m3 twin of the output (bf16; BASELINE.json configs[4]): q[rows][C] = e4m3(bf16(y) * 448 / *amax_prev) (scale 1 while *amax_prev
 * <= 0), |max| of y recorded into *amax_cur -- the bytes and the bookkeeping of lavt_fp8_quantize(y), without its launch. */
extern "C" int lavt_norm_apply_q8(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, const void* mul, int relu, void* y,
                                  void* q, const float* amax_prev, float* amax_cur, int groups, int rows, int C, void* stream) {
    LAVT_CHECK_ARG(x && mean && rstd && y && q && amax_cur && groups > 0 && rows > 0 && C % 8 == 0 && (!gamma == !beta), "lavt_norm_apply_q8: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int rpb = apply_rows_per_block(rows, groups, C, 8);
    hipLaunchKernelGGL((norm_apply_kernel<bf16, true>), dim3(cdiv(rows, rpb), groups), dim3(256), 0, st, (const bf16*)x, mean, rstd, gamma, beta, (const bf16*)mul, relu,
                       (bf16*)y, rows, C, rpb, (unsigned char*)q, amax_prev, amax_cur);
    LAVT_CHECK_LAUNCH("lavt_norm_apply_q8");
    return LAVT_OK;
}

/* lavt_norm_bwd_apply (bf16) that also records |max| of the stored dx into *amax by atomic max (the caller zeroes it once per step). */
extern "C" int lavt_norm_bwd_apply_amax(const void* dy, const void* x, const void* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                        const void* mul, int relu, const float* s1, const float* s2, float count, void* dx, void* dmul, float* amax, int groups, int rows,
                                        int C, void* stream) {
    LAVT_CHECK_ARG(dy && x && mean && rstd && s1 && s2 && dx && amax && (!relu || y) && count > 0 && C % 8 == 0, "lavt_norm_bwd_apply_amax: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int rpb = apply_rows_per_block(rows, groups, C, 8);
    hipLaunchKernelGGL((norm_bwd_apply_kernel<bf16, true>), dim3(cdiv(rows, rpb), groups), dim3(256), 0, st, (const bf16*)dy, (const bf16*)x, (const bf16*)y, mean, rstd,
                       gamma, beta, (const bf16*)mul, relu, s1, s2, 1.f / count, (bf16*)dx, (bf16*)dmul, rows, C, rpb, amax);
    LAVT_CHECK_LAUNCH("lavt_norm_bwd_apply_amax");
    return LAVT_OK;
}
