// LayerNorm backward as a __device__ body shared by norm.hip (its own launch) and gemm_tn_v2.hip (rider workgroups of the grouped weight-gradient
// launch): dx = rstd (dy gamma - mean_c(dy gamma) - xhat mean_c(dy gamma xhat)) [+ dres], per-workgroup partial sums of d gamma / d beta.
#pragma once
#include "common.h"

namespace {

template <typename T>
__device__ __forceinline__ const T* ln_src(const T* x, const int32_t* gather, int64_t row, int C, int col, bool& ok) {
    if (!gather) { ok = true; return x + row * C + col; }
    const int cq = C >> 2, q = col / cq;
    const int src = gather[row * 4 + q];
    ok = src >= 0;
    return x + (int64_t)src * cq + (col - q * cq);
}

// LPR = lanes per row (16 / 32 / 64): a wave normalises 64/LPR rows at once so that narrow rows (C = 96..256) still use
// every lane; reductions are xor-shuffles inside the LPR-lane group.
template <int LPR> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// FLAGS: bit 0 = the rows may be gathered (PatchMerging's 2x2 form), bit 1 = the LayerNorm output is written on the way; the plain form carries
// neither (a never-taken uniform branch is not free in these one-round kernels: DESIGN.md section 5)
// (a __device__ body with explicit block index / block count: norm.hip's kernel passes its own, the grouped weight-gradient launch runs the body in
// rider workgroups -- gemm_tn_v2.hip, lavt_gemm_tn_grouped_ln; `tid` = thread index inside the WAVES * 64 threads that run the body)
template <typename T, int LPR, int CPL, int WAVES, int FLAGS = 3>
__device__ __forceinline__ void layernorm_bwd_body(const T* __restrict__ dy, const T* __restrict__ x,
                                                   const int32_t* __restrict__ gather, const float* __restrict__ gamma,
                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                   T* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                   float* __restrict__ partials, const T* __restrict__ dres, int rows, int C,
                                                   T* __restrict__ xn_out_, const float* __restrict__ beta, const int bid, const int nblocks, const int tid, float* __restrict__ red) {
    if constexpr (!(FLAGS & 1)) gather = nullptr;
    T* const xn_out = (FLAGS & 2) ? xn_out_ : nullptr;
    // CPL = chunks per lane (compile time: the row arrays are exactly as large as needed; LPR < 64 only with CPL == 1)
    constexpr int EPC = Chunk<T>::N, MAXC = CPL, RPW = 64 / LPR;
    const int lane = tid & 63, lir = lane % LPR, wave = tid >> 6;
    const int nchunk = C / EPC;
    constexpr int cpl = CPL;
    float dg[MAXC * EPC], db[MAXC * EPC];
#pragma unroll
    for (int e = 0; e < MAXC * EPC; ++e) { dg[e] = 0.f; db[e] = 0.f; }
    const int64_t rstride = (int64_t)nblocks * WAVES * RPW;
    for (int64_t row0 = ((int64_t)bid * WAVES + wave) * RPW; row0 < rows; row0 += rstride) {
        const int64_t row = row0 + lane / LPR;
        const bool live = row < rows;
        const float mu = live ? mean[row] : 0.f, rs = live ? rstd[row] : 0.f;
        float xh[MAXC * EPC], g[MAXC * EPC];
        // the residual branch's gradient, requested with the row (round 5: loaded behind the two reductions it was one more exposed round trip per row); rows of
        // more than two chunks per lane (C > 1024: no benchmarked shape) keep the load at its use -- eight more chunks in flight spilled 50-100 registers
        constexpr bool RES_PRE = MAXC <= 2;
        uint4 rres[RES_PRE ? MAXC : 1];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = lir + LPR * c;
            const bool on = live && c < cpl && ch < nchunk;
            float fx[EPC], fg[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) { fx[e] = 0.f; fg[e] = 0.f; }
            if (on) {
                bool ok;
                const T* src = ln_src<T>(x, gather, row, C, ch * EPC, ok);
                if (ok) chunk_to_f<T>(*reinterpret_cast<const uint4*>(src), fx);
                chunk_to_f<T>(*reinterpret_cast<const uint4*>(dy + row * C + ch * EPC), fg);
                if constexpr (RES_PRE) { if (dres) rres[c] = *reinterpret_cast<const uint4*>(dres + row * C + ch * EPC); }
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const int k = c * EPC + e;
                xh[k] = on ? (fx[e] - mu) * rs : 0.f;
                const float gg = on ? fg[e] * gamma[ch * EPC + e] : 0.f;
                g[k] = gg;
                s1 += gg; s2 += gg * xh[k];
                dg[k] += fg[e] * xh[k];
                db[k] += fg[e];
            }
            if (xn_out && on) {          // the LayerNorm OUTPUT, for a consumer whose forward folded the norm into its GEMM (the weight gradient's operand)
                float fy[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) fy[e] = xh[c * EPC + e] * gamma[ch * EPC + e] + beta[ch * EPC + e];
                *reinterpret_cast<uint4*>(xn_out + row * C + ch * EPC) = f_to_chunk<T>(fy);
            }
        }
        s1 = group_sum<LPR>(s1) / C; s2 = group_sum<LPR>(s2) / C;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = lir + LPR * c;
            if (live && c < cpl && ch < nchunk) {
                float f[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) f[e] = rs * (g[c * EPC + e] - s1 - xh[c * EPC + e] * s2);
                if (dres) {                       // gradient of the residual branch that bypassed this LayerNorm: dx = LN'(dy) + dres (no gather form)
                    float fr[EPC];
                    if constexpr (RES_PRE) chunk_to_f<T>(rres[c], fr);
                    else chunk_to_f<T>(*reinterpret_cast<const uint4*>(dres + row * C + ch * EPC), fr);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) f[e] += fr[e];
                }
                if (!gather) *reinterpret_cast<uint4*>(dx + row * C + ch * EPC) = f_to_chunk<T>(f);
                else {
                    const int cq = C >> 2, col = ch * EPC, qd = col / cq;
                    const int src = gather[row * 4 + qd];
                    if (src >= 0) *reinterpret_cast<uint4*>(dx + (int64_t)src * cq + (col - qd * cq)) = f_to_chunk<T>(f);
                }
            }
        }
    }
    // sum the RPW row groups of the wave, then the waves through LDS, then ONE partial (or atomic) per channel per workgroup
#pragma unroll
    for (int e = 0; e < MAXC * EPC; ++e) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) { dg[e] += __shfl_xor(dg[e], o, 64); db[e] += __shfl_xor(db[e], o, 64); }
    }
    // red: (WAVES - 1) * LPR * CPL * EPC floats of LDS lent by the caller (static in norm.hip's kernel, the idle tile ring in a rider workgroup)
    constexpr int RW = LPR * CPL * EPC;                    // >= C
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float* part = pass == 0 ? dg : db;
        float* dst = pass == 0 ? dgamma : dbeta;
        __syncthreads();
        if (wave > 0 && lane < LPR) {
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int ch = lir + LPR * c;
                if (c < cpl && ch < nchunk)
#pragma unroll
                    for (int e = 0; e < EPC; ++e) red[(wave - 1) * RW + ch * EPC + e] = part[c * EPC + e];
            }
        }
        __syncthreads();
        if (wave == 0 && lane < LPR) {
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int ch = lir + LPR * c;
                if (c < cpl && ch < nchunk)
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const int col = ch * EPC + e;
                        float tot = part[c * EPC + e];
#pragma unroll
                        for (int w = 0; w < WAVES - 1; ++w) tot += red[w * RW + col];
                        // hundreds of workgroups adding to the SAME C addresses serialise in L2: write per-workgroup partials instead
                        if (partials) partials[((int64_t)bid * 2 + pass) * C + col] = tot;
                        else atomicAdd(dst + col, tot);
                    }
            }
        }
    }
}


}  // namespace
