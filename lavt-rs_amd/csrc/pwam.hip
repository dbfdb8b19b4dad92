// Fused PWAM (pixel-word attention module + language gate) for gfx950, bf16.
// Reference arithmetic: PWAM.forward lib/backbone.py:1265-1278, SpatialImageLanguageAttention.forward :1329-1372, gate :604-611 / :669.
//
// The reference evaluates, per sample, on T pixels x C channels with <= 32 word slots J:
//     vis = GELU(x Wv^T + bv);   q = IN_T(x Wq^T + bq);   P = softmax_words(q K^T C^-1/2 + mask);   w = (P V) Wo^T + bo;
//     r = GELU((vis * IN_T(w)) Wm^T + bm);   x' = x + tanh(ReLU(r W1^T) W2^T) * r
// as ~15 kernels forward and ~35 backward when composed from GEMMs and normalisation passes.  Two identities collapse the middle part:
//   (1) the instance norm of q folds into the keys:  S = q K''^T + s0,  K'' = C^-1/2 rstd_q K,  s0 = mask - mu_q K''^T        (no normalised q tensor)
//   (2) w = P (V Wo^T) + bo is a [T x 32] x [32 x C] product of the word probabilities, so its instance-norm statistics follow from the word
//       statistics alone: mean_T w = Pbar VW + bo, var_T w[c] = VW[:,c]^T Cov_T(P) VW[:,c], and IN_T(w) = (P - Pbar) VW' with VW' = VW rstd_w
//       -- the [T x C] tensor w, its two statistics passes and its [C x C] GEMM over T rows do not exist.
// What remains besides plain GEMMs are two row-streaming kernel families (HBM-bound; the 32-wide contractions ride on v_mfma_f32_16x16x32_bf16):
//   pwam_words_kernel: [T x C] x [C x 32] -> per-row word vectors (forward: S -> softmax -> P; backward: dP -> softmax' -> dS)
//   pwam_mix_kernel:   [T x 32] x [32 x C] -> per-row channel vectors fused with the element-wise neighbours
//                      (forward: mm = GELU(vpre) * what; backward A: d vpre, d what; backward C: dq = dS K'' + c0 - q c1)
// and three tiny language-side kernels on [32 x C] / [32 x 32] matrices.  tools/pwam_algebra_check.py proves the algebra against autograd.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ bf16x8 ldg8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 lds8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ float bf16_round(float v) { return (float)(bf16)v; }
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p; }

// A lane of a C^T accumulator pair holds, for ONE row, elements 4g .. 4g+3 of a 16-wide tile (p0) and the same of the next tile (p1), g = lane / 16.
// Lanes g and g ^ 1 swap one packed quadruple so that every lane moves 16 contiguous bytes (64 contiguous bytes per row and wave-instruction).
__device__ __forceinline__ void store_pair16(bf16* row32, int g, uint2 p0, uint2 p1, bool valid) {
    const bool odd = g & 1;
    const uint2 send = odd ? p0 : p1;
    const uint2 got = make_uint2((unsigned)__shfl_xor((int)send.x, 16, 64), (unsigned)__shfl_xor((int)send.y, 16, 64));
    const uint4 out = odd ? make_uint4(got.x, got.y, p1.x, p1.y) : make_uint4(p0.x, p0.y, got.x, got.y);
    if (valid) *reinterpret_cast<uint4*>(row32 + (odd ? 16 + 4 * (g - 1) : 4 * g)) = out;
}
// the mirror image: one 16-byte load per lane, then the exchange -> this lane's quadruples of both tiles
__device__ __forceinline__ void load_pair16(const bf16* row32, int g, uint2& p0, uint2& p1) {
    const bool odd = g & 1;
    const uint4 L = *reinterpret_cast<const uint4*>(row32 + (odd ? 16 + 4 * (g - 1) : 4 * g));
    const uint2 send = odd ? make_uint2(L.x, L.y) : make_uint2(L.z, L.w);
    const uint2 got = make_uint2((unsigned)__shfl_xor((int)send.x, 16, 64), (unsigned)__shfl_xor((int)send.y, 16, 64));
    p0 = odd ? got : make_uint2(L.x, L.y);
    p1 = odd ? make_uint2(L.z, L.w) : got;
}
__device__ __forceinline__ void unpack4(uint2 p, float (&f)[4]) { f[0] = bf_lo(p.x); f[1] = bf_hi(p.x); f[2] = bf_lo(p.y); f[3] = bf_hi(p.y); }
__device__ __forceinline__ uint2 pack4(const float (&f)[4]) { return make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3])); }

// ================================================================================================ words kernel
// grid (row blocks, B), 256 threads; a wave owns 16-row tiles.  LDS: the [32][C] word matrix (rows padded by 8 elements) + 32 floats.
//   BWD = false: Wm = K'' * log2 e (built here from K, rstd_q), vec = (maskbias - mu_q K''^T) log2 e;  out = P = softmax over words < n_l
//   BWD = true:  Wm = VW' (word-major, from lavt_pwam_lang_fwd), second contraction with -Q over the words of P, vec = Pbar Q - u;
//                out = dS = P (dP - sum_j P_j dP_j)
struct WordsArgs {
    const bf16* X;        // [B*T][ldx]: q (forward) / d what (backward)
    int64_t ldx;
    const bf16* Wsrc;     // forward: K [B][32][ldw];  backward: VW' [B][32][C] (ldw = C)
    int64_t ldw;
    const float* mean;    // forward: mu_q, rstd_q [B][C]
    const float* rstd;
    const float* vec;     // forward: maskbias [B][32]
    const float* Qf;      // backward: [B][records][1024 Q | 32 u] fp32 partial records of lavt_pwam_lang_bwd1
    const float* pbar;    // backward: Pbar [B][32]
    const bf16* P;        // backward: P [B*T][32]
    bf16* out;            // [B*T][32]
    int T, C, n_l;
    float alpha;
    // forward by-product (rec != null): the second moments of the word probabilities.  Every workgroup leaves [1024 P^T P | 32 colsum(P)] of ITS rows in
    // rec[B][gridDim.x][1056]; the consumer (lavt_pwam_lang_fwd: a few workgroups per sample) adds the records in index order.
    float* rec;
};
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
typedef unsigned long long u64_t;

template <bool BWD, int NT, int PRE>
__global__ __launch_bounds__(NT) void pwam_words_kernel(const WordsArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NW = NT / 64;
    const int C = a.C, LDW = C + 8;
    bf16* Wm = reinterpret_cast<bf16*>(smem_raw);                   // [32][LDW]
    float* vec = reinterpret_cast<float*>(Wm + 32 * LDW);           // [32]
    bf16* Qn = reinterpret_cast<bf16*>(vec + 32);                   // backward: -Q as bf16 [32][40]
    float* meanS = reinterpret_cast<float*>(Qn + 32 * 40);          // forward: mu_q of this sample [C]
    float* rstdS = meanS + C;                                       // forward: rstd_q of this sample [C]
    bf16* Pt = reinterpret_cast<bf16*>(rstdS + C);                  // forward by-product: [NW][16 rows][32 words], a wave's P tile for the transposing reads
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int b = blockIdx.y;

    // ---- (round 5) requests that depend on nothing: the first 16 k-steps of this wave's first row tile, its P fragments (backward), the q means /
    // mask bias (forward), the u records and Pbar (backward).  Issued here they travel under the word-matrix prologue; issued where they are used they
    // were 2 (forward: 5) more exposed L2 round trips in a launch that is a chain of round trips (the backward launches ran 6-8 us behind the forward
    // ones of the same shape, the 8-workgroup launch of the last stage 20 us for 2 us of rows).
    const int ntiles = (a.T + 15) >> 4;
    const int ksteps = C >> 5;
    bf16x8 xpre[PRE] = {};
    bf16x8 pf_pre = {};
    uint2 p0_pre = make_uint2(0u, 0u), p1_pre = make_uint2(0u, 0u);
    {
        const int tile = min((int)blockIdx.x * NW + wave, ntiles - 1);
        const int64_t row = (int64_t)b * a.T + min(tile * 16 + c16, a.T - 1);
        const bf16* xp = a.X + row * a.ldx + 8 * g;
#pragma unroll
        for (int u = 0; u < PRE; ++u)
            if (u < ksteps) xpre[u] = ldg8(xp + 32 * u);          // (wave-uniform: the 128-channel stage has 4 k-steps, not 16 requests)
        if constexpr (BWD) {
            pf_pre = ldg8(a.P + row * 32 + 8 * g);
            p0_pre = *reinterpret_cast<const uint2*>(a.P + row * 32 + 4 * g);
            p1_pre = *reinterpret_cast<const uint2*>(a.P + row * 32 + 16 + 4 * g);
        }
    }
    float uj_pre = 0.f, pb_pre[4] = {0.f, 0.f, 0.f, 0.f}, mb_pre = 0.f;
    float4 mean_pre[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)}, rstd_pre[2] = {mean_pre[0], mean_pre[0]};
    if constexpr (!BWD) {
        mb_pre = a.vec[b * 32 + ((tid & 255) >> 3)];
#pragma unroll
        for (int i = 0; i < 2; ++i)          // (C <= 2048: two float4 per thread)
            if (tid + NT * i < (C >> 2)) {
                mean_pre[i] = *reinterpret_cast<const float4*>(a.mean + (int64_t)b * C + 4 * (tid + NT * i));
                rstd_pre[i] = *reinterpret_cast<const float4*>(a.rstd + (int64_t)b * C + 4 * (tid + NT * i));
            }
    } else {
        const int j = tid >> 3, part = tid & 7;
        float vu[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) vu[u] = a.Qf[((int64_t)b * a.n_l + min(u, a.n_l - 1)) * 1056 + 1024 + j];
#pragma unroll
        for (int i = 0; i < 4; ++i) pb_pre[i] = a.pbar[b * 32 + part + 8 * i];
#pragma unroll
        for (int u = 0; u < 16; ++u) uj_pre += u < a.n_l ? vu[u] : 0.f;          // (fixed order: run-to-run identical)
    }

    // backward: the Q records (requested in front of the word matrix: the vector-memory counter retires in order, so summing them below does not wait
    // for the matrix, and the matrix does not wait for them -- behind the matrix loop they were a second dependent round trip)
    float v[BWD ? 4 : 1][16];
    if constexpr (BWD) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int u = 0; u < 16; ++u) v[k][u] = a.Qf[((int64_t)b * a.n_l + min(u, a.n_l - 1)) * 1056 + tid + 256 * k];
    }

    // ---- prologue: the word matrix and the per-word constants of this sample ----
    const int nch = C >> 3;
    // (sixteen chunks per thread per pass = one pass at C = 1024, every load issued before the first use: as a one-chunk loop the 32 x 1024 matrix of
    // the last stage was a chain of 16 exposed round trips per thread -- 22 us for a launch whose rows take 2.  Forward: rstd_q and mu_q travel with
    // the first pass in registers, are parked in LDS, and the chunks are scaled behind ONE barrier -- as global loads per chunk they halved the pass.)
    constexpr int UW = NT == 256 ? 16 : 2;          // (the 1024-thread form serves C <= 256: 1024 chunks)
    for (int base = 0; base < 32 * nch; base += NT * UW) {          // (uniform trip count: the forward pass holds a barrier)
        const int e0 = base + tid;
        uint4 raw[UW];
        // (loads unconditional on a clamped chunk index, only the LDS store below is predicated: with the loads under `if (e < ...)` hipcc kept raw[] in
        // scratch -- 144 bytes per lane, eight scratch round trips at the head of the kernel)
#pragma unroll
        for (int u = 0; u < UW; ++u) {
            const int e = min(e0 + NT * u, 32 * nch - 1);
            const int j = e / nch, cc = e - j * nch;
            raw[u] = *reinterpret_cast<const uint4*>(a.Wsrc + ((int64_t)b * 32 + j) * a.ldw + cc * 8);
        }
        if constexpr (!BWD) {
            if (base == 0) {                          // (first pass, every thread)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    if (tid + NT * i < (C >> 2)) {
                        *reinterpret_cast<float4*>(meanS + 4 * (tid + NT * i)) = mean_pre[i];
                        *reinterpret_cast<float4*>(rstdS + 4 * (tid + NT * i)) = rstd_pre[i];
                    }
                __syncthreads();
            }
        }
#pragma unroll
        for (int u = 0; u < UW; ++u) {
            const int e = e0 + NT * u;
            uint4 val = raw[u];
            if (e < 32 * nch) {
                const int j = e / nch, cc = e - j * nch;
                if constexpr (!BWD) {
                    float f[8];
                    chunk_to_f<bf16>(val, f);
                    const float s = a.alpha * LOG2E;
                    const float4 r0 = *reinterpret_cast<const float4*>(rstdS + cc * 8), r1 = *reinterpret_cast<const float4*>(rstdS + cc * 8 + 4);
                    f[0] *= r0.x * s; f[1] *= r0.y * s; f[2] *= r0.z * s; f[3] *= r0.w * s;
                    f[4] *= r1.x * s; f[5] *= r1.y * s; f[6] *= r1.z * s; f[7] *= r1.w * s;
                    val = f_to_chunk<bf16>(f);
                }
                *reinterpret_cast<uint4*>(Wm + j * LDW + cc * 8) = val;
            }
        }
    }
    if constexpr (BWD) {
        // (n_l carries the record count in this mode.)  Every load of a thread in flight at once, summed in a fixed order: as a plain loop over
        // the records each element was a chain of `records` exposed round trips at the head of a kernel that sits on the critical chain
        // (lavt_pwam_q_parts caps the record count at 16)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float q = 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) q += u < a.n_l ? v[k][u] : 0.f;          // (fixed order: run-to-run identical)
            const int e = tid + 256 * k;
            Qn[(e >> 5) * 40 + (e & 31)] = (bf16)(-q);
        }
    }
    __syncthreads();
    if (NT == 256 || tid < 256) {
        const int j = tid >> 3, part = tid & 7;
        float s = 0.f;
        if constexpr (!BWD) {
#pragma unroll 4
            for (int cc = part; cc < nch; cc += 8) {
                float f[8];
                chunk_to_f<bf16>(*reinterpret_cast<const uint4*>(Wm + j * LDW + cc * 8), f);
                const float4 m0 = *reinterpret_cast<const float4*>(meanS + cc * 8), m1 = *reinterpret_cast<const float4*>(meanS + cc * 8 + 4);
                s += f[0] * m0.x + f[1] * m0.y + f[2] * m0.z + f[3] * m0.w + f[4] * m1.x + f[5] * m1.y + f[6] * m1.z + f[7] * m1.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) s += pb_pre[i] * (-(float)Qn[(part + 8 * i) * 40 + j]);       // Pbar Q with the bf16 Q the MFMA sees
        }
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if (part == 0) {
            if constexpr (BWD) vec[j] = s - uj_pre;
            else vec[j] = mb_pre * LOG2E - s;
        }
    }
    __syncthreads();

    // forward by-product accumulators: P^T P as 2 x 2 tiles of 16 x 16 words (D layout: row = word 4g + r, column = word c16) and colsum(P) (P^T 1)
    f32x4 pp[2][2] = {{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}};
    f32x4 ps[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const bool moments = !BWD && a.rec != nullptr;
    bool first = true;
    for (int tile = blockIdx.x * NW + wave; tile < ntiles; tile += gridDim.x * NW) {
        const int t = tile * 16 + c16;
        const bool vr = t < a.T;
        const int64_t row = (int64_t)b * a.T + (vr ? t : a.T - 1);
        const bf16* xp = a.X + row * a.ldx + 8 * g;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        const bf16* w0 = Wm + c16 * LDW + 8 * g;
        const bf16* w1 = w0 + 16 * LDW;
        bf16x8 pf = pf_pre;
        uint2 pp0 = p0_pre, pp1 = p1_pre;
        int ks = 0;
        if (first) {                                 // (wave-uniform) the k-steps requested at the head of the kernel
#pragma unroll
            for (int u = 0; u < PRE; ++u)
                if (u < ksteps) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds8(w0 + 32 * u), xpre[u], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds8(w1 + 32 * u), xpre[u], acc[1], 0, 0, 0);
                }
            ks = min(PRE, ksteps);
            first = false;
        } else if constexpr (BWD) {
            pf = ldg8(a.P + row * 32 + 8 * g);
            pp0 = *reinterpret_cast<const uint2*>(a.P + row * 32 + 4 * g);
            pp1 = *reinterpret_cast<const uint2*>(a.P + row * 32 + 16 + 4 * g);
        }
        for (; ks + 16 <= ksteps; ks += 16) {        // (C >= 512: sixteen k-steps of row loads in flight -- the last stage's 1024-channel rows were 4 rounds of 8)
            bf16x8 xf[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) xf[u] = ldg8(xp + 32 * (ks + u));
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds8(w0 + 32 * (ks + u)), xf[u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds8(w1 + 32 * (ks + u)), xf[u], acc[1], 0, 0, 0);
            }
        }
        for (; ks + 4 <= ksteps; ks += 4) {
            bf16x8 xf[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) xf[u] = ldg8(xp + 32 * (ks + u));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds8(w0 + 32 * (ks + u)), xf[u], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds8(w1 + 32 * (ks + u)), xf[u], acc[1], 0, 0, 0);
            }
        }
        for (; ks < ksteps; ++ks) {
            const bf16x8 xf = ldg8(xp + 32 * ks);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds8(w0 + 32 * ks), xf, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds8(w1 + 32 * ks), xf, acc[1], 0, 0, 0);
        }
        // lane: row c16, words 4g + r (tile 0) and 16 + 4g + r (tile 1)
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(vec + 4 * g), v1 = *reinterpret_cast<const f32x4*>(vec + 16 + 4 * g);
        float o0[4], o1[4];
        if constexpr (!BWD) {
            float s0[4], s1[4], mx = -3.0e38f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s0[r] = (4 * g + r < a.n_l) ? acc[0][r] + v0[r] : -3.0e38f;
                s1[r] = (16 + 4 * g + r < a.n_l) ? acc[1][r] + v1[r] : -3.0e38f;
                mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o0[r] = (4 * g + r < a.n_l) ? __builtin_amdgcn_exp2f(s0[r] - mx) : 0.f;
                o1[r] = (16 + 4 * g + r < a.n_l) ? __builtin_amdgcn_exp2f(s1[r] - mx) : 0.f;
                sum += o0[r] + o1[r];
            }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.f / sum;
#pragma unroll
            for (int r = 0; r < 4; ++r) { o0[r] *= inv; o1[r] *= inv; }
        } else {
            // second contraction: - P Q (k = words of P): A = -Q rows (word 16wt + c16, k-words 8g..), B = P row fragment
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds8(Qn + c16 * 40 + 8 * g), pf, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds8(Qn + (16 + c16) * 40 + 8 * g), pf, acc[1], 0, 0, 0);
            float p0[4], p1[4], dot = 0.f;
            unpack4(pp0, p0);
            unpack4(pp1, p1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o0[r] = acc[0][r] + v0[r];
                o1[r] = acc[1][r] + v1[r];
                dot += p0[r] * o0[r] + p1[r] * o1[r];
            }
            dot += __shfl_xor(dot, 16, 64);
            dot += __shfl_xor(dot, 32, 64);
#pragma unroll
            for (int r = 0; r < 4; ++r) { o0[r] = p0[r] * (o0[r] - dot); o1[r] = p1[r] * (o1[r] - dot); }
        }
        store_pair16(a.out + row * 32, g, pack4(o0), pack4(o1), vr);
        if constexpr (!BWD) {
            if (moments) {          // (wave-uniform: the transposing reads need every lane)
                // P^T P sums over the ROWS, which sit on the lanes of the S^T accumulators: one transpose through a 1 KB LDS image per wave.  The
                // image holds the bf16 values that were stored (rows beyond T as zeros); lane 4q + p of a 16-lane group addresses row 4g + q,
                // words 4p .. 4p + 3 of a word tile and receives word c16 of rows 4g .. 4g + 3: the A and the B fragment of v_mfma_f32_16x16x16_bf16.
                bf16* pt = Pt + wave * 512;
                const uint2 q0 = vr ? pack4(o0) : make_uint2(0u, 0u), q1 = vr ? pack4(o1) : make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(pt + c16 * 32 + 4 * g) = q0;
                *reinterpret_cast<uint2*>(pt + c16 * 32 + 16 + 4 * g) = q1;
                const unsigned ad = lds_addr(pt + (4 * g + (c16 >> 2)) * 32 + 4 * (c16 & 3));
                u64_t t0, t1;
                asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:32\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(t0), "=&v"(t1) : "v"(ad) : "memory");
                const s16x4 f0 = __builtin_bit_cast(s16x4, t0), f1 = __builtin_bit_cast(s16x4, t1);
                const s16x4 one4 = {0x3F80, 0x3F80, 0x3F80, 0x3F80};
                pp[0][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(f0, f0, pp[0][0], 0, 0, 0);
                pp[0][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(f0, f1, pp[0][1], 0, 0, 0);
                pp[1][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(f1, f0, pp[1][0], 0, 0, 0);
                pp[1][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(f1, f1, pp[1][1], 0, 0, 0);
                ps[0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(f0, one4, ps[0], 0, 0, 0);
                ps[1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(f1, one4, ps[1], 0, 0, 0);
            }
        }
    }
    if constexpr (!BWD) {
        if (moments) {
            // this workgroup's record: waves in LDS (the word matrix is dead), summed in wave order, stored plainly
            __syncthreads();
            float* red = reinterpret_cast<float*>(smem_raw);               // [NW][1056]
            float* mine = red + wave * 1056;
#pragma unroll
            for (int wa = 0; wa < 2; ++wa)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * wa + 4 * g + r;
                    mine[m * 32 + c16] = pp[wa][0][r];
                    mine[m * 32 + 16 + c16] = pp[wa][1][r];
                    if (c16 == 0) mine[1024 + m] = ps[wa][r];
                }
            __syncthreads();
            float* recp = a.rec + ((int64_t)b * gridDim.x + blockIdx.x) * 1056;
            for (int e = tid; e < 1056; e += NT) {
                float s = red[e];
#pragma unroll
                for (int w = 1; w < NW; ++w) s += red[w * 1056 + e];
                recp[e] = s;
            }
        }
    }
}

// ================================================================================================ mix kernel
// grid (row blocks, B), 256 threads; a wave owns 32-row tile pairs.  acc[channel 4g + r of a 16-channel tile][row c16] = sum_j Wc[channel][j] Wd[row][j].
//   MODE 0 (forward):    what = acc + beta;  mm = GELU(vpre) * what                               out0 = mm
//   MODE 1 (backward A): dmm given;  d what = dmm * GELU(vpre);  d vpre = dmm * what * GELU'(vpre)  out0 = d vpre, out1 = d what
//   MODE 2 (backward C): Wd = dS, Wc = K''^T;  dq = acc + c0 - q * c1                              out0 = dq
struct MixArgs {
    const bf16* Wd;       // [B*T][32] row-major word vectors (P / dS)
    const bf16* Wc;       // [B][C][32] channel-major (VW'^T / K''^T)
    const float* v0;      // [B][C]: beta (modes 0, 1) / c0 (mode 2)
    const float* v1;      // [B][C]: c1 (mode 2)
    const float* xb;      // [C] or null: bias added to X (modes 0, 1: the vis_project bias, so that the producing GEMM needs none)
    const bf16* X;        // [B*T][ldx]: vpre (modes 0, 1) / q (mode 2)
    int64_t ldx;
    const bf16* D;        // [B*T][ldd]: dmm (mode 1)
    int64_t ldd;
    bf16* out0;
    int64_t ld0;
    bf16* out1;
    int64_t ld1;
    int T, C;
};

template <int MODE>
__global__ __launch_bounds__(256) void pwam_mix_kernel(const MixArgs a) {
    // A wave owns items (16-row tile, 64-channel group): every global load of an item is issued before the first use (the kernel is a stream of
    // 16-byte loads and stores with 4 MFMAs in between; as a loop over channel pairs with the loads inside it ran as a chain of exposed latencies:
    // 35 us per launch at 28 800 x 128 where the bytes take ~5).
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int b = blockIdx.y, C = a.C;
    const int ntiles = (a.T + 15) >> 4, ngrp = (C + 63) >> 6;
    const bf16* Wc = a.Wc + (int64_t)b * C * 32;
    const float* v0 = a.v0 + (int64_t)b * C;
    const float* v1 = MODE == 2 ? a.v1 + (int64_t)b * C : nullptr;
    const bool odd = g & 1;
    const int poff = odd ? 16 + 4 * (g - 1) : 4 * g;          // this lane's 16-byte piece of a 32-channel span (load_pair16 / store_pair16)
    for (int item = blockIdx.x * 4 + wave; item < ntiles * ngrp; item += gridDim.x * 4) {
        const int tile = item / ngrp, grp = item - tile * ngrp;
        const int t = tile * 16 + c16;
        const bool vr = t < a.T;
        const int64_t row = (int64_t)b * a.T + (vr ? t : a.T - 1);
        const int chg = 64 * grp;
        const bool two = chg + 64 <= C;                        // C % 32 == 0: the last group may hold one 32-channel span only (wave-uniform)
        const bf16x8 wd = ldg8(a.Wd + row * 32 + 8 * g);
        bf16x8 wa[2][2];
        uint4 xr[2], dr[2];
        float4 va[2][2], vb[2][2], xb4[2][2];
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            const int ch0 = (cp == 1 && !two) ? chg : chg + 32 * cp;      // (a dead second span re-reads the first: no out-of-range access, result unused)
            wa[cp][0] = ldg8(Wc + (int64_t)(ch0 + c16) * 32 + 8 * g);
            wa[cp][1] = ldg8(Wc + (int64_t)(ch0 + 16 + c16) * 32 + 8 * g);
            xr[cp] = *reinterpret_cast<const uint4*>(a.X + row * a.ldx + ch0 + poff);
            if constexpr (MODE == 1) dr[cp] = *reinterpret_cast<const uint4*>(a.D + row * a.ldd + ch0 + poff);
            va[cp][0] = *reinterpret_cast<const float4*>(v0 + ch0 + 4 * g);
            va[cp][1] = *reinterpret_cast<const float4*>(v0 + ch0 + 16 + 4 * g);
            if constexpr (MODE == 2) {
                vb[cp][0] = *reinterpret_cast<const float4*>(v1 + ch0 + 4 * g);
                vb[cp][1] = *reinterpret_cast<const float4*>(v1 + ch0 + 16 + 4 * g);
            } else if (a.xb) {
                xb4[cp][0] = *reinterpret_cast<const float4*>(a.xb + ch0 + 4 * g);
                xb4[cp][1] = *reinterpret_cast<const float4*>(a.xb + ch0 + 16 + 4 * g);
            } else {
                xb4[cp][0] = xb4[cp][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        auto xchg = [&](uint4 L, uint2& p0, uint2& p1) {
            const uint2 send = odd ? make_uint2(L.x, L.y) : make_uint2(L.z, L.w);
            const uint2 got = make_uint2((unsigned)__shfl_xor((int)send.x, 16, 64), (unsigned)__shfl_xor((int)send.y, 16, 64));
            p0 = odd ? got : make_uint2(L.x, L.y);
            p1 = odd ? make_uint2(L.z, L.w) : got;
        };
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            const int ch0 = chg + 32 * cp;
            const bool live = vr && (cp == 0 || two);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cp][0], wd, z, 0, 0, 0);
            const f32x4 e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cp][1], wd, z, 0, 0, 0);
            uint2 xp0, xp1;
            xchg(xr[cp], xp0, xp1);
            float x0[4], x1[4], r0[4], r1[4];
            unpack4(xp0, x0); unpack4(xp1, x1);
            const float c0a[4] = {va[cp][0].x, va[cp][0].y, va[cp][0].z, va[cp][0].w}, c0b[4] = {va[cp][1].x, va[cp][1].y, va[cp][1].z, va[cp][1].w};
            if constexpr (MODE != 2) {
                x0[0] += xb4[cp][0].x; x0[1] += xb4[cp][0].y; x0[2] += xb4[cp][0].z; x0[3] += xb4[cp][0].w;
                x1[0] += xb4[cp][1].x; x1[1] += xb4[cp][1].y; x1[2] += xb4[cp][1].z; x1[3] += xb4[cp][1].w;
            }
            if constexpr (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { r0[r] = gelu_f_fast(x0[r]) * (e0[r] + c0a[r]); r1[r] = gelu_f_fast(x1[r]) * (e1[r] + c0b[r]); }
                store_pair16(a.out0 + row * a.ld0 + ch0, g, pack4(r0), pack4(r1), live);
            } else if constexpr (MODE == 1) {
                uint2 dp0, dp1;
                xchg(dr[cp], dp0, dp1);
                float d0[4], d1[4], w0[4], w1[4];
                unpack4(dp0, d0); unpack4(dp1, d1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float wh0 = e0[r] + c0a[r], wh1 = e1[r] + c0b[r];
                    r0[r] = d0[r] * wh0 * gelu_grad_f_fast(x0[r]); r1[r] = d1[r] * wh1 * gelu_grad_f_fast(x1[r]);
                    w0[r] = d0[r] * gelu_f_fast(x0[r]); w1[r] = d1[r] * gelu_f_fast(x1[r]);
                }
                store_pair16(a.out0 + row * a.ld0 + ch0, g, pack4(r0), pack4(r1), live);
                store_pair16(a.out1 + row * a.ld1 + ch0, g, pack4(w0), pack4(w1), live);
            } else {
                const float c1a[4] = {vb[cp][0].x, vb[cp][0].y, vb[cp][0].z, vb[cp][0].w}, c1b[4] = {vb[cp][1].x, vb[cp][1].y, vb[cp][1].z, vb[cp][1].w};
#pragma unroll
                for (int r = 0; r < 4; ++r) { r0[r] = e0[r] + c0a[r] - x0[r] * c1a[r]; r1[r] = e1[r] + c0b[r] - x1[r] * c1b[r]; }
                store_pair16(a.out0 + row * a.ld0 + ch0, g, pack4(r0), pack4(r1), live);
            }
        }
    }
}

// ---- backward mix A with the word-side reduction as a by-product (round 6).  grid (row chunks, 64-channel groups, B), 512 threads: a workgroup owns ONE
// channel group (the VW' fragments, beta, bias live in registers for all its rows), its waves walk 16-row tiles:
//     d what = dmm * GELU(vpre + bv),   d vpre = dmm * (P VW'^T + beta) * GELU'(vpre + bv)                 (as pwam_mix_kernel<1>)
// and, when `rec` is given, H^T = d what^T P [64 x 32] and s = colsum(d what) [64] of ITS rows: the stored bf16 d what tile and the P tile go through a
// 3 KB LDS image per wave, read back with the transposing read (lane: one channel / word, four consecutive rows) as the operands of
// v_mfma_f32_16x16x16_bf16.  A workgroup leaves one record slab rec[b][chunk][C x 32 H^T | C s] (its channel group's part) that lavt_pwam_lang_bwd1 adds
// in chunk order: no H launch, no reduction launch.
struct Mix1Args {
    const bf16* P;        // [B*T][32]
    const bf16* Wc;       // VW' channel-major [B][C][32]
    const float* beta;    // [B][C]
    const float* xb;      // [C] or null
    const bf16* X;        // vpre [B*T][ldx]
    int64_t ldx;
    const bf16* D;        // dmm [B*T][ldd]
    int64_t ldd;
    bf16* out0;           // d vpre
    int64_t ld0;
    bf16* out1;           // d what
    int64_t ld1;
    float* rec;           // [B][gridDim.x][C * 33] or null
    int T, C;
};
constexpr int MIX1_WAVES = 8;
__global__ __launch_bounds__(MIX1_WAVES * 64) void pwam_mix1_kernel(const Mix1Args a) {
    extern __shared__ __attribute__((aligned(16))) char msm[];          // per wave: d what image [16][64] + P image [16][32] (3 KB); then the waves' sums [8][2112] floats
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int b = blockIdx.z, C = a.C, c0 = 64 * blockIdx.y;
    const int nch = min(64, C - c0);
    const bool two = nch == 64, odd = g & 1;
    const int poff = odd ? 16 + 4 * (g - 1) : 4 * g;
    const int ntiles = (a.T + 15) >> 4;
    const bf16* Wc = a.Wc + ((int64_t)b * C + c0) * 32;
    const float* v0 = a.beta + (int64_t)b * C + c0;
    bf16x8 wa[2][2];
    float c0a[2][2][4], xba[2][2][4];
#pragma unroll
    for (int cp = 0; cp < 2; ++cp)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cl = ((cp == 1 && !two) ? 0 : 32 * cp) + 16 * h;          // (a dead second span re-reads the first: result unused)
            wa[cp][h] = ldg8(Wc + (int64_t)(cl + c16) * 32 + 8 * g);
            const float4 bt = *reinterpret_cast<const float4*>(v0 + cl + 4 * g);
            c0a[cp][h][0] = bt.x; c0a[cp][h][1] = bt.y; c0a[cp][h][2] = bt.z; c0a[cp][h][3] = bt.w;
            float4 xb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.xb) xb = *reinterpret_cast<const float4*>(a.xb + c0 + cl + 4 * g);
            xba[cp][h][0] = xb.x; xba[cp][h][1] = xb.y; xba[cp][h][2] = xb.z; xba[cp][h][3] = xb.w;
        }
    const bool moments = a.rec != nullptr;
    bf16* Dt = reinterpret_cast<bf16*>(msm) + wave * 1536;          // [16][64]
    bf16* Pt = Dt + 1024;                                            // [16][32]
    f32x4 hacc[4][2], sacc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) { hacc[ct][0] = hacc[ct][1] = sacc[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    auto xchg = [&](uint4 L, uint2& p0, uint2& p1) {
        const uint2 send = odd ? make_uint2(L.x, L.y) : make_uint2(L.z, L.w);
        const uint2 got = make_uint2((unsigned)__shfl_xor((int)send.x, 16, 64), (unsigned)__shfl_xor((int)send.y, 16, 64));
        p0 = odd ? got : make_uint2(L.x, L.y);
        p1 = odd ? make_uint2(L.z, L.w) : got;
    };
    // the loads of a tile are requested one tile ahead (a wave walks several tiles when the launch is capped at 32 row chunks)
    auto request = [&](int tile, bf16x8& wd, uint4 (&xr)[2], uint4 (&dr)[2]) {
        const int64_t row = (int64_t)b * a.T + min(tile * 16 + c16, a.T - 1);
        wd = ldg8(a.P + row * 32 + 8 * g);
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            const int ch = c0 + ((cp == 1 && !two) ? 0 : 32 * cp) + poff;
            xr[cp] = *reinterpret_cast<const uint4*>(a.X + row * a.ldx + ch);
            dr[cp] = *reinterpret_cast<const uint4*>(a.D + row * a.ldd + ch);
        }
    };
    const int tstep = gridDim.x * MIX1_WAVES;
    int tile = blockIdx.x * MIX1_WAVES + wave;
    bf16x8 wd_n = {};
    uint4 xr_n[2] = {}, dr_n[2] = {};
    if (tile < ntiles) request(tile, wd_n, xr_n, dr_n);
    for (; tile < ntiles; tile += tstep) {
        const bf16x8 wd = wd_n;
        const uint4 xr[2] = {xr_n[0], xr_n[1]}, dr[2] = {dr_n[0], dr_n[1]};
        if (tile + tstep < ntiles) request(tile + tstep, wd_n, xr_n, dr_n);
        const int t = tile * 16 + c16;
        const bool vr = t < a.T;
        const int64_t row = (int64_t)b * a.T + (vr ? t : a.T - 1);
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            const bool live = vr && (cp == 0 || two);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cp][0], wd, z, 0, 0, 0);
            const f32x4 e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cp][1], wd, z, 0, 0, 0);
            uint2 xp0, xp1, dp0, dp1;
            xchg(xr[cp], xp0, xp1);
            xchg(dr[cp], dp0, dp1);
            float x0[4], x1[4], d0[4], d1[4], r0[4], r1[4], w0[4], w1[4];
            unpack4(xp0, x0); unpack4(xp1, x1); unpack4(dp0, d0); unpack4(dp1, d1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float g0, g1;
                const float a0 = gelu_pair_fast(x0[r] + xba[cp][0][r], g0), a1 = gelu_pair_fast(x1[r] + xba[cp][1][r], g1);
                r0[r] = d0[r] * (e0[r] + c0a[cp][0][r]) * g0; r1[r] = d1[r] * (e1[r] + c0a[cp][1][r]) * g1;
                w0[r] = d0[r] * a0; w1[r] = d1[r] * a1;
            }
            const uint2 q0 = pack4(w0), q1 = pack4(w1);
            store_pair16(a.out0 + row * a.ld0 + c0 + 32 * cp, g, pack4(r0), pack4(r1), live);
            store_pair16(a.out1 + row * a.ld1 + c0 + 32 * cp, g, q0, q1, live);
            if (moments) {
                *reinterpret_cast<uint2*>(Dt + c16 * 64 + 32 * cp + 4 * g) = live ? q0 : make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(Dt + c16 * 64 + 32 * cp + 16 + 4 * g) = live ? q1 : make_uint2(0u, 0u);
            }
        }
        if (moments) {          // (wave-uniform: the transposing reads need every lane)
            *reinterpret_cast<bf16x8*>(Pt + c16 * 32 + 8 * g) = vr ? wd : bf16x8{};
            const unsigned ad = lds_addr(Dt + (4 * g + (c16 >> 2)) * 64 + 4 * (c16 & 3));
            const unsigned ap = lds_addr(Pt + (4 * g + (c16 >> 2)) * 32 + 4 * (c16 & 3));
            u64_t fa0, fa1, fa2, fa3, fb0, fb1;
            asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b64_tr_b16 %0, %6\n\tds_read_b64_tr_b16 %1, %6 offset:32\n\tds_read_b64_tr_b16 %2, %6 offset:64\n\t"
                         "ds_read_b64_tr_b16 %3, %6 offset:96\n\tds_read_b64_tr_b16 %4, %7\n\tds_read_b64_tr_b16 %5, %7 offset:32\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(fa0), "=&v"(fa1), "=&v"(fa2), "=&v"(fa3), "=&v"(fb0), "=&v"(fb1) : "v"(ad), "v"(ap) : "memory");
            const s16x4 A[4] = {__builtin_bit_cast(s16x4, fa0), __builtin_bit_cast(s16x4, fa1), __builtin_bit_cast(s16x4, fa2), __builtin_bit_cast(s16x4, fa3)};
            const s16x4 Bf[2] = {__builtin_bit_cast(s16x4, fb0), __builtin_bit_cast(s16x4, fb1)};
            const s16x4 one4 = {0x3F80, 0x3F80, 0x3F80, 0x3F80};
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                hacc[ct][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A[ct], Bf[0], hacc[ct][0], 0, 0, 0);
                hacc[ct][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A[ct], Bf[1], hacc[ct][1], 0, 0, 0);
                sacc[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A[ct], one4, sacc[ct], 0, 0, 0);
            }
        }
    }
    if (moments) {
        // D layout: row = channel 16 ct + 4 g + r, column = word 16 wt + c16 (s: every column the same)
        __syncthreads();
        float* red = reinterpret_cast<float*>(msm);          // [waves][64 * 32 H^T | 64 s]
        float* mine = red + wave * 2112;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = 16 * ct + 4 * g + r;
                mine[ch * 32 + c16] = hacc[ct][0][r];
                mine[ch * 32 + 16 + c16] = hacc[ct][1][r];
                if (c16 == 0) mine[2048 + ch] = sacc[ct][r];
            }
        __syncthreads();
        float* slab = a.rec + ((int64_t)b * gridDim.x + blockIdx.x) * C * 33;
        for (int e = tid; e < 2112; e += MIX1_WAVES * 64) {
            float v = red[e];
#pragma unroll
            for (int w = 1; w < MIX1_WAVES; ++w) v += red[w * 2112 + e];
            if (e < 2048) { if ((e >> 5) < nch) slab[(int64_t)(c0 + (e >> 5)) * 32 + (e & 31)] = v; }
            else if (e - 2048 < nch) slab[(int64_t)C * 32 + c0 + e - 2048] = v;
        }
    }
}

// ================================================================================================ language side
// forward: VW = V Wo^T on the matrix cores, Cov_T(P) from the second-moment matrix, var_w -> VW' in both layouts, beta = -Pbar VW'.
// grid (C / 16, B), 256 threads: a workgroup owns 16 channels, its four waves a quarter of the reduction each (round 5: as 64 channels per workgroup
// with the whole reduction in every wave the C = 1024 launch was 32 workgroups x 4 dependent rounds of cold loads of Wo -- 18 us; every load of a
// wave is now one round, requested in front of everything else).
constexpr int WORDS_MAX_RECORDS = 32;
constexpr int LF_MAXK = 8;          // k-steps per wave held in registers per round (C <= 1024: one round)
__global__ __launch_bounds__(256) void pwam_lang_fwd_kernel(const bf16* __restrict__ V, int64_t ldv, const bf16* __restrict__ Wo, const float* __restrict__ PP,
                                                            const float* __restrict__ sumP, bf16* __restrict__ VWc, bf16* __restrict__ VWw, float* __restrict__ beta,
                                                            float* __restrict__ rw, float* __restrict__ pbar_out, float* __restrict__ cov_out, int T, int C, float eps,
                                                            const float* __restrict__ rec, int nrec) {
    __shared__ float part_vw[4][16][33];
    __shared__ float vw[16][33];
    __shared__ float cov[32][33];
    __shared__ float pb[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int b = blockIdx.y, c0 = blockIdx.x * 16;
    const float invT = 1.0f / (float)T;
    // this wave's k-steps [k_lo, k_hi) of the C / 32
    const int ksteps = C >> 5, per = (ksteps + 3) >> 2;
    const int k_lo = min(wave * per, ksteps), k_hi = min(k_lo + per, ksteps);
    const int ch = c0 + c16;
    const bool vc = ch < C;
    const bf16* wp = Wo + (int64_t)(vc ? ch : C - 1) * C + 8 * g;
    const bf16* vp0 = V + ((int64_t)b * 32 + c16) * ldv + 8 * g;
    const bf16* vp1 = vp0 + 16 * ldv;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    bf16x8 wf[LF_MAXK], a0[LF_MAXK], a1[LF_MAXK];
    // first round of operand loads (clamped k-step: a dead slot re-reads the last live one and is not used)
#pragma unroll
    for (int u = 0; u < LF_MAXK; ++u) {
        const int ks = min(k_lo + u, max(k_hi - 1, 0));
        wf[u] = ldg8(wp + 32 * ks); a0[u] = ldg8(vp0 + 32 * ks); a1[u] = ldg8(vp1 + 32 * ks);
    }
    // second-moment matrix and word means: 4 + 5 independent loads per thread, no LDS hop in between
    if (PP) {
        float ppv[4], sj[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { ppv[i] = PP[(int64_t)b * 1024 + tid + 256 * i]; sj[i] = sumP[b * 32 + (tid >> 5) + 8 * i]; }
        const float sk = sumP[b * 32 + (tid & 31)];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const float cv = ppv[i] * invT - (sj[i] * invT) * (sk * invT);
            cov[e >> 5][e & 31] = cv;
            if (blockIdx.x == 0) cov_out[(int64_t)b * 1024 + e] = cv;
        }
        if (tid < 32) {
            pb[tid] = sk * invT;
            if (blockIdx.x == 0) pbar_out[b * 32 + tid] = sk * invT;
        }
    } else {
        // (ABI v7) the second moments arrive as per-workgroup records of lavt_pwam_words_fwd_moments: [nrec][1024 P^T P | 32 colsum(P)] per sample, added here in
        // index order (every workgroup of this launch repeats the sum: nrec x C / 16 record reads of 4 KB per sample, L2-resident -- no launch, no hand-over)
        const float* rb = rec + (int64_t)b * nrec * 1056;
        float4 pp4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float sp = 0.f;
        for (int r0 = 0; r0 < nrec; r0 += 16) {
            float4 v[16];
            float w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int r = min(r0 + u, nrec - 1);
                v[u] = *reinterpret_cast<const float4*>(rb + (int64_t)r * 1056 + 4 * tid);
                w[u] = rb[(int64_t)r * 1056 + 1024 + (tid & 31)];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (r0 + u < nrec) { pp4.x += v[u].x; pp4.y += v[u].y; pp4.z += v[u].z; pp4.w += v[u].w; sp += w[u]; }
        }
        if (tid < 32) pb[tid] = sp * invT;
        __syncthreads();
        const int j = tid >> 3, k0 = 4 * (tid & 7);          // float4 tid = row tid / 8, columns 4 (tid % 8) ..
        const float pj = pb[j];
        const float cv[4] = {pp4.x * invT - pj * pb[k0], pp4.y * invT - pj * pb[k0 + 1], pp4.z * invT - pj * pb[k0 + 2], pp4.w * invT - pj * pb[k0 + 3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cov[j][k0 + i] = cv[i];
            if (blockIdx.x == 0) cov_out[(int64_t)b * 1024 + 4 * tid + i] = cv[i];
        }
        if (blockIdx.x == 0 && tid < 32) pbar_out[b * 32 + tid] = pb[tid];
    }
#pragma unroll
    for (int u = 0; u < LF_MAXK; ++u)
        if (k_lo + u < k_hi) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[u], wf[u], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[u], wf[u], acc[1], 0, 0, 0);
        }
    for (int ks = k_lo + LF_MAXK; ks < k_hi; ++ks) {          // (C > 1024)
        const bf16x8 w = ldg8(wp + 32 * ks);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ldg8(vp0 + 32 * ks), w, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ldg8(vp1 + 32 * ks), w, acc[1], 0, 0, 0);
    }
    // lane: channel c16, words 4g + r (tile 0) and 16 + 4g + r (tile 1)
#pragma unroll
    for (int r = 0; r < 4; ++r) { part_vw[wave][c16][4 * g + r] = acc[0][r]; part_vw[wave][c16][16 + 4 * g + r] = acc[1][r]; }
    __syncthreads();
    for (int e = tid; e < 512; e += 256) {
        const int c = e >> 5, j = e & 31;
        vw[c][j] = (part_vw[0][c][j] + part_vw[1][c][j]) + (part_vw[2][c][j] + part_vw[3][c][j]);          // (fixed order)
    }
    __syncthreads();
    // thread (channel tid / 16, word pair tid % 16)
    const int c = tid >> 4, j0 = 2 * (tid & 15);
    float t0 = 0.f, t1 = 0.f;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) { const float v = vw[c][k]; t0 += cov[j0][k] * v; t1 += cov[j0 + 1][k] * v; }
    float var = vw[c][j0] * t0 + vw[c][j0 + 1] * t1;
    var += __shfl_xor(var, 1, 64); var += __shfl_xor(var, 2, 64); var += __shfl_xor(var, 4, 64); var += __shfl_xor(var, 8, 64);
    const float rs = rsqrtf(fmaxf(var, 0.f) + eps);
    const float f0 = bf16_round(vw[c][j0] * rs), f1 = bf16_round(vw[c][j0 + 1] * rs);
    float bsum = -(pb[j0] * f0 + pb[j0 + 1] * f1);
    bsum += __shfl_xor(bsum, 1, 64); bsum += __shfl_xor(bsum, 2, 64); bsum += __shfl_xor(bsum, 4, 64); bsum += __shfl_xor(bsum, 8, 64);
    if (c0 + c < C) {
        *reinterpret_cast<uint32_t*>(VWc + ((int64_t)b * C + c0 + c) * 32 + j0) = pack_bf16x2(f0, f1);
        VWw[((int64_t)b * 32 + j0) * C + c0 + c] = (bf16)f0;
        VWw[((int64_t)b * 32 + j0 + 1) * C + c0 + c] = (bf16)f1;
        if ((tid & 15) == 0) { beta[(int64_t)b * C + c0 + c] = bsum; rw[(int64_t)b * C + c0 + c] = rs; }
    }
}

// backward 1: from H^T = dwhat^T P [C][32] and s = colsum(dwhat) [C]: the IN-backward constants a, b per channel, dVW [32][C] (bf16, word-major),
// and this workgroup's share of the sums over channels Q = VW' diag(b) VW'^T [32][32], u = VW' a [32] (one record per workgroup, no atomics).
__global__ __launch_bounds__(256) void pwam_lang_bwd1_kernel(const float* __restrict__ HT, const float* __restrict__ s, const bf16* __restrict__ VWc,
                                                             const float* __restrict__ rw, const float* __restrict__ pbar, const float* __restrict__ cov_in,
                                                             bf16* __restrict__ dVW, float* __restrict__ Qf, int T, int C, const float* __restrict__ rec, int nrec) {
    // grid (min(C / 16, 8), B): 16 channels per workgroup, thread (channel tid / 16, word pair tid % 16).  (64 channels per workgroup with 8 words per
    // thread ran 13 us at every size: ~1 300 dependent LDS operations per thread.)
    __shared__ float vw[16][33];
    __shared__ float h[16][33];
    __shared__ float cov[32][33];
    __shared__ float pb[32];
    __shared__ float av[16], bv[16];
    const int tid = threadIdx.x, b = blockIdx.y;
    const float Tf = (float)T, invT = 1.0f / Tf;
    float qacc[4] = {0.f, 0.f, 0.f, 0.f}, uacc = 0.f;
    // (round 5) the global loads of a trip are requested one trip ahead (registers), so only the first trip's round trip is exposed: as loads at the
    // head of every trip + the per-channel scalars behind its barrier, the C = 1024 launch (4 trips) was 8 dependent round trips, 14.5 us
    const int c = tid >> 4, j0 = 2 * (tid & 15);
    float n_vw[2], n_h[2], n_sc, n_rs;
    auto request = [&](int c0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + 256 * i;
            n_vw[i] = (float)VWc[((int64_t)b * C + c0 + (e >> 5)) * 32 + (e & 31)];
        }
        n_rs = rw[(int64_t)b * C + c0 + c];
        if (rec == nullptr) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + 256 * i;
                n_h[i] = HT[((int64_t)b * C + c0 + (e >> 5)) * 32 + (e & 31)];
            }
            n_sc = s[(int64_t)b * C + c0 + c];
        } else {
            // (ABI v7) H^T and s arrive as per-workgroup records of lavt_pwam_mix1: [nrec][C x 32 | C] per sample, added here in record order, sixteen in flight
            const float* rb = rec + (int64_t)b * nrec * C * 33;
            n_h[0] = n_h[1] = n_sc = 0.f;
            for (int r0 = 0; r0 < nrec; r0 += 16) {
                float v0[16], v1[16], v2[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const float* q = rb + (int64_t)min(r0 + u, nrec - 1) * C * 33;
                    v0[u] = q[(int64_t)(c0 + (tid >> 5)) * 32 + (tid & 31)];
                    v1[u] = q[(int64_t)(c0 + 8 + (tid >> 5)) * 32 + (tid & 31)];
                    v2[u] = q[(int64_t)C * 32 + c0 + c];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (r0 + u < nrec) { n_h[0] += v0[u]; n_h[1] += v1[u]; n_sc += v2[u]; }
            }
        }
    };
    request(min((int)blockIdx.x * 16, C - 16));
    if (tid < 32) pb[tid] = pbar[b * 32 + tid];
#pragma unroll
    for (int e = tid; e < 1024; e += 256) cov[e >> 5][e & 31] = cov_in[(int64_t)b * 1024 + e];
    for (int c0 = blockIdx.x * 16; c0 < C; c0 += gridDim.x * 16) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i;
        vw[e >> 5][e & 31] = n_vw[i];
        h[e >> 5][e & 31] = n_h[i];
    }
    const float sc = n_sc, rs = n_rs;
    if (c0 + (int)gridDim.x * 16 < C) request(c0 + gridDim.x * 16);
    __syncthreads();
    float bs = vw[c][j0] * (h[c][j0] - pb[j0] * sc) + vw[c][j0 + 1] * (h[c][j0 + 1] - pb[j0 + 1] * sc);
    bs += __shfl_xor(bs, 1, 64); bs += __shfl_xor(bs, 2, 64); bs += __shfl_xor(bs, 4, 64); bs += __shfl_xor(bs, 8, 64);
    const float bc = bs * invT, ac = sc * invT;
    if ((tid & 15) == 0) { av[c] = ac; bv[c] = bc; }
    float t0 = 0.f, t1 = 0.f;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) { const float v = vw[c][k]; t0 += cov[j0][k] * v; t1 += cov[j0 + 1][k] * v; }
    dVW[((int64_t)b * 32 + j0) * C + c0 + c] = (bf16)(rs * (h[c][j0] - Tf * pb[j0] * ac - Tf * bc * t0));
    dVW[((int64_t)b * 32 + j0 + 1) * C + c0 + c] = (bf16)(rs * (h[c][j0 + 1] - Tf * pb[j0 + 1] * ac - Tf * bc * t1));
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, k = e >> 5, j = e & 31;
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) qacc[i] += vw[cc][k] * bv[cc] * vw[cc][j];
    }
    if (tid < 32) {
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) uacc += vw[cc][tid] * av[cc];
    }
    }
    // this workgroup's share of Q and u: plain stores, summed in a fixed order by the consumer (lavt_pwam_words_bwd) -- fp32 atomics here made the
    // bf16 copy of Q, and through it the gradients, differ between runs by a rounding flip (1.5e-3 of the largest gradient, seen eager vs hipGraph)
    float* dst = Qf + ((int64_t)b * gridDim.x + blockIdx.x) * 1056;
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[tid + 256 * i] = qacc[i];
    if (tid < 32) dst[1024 + tid] = uacc;
}

// backward 2: from G = dS^T q [32][C] (raw q) and colsum(dS) [32]: dK, the dq constants c0, c1 and K'' in channel-major layout.
__global__ __launch_bounds__(256) void pwam_lang_bwd2_kernel(const float* __restrict__ G, const float* __restrict__ sdS, const bf16* __restrict__ K, int64_t ldk,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd, bf16* __restrict__ dK, int64_t lddk,
                                                             bf16* __restrict__ K2c, float* __restrict__ c0o, float* __restrict__ c1o, int T, int C, float alpha) {
    __shared__ float sd[32];
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < 32) sd[threadIdx.x] = sdS[b * 32 + threadIdx.x];
    __syncthreads();
    if (c >= C) return;
    const float mu = mean[(int64_t)b * C + c], rq = rstd[(int64_t)b * C + c], invT = 1.0f / (float)T;
    float a2 = 0.f, b2 = 0.f, k2[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const float kv = (float)K[((int64_t)b * 32 + j) * ldk + c];
        const float gh = (G[((int64_t)b * 32 + j) * C + c] - sd[j] * mu) * rq;
        dK[((int64_t)b * 32 + j) * lddk + c] = (bf16)(alpha * gh);
        a2 += sd[j] * kv;
        b2 += kv * gh;
        k2[j] = alpha * rq * kv;
    }
    a2 *= alpha * invT; b2 *= alpha * invT;
    const float c1 = rq * rq * b2;
    c1o[(int64_t)b * C + c] = c1;
    c0o[(int64_t)b * C + c] = -rq * a2 + mu * c1;
    bf16* dst = K2c + ((int64_t)b * C + c) * 32;
#pragma unroll
    for (int j = 0; j < 32; j += 8) *reinterpret_cast<uint4*>(dst + j) = f_to_chunk<bf16>(k2 + j);
}

int rows_grid(int tiles_per_sample, int B, int waves) {
    int gx = (tiles_per_sample + waves - 1) / waves;            // one tile per wave per pass
    const int cap = 1024 / (B > 0 ? B : 1);
    if (gx > cap) gx = cap;
    return gx < 1 ? 1 : gx;
}

}  // namespace

#define ST ((hipStream_t)stream)

// workgroups per sample of lavt_pwam_lang_bwd1 = partial [1024 Q | 32 u] records per sample that lavt_pwam_words_bwd sums
// workgroups (= partial Q / u records) of lavt_pwam_lang_bwd1 per sample: 16 channels per workgroup per trip; up to 16 workgroups (C = 1024: four
// trips each instead of eight -- every trip is a barrier-separated round of loads -- and the consumer sums the records eight at a time)
extern "C" int lavt_pwam_q_parts(int C) { const int n = C / 16; return n < 16 ? (n < 1 ? 1 : n) : 16; }

// LDS of the words kernels: word matrix + per-word constants + -Q + the q statistics + a 1 KB transposition image per wave; the forward's reduction
// scratch ([waves][1056] floats) aliases it
static size_t words_lds(int C, int nt, bool moments) {
    const size_t base = (size_t)32 * (C + 8) * 2 + 128 + 32 * 40 * 2 + (size_t)C * 8 + (size_t)(nt / 64) * 1024;
    const size_t red = (size_t)(nt / 64) * 1056 * 4;
    return moments && red > base ? red : base;
}
template <int NT, int PRE> static int launch_words_fwd(const WordsArgs& a, int B, size_t lds, hipStream_t st) {
    static size_t reserved = 0;
    if (lds > 65536 && lds > reserved) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pwam_words_kernel<false, NT, PRE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { lavt_set_error("lavt_pwam_words_fwd: cannot reserve LDS"); return LAVT_ERR_LAUNCH; }
        reserved = lds;
    }
    int gx = rows_grid((a.T + 15) / 16, B, NT / 64);
    if (a.rec && gx > WORDS_MAX_RECORDS) gx = WORDS_MAX_RECORDS;          // (records per sample the consumer adds, 16 loads per round: the waves walk further instead)
    hipLaunchKernelGGL((pwam_words_kernel<false, NT, PRE>), dim3(gx, B), dim3(NT), lds, st, a);
    return LAVT_OK;
}
// 1024-thread workgroups where 256-thread ones would leave more than 16 records per sample for the consumer (C <= 256: registers, one pass of the word matrix)
static int words_fwd_threads(int T, int C, int B) { return (C <= 256 && rows_grid((T + 15) / 16, B, 4) > 16) ? 1024 : 256; }
extern "C" int lavt_pwam_words_records(int B, int T, int C) {
    const int nt = words_fwd_threads(T, C, B), gx = rows_grid((T + 15) / 16, B, nt / 64);
    return gx > WORDS_MAX_RECORDS ? WORDS_MAX_RECORDS : gx;
}

extern "C" int lavt_pwam_words_fwd(const void* q, int64_t ldq, const void* K, int64_t ldk, const float* mean, const float* rstd, const float* maskbias,
                                   void* P, int B, int T, int C, int n_l, float alpha, void* stream) {
    return lavt_pwam_words_fwd_moments(q, ldq, K, ldk, mean, rstd, maskbias, P, nullptr, B, T, C, n_l, alpha, stream);
}

extern "C" int lavt_pwam_words_fwd_moments(const void* q, int64_t ldq, const void* K, int64_t ldk, const float* mean, const float* rstd, const float* maskbias,
                                           void* P, float* rec, int B, int T, int C, int n_l, float alpha, void* stream) {
    LAVT_CHECK_ARG(q && K && mean && rstd && maskbias && P && B > 0 && T > 0 && C >= 32 && C % 32 == 0 && C <= 2048 && n_l > 0 && n_l <= 32 && ldq % 8 == 0 && ldk % 8 == 0,
                   "lavt_pwam_words_fwd: bad arguments (C %% 32 == 0, C <= 2048, 1 <= n_l <= 32, 16-byte aligned rows)");
    WordsArgs a{};
    a.X = (const bf16*)q; a.ldx = ldq; a.Wsrc = (const bf16*)K; a.ldw = ldk; a.mean = mean; a.rstd = rstd; a.vec = maskbias; a.out = (bf16*)P;
    a.T = T; a.C = C; a.n_l = n_l; a.alpha = alpha; a.rec = rec;
    const int nt = rec ? words_fwd_threads(T, C, B) : 256;
    const size_t lds = words_lds(C, nt, rec != nullptr);
    const int rc = nt == 1024 ? launch_words_fwd<1024, 8>(a, B, lds, ST) : launch_words_fwd<256, 16>(a, B, lds, ST);
    if (rc != LAVT_OK) return rc;
    LAVT_CHECK_LAUNCH("lavt_pwam_words_fwd");
    return LAVT_OK;
}

extern "C" int lavt_pwam_words_bwd(const void* dwhat, int64_t ldx, const void* VWw, const float* Qp, const float* pbar, const void* P,
                                   void* dS, int B, int T, int C, void* stream) {
    LAVT_CHECK_ARG(dwhat && VWw && Qp && pbar && P && dS && B > 0 && T > 0 && C >= 32 && C % 32 == 0 && C <= 2048 && ldx % 8 == 0, "lavt_pwam_words_bwd: bad arguments");
    WordsArgs a{};
    a.X = (const bf16*)dwhat; a.ldx = ldx; a.Wsrc = (const bf16*)VWw; a.ldw = C; a.Qf = Qp; a.pbar = pbar; a.P = (const bf16*)P; a.out = (bf16*)dS;
    a.T = T; a.C = C; a.n_l = lavt_pwam_q_parts(C); a.alpha = 1.f;
    const size_t lds = words_lds(C, 256, false);
    static size_t reserved = 0;
    if (lds > 65536 && lds > reserved) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pwam_words_kernel<true, 256, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { lavt_set_error("lavt_pwam_words_bwd: cannot reserve LDS"); return LAVT_ERR_LAUNCH; }
        reserved = lds;
    }
    hipLaunchKernelGGL((pwam_words_kernel<true, 256, 16>), dim3(rows_grid((T + 15) / 16, B, 4), B), dim3(256), lds, ST, a);
    LAVT_CHECK_LAUNCH("lavt_pwam_words_bwd");
    return LAVT_OK;
}

extern "C" int lavt_pwam_mix(int mode, const void* Wd, const void* Wc, const float* v0, const float* v1, const float* xbias, const void* X, int64_t ldx, const void* D, int64_t ldd,
                             void* out0, int64_t ld0, void* out1, int64_t ld1, int B, int T, int C, void* stream) {
    LAVT_CHECK_ARG(Wd && Wc && v0 && X && out0 && B > 0 && T > 0 && C >= 32 && C % 32 == 0 && ldx % 8 == 0 && ld0 % 8 == 0 && mode >= 0 && mode <= 2 &&
                   (mode != 1 || (D && out1 && ldd % 8 == 0 && ld1 % 8 == 0)) && (mode != 2 || v1), "lavt_pwam_mix: bad arguments");
    MixArgs a{};
    a.Wd = (const bf16*)Wd; a.Wc = (const bf16*)Wc; a.v0 = v0; a.v1 = v1; a.xb = xbias; a.X = (const bf16*)X; a.ldx = ldx; a.D = (const bf16*)D; a.ldd = ldd;
    a.out0 = (bf16*)out0; a.ld0 = ld0; a.out1 = (bf16*)out1; a.ld1 = ld1; a.T = T; a.C = C;
    const long items = (long)((T + 15) / 16) * ((C + 63) / 64);
    const dim3 grid((unsigned)(items / 4 + 1 > 8192 ? 8192 : items / 4 + 1), B);
    if (mode == 0) hipLaunchKernelGGL(pwam_mix_kernel<0>, grid, dim3(256), 0, ST, a);
    else if (mode == 1) hipLaunchKernelGGL(pwam_mix_kernel<1>, grid, dim3(256), 0, ST, a);
    else hipLaunchKernelGGL(pwam_mix_kernel<2>, grid, dim3(256), 0, ST, a);
    LAVT_CHECK_LAUNCH("lavt_pwam_mix");
    return LAVT_OK;
}

extern "C" int lavt_pwam_lang_fwd(const void* V, int64_t ldv, const void* Wo, const float* PP, const float* sumP, void* VWc, void* VWw, float* beta, float* rw,
                                  float* pbar, float* cov, int B, int T, int C, float eps, void* stream) {
    return lavt_pwam_lang_fwd_records(V, ldv, Wo, PP, sumP, nullptr, 0, VWc, VWw, beta, rw, pbar, cov, B, T, C, eps, stream);
}

extern "C" int lavt_pwam_lang_fwd_records(const void* V, int64_t ldv, const void* Wo, const float* PP, const float* sumP, const float* rec, int nrec, void* VWc, void* VWw,
                                          float* beta, float* rw, float* pbar, float* cov, int B, int T, int C, float eps, void* stream) {
    LAVT_CHECK_ARG(V && Wo && ((PP && sumP) || (rec && nrec > 0)) && VWc && VWw && beta && rw && pbar && cov && B > 0 && T > 0 && C >= 32 && C % 32 == 0 && ldv % 8 == 0, "lavt_pwam_lang_fwd: bad arguments");
    hipLaunchKernelGGL(pwam_lang_fwd_kernel, dim3(cdiv(C, 16), B), dim3(256), 0, ST, (const bf16*)V, ldv, (const bf16*)Wo, PP, sumP, (bf16*)VWc, (bf16*)VWw, beta, rw, pbar, cov, T, C, eps, PP ? nullptr : rec, nrec);
    LAVT_CHECK_LAUNCH("lavt_pwam_lang_fwd");
    return LAVT_OK;
}

extern "C" int lavt_pwam_mix1(const void* P, const void* VWc, const float* beta, const float* xbias, const void* X, int64_t ldx, const void* D, int64_t ldd, void* dvpre,
                              int64_t ld0, void* dwhat, int64_t ld1, float* rec, int B, int T, int C, void* stream) {
    LAVT_CHECK_ARG(P && VWc && beta && X && D && dvpre && dwhat && B > 0 && T > 0 && C >= 32 && C % 32 == 0 && ldx % 8 == 0 && ldd % 8 == 0 && ld0 % 8 == 0 && ld1 % 8 == 0,
                   "lavt_pwam_mix1: bad arguments");
    Mix1Args a{};
    a.P = (const bf16*)P; a.Wc = (const bf16*)VWc; a.beta = beta; a.xb = xbias; a.X = (const bf16*)X; a.ldx = ldx; a.D = (const bf16*)D; a.ldd = ldd;
    a.out0 = (bf16*)dvpre; a.ld0 = ld0; a.out1 = (bf16*)dwhat; a.ld1 = ld1; a.rec = rec; a.T = T; a.C = C;
    const size_t lds = (size_t)MIX1_WAVES * 2112 * 4;
    static bool reserved = false;
    if (!reserved) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pwam_mix1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { lavt_set_error("lavt_pwam_mix1: cannot reserve LDS"); return LAVT_ERR_LAUNCH; }
        reserved = true;
    }
    hipLaunchKernelGGL(pwam_mix1_kernel, dim3(lavt_pwam_mix1_records(B, T, C), (C + 63) / 64, B), dim3(MIX1_WAVES * 64), lds, ST, a);
    LAVT_CHECK_LAUNCH("lavt_pwam_mix1");
    return LAVT_OK;
}
// row chunks of lavt_pwam_mix1 = records per sample that lavt_pwam_lang_bwd1_records adds: one 16-row tile per wave per pass, at most 32
extern "C" int lavt_pwam_mix1_records(int B, int T, int C) {
    (void)B; (void)C;
    const int rc = ((T + 15) / 16 + MIX1_WAVES - 1) / MIX1_WAVES;
    return rc > 32 ? 32 : (rc < 1 ? 1 : rc);
}

extern "C" int lavt_pwam_lang_bwd1(const float* HT, const float* s, const void* VWc, const float* rw, const float* pbar, const float* cov, void* dVW, float* Qp,
                                   int B, int T, int C, void* stream) {
    return lavt_pwam_lang_bwd1_records(HT, s, nullptr, 0, VWc, rw, pbar, cov, dVW, Qp, B, T, C, stream);
}
extern "C" int lavt_pwam_lang_bwd1_records(const float* HT, const float* s, const float* rec, int nrec, const void* VWc, const float* rw, const float* pbar, const float* cov,
                                           void* dVW, float* Qp, int B, int T, int C, void* stream) {
    LAVT_CHECK_ARG(((HT && s) || (rec && nrec > 0)) && VWc && rw && pbar && cov && dVW && Qp && B > 0 && T > 0 && C >= 32 && C % 16 == 0, "lavt_pwam_lang_bwd1: bad arguments");
    hipLaunchKernelGGL(pwam_lang_bwd1_kernel, dim3(lavt_pwam_q_parts(C), B), dim3(256), 0, ST, HT, s, (const bf16*)VWc, rw, pbar, cov, (bf16*)dVW, Qp, T, C, HT ? nullptr : rec, nrec);
    LAVT_CHECK_LAUNCH("lavt_pwam_lang_bwd1");
    return LAVT_OK;
}

extern "C" int lavt_pwam_lang_bwd2(const float* G, const float* sdS, const void* K, int64_t ldk, const float* mean, const float* rstd, void* dK, int64_t lddk, void* K2c,
                                   float* c0, float* c1, int B, int T, int C, float alpha, void* stream) {
    LAVT_CHECK_ARG(G && sdS && K && mean && rstd && dK && K2c && c0 && c1 && B > 0 && T > 0 && C >= 32, "lavt_pwam_lang_bwd2: bad arguments");
    hipLaunchKernelGGL(pwam_lang_bwd2_kernel, dim3(cdiv(C, 256), B), dim3(256), 0, ST, G, sdS, (const bf16*)K, ldk, mean, rstd, (bf16*)dK, lddk, (bf16*)K2c, c0, c1, T, C, alpha);
    LAVT_CHECK_LAUNCH("lavt_pwam_lang_bwd2");
    return LAVT_OK;
}
