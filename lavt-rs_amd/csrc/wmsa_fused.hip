// W-MSA / SW-MSA forward as ONE kernel per Swin block (bf16, gfx950): LayerNorm (norm1) + window partition / shift / padding + qkv projection +
// relative-position-bias attention + softmax + PV.  Reference arithmetic: SwinTransformerBlock.forward lib/backbone.py:201-217 (norm1, pad, roll,
// window_partition), WindowAttention.forward :113-140 (qkv, scale, bias, mask, softmax, attn @ v).  The proj GEMM that follows scatters the rows back
// (window_reverse / roll / crop) and adds the residual, as before.
//
// One workgroup (8 waves) per (window, head):
//   phase 1  [q | k | v]_head = LN(x_window) W_head^T + b_head as a 144 x 96 x C GEMM: the window's token rows are gathered through the row map by
//            LDS-DMA (global_load_lds_dwordx4, -1 = padded token = zero page) into a 2-stage ring of 64-wide K tiles, the head's 96 weight rows likewise.
//            LayerNorm is applied algebraically: the MFMA contracts the RAW rows with gamma-folded weights, the row statistics are accumulated from
//            the same LDS tiles while they are resident (v_dot2c_f32_bf16: exact products, fp32 sums over C <= 1024 values), and the epilogue forms
//                rstd_r (acc - mu_r wsum_n) + (b_n + sum_k beta_k W_nk)        [padded tokens: b_n only -- the reference pads AFTER norm1]
//            q, k, v go to LDS (and to HBM once, for the backward pass); the normalised rows xn -- the qkv weight gradient's operand -- are written
//            by column slices, one slice per head.
//   phase 2  the attention core of wattn_fwd_mfma (attention_mfma.hip) on the LDS copies: S^T = K Q^T, in-register softmax, O^T = V^T P^T.
// The [M, C] LayerNorm output, the LayerNorm launch, the qkv GEMM launch and the qkv re-read disappear from the forward pass.
#include <stdlib.h>

#include "gemm_common.h"

using namespace lavt_gemm;

namespace {

constexpr int HD = 32;
constexpr int F_LD = 40;        // bf16 elements per LDS row of Q / K / V (80 B)
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__device__ __forceinline__ void dma16(const void* src, void* lds_dst) { __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)lds_dst, 16, 0, 0); }
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ bf16x8 join4(bf16x4 lo, bf16x4 hi) {
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
__device__ __forceinline__ bf16x8 lds_row8(const bf16* s, int ld, int row, int k0) { return *reinterpret_cast<const bf16x8*>(s + row * ld + k0); }
__device__ __forceinline__ void store_head_row16(bf16* row_head, int g, uint2 p0, uint2 p1, bool valid) {
    const bool odd = g & 1;
    const uint2 send = odd ? p0 : p1;
    const uint2 got = make_uint2((unsigned)__shfl_xor((int)send.x, 16, 64), (unsigned)__shfl_xor((int)send.y, 16, 64));
    const uint4 out = odd ? make_uint4(got.x, got.y, p1.x, p1.y) : make_uint4(p0.x, p0.y, got.x, got.y);
    if (valid) *reinterpret_cast<uint4*>(row_head + (odd ? 16 + 4 * (g - 1) : 4 * g)) = out;
}

struct WmsaArgs {
    const bf16* x;            // [tokens][C] residual stream (input of norm1)
    const int32_t* wmap;      // [nwin * N]: token row of window position, -1 = padded token
    const bf16* Wg;           // [3C][C]: gamma-folded qkv weight (lavt_ln_fold)
    const float* wsum;        // [3C]: sum_k Wg[n][k]
    const float* biasp;       // [3C]: b_n + sum_k beta_k W[n][k]
    const float* bias;        // [3C]: b_n
    const float* gamma;       // [C]
    const float* beta;        // [C]
    const float* table;       // [(2ws-1)^2][heads]
    const int8_t* region;     // [nw_img][N] or null
    bf16* out;                // [nwin * N][C] attention output, window order
    float* lse;               // [nwin][heads][N]
    bf16* qkv;                // [nwin * N][3C] (saved for backward)
    bf16* xn;                 // [tokens][C] LayerNorm output (operand of the qkv weight gradient)
    float* mean;              // [tokens]
    float* rstd;              // [tokens]
    const void* zeros;
    int nw_img, wh, ww, nwin, N, heads, C;
    float eps, scale;
    uint4* fill;              // rider: this many 16-byte words are set to zero by extra workgroups (a slice of the step's gradient buffer)
    int64_t fill_words;
    int fill_blocks;
};

template <int NT, bool REGION, bool FULL>
__global__ __launch_bounds__(512, 4) void wmsa_fwd_fused_kernel(const WmsaArgs a) {
    // Riders: the zero fill of the step's flat gradient buffer (475 MB for Swin-B, 58 us at the HBM rate) is needed by nothing before backward, yet as
    // its own launch it headed the captured chain.  The forward launches of the stage-2 blocks (288 workgroups for 512 resident slots) each zero a slice
    // of it with extra workgroups (lavt_wmsa_fwd_rider).
    if ((int)blockIdx.x >= a.nwin * a.heads) {
        const int64_t rb = (int64_t)blockIdx.x - (int64_t)a.nwin * a.heads;
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        for (int64_t e = rb * 512 + threadIdx.x; e < a.fill_words; e += (int64_t)a.fill_blocks * 512) a.fill[e] = z;
        return;
    }
    // 8 waves (two workgroups per CU at <= 128 registers).  With 4 waves the K loop of phase 1 was bound by the ISSUE cost of the LDS-DMA instructions
    // (8 per wave per K tile at ~130 cycles each against 36 MFMAs: 10.4 of the kernel's 26 us at C = 512); here a wave issues 4 and owns a
    // (m-tile group, n half) block of the 144 x 96 product.
    constexpr int NW = 8, NTHR = NW * 64;
    constexpr int KS = (NT + 1) / 2, NP = KS * 32;
    constexpr int TR = ((NT * 16 + 96 + 63) / 64) * 64;     // rows of the combined [window rows | 96 weight rows] K tile: whole DMA instructions per wave
    constexpr int MR = TR - 96, DI = TR / 64;                // window-row slots, DMA instructions per wave per K tile
    constexpr int STAGE = TR * 128;
    constexpr int MI = (NT + 3) / 4;                         // m-tiles (16 window positions) per wave, at most
    constexpr int SU = (MR * 8 + NTHR - 1) / NTHR;           // 16-byte chunks of the window rows per thread (row statistics)
    static_assert(3 * NP * F_LD * 2 <= 2 * STAGE && SU * 64 <= TR, "Q / K / V alias the ring; statistics reads stay inside a stage");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* Qs = reinterpret_cast<bf16*>(smem);                // (after phase 1) [NP][F_LD] each
    bf16* Ks = Qs + NP * F_LD;
    bf16* Vs = Ks + NP * F_LD;
    int* bs = reinterpret_cast<int*>(smem + 2 * STAGE);
    float* mu = reinterpret_cast<float*>(bs + NP);
    float* rsd = mu + MR;
    int* srcs = reinterpret_cast<int*>(rsd + MR);
    uint8_t* Rs = reinterpret_cast<uint8_t*>(srcs + MR);
    float* tab = reinterpret_cast<float*>(Rs + NP);
    float* evec = tab + ((2 * a.wh - 1) * (2 * a.ww - 1) + 3) / 4 * 4;      // [3][96]: wsum | biasp | bias of this head's q, k, v columns

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int w = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const int C = a.C, N = a.N, wh = a.wh, ww = a.ww;
    const int R = (2 * wh - 1) * (2 * ww - 1);
    const int centre = (wh - 1) * (2 * ww - 1) + (ww - 1);
    const bf16* Z = reinterpret_cast<const bf16*>(a.zeros);

    for (int e = tid; e < MR; e += NTHR) srcs[e] = e < N ? a.wmap[(int64_t)w * N + e] : -1;
    __syncthreads();

    // ---- phase 1: [q | k | v]_head = LN(x_window) W_head^T ----------------------------------------------------------------------------
    const int cl = (lane & 7) ^ (lane >> 3);
    const bf16* d_ptr[DI];
    int d_step[DI];
#pragma unroll
    for (int i = 0; i < DI; ++i) {
        const int r = (wave * DI + i) * 8 + (lane >> 3);          // row of the combined tile: [0, MR) window positions, [MR, TR) weight rows
        if (r < MR) {
            const int src = srcs[r];
            d_ptr[i] = src >= 0 ? a.x + (int64_t)src * C + cl * 8 : Z;
            d_step[i] = src >= 0 ? 64 : 0;
        } else {
            const int n = r - MR;                                    // 0 .. 95: q | k | v rows of this head
            d_ptr[i] = a.Wg + ((int64_t)(n >> 5) * C + h * HD + (n & 31)) * C + cl * 8;
            d_step[i] = 64;
        }
    }
    auto issue = [&](int stage) {
        char* sb = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < DI; ++i) { dma16(d_ptr[i], sb + (wave * DI + i) * 1024); d_ptr[i] += d_step[i]; }
    };
    const int QT = (N + 15) / 16;
    const int mg = wave >> 1, nh = wave & 1;
    const int m_cnt = QT / 4 + (mg < (QT & 3) ? 1 : 0), m_beg = mg * (QT / 4) + min(mg, QT & 3);      // this wave's m-tiles; n-tiles 3 nh .. 3 nh + 2
    f32x4 acc[MI][3];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s1[SU], s2[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) { s1[u] = 0.f; s2[u] = 0.f; }
    typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2;
    const bf16x2 ones2 = {(bf16)1.0f, (bf16)1.0f};
    const int ktiles = C >> 6;
    if (ktiles) issue(0);
    // the small tables are staged while the first K tile is in flight (they are first read after the loop's barriers)
    for (int e = tid; e < NP; e += NTHR) {
        const int hy = (e / ww) % wh, wx = e % ww;
        bs[e] = hy * (2 * ww - 1) + wx;
        Rs[e] = (REGION && e < N) ? (uint8_t)a.region[(int64_t)(w % a.nw_img) * N + e] : 0;
    }
    for (int e = tid; e < R; e += NTHR) tab[e] = a.table[(int64_t)e * a.heads + h] * LOG2E;
    for (int e = tid; e < 3 * 96; e += NTHR) {
        const int which = e / 96, n = e - which * 96, col = (n >> 5) * C + h * HD + (n & 31);
        evec[e] = (which == 0 ? a.wsum : (which == 1 ? a.biasp : a.bias))[col];
    }
    for (int kt = 0; kt < ktiles; ++kt) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < ktiles) issue((kt + 1) & 1);
        const bf16* cA = reinterpret_cast<const bf16*>(smem + (kt & 1) * STAGE);
        const bf16* cB = cA + MR * 64;
        // row statistics from the resident tile: thread t owns the 16-byte chunk t % 8 of rows t / 8 + 64 u (any physical chunk order: sums only).
        // (v_dot2c_f32_bf16: one instruction per PAIR and sum -- exact bf16 products, fp32 accumulation.  The reads go through inline asm: in
        // front of a plain C++ LDS load hipcc put s_waitcnt vmcnt(0) here, serialising the ring.)
        uint4 sc[SU];
        lds_read16_n<SU, 64 * 128>(lds_byte_addr(cA) + (unsigned)tid * 16u, sc);
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const unsigned wv[4] = {sc[u].x, sc[u].y, sc[u].z, sc[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bf16x2 v2 = __builtin_bit_cast(bf16x2, wv[e]);
                s1[u] = __builtin_amdgcn_fdot2_f32_bf16(v2, ones2, s1[u], false);
                s2[u] = __builtin_amdgcn_fdot2_f32_bf16(v2, v2, s2[u], false);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fb[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) fb[j] = frag_kc<bf16>(cB, (nh * 3 + j) * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (i < m_cnt) {
                    const bf16x8 fa = frag_kc<bf16>(cA, (m_beg + i) * 16, ks, lane);
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa, acc[i][j], 0, 0, 0);
                }
            }
        }
    }
    {
        const float invC = 1.0f / (float)C;
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            float t1 = s1[u], t2 = s2[u];
            t1 += __shfl_xor(t1, 1, 64); t1 += __shfl_xor(t1, 2, 64); t1 += __shfl_xor(t1, 4, 64);
            t2 += __shfl_xor(t2, 1, 64); t2 += __shfl_xor(t2, 2, 64); t2 += __shfl_xor(t2, 4, 64);
            const int row = (tid >> 3) + 64 * u;
            if ((tid & 7) == 0 && row < MR) {
                const float m1 = t1 * invC;
                mu[row] = m1;
                rsd[row] = rsqrtf(fmaxf(t2 * invC - m1 * m1, 0.f) + a.eps);
            }
        }
    }
    __syncthreads();           // statistics visible; every wave is done with the ring: Q / K / V may overwrite it

    // epilogue of phase 1: LayerNorm algebra + bias -> LDS (attention operands) and HBM (saved for backward)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if (i >= m_cnt) continue;
        const int m = (m_beg + i) * 16 + c16;
        const bool inwin = m < N, tok = inwin && srcs[m] >= 0;
        const float mr = mu[m], rr = rsd[m];
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            const int j = nh * 3 + jj, jp = j >> 1, hf = j & 1;
            const int col = jp * 32 + hf * 16 + 4 * g;
            const float4 ws4 = *reinterpret_cast<const float4*>(evec + col), bp4 = *reinterpret_cast<const float4*>(evec + 96 + col), b4 = *reinterpret_cast<const float4*>(evec + 192 + col);
            const float wsv[4] = {ws4.x, ws4.y, ws4.z, ws4.w}, bpv[4] = {bp4.x, bp4.y, bp4.z, bp4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = tok ? rr * (acc[i][jj][r] - mr * wsv[r]) + bpv[r] : (inwin ? bv[r] : 0.f);
            const uint2 pk = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            bf16* Xs = jp == 0 ? Qs : (jp == 1 ? Ks : Vs);
            *reinterpret_cast<uint2*>(Xs + m * F_LD + hf * 16 + 4 * g) = pk;
            if (inwin) *reinterpret_cast<uint2*>(a.qkv + ((int64_t)w * N + m) * 3 * C + jp * C + h * HD + hf * 16 + 4 * g) = pk;
        }
    }
    for (int e = tid; e < (NP - QT * 16) * 4; e += NTHR) {      // rows beyond the last computed tile read zero in the attention phase
        const int row = QT * 16 + (e >> 2), c = e & 3;
        const uint4 z = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(Qs + row * F_LD + c * 8) = z;
        *reinterpret_cast<uint4*>(Ks + row * F_LD + c * 8) = z;
        *reinterpret_cast<uint4*>(Vs + row * F_LD + c * 8) = z;
    }
    __syncthreads();

    // (round 6) the raw rows of this head's 32-column slice for the LayerNorm-output tail below: requested here, used after the attention phase
    constexpr int XS = (NP * 4 + NTHR - 1) / NTHR;
    uint4 xs[XS];
#pragma unroll
    for (int k = 0; k < XS; ++k) {
        const int e = tid + k * NTHR, row = min(e >> 2, N - 1), c = e & 3;
        const int src = srcs[row];
        xs[k] = *reinterpret_cast<const uint4*>(a.x + (int64_t)(src >= 0 ? src : 0) * C + h * HD + c * 8);
    }
    // ---- phase 2: attention core on the LDS copies (as wattn_fwd_mfma, attention_mfma.hip; a wave owns at most two query tiles, so the K / V^T
    // fragments are read per tile instead of being kept in registers) ---------------------------------------------------------------------------
    const float sc2 = a.scale * LOG2E;
    const int heads = a.heads;
    for (int it = wave; it < QT; it += NW) {
        const int i = 16 * it + c16;
        const bool vi = i < N;
        const bf16x8 qf = lds_row8(Qs, F_LD, i, 8 * g);
        const int rid_i = Rs[i], bi = bs[i] + centre;
        f32x4 s[2 * KS];
        float mx = -1e30f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row8(Ks, F_LD, 16 * t + c16, 8 * g), qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            const int j0 = 16 * t + 4 * g;
            const int4 bj = *reinterpret_cast<const int4*>(bs + j0);
            const f32x4 bb = {tab[bi - bj.x], tab[bi - bj.y], tab[bi - bj.z], tab[bi - bj.w]};
            f32x4 v = sacc * sc2 + bb;
            if constexpr (REGION) {
                const uint32_t rj = *reinterpret_cast<const uint32_t*>(Rs + j0);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += ((int)((rj >> (8 * r)) & 0xFF) != rid_i) ? -100.0f * LOG2E : 0.f;
            }
            if constexpr (!FULL) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (j0 + r < N) ? v[r] : -1e30f;
            }
            s[t] = v;
            mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
            if (t % 3 == 2) __builtin_amdgcn_sched_barrier(0);          // bound the hoisting of K fragments / table gathers: 128-register budget (two workgroups per CU)
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 2 * KS; ++t) {
            if (t < NT) {
                const f32x4 d = s[t] - mx;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float pv = __builtin_amdgcn_exp2f(d[r]); s[t][r] = pv; sum += pv; }
            } else s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        if (vi && g == 0) a.lse[((int64_t)w * heads + h) * N + i] = (mx + __log2f(sum)) * LN2;
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 pf;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) pf[jj] = (bf16)s[2 * ks + (jj >> 2)][jj & 3];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bf16* p = Vs + (32 * ks + 4 * g + (c16 >> 2)) * F_LD + 16 * u + 4 * (c16 & 3);
                const bf16x8 vf = join4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p), __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 16 * F_LD)));
                o[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[u], 0, 0, 0);
            }
        }
        const float inv = 1.f / sum;
        store_head_row16(a.out + ((int64_t)w * N + (vi ? i : 0)) * C + h * HD, g,
                         make_uint2(pack_bf16x2(o[0][0] * inv, o[0][1] * inv), pack_bf16x2(o[0][2] * inv, o[0][3] * inv)),
                         make_uint2(pack_bf16x2(o[1][0] * inv, o[1][1] * inv), pack_bf16x2(o[1][2] * inv, o[1][3] * inv)), vi);
    }
    // this head's 32-column slice of the LayerNorm output (operand of the qkv weight gradient) + the row statistics (LayerNorm backward): the raw rows were
    // requested in front of the attention phase (xs), so only the arithmetic and the store are left here
#pragma unroll
    for (int k = 0; k < XS; ++k) {
        const int e = tid + k * NTHR;
        if (e >= N * 4) continue;
        const int row = e >> 2, c = e & 3;
        const int src = srcs[row];
        if (src < 0) continue;
        float f[8];
        chunk_to_f<bf16>(xs[k], f);
        const float4 g0 = *reinterpret_cast<const float4*>(a.gamma + h * HD + c * 8), g1 = *reinterpret_cast<const float4*>(a.gamma + h * HD + c * 8 + 4);
        const float4 e0 = *reinterpret_cast<const float4*>(a.beta + h * HD + c * 8), e1 = *reinterpret_cast<const float4*>(a.beta + h * HD + c * 8 + 4);
        const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
        const float mr = mu[row], rr = rsd[row];
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = (f[q] - mr) * rr * gm[q] + bt[q];
        *reinterpret_cast<uint4*>(a.xn + (int64_t)src * C + h * HD + c * 8) = f_to_chunk<bf16>(f);
        if (h == 0 && c == 0) { a.mean[src] = mr; a.rstd[src] = rr; }
    }
}

// gamma-folded weight of a Linear that follows a LayerNorm: Wg[n][k] = bf16(gamma_k W[n][k]), wsum[n] = sum_k Wg[n][k] (of the ROUNDED values: the
// epilogue subtracts mu * wsum from a contraction over exactly these), biasp[n] = b_n + sum_k beta_k W[n][k].  One wave per output row.
__global__ __launch_bounds__(256) void ln_fold_kernel(const float* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ bias, bf16* __restrict__ Wg, float* __restrict__ wsum, float* __restrict__ biasp, int N, int Kd) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float s = 0.f, bp = 0.f;
    for (int k = lane * 4; k < Kd; k += 256) {
        const float4 w4 = *reinterpret_cast<const float4*>(W + (int64_t)n * Kd + k), g4 = *reinterpret_cast<const float4*>(gamma + k), b4 = *reinterpret_cast<const float4*>(beta + k);
        const float v[4] = {w4.x * g4.x, w4.y * g4.y, w4.z * g4.z, w4.w * g4.w};
        const uint2 pk = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        *reinterpret_cast<uint2*>(Wg + (int64_t)n * Kd + k) = pk;
        s += __uint_as_float(pk.x << 16) + __uint_as_float(pk.x & 0xFFFF0000u) + __uint_as_float(pk.y << 16) + __uint_as_float(pk.y & 0xFFFF0000u);
        bp += w4.x * b4.x + w4.y * b4.y + w4.z * b4.z + w4.w * b4.w;
    }
    s = wave_sum(s);
    bp = wave_sum(bp);
    if (lane == 0) { wsum[n] = s; biasp[n] = bp + (bias ? bias[n] : 0.f); }
}

// every fold of a model in one launch: desc int64 [count][9] = {W, gamma, beta, bias, Wg, wsum, biasp, N, K} (device table), blockIdx.y = fold
__global__ __launch_bounds__(256) void ln_fold_multi_kernel(const int64_t* __restrict__ desc) {
    const int64_t* d = desc + 9 * (int64_t)blockIdx.y;
    const float* W = reinterpret_cast<const float*>(d[0]);
    const float* gamma = reinterpret_cast<const float*>(d[1]);
    const float* beta = reinterpret_cast<const float*>(d[2]);
    const float* bias = reinterpret_cast<const float*>(d[3]);
    bf16* Wg = reinterpret_cast<bf16*>(d[4]);
    float* wsum = reinterpret_cast<float*>(d[5]);
    float* biasp = reinterpret_cast<float*>(d[6]);
    const int N = (int)d[7], Kd = (int)d[8];
    const int lane = threadIdx.x & 63;
    for (int n = blockIdx.x * 4 + (threadIdx.x >> 6); n < N; n += gridDim.x * 4) {
        float s = 0.f, bp = 0.f;
        for (int k = lane * 4; k < Kd; k += 256) {
            const float4 w4 = *reinterpret_cast<const float4*>(W + (int64_t)n * Kd + k), g4 = *reinterpret_cast<const float4*>(gamma + k), b4 = *reinterpret_cast<const float4*>(beta + k);
            const float v[4] = {w4.x * g4.x, w4.y * g4.y, w4.z * g4.z, w4.w * g4.w};
            const uint2 pk = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            *reinterpret_cast<uint2*>(Wg + (int64_t)n * Kd + k) = pk;
            s += __uint_as_float(pk.x << 16) + __uint_as_float(pk.x & 0xFFFF0000u) + __uint_as_float(pk.y << 16) + __uint_as_float(pk.y & 0xFFFF0000u);
            bp += w4.x * b4.x + w4.y * b4.y + w4.z * b4.z + w4.w * b4.w;
        }
        s = wave_sum(s);
        bp = wave_sum(bp);
        if (lane == 0) { wsum[n] = s; biasp[n] = bp + (bias ? bias[n] : 0.f); }
    }
}

template <int NT, bool FULL> int launch_wmsa(const WmsaArgs& a, hipStream_t st) {
    constexpr int KS = (NT + 1) / 2, NP = KS * 32, TR = ((NT * 16 + 96 + 63) / 64) * 64, MR = TR - 96, STAGE = TR * 128;
    const int R = (2 * a.wh - 1) * (2 * a.ww - 1);
    const size_t lds = (size_t)2 * STAGE + (size_t)NP * 4 + (size_t)MR * 12 + NP + (size_t)R * 4 + 32 + 3 * 96 * 4;
    static size_t reserved = 0;
    if (lds > 65536 && lds > reserved) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wmsa_fwd_fused_kernel<NT, true, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&wmsa_fwd_fused_kernel<NT, false, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            lavt_set_error("lavt_wmsa_fwd: cannot reserve %zu bytes of LDS", lds);
            return LAVT_ERR_LAUNCH;
        }
        reserved = lds;
    }
    const dim3 grid(a.nwin * a.heads + a.fill_blocks);
    if (a.region) hipLaunchKernelGGL((wmsa_fwd_fused_kernel<NT, true, FULL>), grid, dim3(512), lds, st, a);
    else hipLaunchKernelGGL((wmsa_fwd_fused_kernel<NT, false, FULL>), grid, dim3(512), lds, st, a);
    LAVT_CHECK_LAUNCH("lavt_wmsa_fwd");
    return LAVT_OK;
}

}  // namespace

extern "C" int lavt_ln_fold(const float* W, const float* gamma, const float* beta, const float* bias, void* Wg, float* wsum, float* biasp, int N, int K, void* stream) {
    LAVT_CHECK_ARG(W && gamma && beta && Wg && wsum && biasp && N > 0 && K > 0 && K % 4 == 0, "lavt_ln_fold: bad arguments");
    hipLaunchKernelGGL(ln_fold_kernel, dim3(cdiv(N, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), W, gamma, beta, bias, (bf16*)Wg, wsum, biasp, N, K);
    LAVT_CHECK_LAUNCH("lavt_ln_fold");
    return LAVT_OK;
}

extern "C" int lavt_ln_fold_multi(const int64_t* desc, int count, void* stream) {
    LAVT_CHECK_ARG(desc && count > 0, "lavt_ln_fold_multi: bad arguments");
    hipLaunchKernelGGL(ln_fold_multi_kernel, dim3(128, count), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), desc);
    LAVT_CHECK_LAUNCH("lavt_ln_fold_multi");
    return LAVT_OK;
}

static int wmsa_fwd_impl(const void* x, const int32_t* wmap, const void* Wg, const float* wsum, const float* biasp, const float* bias, const float* gamma,
                         const float* beta, const float* table, const int8_t* region, int nw_img, void* out, float* lse, void* qkv, void* xn, float* mean,
                         float* rstd, const void* zeros, int ws, int nwin, int N, int heads, int C, float eps, float scale, void* fill, int64_t fill_bytes, void* stream);
extern "C" int lavt_wmsa_fwd(const void* x, const int32_t* wmap, const void* Wg, const float* wsum, const float* biasp, const float* bias, const float* gamma,
                             const float* beta, const float* table, const int8_t* region, int nw_img, void* out, float* lse, void* qkv, void* xn, float* mean,
                             float* rstd, const void* zeros, int ws, int nwin, int N, int heads, int C, float eps, float scale, void* stream) {
    return wmsa_fwd_impl(x, wmap, Wg, wsum, biasp, bias, gamma, beta, table, region, nw_img, out, lse, qkv, xn, mean, rstd, zeros, ws, nwin, N, heads, C, eps, scale, nullptr, 0,
                         stream);
}
// the same launch with a zero-fill rider: `fill_bytes` bytes at `fill` (both multiples of 16) are set to zero by extra workgroups of the launch
extern "C" int lavt_wmsa_fwd_rider(const void* x, const int32_t* wmap, const void* Wg, const float* wsum, const float* biasp, const float* bias, const float* gamma,
                                   const float* beta, const float* table, const int8_t* region, int nw_img, void* out, float* lse, void* qkv, void* xn, float* mean,
                                   float* rstd, const void* zeros, int ws, int nwin, int N, int heads, int C, float eps, float scale, void* fill, int64_t fill_bytes,
                                   void* stream) {
    LAVT_CHECK_ARG(fill_bytes == 0 || (fill && fill_bytes > 0 && fill_bytes % 16 == 0 && ((uintptr_t)fill % 16) == 0), "lavt_wmsa_fwd_rider: fill region must be 16-byte aligned");
    return wmsa_fwd_impl(x, wmap, Wg, wsum, biasp, bias, gamma, beta, table, region, nw_img, out, lse, qkv, xn, mean, rstd, zeros, ws, nwin, N, heads, C, eps, scale, fill,
                         fill_bytes, stream);
}
static int wmsa_fwd_impl(const void* x, const int32_t* wmap, const void* Wg, const float* wsum, const float* biasp, const float* bias, const float* gamma,
                         const float* beta, const float* table, const int8_t* region, int nw_img, void* out, float* lse, void* qkv, void* xn, float* mean,
                         float* rstd, const void* zeros, int ws, int nwin, int N, int heads, int C, float eps, float scale, void* fill, int64_t fill_bytes, void* stream) {
    LAVT_CHECK_ARG(x && wmap && Wg && wsum && biasp && bias && gamma && beta && table && out && lse && qkv && xn && mean && rstd && zeros, "lavt_wmsa_fwd: null argument");
    LAVT_CHECK_ARG(nwin > 0 && heads > 0 && C == heads * HD && C % 64 == 0 && ws > 0 && N > 0 && N <= ws * ws && N <= 160 && (!region || nw_img > 0),
                   "lavt_wmsa_fwd: needs C = 32 * heads, C %% 64 == 0, windows of <= 160 tokens");
    WmsaArgs a{};
    a.x = (const bf16*)x; a.wmap = wmap; a.Wg = (const bf16*)Wg; a.wsum = wsum; a.biasp = biasp; a.bias = bias; a.gamma = gamma; a.beta = beta;
    a.table = table; a.region = region; a.out = (bf16*)out; a.lse = lse; a.qkv = (bf16*)qkv; a.xn = (bf16*)xn; a.mean = mean; a.rstd = rstd; a.zeros = zeros;
    a.nw_img = nw_img; a.wh = ws; a.ww = ws; a.nwin = nwin; a.N = N; a.heads = heads; a.C = C; a.eps = eps; a.scale = scale;
    a.fill = reinterpret_cast<uint4*>(fill); a.fill_words = fill_bytes / 16;
    a.fill_blocks = fill_bytes > 0 ? (int)((a.fill_words + 16383) / 16384 < 224 ? (a.fill_words + 16383) / 16384 : 224) : 0;      // <= 224 riders (the slots 288 workgroups leave free), >= 256 KB each
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (N <= 64) return launch_wmsa<4, false>(a, st);
    if (N == 144) return launch_wmsa<9, true>(a, st);
    if (N <= 144) return launch_wmsa<9, false>(a, st);
    return launch_wmsa<10, false>(a, st);
}
