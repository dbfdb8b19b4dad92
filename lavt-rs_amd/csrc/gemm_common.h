// Shared tile / fragment helpers of the gather-GEMM kernels (gemm.hip: register-staged v1; gemm_v2.hip: LDS-DMA pipeline).
#pragma once
#include "common.h"

namespace lavt_gemm {

template <typename T> struct Cfg;
template <> struct Cfg<bf16> { static constexpr int BK = 64, KSTEP = 32, EPC = 8, KC_LD = 64; };
template <> struct Cfg<float> { static constexpr int BK = 16, KSTEP = 4, EPC = 4, KC_LD = 20; };
constexpr int KM_PAD = 16;

// ---- LDS addressing -----------------------------------------------------------------------------
// "KC" tile: [rows][BK], k contiguous.  bf16 rows are 128 B = 8 chunks, chunk index XOR (row & 7).
template <typename T> __device__ __forceinline__ int kc_off(int row, int chunk) {
    if constexpr (std::is_same<T, bf16>::value) return row * 64 + ((chunk ^ (row & 7)) << 3);
    else return row * 20 + (chunk << 2);
}

template <typename T> struct FragT;
template <> struct FragT<bf16> { typedef bf16x8 type; };
template <> struct FragT<float> { typedef float type; };

// fragment of a KC tile: 16 rows starting at `row0`, k-step ks
template <typename T> __device__ __forceinline__ typename FragT<T>::type frag_kc(const T* s, int row0, int ks, int lane) {
    const int row = row0 + (lane & 15);
    if constexpr (std::is_same<T, bf16>::value) return *reinterpret_cast<const bf16x8*>(s + kc_off<bf16>(row, ks * 4 + (lane >> 4)));
    else return s[row * 20 + ks * 4 + (lane >> 4)];
}
// fragment of a "KM" tile stored [k][ld] (k-major): 16 columns starting at col0, k-step ks
template <typename T> __device__ __forceinline__ typename FragT<T>::type frag_km(const T* s, int ld, int col0, int ks, int lane) {
    if constexpr (std::is_same<T, bf16>::value) {
        const int k = ks * 32 + 8 * (lane >> 4) + ((lane & 15) >> 2);
        const bf16* p = s + k * ld + col0 + 4 * (lane & 3);
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 4 * ld));
        bf16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    } else {
        return s[(ks * 4 + (lane >> 4)) * ld + col0 + (lane & 15)];
    }
}
template <typename T> __device__ __forceinline__ f32x4 mfma16(typename FragT<T>::type a, typename FragT<T>::type b, f32x4 c) {
    if constexpr (std::is_same<T, bf16>::value) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ uint4 ldg16(const void* p) { return *reinterpret_cast<const uint4*>(p); }

// N 16-byte LDS reads at addr + i * STRIDE bytes issued from ONE inline-asm statement that also waits for them (early-clobber outputs: no consumer
// can be scheduled above the wait).  For reads of an LDS-DMA ring inside the K loop that are not MFMA fragments: hipcc (ROCm 7.2) puts
// `s_waitcnt vmcnt(0)` in front of such a plain C++ LDS load when a global_load_lds is in flight -- seen in front of the row-statistics reads of the
// fused W-MSA kernel and of the LayerNorm-folded GEMM, where it serialised the ring (the tile just issued had to land before the current one was
// read: 10 us instead of ~5 for 8 K tiles) -- asm reads are invisible to that pass (cdna_hip_programming.md 5.4 trap (a), 5.7).
template <int N, int STRIDE> __device__ __forceinline__ void lds_read16_n(unsigned addr, uint4 (&v)[N]) {
    static_assert(N == 1 || N == 2 || N == 3 || N == 5, "instantiated counts");
    if constexpr (N == 3) {
        asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:%c4\n\tds_read_b128 %2, %3 offset:%c4*2\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]) : "v"(addr), "n"(STRIDE) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        return;
    }
    if constexpr (N == 1) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v[0]) : "v"(addr) : "memory");
    else if constexpr (N == 2)
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%c3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v[0]), "=&v"(v[1]) : "v"(addr), "n"(STRIDE) : "memory");
    else
        asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:%c6\n\tds_read_b128 %2, %5 offset:%c6*2\n\tds_read_b128 %3, %5 offset:%c6*3\n\t"
                     "ds_read_b128 %4, %5 offset:%c6*4\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]) : "v"(addr), "n"(STRIDE) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ unsigned lds_byte_addr(const void* p) { return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p; }
__device__ __forceinline__ uint4 zero16() { return make_uint4(0, 0, 0, 0); }

template <bool FAST = false> __device__ __forceinline__ float apply_act(int act, float v) {
    switch (act) {
        case LAVT_ACT_GELU: case LAVT_ACT_GELU_D: return FAST ? gelu_f_fast(v) : gelu_f(v);
        case LAVT_ACT_RELU: return fmaxf(v, 0.f);
        case LAVT_ACT_TANH: return tanhf(v);
        default: return v;
    }
}


// ---- implicit-GEMM convolution geometry (2-D 3x3 and the 3-D kernels of SepTPWAM) ---------------------------------------
struct ConvGeom {
    int d, h, w, kd, kh, kw, taps, vox;     // vox = d*h*w voxels per sample
    float inv_vox, inv_hw, inv_w;           // reciprocals for fdiv()
};
// n / d for 0 <= n < 2^23 with a precomputed float reciprocal: one multiply + a +-1 fix-up instead of the ~35-instruction
// exact integer division sequence (these sit in the K loops of the conv kernels, once per DMA instruction per K tile)
__device__ __forceinline__ int fdiv(int n, int d, float inv) {
    int q = (int)((float)n * inv);
    const int r = n - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}
template <typename P> __device__ __forceinline__ ConvGeom conv_geom(const P& p) {
    ConvGeom g;
    g.kh = p.conv_kh > 0 ? p.conv_kh : 3; g.kw = p.conv_kw > 0 ? p.conv_kw : 3; g.kd = p.conv_kd > 0 ? p.conv_kd : 1;
    g.d = p.conv_d > 0 ? p.conv_d : 1; g.h = p.conv_h; g.w = p.conv_w;
    g.taps = g.kd * g.kh * g.kw; g.vox = g.d * g.h * g.w;
    g.inv_vox = 1.0f / (float)(g.vox > 0 ? g.vox : 1); g.inv_hw = 1.0f / (float)(g.h * g.w > 0 ? g.h * g.w : 1); g.inv_w = 1.0f / (float)(g.w > 0 ? g.w : 1);
    return g;
}
__device__ __forceinline__ void conv_tap(const ConvGeom& g, int tap, int& dz, int& dy, int& dx) {
    dx = tap % g.kw - (g.kw >> 1);
    dy = (tap / g.kw) % g.kh - (g.kh >> 1);
    dz = tap / (g.kw * g.kh) - (g.kd >> 1);
}
// (z, y, x) of voxel row `src` >= 0 inside its sample
__device__ __forceinline__ void conv_coords(const ConvGeom& g, int src, int& z, int& y, int& x) {
    const int pix = src - fdiv(src, g.vox, g.inv_vox) * g.vox;
    z = fdiv(pix, g.h * g.w, g.inv_hw);
    const int rem = pix - z * g.h * g.w;
    y = fdiv(rem, g.w, g.inv_w);
    x = rem - y * g.w;
}
// source row of the (dz,dy,dx) neighbour of voxel row `src` (-1 outside the zero-padded volume or if src < 0)
__device__ __forceinline__ int conv_nbr(const ConvGeom& g, int src, int dz, int dy, int dx) {
    if (src < 0) return -1;
    int z, y, x;
    conv_coords(g, src, z, y, x);
    z += dz; y += dy; x += dx;
    return ((unsigned)z < (unsigned)g.d && (unsigned)y < (unsigned)g.h && (unsigned)x < (unsigned)g.w) ? src + (dz * g.h + dy) * g.w + dx : -1;
}

// ---- XCD-aware workgroup -> tile order ----------------------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2): blocks b and b+8 share an L2.  Give each
// XCD a CONTIGUOUS range of logical tiles (n fastest), so the tiles resident on one XCD share A row-panels and the whole
// of B instead of every XCD streaming every panel (bijective for any grid size; speed only, never correctness).
__device__ __forceinline__ int xcd_tile_id(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, xcd = b & 7, i = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

// ---- NT epilogue shared by both generations: accumulators hold C^T tiles (rows = 4 consecutive n per lane, column = m) ----
__device__ __forceinline__ float act_grad(int act, float pre) {
    switch (act) {
        case LAVT_ACT_GELU: return gelu_grad_f_fast(pre);          // bf16 data-gradient epilogue only
        case LAVT_ACT_RELU: return pre > 0.f ? 1.f : 0.f;
        case LAVT_ACT_TANH: { const float t = tanhf(pre); return 1.f - t * t; }
        default: return 1.f;
    }
}
// ---- wide bf16 stores straight from the accumulators ----------------------------------------------------------------------------------
// In the C^T accumulator layout a lane holds 4 consecutive n of one row m, and the four lane groups g = lane / 16 hold columns 4g .. 4g+3 of
// the same 16 rows: stored as they stand, a wave-instruction writes 16 rows x 32 B in 8-byte pieces (measured: a second [1800 x 2048] output
// -- the pre-activation -- cost a 12 us GEMM another 6 us, i.e. ~1.2 TB/s).  Two neighbouring 16-column fragments are therefore combined:
// lanes g and g ^ 1 swap one packed pair (ds_bpermute, no memory), after which an even group owns columns 4g .. 4g+7 of the first fragment and
// an odd group columns 4(g-1) .. 4(g-1)+7 of the second -- one 16-byte store per lane, 64 contiguous bytes per row per wave-instruction, half
// the store instructions.  Bias is read as float4.  Conditions are checked by the caller (whole 32-column pairs inside N, 16-byte aligned rows).
// 16-byte output store of the wide epilogue: write-through (`sc0 sc1`) -- the bytes leave the XCD's L2 as they are stored instead of at the kernel's
// end-of-launch release (MI355X_MICROARCH.md, price list: a dependent kernel boundary costs + B / 6 TB/s behind B dirty bytes; the consumer launch
// reads them from another XCD's side of the fabric seven times out of eight anyway).  Round 5, same box, alternating libraries: 7.832 / 7.828 ms per
// step against 7.866 / 7.851 with plain stores (-DLAVT_ST_PLAIN builds the plain form).
__device__ __forceinline__ void st16_out(void* p, const uint4& v) {
#if !defined(LAVT_ST_PLAIN)
    typedef unsigned st16_u32x4 __attribute__((ext_vector_type(4)));
    const st16_u32x4 w = {v.x, v.y, v.z, v.w};
    // (s_nop 1: a VMEM store of more than 8 bytes reads its data registers after issue -- the next instruction must not overwrite them; hipcc's hazard
    // recogniser inserts the wait state for its own stores, it cannot see into an asm statement.  Without it the 256x256 instantiations stored garbage.)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
#else
    *reinterpret_cast<uint4*>(p) = v;
#endif
}
__device__ __forceinline__ uint2 xchg16(uint2 v) { return make_uint2((unsigned)__shfl_xor((int)v.x, 16, 64), (unsigned)__shfl_xor((int)v.y, 16, 64)); }
// LEAN: the launch uses none of {activation, multiplier, second (pre-activation) output, split output}: those branches are compiled out.  Carrying
// them as not-taken uniform branches costs every plain GEMM of the step (tools/ab_lib.sh: -0.2 ms per step with all of them compiled out, of which
// about half is the launches that do use them).
// LEAN = 2 (data gradients, convolutions without bias): additionally no bias / residual / row scale / output row map -- a plain store.
// LEAN = 3 (round 5): LEAN 2 with the split output kept -- the data gradient of a concat convolution in ONE launch (Swin-T's conv1_2: 384 + 96 columns;
// on the full epilogue that launch ran at 0.36 of peak where its lean siblings reach 0.45).
// Side inputs of the wide epilogue (pre-activation of the activation gradient, residual, multiplier) for a wave tile of <= 8 fragments.  Loaded at the
// head of the epilogue they are one exposed round trip to L2 / HBM behind the K loop of a launch whose K loop lasts 3-10 us (proj forward 1800 x 512 x
// 512: 5.2 us as a plain GEMM, ~10 with residual + DropPath scale); `nt_side_load` issues the same loads BEFORE the K loop (round 5) -- they are the
// oldest vector-memory operations of the wave, so the first counted vmcnt wait of the loop covers them and they cost 8 bytes of registers per fragment
// across the loop.  Kernels call it only where those registers are free (64x64 / 4-wave tiles: 100 -> 116 VGPRs; the activation-gradient kernels).
template <int MI, int NI> struct NtSide {
    uint2 d[MI][NI], r[MI][NI], m[MI][NI];
    bool have;
};
template <int MI, int NI, bool DACT, bool GD = false, int LEAN = 0>
__device__ __forceinline__ void nt_side_load(const lavt_gemm_nt_t& p_in, NtSide<MI, NI>& s, int m_base, int n_base, int lane) {
    lavt_gemm_nt_t p = p_in;
    if constexpr (DACT && GD) { p.R = nullptr; p.c_rowmap = nullptr; p.mul = nullptr; }
    if constexpr (GD && !DACT) { p.mul = nullptr; p.R = nullptr; p.c_rowmap = nullptr; }
    if constexpr (LEAN >= 1) p.mul = nullptr;
    if constexpr (LEAN >= 2) { p.R = nullptr; p.c_rowmap = nullptr; }
    const int g = lane >> 4;
    int64_t srow[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m_base + i * 16 + (lane & 15);
        int orow = -1;
        if (m < p.M) orow = p.c_rowmap ? p.c_rowmap[m] : m;
        srow[i] = orow >= 0 ? orow : 0;
    }
    if constexpr (DACT) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                s.d[i][j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16*>(p.dact_pre) + srow[i] * p.lddact + n_base + j * 16 + 4 * g);
    }
    if (p.R) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                s.r[i][j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16*>(p.R) + srow[i] * p.ldr + n_base + j * 16 + 4 * g);
    }
    if (p.mul) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                s.m[i][j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16*>(p.mul) + srow[i] * p.ldmul + n_base + j * 16 + 4 * g);
    }
    s.have = true;
}
template <int MI, int NI, bool DACT, bool GD = false, int LEAN = 0>
__device__ __forceinline__ void nt_epilogue_wide(const lavt_gemm_nt_t& p_in, f32x4 (&acc)[MI][NI], int m_base, int n_base, int lane, int bz,
                                                 const NtSide<(MI * NI <= 8 ? MI : 1), NI>* pre = nullptr) {
    lavt_gemm_nt_t p = p_in;
    if constexpr (DACT && GD) {          // stored-derivative data gradient (launch_nt_v2_dact): C = acc * alpha * row_scale * dact_pre, nothing else
        p.dact = LAVT_ACT_STORED; p.R = nullptr; p.bias = nullptr; p.c_rowmap = nullptr; p.act = 0; p.mul = nullptr; p.Cpre = nullptr; p.C2 = nullptr;
    }
    if constexpr (GD && !DACT) {         // fc1 of the LayerNorm-folded MLP node: folded bias, GELU + its derivative as second output, nothing else
        p.mul = nullptr; p.C2 = nullptr; p.R = nullptr; p.row_scale = nullptr; p.c_rowmap = nullptr;
    }
    if constexpr (LEAN >= 1) { p.act = 0; p.mul = nullptr; p.Cpre = nullptr; }
    if constexpr (LEAN == 1 || LEAN == 2) p.C2 = nullptr;
    if constexpr (LEAN >= 2) { p.bias = nullptr; p.R = nullptr; p.row_scale = nullptr; p.c_rowmap = nullptr; }
    const float* bias = p.bias ? p.bias + (int64_t)bz * p.strideBias : nullptr;
    const float* rscale = p.row_scale ? p.row_scale + (int64_t)bz * p.strideRowScale : nullptr;
    const int64_t c_off = (int64_t)bz * p.strideC;
    const int g = lane >> 4;
    const bool odd = g & 1;
    // Side inputs of the whole wave tile first (pre-activation of the activation gradient, residual, multiplier), unconditionally -- dead lanes
    // read row 0: loaded where they are used, each sits behind a per-lane `live` test, and the compiler then waits for every one of them on
    // its own (vmcnt(0) per fragment: 8 round trips to L2 per wave on the fc2 data gradient, +6 us on a 10 us GEMM; tools/mlp_gemm_probe.py).
    // Up to 8 fragments per wave (the 128x128 / 8-wave and 64x64 / 4-wave tiles); the 16-fragment tiles keep the loads at the use.
    constexpr bool PF = MI * NI <= 8;
    uint2 q_d[PF ? MI : 1][NI], q_r[PF ? MI : 1][NI], q_m[PF ? MI : 1][NI];
    if constexpr (PF) {
      if (pre != nullptr && pre->have) {          // (wave-uniform: loaded before the K loop)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) { q_d[i][j] = pre->d[i][j]; q_r[i][j] = pre->r[i][j]; q_m[i][j] = pre->m[i][j]; }
      } else {
        int64_t srow[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m_base + i * 16 + (lane & 15);
            int orow = -1;
            if (m < p.M) orow = p.c_rowmap ? p.c_rowmap[m] : m;
            srow[i] = orow >= 0 ? orow : 0;
        }
        if constexpr (DACT) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    q_d[i][j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16*>(p.dact_pre) + srow[i] * p.lddact + n_base + j * 16 + 4 * g);
        }
        if (p.R) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    q_r[i][j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16*>(p.R) + srow[i] * p.ldr + n_base + j * 16 + 4 * g);
        }
        if (p.mul) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    q_m[i][j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16*>(p.mul) + srow[i] * p.ldmul + n_base + j * 16 + 4 * g);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m_base + i * 16 + (lane & 15);
        int orow = -1;
        if (m < p.M) orow = p.c_rowmap ? p.c_rowmap[m] : m;
        const bool live = orow >= 0;                         // dead lanes still take part in the exchanges
        const float rs = (rscale && m < p.M) ? rscale[p.row_scale_div > 1 ? m / p.row_scale_div : m] : 1.f;
#pragma unroll
        for (int jp = 0; jp < NI / 2; ++jp) {
            uint2 packed[2], packed_pre[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = n_base + (2 * jp + h) * 16 + 4 * g;
                float v[4];
                float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bias) b4 = *reinterpret_cast<const float4*>(bias + n);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (p.alpha * acc[i][2 * jp + h][r] + bb[r]) * rs;
                if constexpr (DACT) {
                    if (PF || live) {
                        if (p.res_first && p.R) {          // gradient arriving beside the GEMM's own (a residual branch) joins BEFORE the activation gradient
                            uint2 q;
                            if constexpr (PF) q = q_r[i][2 * jp + h]; else q = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16*>(p.R) + (int64_t)orow * p.ldr + n);
                            v[0] += __uint_as_float(q.x << 16); v[1] += __uint_as_float(q.x & 0xFFFF0000u);
                            v[2] += __uint_as_float(q.y << 16); v[3] += __uint_as_float(q.y & 0xFFFF0000u);
                        }
                        uint2 q;
                        if constexpr (PF) q = q_d[i][2 * jp + h]; else q = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16*>(p.dact_pre) + (int64_t)orow * p.lddact + n);
                        if (p.dact == LAVT_ACT_STORED) {       // the producing launch (LAVT_ACT_GELU_D) stored the derivative itself: a uniform branch around
                            v[0] *= __uint_as_float(q.x << 16); v[1] *= __uint_as_float(q.x & 0xFFFF0000u);      // the transcendental forms, not a case of
                            v[2] *= __uint_as_float(q.y << 16); v[3] *= __uint_as_float(q.y & 0xFFFF0000u);      // act_grad's switch
                        } else {                                                                                 // (as a case of it, both forms ran slower)
                            v[0] *= act_grad(p.dact, __uint_as_float(q.x << 16)); v[1] *= act_grad(p.dact, __uint_as_float(q.x & 0xFFFF0000u));
                            v[2] *= act_grad(p.dact, __uint_as_float(q.y << 16)); v[3] *= act_grad(p.dact, __uint_as_float(q.y & 0xFFFF0000u));
                        }
                    }
                }
                if constexpr (GD && !DACT) {               // the second output carries GELU'(pre), computed beside the activation (GD: only the
                                                          // LayerNorm-folded launch is built with this branch -- in every NT kernel it cost 0.16 ms per step)
                    float d[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_pair_fast(v[r], d[r]);
                    packed_pre[h] = make_uint2(pack_bf16x2(d[0], d[1]), pack_bf16x2(d[2], d[3]));
                } else {
                    packed_pre[h] = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                    if (p.act) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = apply_act<true>(p.act, v[r]);
                    }
                }
                if (p.mul && (PF || live)) {        // language gate: x + tanh(g) * r -- the multiplier r rides between the activation and the residual
                    uint2 q;
                    if constexpr (PF) q = q_m[i][2 * jp + h]; else q = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16*>(p.mul) + (int64_t)orow * p.ldmul + n);
                    v[0] *= __uint_as_float(q.x << 16); v[1] *= __uint_as_float(q.x & 0xFFFF0000u);
                    v[2] *= __uint_as_float(q.y << 16); v[3] *= __uint_as_float(q.y & 0xFFFF0000u);
                }
                if (p.R && (PF || live) && !(DACT && p.res_first)) {          // (an fp32 exchange + one 16-byte residual load measured slower than these 8-byte loads: 2.6 vs 2.0 us on the fc1 shape)
                    uint2 q;
                    if constexpr (PF) q = q_r[i][2 * jp + h]; else q = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16*>(p.R) + (int64_t)orow * p.ldr + n);
                    v[0] += __uint_as_float(q.x << 16); v[1] += __uint_as_float(q.x & 0xFFFF0000u);
                    v[2] += __uint_as_float(q.y << 16); v[3] += __uint_as_float(q.y & 0xFFFF0000u);
                }
                packed[h] = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
            // even groups keep fragment 0 and send their fragment-1 piece; odd groups keep fragment 1 and send their fragment-0 piece
            const uint2 got = xchg16(odd ? packed[0] : packed[1]);
            const uint4 out = odd ? make_uint4(got.x, got.y, packed[1].x, packed[1].y) : make_uint4(packed[0].x, packed[0].y, got.x, got.y);
            const int n_out = n_base + (2 * jp + (odd ? 1 : 0)) * 16 + 4 * (g & ~1);
            if (p.Cpre) {
                const uint2 gp = xchg16(odd ? packed_pre[0] : packed_pre[1]);
                const uint4 op = odd ? make_uint4(gp.x, gp.y, packed_pre[1].x, packed_pre[1].y) : make_uint4(packed_pre[0].x, packed_pre[0].y, gp.x, gp.y);
                if (live) st16_out(reinterpret_cast<bf16*>(p.Cpre) + (int64_t)orow * p.ldcpre + n_out, op);
            }
            if (live) {
                const bool second = p.C2 != nullptr && n_out >= p.c_split;
                bf16* cp = reinterpret_cast<bf16*>(second ? p.C2 : p.C) + c_off + (int64_t)orow * (second ? p.ldc2 : p.ldc) + (second ? n_out - p.c_split : n_out);
                st16_out(cp, out);
            }
        }
    }
}

// wave-uniform: this wave's tile takes the wide epilogue
template <typename T, int MI, int NI> __device__ __forceinline__ bool nt_takes_wide(const lavt_gemm_nt_t& p, int n_base) {
    if constexpr (std::is_same<T, bf16>::value && NI % 2 == 0) return !p.c_f32 && p.epi_wide && n_base + NI * 16 <= p.N;
    else return false;
}
template <typename T, int MI, int NI, bool DACT = false, bool GD = false, int LEAN = 0>
__device__ __forceinline__ void nt_epilogue(const lavt_gemm_nt_t& p, f32x4 (&acc)[MI][NI], int m_base, int n_base, int lane, int bz,
                                            const NtSide<(MI * NI <= 8 ? MI : 1), NI>* pre = nullptr) {
    if constexpr (std::is_same<T, bf16>::value && NI % 2 == 0) {
        // wave-uniform conditions: the whole wave takes one path (the exchanges need every lane)
        if (!p.c_f32 && p.epi_wide && n_base + NI * 16 <= p.N) { nt_epilogue_wide<MI, NI, DACT, GD, LEAN>(p, acc, m_base, n_base, lane, bz, pre); return; }
    }
    const float* bias = p.bias ? p.bias + (int64_t)bz * p.strideBias : nullptr;
    const float* rscale = p.row_scale ? p.row_scale + (int64_t)bz * p.strideRowScale : nullptr;
    const int64_t c_off = (int64_t)bz * p.strideC;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m_base + i * 16 + (lane & 15);
        if (m >= p.M) continue;
        const int orow = p.c_rowmap ? p.c_rowmap[m] : m;
        if (orow < 0) continue;
        const float rs = rscale ? rscale[p.row_scale_div > 1 ? m / p.row_scale_div : m] : 1.f;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n_base + j * 16 + 4 * (lane >> 4);
            if (n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = p.alpha * acc[i][j][r];
                if (bias && n + r < p.N) v[r] += bias[n + r];
                v[r] *= rs;
            }
            const bool full = (n + 3 < p.N);
            if constexpr (DACT) {          // gradient w.r.t. the pre-activation of the producing layer: v *= act'(pre[orow][n..n+3])
                if (p.res_first && p.R) {
                    const T* rp = reinterpret_cast<const T*>(p.R) + (int64_t)orow * p.ldr + n;
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += to_f<T>(rp[r]);
                }
                const T* dp = reinterpret_cast<const T*>(p.dact_pre) + (int64_t)orow * p.lddact + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] *= p.dact == LAVT_ACT_STORED ? to_f<T>(dp[r]) : act_grad(p.dact, to_f<T>(dp[r]));
            }
            if (p.Cpre) {
                T* cp = reinterpret_cast<T*>(p.Cpre) + (int64_t)orow * p.ldcpre + n;
                float w[4] = {v[0], v[1], v[2], v[3]};
                if constexpr (GD) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[r] = std::is_same<T, bf16>::value ? gelu_grad_f_fast(v[r]) : gelu_grad_f(v[r]);
                }
                if (full) {
                    if constexpr (std::is_same<T, float>::value) *reinterpret_cast<float4*>(cp) = make_float4(w[0], w[1], w[2], w[3]);
                    else *reinterpret_cast<uint2*>(cp) = make_uint2(pack_bf16x2(w[0], w[1]), pack_bf16x2(w[2], w[3]));
                } else for (int r = 0; r < 4; ++r) if (n + r < p.N) cp[r] = from_f<T>(w[r]);
            }
            if (p.act) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = apply_act<std::is_same<T, bf16>::value>(p.act, v[r]);
            }
            if (p.mul) {
                const T* mp = reinterpret_cast<const T*>(p.mul) + (int64_t)orow * p.ldmul + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] *= to_f<T>(mp[r]);
            }
            if (p.R && !(DACT && p.res_first)) {
                const T* rp = reinterpret_cast<const T*>(p.R) + (int64_t)orow * p.ldr + n;
                if (full) {
                    if constexpr (std::is_same<T, float>::value) { const float4 q = *reinterpret_cast<const float4*>(rp); v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w; }
                    else { const uint2 q = *reinterpret_cast<const uint2*>(rp); v[0] += __uint_as_float(q.x << 16); v[1] += __uint_as_float(q.x & 0xFFFF0000u); v[2] += __uint_as_float(q.y << 16); v[3] += __uint_as_float(q.y & 0xFFFF0000u); }
                } else for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += to_f<T>(rp[r]);
            }
            const bool second = p.C2 != nullptr && n >= p.c_split;
            const int64_t ldc = second ? p.ldc2 : p.ldc;
            const int nn = second ? n - p.c_split : n;
            if (p.c_f32) {
                float* cp = reinterpret_cast<float*>(second ? p.C2 : p.C) + c_off + (int64_t)orow * ldc + nn;
                if (full) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                else for (int r = 0; r < 4; ++r) if (n + r < p.N) cp[r] = v[r];
            } else {
                T* cp = reinterpret_cast<T*>(second ? p.C2 : p.C) + c_off + (int64_t)orow * ldc + nn;
                if (full) {
                    if constexpr (std::is_same<T, float>::value) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                    else *reinterpret_cast<uint2*>(cp) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                } else for (int r = 0; r < 4; ++r) if (n + r < p.N) cp[r] = from_f<T>(v[r]);
            }
        }
    }
}


// ---- bf16 NT epilogue staged through LDS: full-row 16-byte stores -------------------------------------------------------
// The direct epilogue above issues 8-byte stores that touch 16 different rows per wave-instruction (32-B segments): the
// store tail is issue-bound (cdna_hip_programming.md T21).  Here the C tile goes registers -> LDS ([BM][BN+8] bf16) -> global
// as 16 B per lane, 4 full 256-B rows per wave-instruction; the residual is read with the same coalesced pattern.
template <int BM, int BN, int MI, int NI, bool GD = false>
__device__ __forceinline__ void nt_epilogue_lds(const lavt_gemm_nt_t& p, f32x4 (&acc)[MI][NI], bf16* sC, int m0, int n0, int wm_off, int wn_off,
                                                int tid, int lane, int bz) {
    constexpr int LD = BN + 8, CPR = BN / 8, CHUNKS = BM * CPR;
    const float* bias = p.bias ? p.bias + (int64_t)bz * p.strideBias : nullptr;
    const float* rscale = p.row_scale ? p.row_scale + (int64_t)bz * p.strideRowScale : nullptr;
    const int64_t c_off = (int64_t)bz * p.strideC;
    float rs[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm_off + i * 16 + (lane & 15);
        rs[i] = (rscale && m < p.M) ? rscale[p.row_scale_div > 1 ? m / p.row_scale_div : m] : 1.f;
    }
    const int npass = p.Cpre ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        const bool pre = p.Cpre && pass == 0;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int ml = wm_off + i * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int nl = wn_off + j * 16 + 4 * (lane >> 4);
                const int n = n0 + nl;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = p.alpha * acc[i][j][r];
                    if (bias && n + r < p.N) v[r] += bias[n + r];
                    v[r] *= rs[i];
                    if (!pre && p.act) v[r] = apply_act<true>(p.act, v[r]);
                    if (GD && pre) v[r] = gelu_grad_f_fast(v[r]);
                }
                *reinterpret_cast<uint2*>(sC + ml * LD + nl) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
        }
        __syncthreads();
        for (int idx = tid; idx < CHUNKS; idx += (int)blockDim.x) {
            const int row = idx / CPR, c = idx - row * CPR;
            const int m = m0 + row, n = n0 + c * 8;
            if (m >= p.M || n >= p.N) continue;
            const int orow = p.c_rowmap ? p.c_rowmap[m] : m;
            if (orow < 0) continue;
            uint4 v = *reinterpret_cast<const uint4*>(sC + row * LD + c * 8);
            const bool full = n + 8 <= p.N;
            bf16* dst;
            if (pre) dst = reinterpret_cast<bf16*>(p.Cpre) + (int64_t)orow * p.ldcpre + n;
            else {
                const bool second = p.C2 != nullptr && n >= p.c_split;
                dst = reinterpret_cast<bf16*>(second ? p.C2 : p.C) + c_off + (int64_t)orow * (second ? p.ldc2 : p.ldc) + (second ? n - p.c_split : n);
            }
            if (!pre && p.R) {
                const bf16* rp = reinterpret_cast<const bf16*>(p.R) + (int64_t)orow * p.ldr + n;
                float a[8], b[8];
                chunk_to_f<bf16>(v, a);
                if (full) chunk_to_f<bf16>(*reinterpret_cast<const uint4*>(rp), b);
                else for (int e = 0; e < 8; ++e) b[e] = n + e < p.N ? to_f<bf16>(rp[e]) : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] += b[e];
                v = f_to_chunk<bf16>(a);
            }
            if (full) *reinterpret_cast<uint4*>(dst) = v;
            else {
                const bf16* sv = reinterpret_cast<const bf16*>(&v);
                for (int e = 0; e < 8; ++e) if (n + e < p.N) dst[e] = sv[e];
            }
        }
    }
}

}  // namespace lavt_gemm
