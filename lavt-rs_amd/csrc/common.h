// Shared device/host helpers for the LAVT gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/lavt_hip.h"

typedef __bf16 bf16;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

#define LAVT_WAVE 64

// ---- error plumbing -------------------------------------------------------------------------
void lavt_set_error(const char* fmt, ...);
#define LAVT_CHECK_ARG(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            lavt_set_error(__VA_ARGS__);          \
            return LAVT_ERR_INVALID;              \
        }                                         \
    } while (0)
#define LAVT_CHECK_LAUNCH(name)                                                  \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            lavt_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return LAVT_ERR_LAUNCH;                                              \
        }                                                                        \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- tuning / A-B switches (tuning.hip): the LAVT_* environment variables, read once per process (lavt_tuning_reload() re-reads) ---------
struct lavt_tuning_t {
    bool attn_simple, unpack_tiled, gemm_epi_lds, gemm_epi_narrow, gemm_v2_off, tn_big, gemm_general, gemm_wide, fp8_pipe_off, conv_tail_off, side_pre_off, upce_tile_off, attn_bwd_split_off;
    int attn_bwd_waves, gemm_tile, tn_split, tng_tile, tng_waves, tng_stages, tng_chain, tng_piece, tn_big_min, tn_target, gemm_big_long, gemm_stages,
        gemm_waves, ln_bwd_waves, tn_streamk, gemm_pipe, tn_pipe, tn_pipe_min_tiles, tn_pipe_stages, tn_pipe_min_ktiles;
    int probe[8];          // LAVT_PROBE=a,b,...: free integers for experiment builds (unused by the shipped dispatch)
};
const lavt_tuning_t& lavt_tuning();

// ---- scalar conversions -----------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16>(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return (bf16)v; }

// A 16-byte chunk of T: 4 floats or 8 bf16.
template <typename T> struct Chunk { static constexpr int N = 16 / sizeof(T); };

template <typename T> __device__ __forceinline__ void chunk_to_f(const uint4& c, float* f) {
    if constexpr (std::is_same<T, float>::value) {
        f[0] = __uint_as_float(c.x); f[1] = __uint_as_float(c.y); f[2] = __uint_as_float(c.z); f[3] = __uint_as_float(c.w);
    } else {
        const uint32_t w[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
        }
    }
}
// (as a two-element vector conversion the pair is ONE v_cvt_pk_bf16_f32; converted one by one and joined with shift / or it was four instructions)
typedef __bf16 lavt_bf16x2 __attribute__((ext_vector_type(2)));
typedef float lavt_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(lavt_f32x2{lo, hi}, lavt_bf16x2));
}
template <typename T> __device__ __forceinline__ uint4 f_to_chunk(const float* f) {
    uint4 c;
    if constexpr (std::is_same<T, float>::value) {
        c.x = __float_as_uint(f[0]); c.y = __float_as_uint(f[1]); c.z = __float_as_uint(f[2]); c.w = __float_as_uint(f[3]);
    } else {
        c.x = pack_bf16x2(f[0], f[1]); c.y = pack_bf16x2(f[2], f[3]);
        c.z = pack_bf16x2(f[4], f[5]); c.w = pack_bf16x2(f[6], f[7]);
    }
    return c;
}

// ---- wave / block reductions ----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- activations ------------------------------------------------------------------------------
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7): 1 reciprocal (v_rcp_f32, 1 ulp: __frcp_rn / a division expand to the 11-instruction
// IEEE sequence, more than the rest of the function), 1 exponential, 7 fma -- the library erff is ~40 instructions, and inside a
// GEMM epilogue (two waves per SIMD, nothing to overlap with) the exact form cost 4-5 us on a 1800 x 2048 tile set (tools/gemm_cold_probe.py).
// Used on the bf16 path only (bf16 keeps 8 significant bits); the fp32 parity path keeps erff.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float y = 1.0f - poly * __expf(-ax * ax);
    return copysignf(y, x);
}
__device__ __forceinline__ float gelu_f_fast(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f_fast(float x) {
    const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
// GELU and its derivative together: Phi(x) from the fast erf, phi(x) from the SAME exponential (exp(-z^2) with z = x / sqrt 2 is exp(-x^2 / 2))
__device__ __forceinline__ float gelu_pair_fast(float x, float& d) {
    const float z = x * 0.70710678118654752440f, az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * az);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float e = __expf(-az * az);
    const float cdf = 0.5f * (1.0f + copysignf(1.0f - poly * e, z));
    d = cdf + x * (0.39894228040143267794f * e);
    return x * cdf;
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
