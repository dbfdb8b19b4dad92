// OCP e4m3 operand preparation for the fp8 GEMM / convolution path (BASELINE.json configs[4]): per-tensor scaling.
//   activations: DELAYED scaling -- a tensor is quantised against the |max| its call site saw in the previous step (no reduction in front of
//     the quantiser, no host round trip: the scale lives in device memory and the GEMM epilogue reads the same float), and this step's |max|
//     is recorded on the way for the next one;
//   weights: CURRENT scaling -- |max| over the tensor, computed when the compute copies are refreshed (with the optimizer step).
// gfx950 converts with v_cvt_pk_fp8_f32 (OCP e4m3fn, round to nearest even); values are clamped to +-448 first (no NaN / inf encodings produced).
#include "common.h"

namespace {

constexpr float E4M3_MAX = 448.f;

__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned)v;
}
__device__ __forceinline__ float clamp448(float x) { return fminf(fmaxf(x, -E4M3_MAX), E4M3_MAX); }

// non-negative floats order like their bit patterns
__device__ __forceinline__ void atomic_max_nonneg(float* dst, float v) { atomicMax(reinterpret_cast<unsigned*>(dst), __float_as_uint(v)); }

// 16 source elements per thread per iteration -> one 16-byte store of 16 e4m3 values
template <typename T>
__global__ __launch_bounds__(256) void fp8_quantize_kernel(const T* __restrict__ src, unsigned char* __restrict__ dst, int64_t n16,
                                                           const float* __restrict__ amax_prev, float* __restrict__ amax_cur) {
    const float ap = amax_prev ? *amax_prev : 0.f;
    const float s = ap > 0.f ? E4M3_MAX / ap : 1.f;
    float m = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
        float f[16];
        if constexpr (std::is_same<T, float>::value) {
#pragma unroll
            for (int q = 0; q < 4; ++q) chunk_to_f<float>(*reinterpret_cast<const uint4*>(src + i * 16 + q * 4), f + q * 4);
        } else {
            chunk_to_f<T>(*reinterpret_cast<const uint4*>(src + i * 16), f);
            chunk_to_f<T>(*reinterpret_cast<const uint4*>(src + i * 16 + 8), f + 8);
        }
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(f[q * 4 + e]));
            w[q] = pack4_e4m3(clamp448(f[q * 4] * s), clamp448(f[q * 4 + 1] * s), clamp448(f[q * 4 + 2] * s), clamp448(f[q * 4 + 3] * s));
        }
        *reinterpret_cast<uint4*>(dst + i * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (amax_cur) {
        // ONE atomic per workgroup, and only when it would raise the value: thousands of same-address atomics serialise at the memory side
        // (one per wave made this kernel 98 us on a 14.7 M-element map; MI355X_MICROARCH.md, global float atomics: 14x slower into one row)
        __shared__ float red[4];
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            if (m > __builtin_nontemporal_load(amax_cur)) atomic_max_nonneg(amax_cur, m);
        }
    }
}

// |max| of a bf16 / fp32 tensor, 16 elements per thread per iteration (the pass in front of a CURRENT-scaling quantisation: lavt_fp8_quantize_current)
template <typename T>
__global__ __launch_bounds__(256) void fp8_amax16_kernel(const T* __restrict__ src, int64_t n16, float* __restrict__ amax) {
    float m = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
        float f[16];
        if constexpr (std::is_same<T, float>::value) {
#pragma unroll
            for (int q = 0; q < 4; ++q) chunk_to_f<float>(*reinterpret_cast<const uint4*>(src + i * 16 + q * 4), f + q * 4);
        } else {
            chunk_to_f<T>(*reinterpret_cast<const uint4*>(src + i * 16), f);
            chunk_to_f<T>(*reinterpret_cast<const uint4*>(src + i * 16 + 8), f + 8);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) m = fmaxf(m, fabsf(f[e]));
    }
    __shared__ float red[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m > __builtin_nontemporal_load(amax)) atomic_max_nonneg(amax, m);
    }
}

__global__ void fp8_advance_kernel(float* prev, float* cur, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float c = cur[i];
    if (c > 0.f) prev[i] = c;
    cur[i] = 0.f;
}

__global__ __launch_bounds__(256) void fp8_amax_kernel(const float* __restrict__ src, int64_t n, float* __restrict__ amax) {
    float m = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(src[i]));
    __shared__ float red[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m > __builtin_nontemporal_load(amax)) atomic_max_nonneg(amax, m);
    }
}
// dst[o][t][c] = e4m3(src[o][c][t] * 448 / amax): thread per 4 consecutive c of one (o, t)
__global__ __launch_bounds__(256) void fp8_weight_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, const float* __restrict__ amax,
                                                         int cout, int cin, int taps) {
    const float a = *amax;
    const float s = a > 0.f ? E4M3_MAX / a : 1.f;
    const int64_t n4 = (int64_t)cout * taps * (cin / 4);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % (cin / 4)), t = (int)((i / (cin / 4)) % taps), o = (int)(i / (cin / 4) / taps);
        const float* sp = src + ((int64_t)o * cin + c4 * 4) * taps + t;
        *reinterpret_cast<unsigned*>(dst + ((int64_t)o * taps + t) * cin + c4 * 4) =
            pack4_e4m3(clamp448(sp[0] * s), clamp448(sp[taps] * s), clamp448(sp[2 * taps] * s), clamp448(sp[3 * taps] * s));
    }
}

// transposed pack for the data gradient: dst[c][t][o] = e4m3(src[o][c][t] * 448 / amax): thread per 4 consecutive o of one (c, t)
__global__ __launch_bounds__(256) void fp8_weight_t_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, const float* __restrict__ amax,
                                                           int cout, int cin, int taps) {
    const float a = *amax;
    const float s = a > 0.f ? E4M3_MAX / a : 1.f;
    const int64_t n4 = (int64_t)cin * taps * (cout / 4), os = (int64_t)cin * taps;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int o4 = (int)(i % (cout / 4)), t = (int)((i / (cout / 4)) % taps), c = (int)(i / (cout / 4) / taps);
        const float* sp = src + ((int64_t)o4 * 4 * cin + c) * taps + t;
        *reinterpret_cast<unsigned*>(dst + ((int64_t)c * taps + t) * cout + o4 * 4) =
            pack4_e4m3(clamp448(sp[0] * s), clamp448(sp[os] * s), clamp448(sp[2 * os] * s), clamp448(sp[3 * os] * s));
    }
}

}  // namespace

static inline int fp8_grid(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b)); }

extern "C" int lavt_fp8_quantize(int src_dtype, const void* src, void* dst, int64_t n, const float* amax_prev, float* amax_cur, void* stream) {
    LAVT_CHECK_ARG(src && dst && n > 0 && n % 16 == 0, "lavt_fp8_quantize: bad arguments (n=%ld must be a multiple of 16)", (long)n);
    LAVT_CHECK_ARG(src_dtype == LAVT_F32 || src_dtype == LAVT_BF16, "lavt_fp8_quantize: source dtype %d", src_dtype);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (src_dtype == LAVT_F32) hipLaunchKernelGGL(fp8_quantize_kernel<float>, dim3(fp8_grid(n / 16)), dim3(256), 0, st, (const float*)src, (unsigned char*)dst, n / 16, amax_prev, amax_cur);
    else hipLaunchKernelGGL(fp8_quantize_kernel<bf16>, dim3(fp8_grid(n / 16)), dim3(256), 0, st, (const bf16*)src, (unsigned char*)dst, n / 16, amax_prev, amax_cur);
    LAVT_CHECK_LAUNCH("lavt_fp8_quantize");
    return LAVT_OK;
}
extern "C" int lavt_fp8_advance(float* amax_prev, float* amax_cur, int n, void* stream) {
    LAVT_CHECK_ARG(amax_prev && amax_cur && n > 0, "lavt_fp8_advance: bad arguments");
    hipLaunchKernelGGL(fp8_advance_kernel, dim3(cdiv(n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), amax_prev, amax_cur, n);
    LAVT_CHECK_LAUNCH("lavt_fp8_advance");
    return LAVT_OK;
}
extern "C" int lavt_fp8_quantize_weight(const float* src, void* dst, float* amax, int cout, int cin, int taps, void* stream) {
    LAVT_CHECK_ARG(src && dst && amax && cout > 0 && cin > 0 && cin % 4 == 0 && taps > 0, "lavt_fp8_quantize_weight: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t n = (int64_t)cout * cin * taps;
    if (hipMemsetAsync(amax, 0, sizeof(float), st) != hipSuccess) { lavt_set_error("lavt_fp8_quantize_weight: memset failed"); return LAVT_ERR_LAUNCH; }
    hipLaunchKernelGGL(fp8_amax_kernel, dim3(fp8_grid(n)), dim3(256), 0, st, src, n, amax);
    hipLaunchKernelGGL(fp8_weight_kernel, dim3(fp8_grid(n / 4)), dim3(256), 0, st, src, (unsigned char*)dst, amax, cout, cin, taps);
    LAVT_CHECK_LAUNCH("lavt_fp8_quantize_weight");
    return LAVT_OK;
}
extern "C" int lavt_fp8_quantize_weight_t(const float* src, void* dst, float* amax, int cout, int cin, int taps, void* stream) {
    LAVT_CHECK_ARG(src && dst && amax && cout > 0 && cout % 4 == 0 && cin > 0 && taps > 0, "lavt_fp8_quantize_weight_t: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t n = (int64_t)cout * cin * taps;
    if (hipMemsetAsync(amax, 0, sizeof(float), st) != hipSuccess) { lavt_set_error("lavt_fp8_quantize_weight_t: memset failed"); return LAVT_ERR_LAUNCH; }
    hipLaunchKernelGGL(fp8_amax_kernel, dim3(fp8_grid(n)), dim3(256), 0, st, src, n, amax);
    hipLaunchKernelGGL(fp8_weight_t_kernel, dim3(fp8_grid(n / 4)), dim3(256), 0, st, src, (unsigned char*)dst, amax, cout, cin, taps);
    LAVT_CHECK_LAUNCH("lavt_fp8_quantize_weight_t");
    return LAVT_OK;
}

extern "C" int lavt_fp8_quantize_current(int src_dtype, const void* src, void* dst, int64_t n, float* amax, void* stream) {
    LAVT_CHECK_ARG(src && dst && amax && n > 0 && n % 16 == 0, "lavt_fp8_quantize_current: bad arguments (n=%ld must be a multiple of 16)", (long)n);
    LAVT_CHECK_ARG(src_dtype == LAVT_F32 || src_dtype == LAVT_BF16, "lavt_fp8_quantize_current: source dtype %d", src_dtype);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(amax, 0, sizeof(float), st) != hipSuccess) { lavt_set_error("lavt_fp8_quantize_current: memset failed"); return LAVT_ERR_LAUNCH; }
    if (src_dtype == LAVT_F32) {
        hipLaunchKernelGGL(fp8_amax16_kernel<float>, dim3(fp8_grid(n / 16)), dim3(256), 0, st, (const float*)src, n / 16, amax);
        hipLaunchKernelGGL(fp8_quantize_kernel<float>, dim3(fp8_grid(n / 16)), dim3(256), 0, st, (const float*)src, (unsigned char*)dst, n / 16, (const float*)amax, (float*)nullptr);
    } else {
        hipLaunchKernelGGL(fp8_amax16_kernel<bf16>, dim3(fp8_grid(n / 16)), dim3(256), 0, st, (const bf16*)src, n / 16, amax);
        hipLaunchKernelGGL(fp8_quantize_kernel<bf16>, dim3(fp8_grid(n / 16)), dim3(256), 0, st, (const bf16*)src, (unsigned char*)dst, n / 16, (const float*)amax, (float*)nullptr);
    }
    LAVT_CHECK_LAUNCH("lavt_fp8_quantize_current");
    return LAVT_OK;
}
