// e4m3 packing shared by the producers that write a quantised twin of their bf16 output (norm.hip, elementwise.hip): the same arithmetic as
// fp8_quantize_kernel (fp8.hip) applied to the bf16-ROUNDED value, so a twin holds exactly the bytes lavt_fp8_quantize would write for that output.
#pragma once
#include "common.h"

namespace {

__device__ __forceinline__ unsigned q8_pack4(float a, float b, float c, float d) {
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned)v;
}
__device__ __forceinline__ float q8_clamp(float x) { return fminf(fmaxf(x, -448.f), 448.f); }
__device__ __forceinline__ float q8_scale(const float* amax_prev) {
    const float ap = amax_prev ? *amax_prev : 0.f;
    return ap > 0.f ? 448.f / ap : 1.f;
}
// 8 values (one bf16 chunk) -> 8 e4m3 bytes; m collects |max| of the values
__device__ __forceinline__ uint2 q8_chunk8(const float* f, float s, float& m) {
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(f[e]));
    return make_uint2(q8_pack4(q8_clamp(f[0] * s), q8_clamp(f[1] * s), q8_clamp(f[2] * s), q8_clamp(f[3] * s)),
                      q8_pack4(q8_clamp(f[4] * s), q8_clamp(f[5] * s), q8_clamp(f[6] * s), q8_clamp(f[7] * s)));
}
// ONE atomic per workgroup (256 threads), and only when it would raise the value (non-negative floats order like their bit patterns)
__device__ __forceinline__ void q8_block_amax(float m, float* amax) {
    __shared__ float q8_red[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) q8_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(q8_red[0], q8_red[1]), fmaxf(q8_red[2], q8_red[3]));
        if (m > __builtin_nontemporal_load(amax)) atomicMax(reinterpret_cast<unsigned*>(amax), __float_as_uint(m));
    }
}

}  // namespace
