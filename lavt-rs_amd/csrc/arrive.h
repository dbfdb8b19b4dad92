// Hand-over between the workgroups of ONE launch (gfx950): every workgroup of a set stores a partial record, then arrives at the set's counter;
// the workgroup that arrives last -- exactly one -- continues alone with the records of all of them ("tail").  What this replaces is a kernel
// boundary (4.6-5 us per launch in a replayed hipGraph) in front of a reduction whose work is a few KB: PWAM's word-side matrices (csrc/pwam.hip).
//
// Memory model: the records are plain stores into coarse-grained device memory, cached in the L2 of the XCD that wrote them; the eight L2s are not
// coherent among themselves.  Release = every wave waits for its stores to be acknowledged by the L2 (vmcnt), the workgroup meets at a barrier, ONE
// thread writes the L2's dirty lines back (`buffer_wbl2 sc1`: the agent-scope release fence) and adds 1 to the counter (device-scope atomic, executed
// at the memory side).  Acquire = the last workgroup invalidates its non-coherent lines (`buffer_inv sc1`) before the first record load.  The last
// arriver puts the counter back to zero, so a launch leaves the counters as it found them (hipGraph replays need no memset).
// Sums over records run in a FIXED order (record index), whatever the arrival order was: results are run-to-run identical.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ bool arrive_last(unsigned* counter, unsigned n, int* s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old == n - 1;
        if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = last ? 1 : 0;
    }
    __syncthreads();
    const bool last = *s_flag != 0;
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return last;
}

// out[i] = sum_r recs[r * stride + i] for the nf4 float4 of a record, r = 0 .. R-1 in a fixed association: the NT / 256 thread sets take the records
// set, set + nsets, ... (eight loads in flight per float4 lane), the sets meet in LDS in set order.  MAXI = ceil(nf4 / 256).
// red: LDS [NT / 256][nf4] float4, out: LDS [nf4] float4 (may alias nothing else that is live).  Ends with a barrier.
template <int NT, int MAXI>
__device__ __forceinline__ void sum_records(const float* __restrict__ recs, const int R, const int64_t stride, const int nf4, float4* red, float4* out) {
    constexpr int NSETS = NT / 256;
    const int tid = threadIdx.x, set = tid >> 8, lane = tid & 255;
    float4 acc[MAXI];
#pragma unroll
    for (int m = 0; m < MAXI; ++m) acc[m] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r0 = set; r0 < R; r0 += NSETS * 8) {
        float4 v[MAXI][8];
#pragma unroll
        for (int m = 0; m < MAXI; ++m) {
            const int i = min(lane + 256 * m, nf4 - 1);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = min(r0 + u * NSETS, R - 1);
                v[m][u] = *reinterpret_cast<const float4*>(recs + (int64_t)r * stride + 4 * i);
            }
        }
#pragma unroll
        for (int m = 0; m < MAXI; ++m)
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r0 + u * NSETS < R) { acc[m].x += v[m][u].x; acc[m].y += v[m][u].y; acc[m].z += v[m][u].z; acc[m].w += v[m][u].w; }
    }
#pragma unroll
    for (int m = 0; m < MAXI; ++m)
        if (lane + 256 * m < nf4) red[set * nf4 + lane + 256 * m] = acc[m];
    __syncthreads();
    for (int i = tid; i < nf4; i += NT) {
        float4 s = red[i];
#pragma unroll
        for (int q = 1; q < NSETS; ++q) { const float4 t = red[q * nf4 + i]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
        out[i] = s;
    }
    __syncthreads();
}
