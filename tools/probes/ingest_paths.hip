// Per-CU ingest from L2: LDS-DMA (global_load_lds_dwordx4), plain vector loads into VGPRs (global_load_dwordx4), and both at once.
// Round-4 review item 6 ("second ingest path for the decoder convolutions"): does a wave that loads one operand straight into registers ADD bytes per
// second to what the LDS-DMA stream of the same CU already moves?  One 512-thread workgroup per CU (the geometry of gemm_nt_pipe / gemm_tn_pipe /
// conv_wgrad), every workgroup sweeping the same `region` bytes (L2-resident per XCD when region <= 2 MB; Infinity Cache beyond), 16 requests of
// 1 KiB per wave in flight.
//   mode 0: all 8 waves LDS-DMA          mode 1: all 8 waves register loads
//   mode 2: waves 0-3 LDS-DMA, 4-7 register loads (the review's proposal: B fragments beside the DMA ring)
//   mode 3: every wave alternates the two      mode 4: waves 0-3 LDS-DMA only (half the issuers, for reference)
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/ingest_paths.hip -o .ab/ingest_paths ; run on the GPU box: ./.ab/ingest_paths
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

template <int MODE>
__global__ __launch_bounds__(512) void ingest(const char* __restrict__ src, size_t region, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool dma_wave = MODE == 0 || MODE == 3 || ((MODE == 2 || MODE == 4) && wave < 4);
    const bool reg_wave = MODE == 1 || MODE == 3 || (MODE == 2 && wave >= 4);
    char* ring = smem + wave * 16 * 1024;                       // 16 KiB per wave
    uint4 acc = make_uint4(0, 0, 0, 0);
    // every (workgroup, wave) starts at its own offset so that the chip does not hammer one channel; all stay inside `region`
    size_t off = ((size_t)blockIdx.x * 8 + wave) * 16384 % region;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const char* p = src + (off + (size_t)u * 1024) % region + lane * 16;
            if (MODE == 3) {
                if (u & 1) { if (reg_wave) { const uint4 v = *reinterpret_cast<const uint4*>(p); acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; } }
                else __builtin_amdgcn_global_load_lds((gbl_void*)p, (lds_void*)(ring + u * 1024), 16, 0, 0);
            } else if (dma_wave) {
                __builtin_amdgcn_global_load_lds((gbl_void*)p, (lds_void*)(ring + u * 1024), 16, 0, 0);
            } else if (reg_wave) {
                const uint4 v = *reinterpret_cast<const uint4*>(p);
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        off = (off + 16 * 1024 * 8) % region;
    }
    __syncthreads();
    if (acc.x == 0x12345678u && sink) sink[0] = acc.y ^ acc.z ^ acc.w ^ (unsigned)smem[tid];
}

template <int MODE> double run(const char* d, size_t region, int iters, unsigned* sink) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&ingest<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    ingest<MODE><<<256, 512, 128 * 1024>>>(d, region, 8, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    ingest<MODE><<<256, 512, 128 * 1024>>>(d, region, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int issuers = MODE == 4 ? 4 : 8;
    const double per = MODE == 3 ? 16.0 : 16.0;
    const double bytes = 256.0 * issuers * per * 1024.0 * iters;
    return bytes / (ms * 1e-3) / 1e9 / 256.0;                    // GB/s per CU
}
int main() {
    char* d; unsigned* sink;
    hipMalloc(&d, 512u << 20); hipMemset(d, 1, 512u << 20); hipMalloc(&sink, 64);
    const char* names[5] = {"8 waves LDS-DMA", "8 waves register loads", "4 waves LDS-DMA + 4 waves register loads", "every wave alternates", "4 waves LDS-DMA only"};
    const size_t regions[4] = {1u << 20, 2u << 20, 32u << 20, 256u << 20};
    for (size_t region : regions) {
        printf("region %4zu MB (every workgroup sweeps it):\n", region >> 20);
        double r[5];
        r[0] = run<0>(d, region, 400, sink); r[1] = run<1>(d, region, 400, sink); r[2] = run<2>(d, region, 400, sink); r[3] = run<3>(d, region, 400, sink); r[4] = run<4>(d, region, 400, sink);
        for (int m = 0; m < 5; ++m) printf("  %-44s %7.1f GB/s per CU  (%6.2f TB/s chip)\n", names[m], r[m], r[m] * 256 / 1e3);
    }
    return 0;
}
