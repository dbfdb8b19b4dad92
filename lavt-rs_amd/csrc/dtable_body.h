// Table-gradient binning of the window-attention backward as a __device__ body shared by attention_mfma.hip (its own launch, rider workgroups of an
// attention-backward launch) and norm.hip (rider workgroups of a LayerNorm-backward launch: round 6).
#pragma once
#include "common.h"

namespace {

// Table gradient straight from the per-(window, head) dS slabs: dtable[idx][h] += sum_w sum_{(i,j): idx(i,j) = idx} slab[w][h][i][j].
// grid = (row chunks, heads, window groups).  A wave walks rows i of its chunk, lanes cover the keys j (coalesced row reads), the sum over
// the group's windows stays in registers; the (i, j) -> table-index binning then costs one LDS atomic per (i, j) per workgroup (not per
// window), and one global atomic per touched table entry per workgroup.
__device__ __forceinline__ void dtable_block(const bf16* __restrict__ slab, float* __restrict__ part, int slab_ld, int wd, int wh, int ww, int nwin, int N, int heads,
                             int rows_per_block, int win_per_group, int bx, int h, int bz, int gx, int tid, char* smem_raw) {
    const int R = (2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1);
    const int centre = ((wd - 1) * (2 * wh - 1) + (wh - 1)) * (2 * ww - 1) + (ww - 1);
    float* hist = reinterpret_cast<float*>(smem_raw);
    int* bs = reinterpret_cast<int*>(hist + R);
    const int lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < R; e += 256) hist[e] = 0.f;
    for (int e = tid; e < N; e += 256) {
        const int dz = e / (wh * ww), hy = (e / ww) % wh, wx = e % ww;
        bs[e] = (dz * (2 * wh - 1) + hy) * (2 * ww - 1) + wx;
    }
    __syncthreads();
    const int r0 = bx * rows_per_block, r1 = min(N, r0 + rows_per_block);
    const int w0 = bz * win_per_group, w1 = min(nwin, w0 + win_per_group);
    const int64_t wstride = (int64_t)heads * N * slab_ld;
    for (int i = r0 + wave; i < r1; i += 4) {
        const bf16* row = slab + ((int64_t)w0 * heads + h) * N * slab_ld + (int64_t)i * slab_ld;
        const int bi = bs[i] + centre;
        // a lane owns a QUAD of keys (one 8-byte load: 36 lanes cover a 144-token row in one pass); sixteen windows at a time, then ONE predicated
        // round for the rest: the kernel is a chain of dependent-latency rounds (18 windows were 2 passes x 3 rounds with key pairs and 8-window steps)
        for (int j = 4 * lane; j < N; j += 256) {
            const bf16* q = row + j;
            float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
            auto add = [&](float (&t)[4], uint2 v) {
                t[0] += __uint_as_float(v.x << 16); t[1] += __uint_as_float(v.x & 0xFFFF0000u);
                t[2] += __uint_as_float(v.y << 16); t[3] += __uint_as_float(v.y & 0xFFFF0000u);
            };
            int w = w0;
            for (; w + 15 < w1; w += 16, q += 16 * wstride) {
                uint2 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const uint2*>(q + u * wstride);
#pragma unroll
                for (int u = 0; u < 16; u += 2) { add(a, v[u]); add(b, v[u + 1]); }
            }
            if (w < w1) {
                uint2 v[15];
#pragma unroll
                for (int u = 0; u < 15; ++u) v[u] = (w + u < w1) ? *reinterpret_cast<const uint2*>(q + u * wstride) : make_uint2(0u, 0u);
#pragma unroll
                for (int u = 0; u < 15; ++u) add((u & 1) ? b : a, v[u]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (j + r < N) atomicAdd(hist + (bi - bs[j + r]), a[r] + b[r]);
        }
    }
    __syncthreads();
    // this workgroup's histogram -> part[(z * chunks + chunk) * heads + h][R] (contiguous, plain stores); wattn_dtable_finish adds the pieces up.
    // (Flushing with atomics straight into dtable[R][heads] puts every lane in a different 64-byte segment: ~17x below the atomic rate.)
    if (part == nullptr) return;                           // (a rider half without a unit)
    float* dst = part + (((int64_t)bz * gx + bx) * heads + h) * R;
    for (int e = tid; e < R; e += 256) dst[e] = hist[e];
}

}  // namespace
