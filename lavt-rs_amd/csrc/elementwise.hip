// HBM-bound element-wise / data-movement kernels of the LAVT path (16-byte accesses, grid-stride).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"
#include "fp8_pack.h"

// ---- error string (the only global state of the library) ----------------------------------------
static thread_local char g_err[512] = "";
void lavt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* lavt_last_error(void) { return g_err; }
extern "C" int lavt_abi_version(void) { return 7; }

namespace {

inline int ew_grid(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }
#define GRID_STRIDE(i, n) for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// ---------------------------------------------------------------------------------------------- activations / gate
template <typename T> __global__ void act_bwd_kernel(int act, const T* dy, const T* pre, T* dx, int64_t nchunks) {
    constexpr int EPC = Chunk<T>::N;
    GRID_STRIDE(i, nchunks) {
        float g[EPC], x[EPC];
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(dy + i * EPC), g);
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(pre + i * EPC), x);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            if (act == LAVT_ACT_GELU) g[e] *= std::is_same<T, bf16>::value ? gelu_grad_f_fast(x[e]) : gelu_grad_f(x[e]);
            else if (act == LAVT_ACT_RELU) g[e] = x[e] > 0.f ? g[e] : 0.f;
            else if (act == LAVT_ACT_TANH) { const float t = tanhf(x[e]); g[e] *= 1.f - t * t; }
        }
        *reinterpret_cast<uint4*>(dx + i * EPC) = f_to_chunk<T>(g);
    }
}
template <typename T> __global__ void gate_fwd_kernel(const T* x, const T* gpre, const T* r, T* xo, int64_t nchunks) {
    constexpr int EPC = Chunk<T>::N;
    GRID_STRIDE(i, nchunks) {
        float a[EPC], g[EPC], b[EPC];
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(x + i * EPC), a);
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(gpre + i * EPC), g);
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(r + i * EPC), b);
#pragma unroll
        for (int e = 0; e < EPC; ++e) a[e] += tanhf(g[e]) * b[e];
        *reinterpret_cast<uint4*>(xo + i * EPC) = f_to_chunk<T>(a);
    }
}
template <typename T> __global__ void gate_bwd_kernel(const T* dxo, const T* gpre, const T* r, const T* dr_add, T* dgpre, T* dr, int64_t nchunks) {
    constexpr int EPC = Chunk<T>::N;
    GRID_STRIDE(i, nchunks) {
        float d[EPC], g[EPC], b[EPC], o1[EPC], o2[EPC], ad[EPC];
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(dxo + i * EPC), d);
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(gpre + i * EPC), g);
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(r + i * EPC), b);
        if (dr_add) chunk_to_f<T>(*reinterpret_cast<const uint4*>(dr_add + i * EPC), ad);
#pragma unroll
        for (int e = 0; e < EPC; ++e) { const float t = tanhf(g[e]); o1[e] = d[e] * b[e] * (1.f - t * t); o2[e] = d[e] * t + (dr_add ? ad[e] : 0.f); }
        *reinterpret_cast<uint4*>(dgpre + i * EPC) = f_to_chunk<T>(o1);
        *reinterpret_cast<uint4*>(dr + i * EPC) = f_to_chunk<T>(o2);
    }
}

// ---------------------------------------------------------------------------------------------- PWAM word softmax
// A thread per row, the whole row (ld <= 32 elements: 4 / 8 sixteen-byte chunks) in registers: loaded at once, stored at once.  (The first
// version walked the row three times with 2-byte loads and stored it element by element: a ~15 us dependent chain whatever the row count.)
template <typename T> __global__ __launch_bounds__(256) void rowsoftmax_fwd_kernel(const T* __restrict__ s, T* __restrict__ p, int64_t rows, int n_l, int ld) {
    constexpr int EPC = Chunk<T>::N, MAXC = 32 / EPC;
    const int nch = ld / EPC;
    GRID_STRIDE(r, rows) {
        float v[32];
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < nch) chunk_to_f<T>(*reinterpret_cast<const uint4*>(s + r * ld + c * EPC), v + c * EPC);
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 32; ++j) if (j < n_l) mx = fmaxf(mx, v[j]);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) { v[j] = j < n_l ? __expf(v[j] - mx) : 0.f; sum += v[j]; }
        const float inv = 1.f / sum;
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] *= inv;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < nch) *reinterpret_cast<uint4*>(p + r * ld + c * EPC) = f_to_chunk<T>(v + c * EPC);
    }
}
template <typename T> __global__ __launch_bounds__(256) void rowsoftmax_bwd_kernel(const T* __restrict__ p, const T* __restrict__ dp, T* __restrict__ ds, int64_t rows, int n_l, int ld) {
    constexpr int EPC = Chunk<T>::N, MAXC = 32 / EPC;
    const int nch = ld / EPC;
    GRID_STRIDE(r, rows) {
        float pv[32], dv[32];
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < nch) {
                chunk_to_f<T>(*reinterpret_cast<const uint4*>(p + r * ld + c * EPC), pv + c * EPC);
                chunk_to_f<T>(*reinterpret_cast<const uint4*>(dp + r * ld + c * EPC), dv + c * EPC);
            }
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) if (j < n_l) dot += pv[j] * dv[j];
#pragma unroll
        for (int j = 0; j < 32; ++j) dv[j] = j < n_l ? pv[j] * (dv[j] - dot) : 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < nch) *reinterpret_cast<uint4*>(ds + r * ld + c * EPC) = f_to_chunk<T>(dv + c * EPC);
    }
}

// ---------------------------------------------------------------------------------------------- bilinear (align_corners=True)
__device__ __forceinline__ void bl_coord(int o, float scale, int n_in, int& i0, int& i1, float& lam) {
    const float src = scale * (float)o;
    i0 = (int)src;
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = min(i0 + 1, n_in - 1);
    lam = src - (float)i0;
}
__host__ __device__ inline float bl_scale(int n_in, int n_out) { return n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f; }

// Q8 (bf16; configs[4]): also writes the e4m3 twin of the output (the bytes of lavt_fp8_quantize(y)) and records its |max| (delayed scaling)
template <typename T, bool Q8 = false>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const T* x, T* y, int B, int Hi, int Wi, int Ho, int Wo, int C, float sh, float sw, unsigned char* q = nullptr,
                                                           const float* amax_prev = nullptr, float* amax_cur = nullptr) {
    constexpr int EPC = Chunk<T>::N;
    float q_s = 1.f, q_m = 0.f;
    if constexpr (Q8) q_s = q8_scale(amax_prev);
    const unsigned cpr = C / EPC;
    // 32-bit index arithmetic (the entry points refuse more than 2^31 chunks): with int64 indices the three divisions of the decode were ~300 vector
    // instructions per 16-byte chunk and the kernel ran at 2.5 TB/s, VALU-bound (28.3 us for the 4 x 60 x 60 -> 120 x 120 x 512 map)
    const unsigned n = (unsigned)B * Ho * Wo * cpr;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned pix = i / cpr, rowo = pix / (unsigned)Wo;
        const int c = (int)(i - pix * cpr) * EPC;
        const int xo = (int)(pix - rowo * Wo), b = (int)(rowo / (unsigned)Ho), yo = (int)(rowo - (unsigned)b * Ho);
        int y0, y1, x0, x1; float ly, lx;
        bl_coord(yo, sh, Hi, y0, y1, ly);
        bl_coord(xo, sw, Wi, x0, x1, lx);
        const T* base = x + (int64_t)b * Hi * Wi * C + c;
        float f00[EPC], f01[EPC], f10[EPC], f11[EPC], o[EPC];
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(base + ((int64_t)y0 * Wi + x0) * C), f00);
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(base + ((int64_t)y0 * Wi + x1) * C), f01);
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(base + ((int64_t)y1 * Wi + x0) * C), f10);
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(base + ((int64_t)y1 * Wi + x1) * C), f11);
#pragma unroll
        for (int e = 0; e < EPC; ++e)
            o[e] = (1.f - ly) * ((1.f - lx) * f00[e] + lx * f01[e]) + ly * ((1.f - lx) * f10[e] + lx * f11[e]);
        const uint4 out = f_to_chunk<T>(o);
        *reinterpret_cast<uint4*>(y + (int64_t)i * EPC) = out;
        if constexpr (Q8) {
            chunk_to_f<T>(out, o);
            *reinterpret_cast<uint2*>(q + (int64_t)i * EPC) = q8_chunk8(o, q_s, q_m);
        }
    }
    if constexpr (Q8) q8_block_amax(q_m, amax_cur);
}
// Row-staged form (bf16, round 5): the element-indexed kernel above reads every input byte 16 times through register loads -- 43 GB/s per CU of L1 / L2
// traffic, the rate that path reaches (tools/probes/ingest_paths.hip: 32 GB/s per CU for register loads from L2, 71 for LDS-DMA) -- and ran the decoder's
// 60 -> 120 upsample at 2.6 TB/s.  Here a workgroup owns ONE input row interval (rows y0, y0 + 1) of one image and one block of channels: the two rows
// arrive in LDS by LDS-DMA (every input byte fetched twice, lane-linear image [row][pixel][16-byte chunk]), then every output row that samples the
// interval is formed from LDS with the element-indexed kernel's own expression (bit-identical results) and stored.  grid (Hi, B, C / cblk).
typedef __attribute__((address_space(3))) void ew_lds_void;
typedef __attribute__((address_space(1))) const void ew_gbl_void;
template <bool Q8>
__global__ __launch_bounds__(256) void bilinear_rows_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int Hi, int Wi, int Ho, int Wo, int C, int cblk, float sh, float sw,
                                                                unsigned char* __restrict__ q, const float* __restrict__ amax_prev, float* __restrict__ amax_cur) {
    extern __shared__ __attribute__((aligned(16))) char ew_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int y0b = blockIdx.x, b = blockIdx.y, c0 = blockIdx.z * cblk, cpp = cblk >> 3;          // cpp: 16-byte chunks per pixel of this channel block (divides 256)
    const int y1b = min(y0b + 1, Hi - 1);
    const int row_chunks = Wi * cpp, total = 2 * row_chunks;
    float q_s = 1.f, q_m = 0.f;
    if constexpr (Q8) q_s = q8_scale(amax_prev);
    for (int q0 = wave * 64; q0 < total; q0 += 256) {          // one 1 KiB LDS-DMA per wave and trip; chunks beyond the image re-load the last one (the LDS image is rounded up)
        const int qq = min(q0 + lane, total - 1);
        const int r = qq >= row_chunks ? 1 : 0, e = qq - r * row_chunks, xi = e / cpp, cc = e - xi * cpp;
        const bf16* src = x + (((int64_t)b * Hi + (r ? y1b : y0b)) * Wi + xi) * C + c0 + cc * 8;
        __builtin_amdgcn_global_load_lds((ew_gbl_void*)src, (ew_lds_void*)(ew_smem + (size_t)q0 * 16), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const uint4* r0 = reinterpret_cast<const uint4*>(ew_smem);
    const uint4* r1 = r0 + row_chunks;
    const int cc = tid % cpp, xstep = 256 / cpp;
    // output rows that sample this interval: floor(sh * yo) == y0b (bl_coord's own arithmetic decides; the candidate range is generous)
    const int lo = sh > 0.f ? max(0, (int)floorf((float)y0b / sh) - 1) : 0, hi = sh > 0.f ? min(Ho - 1, (int)ceilf((float)(y0b + 1) / sh) + 1) : Ho - 1;
    for (int yo = lo; yo <= hi; ++yo) {
        int yy0, yy1; float ly;
        bl_coord(yo, sh, Hi, yy0, yy1, ly);
        if (yy0 != y0b) continue;
        for (int xo = tid / cpp; xo < Wo; xo += xstep) {
            int x0, x1; float lx;
            bl_coord(xo, sw, Wi, x0, x1, lx);
            float f00[8], f01[8], f10[8], f11[8], o[8];
            chunk_to_f<bf16>(r0[x0 * cpp + cc], f00);
            chunk_to_f<bf16>(r0[x1 * cpp + cc], f01);
            chunk_to_f<bf16>(r1[x0 * cpp + cc], f10);
            chunk_to_f<bf16>(r1[x1 * cpp + cc], f11);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                o[e] = (1.f - ly) * ((1.f - lx) * f00[e] + lx * f01[e]) + ly * ((1.f - lx) * f10[e] + lx * f11[e]);
            const uint4 out = f_to_chunk<bf16>(o);
            const int64_t off = (((int64_t)b * Ho + yo) * Wo + xo) * C + c0 + cc * 8;
            *reinterpret_cast<uint4*>(y + off) = out;
            if constexpr (Q8) {
                chunk_to_f<bf16>(out, o);
                *reinterpret_cast<uint2*>(q + off) = q8_chunk8(o, q_s, q_m);
            }
        }
    }
    if constexpr (Q8) q8_block_amax(q_m, amax_cur);
}
// channel block of the row-staged form: the widest of 512 / 256 / 128 / 64 channels that divides C, keeps two input rows within 64 KB of LDS and leaves
// at least ~256 workgroups; 0 = the shape stays on the element-indexed kernel
static int bl_rows_cblk(int B, int Hi, int Wi, int C) {
    for (int cb = 512; cb >= 64; cb >>= 1) {
        if (C % cb) continue;
        const long lds = 2L * Wi * cb * 2;
        const long blocks = (long)Hi * B * (C / cb);
        if (lds <= 64 * 1024 && (blocks >= 256 || cb == 64)) return cb;
    }
    return 0;
}
static inline size_t bl_rows_lds(int Wi, int cblk) { return ((size_t)2 * Wi * cblk * 2 + 1023) / 1024 * 1024; }

// gather form of the transpose: every input pixel sums the output pixels that sampled it (deterministic, no atomics)
__device__ __forceinline__ void bl_range(int i, float scale, int n_in, int n_out, int& lo, int& hi) {
    if (scale <= 0.f) { lo = 0; hi = n_out - 1; return; }
    lo = max(0, (int)floorf((float)(i - 1) / scale) - 1);
    hi = min(n_out - 1, (int)ceilf((float)(i + 1) / scale) + 1);
}
template <typename T> __global__ void bilinear_bwd_kernel(const T* dy, T* dx, int B, int Hi, int Wi, int Ho, int Wo, int C, float sh, float sw) {
    constexpr int EPC = Chunk<T>::N;
    const unsigned cpr = C / EPC;
    const unsigned n = (unsigned)B * Hi * Wi * cpr;          // (32-bit index arithmetic: see bilinear_fwd_kernel)
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned pix = i / cpr, rowi = pix / (unsigned)Wi;
        const int c = (int)(i - pix * cpr) * EPC;
        const int xi = (int)(pix - rowi * Wi), b = (int)(rowi / (unsigned)Hi), yi = (int)(rowi - (unsigned)b * Hi);
        int ylo, yhi, xlo, xhi;
        bl_range(yi, sh, Hi, Ho, ylo, yhi);
        bl_range(xi, sw, Wi, Wo, xlo, xhi);
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        for (int yo = ylo; yo <= yhi; ++yo) {
            int y0, y1; float ly;
            bl_coord(yo, sh, Hi, y0, y1, ly);
            const float wy = (y0 == yi ? 1.f - ly : 0.f) + (y1 == yi ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int xo = xlo; xo <= xhi; ++xo) {
                int x0, x1; float lx;
                bl_coord(xo, sw, Wi, x0, x1, lx);
                const float wx = (x0 == xi ? 1.f - lx : 0.f) + (x1 == xi ? lx : 0.f);
                if (wx == 0.f) continue;
                float g[EPC];
                chunk_to_f<T>(*reinterpret_cast<const uint4*>(dy + (((int64_t)b * Ho + yo) * Wo + xo) * C + c), g);
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[e] += wy * wx * g[e];
            }
        }
        *reinterpret_cast<uint4*>(dx + (int64_t)i * EPC) = f_to_chunk<T>(acc);
    }
}
// logits: NHWC [B,Hi,Wi,2] (T) -> NCHW fp32 [B,2,Ho,Wo]
// ---- fused final step of the caller: bilinear upsample (align_corners) of the 2-class low-resolution logits + class-weighted
// cross-entropy (losses.py:7-11: F.cross_entropy(out, target, weight=[0.9, 1.1])) + the I / U pixel counts of train.py:64-76.
// The (B, 2, H, W) logits are never written: every full-resolution pixel is recomputed from its four low-resolution neighbours.
struct UpCe { float up0, up1, lse; };
template <typename T>
__device__ __forceinline__ UpCe upce_at(const T* base, int Wi, int y0, int y1, float ly, int x0, int x1, float lx) {
    const T* r0 = base + ((int64_t)y0 * Wi) * 2;
    const T* r1 = base + ((int64_t)y1 * Wi) * 2;
    const float a0 = to_f<T>(r0[x0 * 2]), a1 = to_f<T>(r0[x0 * 2 + 1]), b0 = to_f<T>(r0[x1 * 2]), b1 = to_f<T>(r0[x1 * 2 + 1]);
    const float c0 = to_f<T>(r1[x0 * 2]), c1 = to_f<T>(r1[x0 * 2 + 1]), d0 = to_f<T>(r1[x1 * 2]), d1 = to_f<T>(r1[x1 * 2 + 1]);
    UpCe u;
    u.up0 = (1.f - ly) * ((1.f - lx) * a0 + lx * b0) + ly * ((1.f - lx) * c0 + lx * d0);
    u.up1 = (1.f - ly) * ((1.f - lx) * a1 + lx * b1) + ly * ((1.f - lx) * c1 + lx * d1);
    const float m = fmaxf(u.up0, u.up1);
    u.lse = m + logf(expf(u.up0 - m) + expf(u.up1 - m));
    return u;
}
// partial[blk] = {sum w*nll, sum w, I, U}
template <typename T>
__global__ __launch_bounds__(256) void upsample_ce_fwd_kernel(const T* __restrict__ x, const int64_t* __restrict__ target, float w0, float w1,
                                                              float* __restrict__ partial, int B, int Hi, int Wi, int Ho, int Wo, float sh, float sw) {
    const int64_t n = (int64_t)B * Ho * Wo;
    float num = 0.f, den = 0.f, inter = 0.f, uni = 0.f;
    GRID_STRIDE(i, n) {
        const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho), b = (int)(i / Wo / Ho);
        int y0, y1, x0, x1; float ly, lx;
        bl_coord(yo, sh, Hi, y0, y1, ly);
        bl_coord(xo, sw, Wi, x0, x1, lx);
        const UpCe u = upce_at<T>(x + (int64_t)b * Hi * Wi * 2, Wi, y0, y1, ly, x0, x1, lx);
        const int64_t t = target[i];
        if (t == 0 || t == 1) {                                  // anything else is ignored, like F.cross_entropy's ignore_index
            const float w = t ? w1 : w0;
            num += w * (u.lse - (t ? u.up1 : u.up0));
            den += w;
        }
        const bool pred = u.up1 > u.up0, tgt = t == 1;            // argmax picks class 0 on ties
        inter += (pred && tgt) ? 1.f : 0.f;
        uni += (pred || tgt) ? 1.f : 0.f;
    }
    __shared__ float red[4][4];
    num = wave_sum(num); den = wave_sum(den); inter = wave_sum(inter); uni = wave_sum(uni);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = num; red[wave][1] = den; red[wave][2] = inter; red[wave][3] = uni; }
    __syncthreads();
    if (threadIdx.x < 4) partial[blockIdx.x * 4 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// out = {loss, sum w, I, U}
__global__ __launch_bounds__(256) void upsample_ce_finish_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ out) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < nblk; b += 256)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += partial[b * 4 + k];
    __shared__ float red[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = wave_sum(a[k]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[threadIdx.x >> 6][k] = a[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
        out[0] = v[1] > 0.f ? v[0] / v[1] : 0.f;
        out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
    }
}
// dx[b, yi, xi, c] = dloss / sum_w * sum over the full-resolution pixels that sampled (yi, xi) of coef * w_t * (softmax_c - [c == t])
// One WAVE per low-resolution pixel: its lanes share out the ~100 candidate full-resolution pixels (about 64 of them have a non-zero bilinear
// weight at 4x upsampling) and meet in a wave reduction.  (A thread per pixel -- 28 800 threads = 113 workgroups on 256 CUs, each walking its
// candidates serially -- took 55 us at 2x480x480.)
template <typename T>
__global__ __launch_bounds__(256) void upsample_ce_bwd_kernel(const T* __restrict__ x, const int64_t* __restrict__ target, float w0, float w1,
                                                              const float* __restrict__ stats, const float* __restrict__ dloss, T* __restrict__ dx,
                                                              int B, int Hi, int Wi, int Ho, int Wo, float sh, float sw) {
    const int64_t n = (int64_t)B * Hi * Wi;
    const float gscale = (stats[1] > 0.f ? 1.f / stats[1] : 0.f) * (dloss ? dloss[0] : 1.f);
    const int lane = threadIdx.x & 63;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
        const int xi = (int)(i % Wi), yi = (int)((i / Wi) % Hi), b = (int)(i / Wi / Hi);
        int ylo, yhi, xlo, xhi;
        bl_range(yi, sh, Hi, Ho, ylo, yhi);
        bl_range(xi, sw, Wi, Wo, xlo, xhi);
        const T* base = x + (int64_t)b * Hi * Wi * 2;
        const int nx = xhi - xlo + 1, cand = (yhi - ylo + 1) * nx;
        float a0 = 0.f, a1 = 0.f;
        for (int c = lane; c < cand; c += 64) {
            const int yo = ylo + c / nx, xo = xlo + c % nx;
            int y0, y1, x0, x1; float ly, lx;
            bl_coord(yo, sh, Hi, y0, y1, ly);
            bl_coord(xo, sw, Wi, x0, x1, lx);
            const float wy = (y0 == yi ? 1.f - ly : 0.f) + (y1 == yi ? ly : 0.f);
            const float wx = (x0 == xi ? 1.f - lx : 0.f) + (x1 == xi ? lx : 0.f);
            if (wy == 0.f || wx == 0.f) continue;
            const int64_t t = target[((int64_t)b * Ho + yo) * Wo + xo];
            if (t != 0 && t != 1) continue;
            const UpCe u = upce_at<T>(base, Wi, y0, y1, ly, x0, x1, lx);
            const float w = (t ? w1 : w0) * wy * wx;
            a0 += w * (expf(u.up0 - u.lse) - (t == 0 ? 1.f : 0.f));
            a1 += w * (expf(u.up1 - u.lse) - (t == 1 ? 1.f : 0.f));
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1);
        if (lane == 0) {
            dx[i * 2] = from_f<T>(a0 * gscale);
            dx[i * 2 + 1] = from_f<T>(a1 * gscale);
        }
    }
}

// Tiled form (round 5): a workgroup owns TL x TL low-resolution pixels.  Phase 1 evaluates every full-resolution pixel that can touch the tile ONCE
// (target, four neighbours, softmax) and parks w_t (softmax_c - [c == t]) in LDS; phase 2 gathers each low-resolution pixel's candidates from LDS, four
// lanes per pixel.  The wave-per-pixel form above evaluates ~100 candidates per low-resolution pixel (every full-resolution pixel up to 4 x, plus the
// zero-weight ring): 27.7 us at 2 x 480 x 480 -> 120 x 120 where this form evaluates ~27 per pixel.
constexpr int UPCE_TL = 8;
template <typename T>
__global__ __launch_bounds__(256) void upsample_ce_bwd_tile_kernel(const T* __restrict__ x, const int64_t* __restrict__ target, float w0, float w1,
                                                                   const float* __restrict__ stats, const float* __restrict__ dloss, T* __restrict__ dx,
                                                                   int B, int Hi, int Wi, int Ho, int Wo, float sh, float sw, int lds_pixels) {
    extern __shared__ __attribute__((aligned(16))) char upce_smem[];
    float2* g = reinterpret_cast<float2*>(upce_smem);
    const int tiles_x = (Wi + UPCE_TL - 1) / UPCE_TL, tiles_y = (Hi + UPCE_TL - 1) / UPCE_TL;
    const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, b = blockIdx.x / (tiles_x * tiles_y);
    const int yi0 = ty * UPCE_TL, xi0 = tx * UPCE_TL, yi1 = min(yi0 + UPCE_TL - 1, Hi - 1), xi1 = min(xi0 + UPCE_TL - 1, Wi - 1);
    int ylo, yhi, xlo, xhi, tmp;
    bl_range(yi0, sh, Hi, Ho, ylo, tmp); bl_range(yi1, sh, Hi, Ho, tmp, yhi);
    bl_range(xi0, sw, Wi, Wo, xlo, tmp); bl_range(xi1, sw, Wi, Wo, tmp, xhi);
    const int nx = xhi - xlo + 1, ny = yhi - ylo + 1;
    const T* base = x + (int64_t)b * Hi * Wi * 2;
    const float gscale = (stats[1] > 0.f ? 1.f / stats[1] : 0.f) * (dloss ? dloss[0] : 1.f);
    for (int idx = threadIdx.x; idx < nx * ny && idx < lds_pixels; idx += 256) {
        const int yo = ylo + idx / nx, xo = xlo + idx % nx;
        const int64_t t = target[((int64_t)b * Ho + yo) * Wo + xo];
        float2 v = make_float2(0.f, 0.f);
        if (t == 0 || t == 1) {
            int y0, y1, x0, x1; float ly, lx;
            bl_coord(yo, sh, Hi, y0, y1, ly);
            bl_coord(xo, sw, Wi, x0, x1, lx);
            const UpCe u = upce_at<T>(base, Wi, y0, y1, ly, x0, x1, lx);
            const float w = t ? w1 : w0;
            v.x = w * (expf(u.up0 - u.lse) - (t == 0 ? 1.f : 0.f));
            v.y = w * (expf(u.up1 - u.lse) - (t == 1 ? 1.f : 0.f));
        }
        g[idx] = v;
    }
    __syncthreads();
    const int p = threadIdx.x >> 2, sub = threadIdx.x & 3;
    const int yi = yi0 + p / UPCE_TL, xi = xi0 + p % UPCE_TL;
    const bool live = yi < Hi && xi < Wi;
    float a0 = 0.f, a1 = 0.f;
    if (live) {
        int cylo, cyhi, cxlo, cxhi;
        bl_range(yi, sh, Hi, Ho, cylo, cyhi);
        bl_range(xi, sw, Wi, Wo, cxlo, cxhi);
        for (int yo = cylo; yo <= cyhi; ++yo) {
            int y0, y1; float ly;
            bl_coord(yo, sh, Hi, y0, y1, ly);
            const float wy = (y0 == yi ? 1.f - ly : 0.f) + (y1 == yi ? ly : 0.f);
            if (wy == 0.f) continue;
            const float2* row = g + (yo - ylo) * nx - xlo;
            for (int xo = cxlo + sub; xo <= cxhi; xo += 4) {
                int x0, x1; float lx;
                bl_coord(xo, sw, Wi, x0, x1, lx);
                const float wx = (x0 == xi ? 1.f - lx : 0.f) + (x1 == xi ? lx : 0.f);
                const float2 v = row[xo];
                a0 += wy * wx * v.x;
                a1 += wy * wx * v.y;
            }
        }
    }
    a0 += __shfl_xor(a0, 1, 64); a0 += __shfl_xor(a0, 2, 64);
    a1 += __shfl_xor(a1, 1, 64); a1 += __shfl_xor(a1, 2, 64);
    if (live && sub == 0) {
        const int64_t i = ((int64_t)b * Hi + yi) * Wi + xi;
        dx[i * 2] = from_f<T>(a0 * gscale);
        dx[i * 2 + 1] = from_f<T>(a1 * gscale);
    }
}

template <typename T> __global__ void logits_up_fwd_kernel(const T* x, float* y, int B, int Hi, int Wi, int Ho, int Wo, float sh, float sw) {
    const int64_t n = (int64_t)B * Ho * Wo;
    GRID_STRIDE(i, n) {
        const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho), b = (int)(i / Wo / Ho);
        int y0, y1, x0, x1; float ly, lx;
        bl_coord(yo, sh, Hi, y0, y1, ly);
        bl_coord(xo, sw, Wi, x0, x1, lx);
        const T* base = x + (int64_t)b * Hi * Wi * 2;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float f00 = to_f<T>(base[((int64_t)y0 * Wi + x0) * 2 + c]), f01 = to_f<T>(base[((int64_t)y0 * Wi + x1) * 2 + c]);
            const float f10 = to_f<T>(base[((int64_t)y1 * Wi + x0) * 2 + c]), f11 = to_f<T>(base[((int64_t)y1 * Wi + x1) * 2 + c]);
            y[(((int64_t)b * 2 + c) * Ho + yo) * Wo + xo] = (1.f - ly) * ((1.f - lx) * f00 + lx * f01) + ly * ((1.f - lx) * f10 + lx * f11);
        }
    }
}
template <typename T> __global__ void logits_up_bwd_kernel(const float* dy, T* dx, int B, int Hi, int Wi, int Ho, int Wo, float sh, float sw) {
    const int64_t n = (int64_t)B * Hi * Wi;
    GRID_STRIDE(i, n) {
        const int xi = (int)(i % Wi), yi = (int)((i / Wi) % Hi), b = (int)(i / Wi / Hi);
        int ylo, yhi, xlo, xhi;
        bl_range(yi, sh, Hi, Ho, ylo, yhi);
        bl_range(xi, sw, Wi, Wo, xlo, xhi);
        float a0 = 0.f, a1 = 0.f;
        for (int yo = ylo; yo <= yhi; ++yo) {
            int y0, y1; float ly;
            bl_coord(yo, sh, Hi, y0, y1, ly);
            const float wy = (y0 == yi ? 1.f - ly : 0.f) + (y1 == yi ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int xo = xlo; xo <= xhi; ++xo) {
                int x0, x1; float lx;
                bl_coord(xo, sw, Wi, x0, x1, lx);
                const float wx = (x0 == xi ? 1.f - lx : 0.f) + (x1 == xi ? lx : 0.f);
                if (wx == 0.f) continue;
                a0 += wy * wx * dy[(((int64_t)b * 2 + 0) * Ho + yo) * Wo + xo];
                a1 += wy * wx * dy[(((int64_t)b * 2 + 1) * Ho + yo) * Wo + xo];
            }
        }
        dx[i * 2] = from_f<T>(a0);
        dx[i * 2 + 1] = from_f<T>(a1);
    }
}

// ---------------------------------------------------------------------------------------------- classifier head (hidden -> 2)
template <typename T> __global__ __launch_bounds__(256) void cls_head_fwd_kernel(const T* x, const float* w, const float* bias, T* y, int64_t rows, int C) {
    constexpr int EPC = Chunk<T>::N;
    const int lane = threadIdx.x & 63;
    const int nchunk = C / EPC;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        float a0 = 0.f, a1 = 0.f;
        for (int ch = lane; ch < nchunk; ch += 64) {
            float f[EPC];
            chunk_to_f<T>(*reinterpret_cast<const uint4*>(x + row * C + ch * EPC), f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) { a0 = fmaf(f[e], w[ch * EPC + e], a0); a1 = fmaf(f[e], w[C + ch * EPC + e], a1); }
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1);
        if (lane == 0) { y[row * 2] = from_f<T>(a0 + bias[0]); y[row * 2 + 1] = from_f<T>(a1 + bias[1]); }
    }
}
// PART: the workgroup's sums go to its own record of a partial table (pw [blocks][2 C], pb [blocks][2]; reduced with the LayerNorm partial sums by
// lavt_reduce_partials_multi at the end of backward) instead of 2 C + 2 global atomics per workgroup: 450 workgroups adding to the same 1026 addresses
// were most of the launch (31 us for 59 MB of rows at 2 x 120 x 120 x 512), and their order changed the sums from run to run.
template <typename T, bool PART> __global__ __launch_bounds__(256) void cls_head_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, float* dw, float* db, int64_t rows, int C) {
    // thread owns one chunk column for a strip of rows: dx = dy0*w0 + dy1*w1 ; dw[c] += dy[c]*x ; db += dy
    constexpr int EPC = Chunk<T>::N, U = 8;
    const int cpr = C / EPC;
    const int tc = threadIdx.x % cpr, tr = threadIdx.x / cpr, rstep = blockDim.x / cpr;
    const bool live = tr < rstep;
    float w0[EPC], w1[EPC], g0[EPC], g1[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { w0[e] = w[tc * EPC + e]; w1[e] = w[C + tc * EPC + e]; g0[e] = 0.f; g1[e] = 0.f; }
    float b0 = 0.f, b1 = 0.f;
    // eight rows per trip, every load of the trip issued before the first use (one row per trip was a chain of ~14 dependent L2 / HBM round trips per
    // thread: 40 us for the 2 x 120 x 120 x 512 map where the bytes take ~12)
    const int64_t rs = (int64_t)gridDim.x * rstep;
    for (int64_t row = (int64_t)blockIdx.x * rstep + tr; live && row < rows; row += U * rs) {
        uint4 xv[U];
        float d0[U], d1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = row + u * rs;
            const bool ok = r < rows;
            const int64_t rr = ok ? r : row;
            xv[u] = *reinterpret_cast<const uint4*>(x + rr * C + tc * EPC);
            d0[u] = ok ? to_f<T>(dy[rr * 2]) : 0.f;
            d1[u] = ok ? to_f<T>(dy[rr * 2 + 1]) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = row + u * rs;
            if (r >= rows) break;
            float f[EPC], o[EPC];
            chunk_to_f<T>(xv[u], f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) { o[e] = d0[u] * w0[e] + d1[u] * w1[e]; g0[e] += d0[u] * f[e]; g1[e] += d1[u] * f[e]; }
            *reinterpret_cast<uint4*>(dx + r * C + tc * EPC) = f_to_chunk<T>(o);
            if (tc == 0) { b0 += d0[u]; b1 += d1[u]; }
        }
    }
    // combine the row-lanes of the workgroup in LDS first (fixed order: row-lane by row-lane)
    __shared__ float red[2 * 2048 + 2];
    for (int e = threadIdx.x; e < 2 * C + 2; e += blockDim.x) red[e] = 0.f;
    __syncthreads();
    for (int k = 0; k < rstep; ++k) {
        if (live && tr == k) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) { red[tc * EPC + e] += g0[e]; red[C + tc * EPC + e] += g1[e]; }
            if (tc == 0) { red[2 * C] += b0; red[2 * C + 1] += b1; }
        }
        __syncthreads();
    }
    if constexpr (PART) {
        for (int e = threadIdx.x; e < 2 * C; e += blockDim.x) dw[(int64_t)blockIdx.x * 2 * C + e] = red[e];
        if (threadIdx.x < 2) db[blockIdx.x * 2 + threadIdx.x] = red[2 * C + threadIdx.x];
    } else {
        for (int e = threadIdx.x; e < 2 * C; e += blockDim.x) atomicAdd(dw + e, red[e]);
        if (threadIdx.x < 2) atomicAdd(db + threadIdx.x, red[2 * C + threadIdx.x]);
    }
}

// ---------------------------------------------------------------------------------------------- patch-embed im2col (4x4 / stride 4)
// cols[(b*H4+py)*W4+px][c*16 + ky*4 + kx] = img[b][c][4py+ky][4px+kx]   (zero beyond H,W); matches Conv2d weight.view(Cout, 48)
template <typename T> __global__ void im2col4_kernel(const float* img, T* cols, int B, int H, int W, int H4, int W4) {
    const int64_t n = (int64_t)B * H4 * W4 * 12;      // 12 = 3 channels * 4 kernel rows; each item = 4 contiguous pixels
    GRID_STRIDE(i, n) {
        const int ky = (int)(i % 4), c = (int)((i / 4) % 3);
        const int64_t pix = i / 12;
        const int px = (int)(pix % W4), py = (int)((pix / W4) % H4), b = (int)(pix / W4 / H4);
        const int y = 4 * py + ky;
        T* dst = cols + pix * 48 + c * 16 + ky * 4;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
            const int x = 4 * px + kx;
            dst[kx] = from_f<T>((y < H && x < W) ? img[(((int64_t)b * 3 + c) * H + y) * W + x] : 0.f);
        }
    }
}
template <typename T> __global__ void col2im4_kernel(const T* dcols, float* dimg, int B, int H, int W, int H4, int W4) {
    const int64_t n = (int64_t)B * 3 * H * W;
    GRID_STRIDE(i, n) {
        const int x = (int)(i % W), y = (int)((i / W) % H), c = (int)((i / W / H) % 3), b = (int)(i / W / H / 3);
        const int64_t pix = ((int64_t)b * H4 + y / 4) * W4 + x / 4;
        dimg[i] = to_f<T>(dcols[pix * 48 + c * 16 + (y % 4) * 4 + (x % 4)]);
    }
}

// ---------------------------------------------------------------------------------------------- casts / layout
template <typename S, typename D> __global__ void cast_kernel(const S* src, D* dst, int64_t n) {
    GRID_STRIDE(i, n) dst[i] = from_f<D>(to_f<S>(src[i]));
}
// [B][C][HW] <-> [B][HW][C] through a 32x33 LDS tile
template <typename S, typename D> __global__ void transpose_kernel(const S* src, D* dst, int R, int Cc) {
    // src: [batch][R][Cc] -> dst: [batch][Cc][R]
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const S* s = src + (int64_t)b * R * Cc;
    D* d = dst + (int64_t)b * R * Cc;
    for (int k = threadIdx.y; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + threadIdx.x;
        if (r < R && c < Cc) tile[k][threadIdx.x] = to_f<S>(s[(int64_t)r * Cc + c]);
    }
    __syncthreads();
    for (int k = threadIdx.y; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + threadIdx.x;
        if (r < R && c < Cc) d[(int64_t)c * R + r] = from_f<D>(tile[threadIdx.x][k]);
    }
}
template <typename T> __global__ void pack_conv3x3_kernel(const float* w, T* packed, int Cout, int Cin, int taps) {
    const int64_t n = (int64_t)Cout * taps * Cin;
    GRID_STRIDE(i, n) {
        const int ci = (int)(i % Cin), tap = (int)((i / Cin) % taps), co = (int)(i / Cin / taps);
        packed[i] = from_f<T>(w[((int64_t)co * Cin + ci) * taps + tap]);
    }
}
// inverse of the packing for gradients: dw[co][ci][tap] += packed[co][tap][ci] (the conv weight-gradient GEMM writes the packed layout so that
// its split-K atomics stay contiguous: scattered 4-byte atomics, 36 bytes apart for 9 taps, run ~17x below the contiguous atomic rate)
__global__ void unpack_conv_grad_kernel(const float* __restrict__ packed, float* __restrict__ dw, int Cout, int Cin, int taps) {
    const int64_t n = (int64_t)Cout * taps * Cin;
    GRID_STRIDE(i, n) {
        const int tap = (int)(i % taps), ci = (int)((i / taps) % Cin), co = (int)(i / taps / Cin);          // i walks dw (coalesced writes)
        dw[i] += packed[((int64_t)co * taps + tap) * Cin + ci];
    }
}
// the same through an LDS tile: one workgroup per (co, 128 input channels) reads `taps` contiguous 512-byte rows of the packed gradient and writes
// 128 * taps contiguous floats of dw (the element-wise form reads 4-byte pieces Cin floats apart and divides three times per element: 12 us for a
// 512 x 4608 matrix that is 19 MB of traffic)
__global__ __launch_bounds__(256) void unpack_conv_grad_tiled_kernel(const float* __restrict__ packed, float* __restrict__ dw, int Cin, int taps) {
    extern __shared__ float tile[];                                  // [taps][129]
    const int co = blockIdx.y, ci0 = blockIdx.x * 128, nci = min(128, Cin - ci0);
    const float* src = packed + (int64_t)co * taps * Cin + ci0;
    for (int e = threadIdx.x; e < taps * 128; e += 256) {
        const int t = e >> 7, c = e & 127;
        if (c < nci) tile[t * 129 + c] = src[(int64_t)t * Cin + c];
    }
    __syncthreads();
    float* dst = dw + ((int64_t)co * Cin + ci0) * taps;
    const int n = nci * taps;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int c = e / taps, t = e - c * taps;
        dst[e] += tile[t * 129 + c];
    }
}
// ---- multi-tensor AdamW with the poly learning-rate schedule of the caller (train.py:688-700: torch.optim.AdamW + LambdaLR((1 - it/T)^0.9)).
// One launch for the whole parameter list: blockIdx.y = tensor, descriptor {param, grad, exp_avg, exp_avg_sq, numel} (fp32 pointers),
// per-tensor hyper-parameters {base_lr, weight_decay, beta1, beta2, eps}.  The step counter lives on the device (incremented by
// adamw_tick_kernel), so the optimizer step can be part of the captured hipGraph and still advance its schedule on every replay.
__global__ __launch_bounds__(256) void adamw_multi_kernel(const int64_t* __restrict__ desc, const float* __restrict__ hyper, int count,
                                                          const float* __restrict__ step, float total_steps, float power) {
    const int t = blockIdx.y;
    if (t >= count) return;
    float* p = reinterpret_cast<float*>(desc[5 * t]);
    const float* g = reinterpret_cast<const float*>(desc[5 * t + 1]);
    float* m = reinterpret_cast<float*>(desc[5 * t + 2]);
    float* v = reinterpret_cast<float*>(desc[5 * t + 3]);
    const int64_t n = desc[5 * t + 4];
    const float b1 = hyper[5 * t + 2], b2 = hyper[5 * t + 3], eps = hyper[5 * t + 4], wd = hyper[5 * t + 1];
    const float k = step[0];                                       // optimizer steps taken so far
    const float sched = total_steps > 0.f ? powf(fmaxf(1.f - k / total_steps, 0.f), power) : 1.f;
    const float lr = hyper[5 * t] * sched;
    const float bc1 = 1.f - powf(b1, k + 1.f), bc2 = 1.f - powf(b2, k + 1.f);
    const float step_size = lr / bc1, rbc2 = rsqrtf(bc2), decay = 1.f - lr * wd;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        pp *= decay;
        mm = b1 * mm + (1.f - b1) * gg;
        vv = b2 * vv + (1.f - b2) * gg * gg;
        pp -= step_size * mm / (sqrtf(vv) * rbc2 + eps);
    };
    const bool vec = ((desc[5 * t] | desc[5 * t + 1] | desc[5 * t + 2] | desc[5 * t + 3]) & 15) == 0;
    if (vec) {
        const int64_t n4 = n >> 2;
        GRID_STRIDE(i, n4) {
            float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
            const float4 gg = reinterpret_cast<const float4*>(g)[i];
            upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
            reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
        }
        for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) upd(p[i], g[i], m[i], v[i]);
    } else {
        GRID_STRIDE(i, n) upd(p[i], g[i], m[i], v[i]);
    }
}
__global__ void adamw_tick_kernel(float* step) { step[0] += 1.f; }

// The same update over a CHUNK table: workgroup c updates elements [chunk * CE, (chunk + 1) * CE) of tensor chunks[c].x -- every workgroup has work
// (the per-tensor grid above launches 256 workgroups per tensor: ~100 000 of them find nothing to do for the ~400 small tensors of a Swin-B LAVT)
// and all of a thread's loads are issued before the first use.  desc: int64 [count][6] = {param, grad, exp_avg, exp_avg_sq, numel, copy}: a
// non-zero `copy` is the parameter's bf16 compute copy in the same layout (Linear / 1x1 weights), written from the registers that hold the
// updated value -- the separate re-cast pass (a second read of every parameter) disappears.
constexpr int ADAMW_CE = 8192;            // elements per chunk: 256 threads x 2 rounds x 4 float4
__global__ __launch_bounds__(256) void adamw_chunks_kernel(const int64_t* __restrict__ desc, const float* __restrict__ hyper, const int2* __restrict__ chunks,
                                                           const float* __restrict__ step, float total_steps, float power) {
    const int2 ch = chunks[blockIdx.x];
    const int t = ch.x;
    float* p = reinterpret_cast<float*>(desc[6 * t]);
    const float* g = reinterpret_cast<const float*>(desc[6 * t + 1]);
    float* m = reinterpret_cast<float*>(desc[6 * t + 2]);
    float* v = reinterpret_cast<float*>(desc[6 * t + 3]);
    const int64_t n = desc[6 * t + 4];
    bf16* cp = reinterpret_cast<bf16*>(desc[6 * t + 5]);
    const float b1 = hyper[5 * t + 2], b2 = hyper[5 * t + 3], eps = hyper[5 * t + 4], wd = hyper[5 * t + 1];
    const float k = step[0];
    const float sched = total_steps > 0.f ? powf(fmaxf(1.f - k / total_steps, 0.f), power) : 1.f;
    const float lr = hyper[5 * t] * sched;
    const float bc1 = 1.f - powf(b1, k + 1.f), bc2 = 1.f - powf(b2, k + 1.f);
    const float step_size = lr / bc1, rbc2 = rsqrtf(bc2), decay = 1.f - lr * wd;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        pp *= decay;
        mm = b1 * mm + (1.f - b1) * gg;
        vv = b2 * vv + (1.f - b2) * gg * gg;
        pp -= step_size * mm / (sqrtf(vv) * rbc2 + eps);
    };
    const int64_t e0 = (int64_t)ch.y * ADAMW_CE, e1 = min(n, e0 + ADAMW_CE);
    const bool vec = ((desc[6 * t] | desc[6 * t + 1] | desc[6 * t + 2] | desc[6 * t + 3]) & 15) == 0 && (desc[6 * t + 5] & 7) == 0;
    if (vec && e1 - e0 == ADAMW_CE) {          // a whole chunk: no per-element conditions (a conditional load costs a branch and a full wait each)
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            float4 pp[4], gg[4], mm[4], vv[4];
            const int64_t base = e0 + (int64_t)round * 4096 + threadIdx.x * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = base + u * 1024;
                pp[u] = *reinterpret_cast<const float4*>(p + i); gg[u] = *reinterpret_cast<const float4*>(g + i);
                mm[u] = *reinterpret_cast<const float4*>(m + i); vv[u] = *reinterpret_cast<const float4*>(v + i);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = base + u * 1024;
                upd(pp[u].x, gg[u].x, mm[u].x, vv[u].x); upd(pp[u].y, gg[u].y, mm[u].y, vv[u].y);
                upd(pp[u].z, gg[u].z, mm[u].z, vv[u].z); upd(pp[u].w, gg[u].w, mm[u].w, vv[u].w);
                *reinterpret_cast<float4*>(p + i) = pp[u]; *reinterpret_cast<float4*>(m + i) = mm[u]; *reinterpret_cast<float4*>(v + i) = vv[u];
                if (cp) *reinterpret_cast<uint2*>(cp + i) = make_uint2(pack_bf16x2(pp[u].x, pp[u].y), pack_bf16x2(pp[u].z, pp[u].w));
            }
        }
    } else {                                   // a tensor's last chunk / unaligned tensors
        for (int64_t i = e0 + threadIdx.x; i < e1; i += 256) {
            upd(p[i], g[i], m[i], v[i]);
            if (cp) cp[i] = from_f<bf16>(p[i]);
        }
    }
}

template <typename D> __global__ void cast_multi_kernel(const int64_t* desc, int count) {
    // blockIdx.y = tensor; grid-stride over its elements
    const int t = blockIdx.y;
    if (t >= count) return;
    const float* src = reinterpret_cast<const float*>(desc[3 * t]);
    D* dst = reinterpret_cast<D*>(desc[3 * t + 1]);
    const int64_t n = desc[3 * t + 2];
    GRID_STRIDE(i, n) dst[i] = from_f<D>(src[i]);
}

}  // namespace

#define DISPATCH_T(dtype, NAME, ...)                                   \
    if (dtype == LAVT_F32) { using T = float; __VA_ARGS__; }           \
    else if (dtype == LAVT_BF16) { using T = bf16; __VA_ARGS__; }      \
    else { lavt_set_error(NAME ": bad dtype %d", dtype); return LAVT_ERR_INVALID; }
#define ST reinterpret_cast<hipStream_t>(stream)
#define EPC_OF(dtype) ((dtype) == LAVT_F32 ? 4 : 8)

extern "C" int lavt_act_bwd(int dtype, int act, const void* dy, const void* pre, void* dx, int64_t n, void* stream) {
    LAVT_CHECK_ARG(dy && pre && dx && n > 0 && n % EPC_OF(dtype) == 0, "lavt_act_bwd: bad arguments (n=%ld)", (long)n);
    const int64_t nc = n / EPC_OF(dtype);
    DISPATCH_T(dtype, "lavt_act_bwd", hipLaunchKernelGGL(act_bwd_kernel<T>, dim3(ew_grid(nc)), dim3(256), 0, ST, act, (const T*)dy, (const T*)pre, (T*)dx, nc));
    LAVT_CHECK_LAUNCH("lavt_act_bwd");
    return LAVT_OK;
}
// ---------------------------------------------------------------------------------------------- O(batch) glue of a forward, one launch each
// (as torch ops these were ~10 element-wise launches of ~5 us per step, each a node of the captured chain)
// language mask l_mask [B][n_l] (float32 or int64, 0 / 1) -> mask_rows [B * n_l] (the float mask) and maskbias [B][ld] = 1e4 * m - 1e4, -1e4 beyond
// n_l (reference lib/backbone.py:1360: padded words get -1e4 before the softmax over words)
__global__ void lang_mask_kernel(const float* mf, const int64_t* mi, float* rows, float* bias, int B, int n_l, int ld) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * ld) return;
    const int b = e / ld, j = e - b * ld;
    float m = 0.f;
    if (j < n_l) {
        m = mf ? mf[b * n_l + j] : (float)mi[b * n_l + j];
        rows[b * n_l + j] = m;
    }
    bias[e] = j < n_l ? 1e4f * m - 1e4f : -1e4f;
}
extern "C" int lavt_lang_mask(const void* l_mask, int is_int64, float* mask_rows, float* maskbias, int B, int n_l, int ld, void* stream) {
    LAVT_CHECK_ARG(l_mask && mask_rows && maskbias && B > 0 && n_l > 0 && ld >= n_l, "lavt_lang_mask: bad arguments");
    hipLaunchKernelGGL(lang_mask_kernel, dim3(cdiv((long)B * ld, 256)), dim3(256), 0, ST, is_int64 ? nullptr : (const float*)l_mask, is_int64 ? (const int64_t*)l_mask : nullptr,
                       mask_rows, maskbias, B, n_l, ld);
    LAVT_CHECK_LAUNCH("lavt_lang_mask");
    return LAVT_OK;
}
// DropPath factors of all branches of a forward from one uniform draw u [n][B]: f = floor(keep[n] + u) / keep[n] (timm's drop_path; reference
// lib/backbone.py:6, 240-245)
__global__ void droppath_factors_kernel(const float* u, const float* keep, float* f, int n, int B) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * B) return;
    const float k = keep[e / B];
    f[e] = floorf(k + u[e]) / k;
}
extern "C" int lavt_droppath_factors(const float* u, const float* keep, float* f, int n, int B, void* stream) {
    LAVT_CHECK_ARG(u && keep && f && n > 0 && B > 0, "lavt_droppath_factors: bad arguments");
    hipLaunchKernelGGL(droppath_factors_kernel, dim3(cdiv((long)n * B, 256)), dim3(256), 0, ST, u, keep, f, n, B);
    LAVT_CHECK_LAUNCH("lavt_droppath_factors");
    return LAVT_OK;
}

// DropPath factors with the uniform draw made here (round 5): Philox4x32-10 keyed by state[0] (seed), counter = (state[1], element index); the kernel
// advances state[1] itself, so a hipGraph replay draws fresh numbers with no host involvement.  (torch.rand under capture costs two bookkeeping fills per
// replay -- the graph's seed / offset tensors -- besides its own launch: 3 x 4.7 us at the head of every step.)  One workgroup: every thread has read
// the counter before thread 0 writes it.
__device__ __forceinline__ uint32_t philox_u32(uint32_t seed_lo, uint32_t seed_hi, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    uint32_t k0 = seed_lo, k1 = seed_hi;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}
__global__ __launch_bounds__(256) void droppath_draw_kernel(unsigned long long* __restrict__ state, const float* __restrict__ keep, float* __restrict__ f, int n, int B) {
    const unsigned long long seed = state[0], ctr = state[1];
    for (int e = threadIdx.x; e < n * B; e += 256) {
        const uint32_t x = philox_u32((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)e, 0u);
        const float u = (float)(x >> 8) * (1.0f / 16777216.0f);          // 24 bits: [0, 1)
        const float k = keep[e / B];
        f[e] = floorf(k + u) / k;
    }
    __syncthreads();
    if (threadIdx.x == 0) state[1] = ctr + 1ull;
}
extern "C" int lavt_droppath_draw(void* state, const float* keep, float* f, int n, int B, void* stream) {
    LAVT_CHECK_ARG(state && keep && f && n > 0 && B > 0, "lavt_droppath_draw: bad arguments");
    hipLaunchKernelGGL(droppath_draw_kernel, dim3(1), dim3(256), 0, ST, (unsigned long long*)state, keep, f, n, B);
    LAVT_CHECK_LAUNCH("lavt_droppath_draw");
    return LAVT_OK;
}

extern "C" int lavt_gate_fwd(int dtype, const void* x, const void* gpre, const void* r, void* xo, int64_t n, void* stream) {
    LAVT_CHECK_ARG(x && gpre && r && xo && n > 0 && n % EPC_OF(dtype) == 0, "lavt_gate_fwd: bad arguments");
    const int64_t nc = n / EPC_OF(dtype);
    DISPATCH_T(dtype, "lavt_gate_fwd", hipLaunchKernelGGL(gate_fwd_kernel<T>, dim3(ew_grid(nc)), dim3(256), 0, ST, (const T*)x, (const T*)gpre, (const T*)r, (T*)xo, nc));
    LAVT_CHECK_LAUNCH("lavt_gate_fwd");
    return LAVT_OK;
}
extern "C" int lavt_gate_bwd(int dtype, const void* dxo, const void* gpre, const void* r, const void* dr_add, void* dgpre, void* dr, int64_t n, void* stream) {
    LAVT_CHECK_ARG(dxo && gpre && r && dgpre && dr && n > 0 && n % EPC_OF(dtype) == 0, "lavt_gate_bwd: bad arguments");
    const int64_t nc = n / EPC_OF(dtype);
    DISPATCH_T(dtype, "lavt_gate_bwd", hipLaunchKernelGGL(gate_bwd_kernel<T>, dim3(ew_grid(nc)), dim3(256), 0, ST, (const T*)dxo, (const T*)gpre, (const T*)r, (const T*)dr_add, (T*)dgpre, (T*)dr, nc));
    LAVT_CHECK_LAUNCH("lavt_gate_bwd");
    return LAVT_OK;
}
extern "C" int lavt_rowsoftmax_fwd(int dtype, const void* s, void* p, int64_t rows, int n_l, int ld, void* stream) {
    LAVT_CHECK_ARG(s && p && rows > 0 && n_l > 0 && n_l <= ld && ld <= 32 && ld % EPC_OF(dtype) == 0, "lavt_rowsoftmax_fwd: bad arguments (ld <= 32, a multiple of the 16-byte chunk)");
    DISPATCH_T(dtype, "lavt_rowsoftmax_fwd", hipLaunchKernelGGL(rowsoftmax_fwd_kernel<T>, dim3(ew_grid(rows)), dim3(256), 0, ST, (const T*)s, (T*)p, rows, n_l, ld));
    LAVT_CHECK_LAUNCH("lavt_rowsoftmax_fwd");
    return LAVT_OK;
}
extern "C" int lavt_rowsoftmax_bwd(int dtype, const void* p, const void* dp, void* ds, int64_t rows, int n_l, int ld, void* stream) {
    LAVT_CHECK_ARG(p && dp && ds && rows > 0 && n_l > 0 && n_l <= ld && ld <= 32 && ld % EPC_OF(dtype) == 0, "lavt_rowsoftmax_bwd: bad arguments (ld <= 32, a multiple of the 16-byte chunk)");
    DISPATCH_T(dtype, "lavt_rowsoftmax_bwd", hipLaunchKernelGGL(rowsoftmax_bwd_kernel<T>, dim3(ew_grid(rows)), dim3(256), 0, ST, (const T*)p, (const T*)dp, (T*)ds, rows, n_l, ld));
    LAVT_CHECK_LAUNCH("lavt_rowsoftmax_bwd");
    return LAVT_OK;
}
extern "C" int lavt_bilinear_fwd(int dtype, const void* x, void* y, int B, int Hi, int Wi, int Ho, int Wo, int C, void* stream) {
    LAVT_CHECK_ARG(x && y && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C % EPC_OF(dtype) == 0, "lavt_bilinear_fwd: bad arguments");
    LAVT_CHECK_ARG((int64_t)B * Ho * Wo * (C / EPC_OF(dtype)) < (1LL << 31) - (1LL << 21), "lavt_bilinear_fwd: more than 2^31 output chunks");
    const int64_t nc = (int64_t)B * Ho * Wo * (C / EPC_OF(dtype));
    if (dtype == LAVT_BF16 && lavt_tuning().probe[7] != 1) {          // row-staged form (LDS-DMA): every decoder shape
        const int cblk = bl_rows_cblk(B, Hi, Wi, C);
        if (cblk) {
            hipLaunchKernelGGL((bilinear_rows_fwd_kernel<false>), dim3(Hi, B, C / cblk), dim3(256), bl_rows_lds(Wi, cblk), ST, (const bf16*)x, (bf16*)y, Hi, Wi, Ho, Wo, C, cblk,
                               bl_scale(Hi, Ho), bl_scale(Wi, Wo), (unsigned char*)nullptr, (const float*)nullptr, (float*)nullptr);
            LAVT_CHECK_LAUNCH("lavt_bilinear_fwd");
            return LAVT_OK;
        }
    }
    DISPATCH_T(dtype, "lavt_bilinear_fwd", hipLaunchKernelGGL(bilinear_fwd_kernel<T>, dim3(ew_grid(nc)), dim3(256), 0, ST, (const T*)x, (T*)y, B, Hi, Wi, Ho, Wo, C, bl_scale(Hi, Ho), bl_scale(Wi, Wo)));
    LAVT_CHECK_LAUNCH("lavt_bilinear_fwd");
    return LAVT_OK;
}
/* lavt_bilinear_fwd (bf16) with an e4m3 twin of the output: see lavt_norm_apply_q8 */
extern "C" int lavt_bilinear_fwd_q8(const void* x, void* y, void* q, const float* amax_prev, float* amax_cur, int B, int Hi, int Wi, int Ho, int Wo, int C, void* stream) {
    LAVT_CHECK_ARG(x && y && q && amax_cur && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C % 8 == 0, "lavt_bilinear_fwd_q8: bad arguments");
    const int64_t nc = (int64_t)B * Ho * Wo * (C / 8);
    LAVT_CHECK_ARG(nc < (1LL << 31) - (1LL << 21), "lavt_bilinear_fwd_q8: more than 2^31 output chunks");
    if (lavt_tuning().probe[7] != 1) {
        const int cblk = bl_rows_cblk(B, Hi, Wi, C);
        if (cblk && (long)Hi * B * (C / cblk) <= 1024) {          // (one same-address |max| atomic per workgroup: keep them few)
            hipLaunchKernelGGL((bilinear_rows_fwd_kernel<true>), dim3(Hi, B, C / cblk), dim3(256), bl_rows_lds(Wi, cblk), ST, (const bf16*)x, (bf16*)y, Hi, Wi, Ho, Wo, C, cblk,
                               bl_scale(Hi, Ho), bl_scale(Wi, Wo), (unsigned char*)q, amax_prev, amax_cur);
            LAVT_CHECK_LAUNCH("lavt_bilinear_fwd_q8");
            return LAVT_OK;
        }
    }
    // at most 1024 workgroups: each ends with one same-address atomic (|max|), and those serialise at ~20 ns apiece (3600 of them made the 60x60 launch 55 us)
    const int grid = ew_grid(nc) < 1024 ? ew_grid(nc) : 1024;
    hipLaunchKernelGGL((bilinear_fwd_kernel<bf16, true>), dim3(grid), dim3(256), 0, ST, (const bf16*)x, (bf16*)y, B, Hi, Wi, Ho, Wo, C, bl_scale(Hi, Ho), bl_scale(Wi, Wo),
                       (unsigned char*)q, amax_prev, amax_cur);
    LAVT_CHECK_LAUNCH("lavt_bilinear_fwd_q8");
    return LAVT_OK;
}

extern "C" int lavt_bilinear_bwd(int dtype, const void* dy, void* dx, int B, int Hi, int Wi, int Ho, int Wo, int C, void* stream) {
    LAVT_CHECK_ARG(dy && dx && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C % EPC_OF(dtype) == 0, "lavt_bilinear_bwd: bad arguments");
    LAVT_CHECK_ARG((int64_t)B * Hi * Wi * (C / EPC_OF(dtype)) < (1LL << 31) - (1LL << 21), "lavt_bilinear_bwd: more than 2^31 input chunks");
    const int64_t nc = (int64_t)B * Hi * Wi * (C / EPC_OF(dtype));
    DISPATCH_T(dtype, "lavt_bilinear_bwd", hipLaunchKernelGGL(bilinear_bwd_kernel<T>, dim3(ew_grid(nc)), dim3(256), 0, ST, (const T*)dy, (T*)dx, B, Hi, Wi, Ho, Wo, C, bl_scale(Hi, Ho), bl_scale(Wi, Wo)));
    LAVT_CHECK_LAUNCH("lavt_bilinear_bwd");
    return LAVT_OK;
}
extern "C" int lavt_logits_up_fwd(int dtype, const void* x, float* y, int B, int Hi, int Wi, int Ho, int Wo, void* stream) {
    LAVT_CHECK_ARG(x && y && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "lavt_logits_up_fwd: bad arguments");
    const int64_t n = (int64_t)B * Ho * Wo;
    DISPATCH_T(dtype, "lavt_logits_up_fwd", hipLaunchKernelGGL(logits_up_fwd_kernel<T>, dim3(ew_grid(n)), dim3(256), 0, ST, (const T*)x, y, B, Hi, Wi, Ho, Wo, bl_scale(Hi, Ho), bl_scale(Wi, Wo)));
    LAVT_CHECK_LAUNCH("lavt_logits_up_fwd");
    return LAVT_OK;
}
extern "C" int lavt_logits_up_bwd(int dtype, const float* dy, void* dx, int B, int Hi, int Wi, int Ho, int Wo, void* stream) {
    LAVT_CHECK_ARG(dy && dx && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "lavt_logits_up_bwd: bad arguments");
    const int64_t n = (int64_t)B * Hi * Wi;
    DISPATCH_T(dtype, "lavt_logits_up_bwd", hipLaunchKernelGGL(logits_up_bwd_kernel<T>, dim3(ew_grid(n)), dim3(256), 0, ST, dy, (T*)dx, B, Hi, Wi, Ho, Wo, bl_scale(Hi, Ho), bl_scale(Wi, Wo)));
    LAVT_CHECK_LAUNCH("lavt_logits_up_bwd");
    return LAVT_OK;
}
extern "C" int lavt_upsample_ce_fwd(int dtype, const void* x, const int64_t* target, float w0, float w1, float* ws, int64_t ws_floats,
                                    float* out4, int B, int Hi, int Wi, int Ho, int Wo, void* stream) {
    LAVT_CHECK_ARG(x && target && ws && out4 && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "lavt_upsample_ce_fwd: bad arguments");
    const int64_t n = (int64_t)B * Ho * Wo;
    int blocks = (int)((n + 1023) / 1024);
    if (blocks > 2048) blocks = 2048;
    LAVT_CHECK_ARG(ws_floats >= (int64_t)blocks * 4, "lavt_upsample_ce_fwd: scratch of %d floats needed", blocks * 4);
    DISPATCH_T(dtype, "lavt_upsample_ce_fwd", hipLaunchKernelGGL(upsample_ce_fwd_kernel<T>, dim3(blocks), dim3(256), 0, ST, (const T*)x, target, w0, w1, ws, B, Hi, Wi, Ho, Wo, bl_scale(Hi, Ho), bl_scale(Wi, Wo)));
    hipLaunchKernelGGL(upsample_ce_finish_kernel, dim3(1), dim3(256), 0, ST, ws, blocks, out4);
    LAVT_CHECK_LAUNCH("lavt_upsample_ce_fwd");
    return LAVT_OK;
}
extern "C" int lavt_upsample_ce_bwd(int dtype, const void* x, const int64_t* target, float w0, float w1, const float* out4, const float* dloss,
                                    void* dx, int B, int Hi, int Wi, int Ho, int Wo, void* stream) {
    LAVT_CHECK_ARG(x && target && out4 && dx && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "lavt_upsample_ce_bwd: bad arguments");
    const int64_t n = (int64_t)B * Hi * Wi;
    // tiled form: the full-resolution region of a TL x TL tile must fit LDS (bl_range: (TL + 1) / scale + 5 rows / columns)
    const float sh = bl_scale(Hi, Ho), sw = bl_scale(Wi, Wo);
    if (sh > 0.f && sw > 0.f && !lavt_tuning().upce_tile_off) {
        const long ny = (long)((UPCE_TL + 1) / sh) + 6, nx = (long)((UPCE_TL + 1) / sw) + 6;
        const long tiles = (long)B * cdiv(Hi, UPCE_TL) * cdiv(Wi, UPCE_TL);
        if (ny * nx * 8 <= 48 * 1024 && tiles < (1L << 30)) {
            DISPATCH_T(dtype, "lavt_upsample_ce_bwd", hipLaunchKernelGGL(upsample_ce_bwd_tile_kernel<T>, dim3((unsigned)tiles), dim3(256), (size_t)(ny * nx * 8), ST, (const T*)x, target, w0, w1, out4, dloss,
                                                                          (T*)dx, B, Hi, Wi, Ho, Wo, sh, sw, (int)(ny * nx)));
            LAVT_CHECK_LAUNCH("lavt_upsample_ce_bwd");
            return LAVT_OK;
        }
    }
    const int blocks = (int)((n + 3) / 4 > 8192 ? 8192 : (n + 3) / 4);          // a wave per low-resolution pixel
    DISPATCH_T(dtype, "lavt_upsample_ce_bwd", hipLaunchKernelGGL(upsample_ce_bwd_kernel<T>, dim3(blocks), dim3(256), 0, ST, (const T*)x, target, w0, w1, out4, dloss, (T*)dx, B, Hi, Wi, Ho, Wo, bl_scale(Hi, Ho), bl_scale(Wi, Wo)));
    LAVT_CHECK_LAUNCH("lavt_upsample_ce_bwd");
    return LAVT_OK;
}
extern "C" int lavt_cls_head_fwd(int dtype, const void* x, const float* w, const float* b, void* y, int64_t rows, int C, void* stream) {
    LAVT_CHECK_ARG(x && w && b && y && rows > 0 && C % EPC_OF(dtype) == 0, "lavt_cls_head_fwd: bad arguments");
    int blocks = cdiv(rows, 4 * 4);
    if (blocks > 2048) blocks = 2048;
    DISPATCH_T(dtype, "lavt_cls_head_fwd", hipLaunchKernelGGL(cls_head_fwd_kernel<T>, dim3(blocks), dim3(256), 0, ST, (const T*)x, w, b, (T*)y, rows, C));
    LAVT_CHECK_LAUNCH("lavt_cls_head_fwd");
    return LAVT_OK;
}
static int cls_head_bwd_blocks(int dtype, int64_t rows, int C) {
    const int cpr = C / EPC_OF(dtype), rstep = 256 / (cpr > 0 ? cpr : 1);
    int blocks = (int)cdiv(rows, (long)(rstep > 0 ? rstep : 1) * 16);
    if (blocks > 512) blocks = 512;       // each workgroup ends with 2 C + 2 atomics / one partial record (after an LDS combine of its row-lanes)
    return blocks < 1 ? 1 : blocks;
}
extern "C" int lavt_cls_head_bwd_blocks(int dtype, int64_t rows, int C) { return cls_head_bwd_blocks(dtype, rows, C); }
extern "C" int lavt_cls_head_bwd(int dtype, const void* x, const void* dy, const float* w, void* dx, float* dw, float* db,
                                 int64_t rows, int C, void* stream) {
    const int cpr = C / EPC_OF(dtype);
    LAVT_CHECK_ARG(x && dy && w && dx && dw && db && rows > 0 && C % EPC_OF(dtype) == 0 && cpr <= 256 && C <= 2048, "lavt_cls_head_bwd: bad arguments");
    const int blocks = cls_head_bwd_blocks(dtype, rows, C);
    DISPATCH_T(dtype, "lavt_cls_head_bwd", hipLaunchKernelGGL((cls_head_bwd_kernel<T, false>), dim3(blocks), dim3(256), 0, ST, (const T*)x, (const T*)dy, w, (T*)dx, dw, db, rows, C));
    LAVT_CHECK_LAUNCH("lavt_cls_head_bwd");
    return LAVT_OK;
}
extern "C" int lavt_cls_head_bwd_partial(int dtype, const void* x, const void* dy, const float* w, void* dx, float* pw, float* pb,
                                         int64_t rows, int C, void* stream) {
    const int cpr = C / EPC_OF(dtype);
    LAVT_CHECK_ARG(x && dy && w && dx && pw && pb && rows > 0 && C % EPC_OF(dtype) == 0 && cpr <= 256 && C <= 2048, "lavt_cls_head_bwd_partial: bad arguments");
    const int blocks = cls_head_bwd_blocks(dtype, rows, C);
    DISPATCH_T(dtype, "lavt_cls_head_bwd_partial", hipLaunchKernelGGL((cls_head_bwd_kernel<T, true>), dim3(blocks), dim3(256), 0, ST, (const T*)x, (const T*)dy, w, (T*)dx, pw, pb, rows, C));
    LAVT_CHECK_LAUNCH("lavt_cls_head_bwd_partial");
    return LAVT_OK;
}
extern "C" int lavt_im2col4(int dtype, const float* img, void* cols, int B, int H, int W, void* stream) {
    LAVT_CHECK_ARG(img && cols && B > 0 && H > 0 && W > 0, "lavt_im2col4: bad arguments");
    const int H4 = (H + 3) / 4, W4 = (W + 3) / 4;
    const int64_t n = (int64_t)B * H4 * W4 * 12;
    DISPATCH_T(dtype, "lavt_im2col4", hipLaunchKernelGGL(im2col4_kernel<T>, dim3(ew_grid(n)), dim3(256), 0, ST, img, (T*)cols, B, H, W, H4, W4));
    LAVT_CHECK_LAUNCH("lavt_im2col4");
    return LAVT_OK;
}
extern "C" int lavt_col2im4(int dtype, const void* dcols, float* dimg, int B, int H, int W, void* stream) {
    LAVT_CHECK_ARG(dcols && dimg && B > 0 && H > 0 && W > 0, "lavt_col2im4: bad arguments");
    const int H4 = (H + 3) / 4, W4 = (W + 3) / 4;
    const int64_t n = (int64_t)B * 3 * H * W;
    DISPATCH_T(dtype, "lavt_col2im4", hipLaunchKernelGGL(col2im4_kernel<T>, dim3(ew_grid(n)), dim3(256), 0, ST, (const T*)dcols, dimg, B, H, W, H4, W4));
    LAVT_CHECK_LAUNCH("lavt_col2im4");
    return LAVT_OK;
}

template <typename S> static int cast_from(const void* src, int dst_dtype, void* dst, int64_t n, hipStream_t st) {
    if (dst_dtype == LAVT_F32) hipLaunchKernelGGL((cast_kernel<S, float>), dim3(ew_grid(n)), dim3(256), 0, st, (const S*)src, (float*)dst, n);
    else hipLaunchKernelGGL((cast_kernel<S, bf16>), dim3(ew_grid(n)), dim3(256), 0, st, (const S*)src, (bf16*)dst, n);
    return 0;
}
extern "C" int lavt_cast(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t n, void* stream) {
    LAVT_CHECK_ARG(src && dst && n > 0 && (src_dtype | 1) == 1 && (dst_dtype | 1) == 1, "lavt_cast: bad arguments");
    if (src_dtype == LAVT_F32) cast_from<float>(src, dst_dtype, dst, n, ST); else cast_from<bf16>(src, dst_dtype, dst, n, ST);
    LAVT_CHECK_LAUNCH("lavt_cast");
    return LAVT_OK;
}
template <typename S> static void transpose_from(const void* src, int dst_dtype, void* dst, int batch, int R, int Cc, hipStream_t st) {
    dim3 grid(cdiv(Cc, 32), cdiv(R, 32), batch), block(32, 8);
    if (dst_dtype == LAVT_F32) hipLaunchKernelGGL((transpose_kernel<S, float>), grid, block, 0, st, (const S*)src, (float*)dst, R, Cc);
    else hipLaunchKernelGGL((transpose_kernel<S, bf16>), grid, block, 0, st, (const S*)src, (bf16*)dst, R, Cc);
}
extern "C" int lavt_nchw_to_nhwc(int src_dtype, const void* src, int dst_dtype, void* dst, int B, int C, int HW, void* stream) {
    LAVT_CHECK_ARG(src && dst && B > 0 && C > 0 && HW > 0 && (src_dtype | 1) == 1 && (dst_dtype | 1) == 1, "lavt_nchw_to_nhwc: bad arguments");
    if (src_dtype == LAVT_F32) transpose_from<float>(src, dst_dtype, dst, B, C, HW, ST); else transpose_from<bf16>(src, dst_dtype, dst, B, C, HW, ST);
    LAVT_CHECK_LAUNCH("lavt_nchw_to_nhwc");
    return LAVT_OK;
}
extern "C" int lavt_nhwc_to_nchw(int src_dtype, const void* src, int dst_dtype, void* dst, int B, int C, int HW, void* stream) {
    LAVT_CHECK_ARG(src && dst && B > 0 && C > 0 && HW > 0 && (src_dtype | 1) == 1 && (dst_dtype | 1) == 1, "lavt_nhwc_to_nchw: bad arguments");
    if (src_dtype == LAVT_F32) transpose_from<float>(src, dst_dtype, dst, B, HW, C, ST); else transpose_from<bf16>(src, dst_dtype, dst, B, HW, C, ST);
    LAVT_CHECK_LAUNCH("lavt_nhwc_to_nchw");
    return LAVT_OK;
}
extern "C" int lavt_pack_conv3x3(const float* w, int dtype, void* packed, int Cout, int Cin, int taps, void* stream) {
    LAVT_CHECK_ARG(w && packed && Cout > 0 && Cin > 0 && taps > 0, "lavt_pack_conv3x3: bad arguments");
    const int64_t n = (int64_t)Cout * Cin * taps;
    DISPATCH_T(dtype, "lavt_pack_conv3x3", hipLaunchKernelGGL(pack_conv3x3_kernel<T>, dim3(ew_grid(n)), dim3(256), 0, ST, w, (T*)packed, Cout, Cin, taps));
    LAVT_CHECK_LAUNCH("lavt_pack_conv3x3");
    return LAVT_OK;
}
extern "C" int lavt_unpack_conv_grad(const float* packed, float* dw, int Cout, int Cin, int taps, void* stream) {
    LAVT_CHECK_ARG(packed && dw && Cout > 0 && Cin > 0 && taps > 0, "lavt_unpack_conv_grad: bad arguments");
    const int64_t n = (int64_t)Cout * Cin * taps;
    const bool tiled = lavt_tuning().unpack_tiled;
    if (tiled && taps <= 32 && Cout <= 65535) hipLaunchKernelGGL(unpack_conv_grad_tiled_kernel, dim3((Cin + 127) / 128, Cout), dim3(256), (size_t)taps * 129 * 4, ST, packed, dw, Cin, taps);
    else hipLaunchKernelGGL(unpack_conv_grad_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, packed, dw, Cout, Cin, taps);
    LAVT_CHECK_LAUNCH("lavt_unpack_conv_grad");
    return LAVT_OK;
}
// second stage of a split-K lavt_gemm_nt (fp32 partial outputs [splits][M][N]): 8 columns per thread, 16-byte loads, bf16 / fp32 out
template <typename T> __global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ parts, int splits, int64_t MN, int64_t chunks, int N, T* __restrict__ out, int64_t ldc) {
    GRID_STRIDE(i, chunks) {
        const int64_t e = i * 8;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < splits; ++s) {
            const float4 v0 = *reinterpret_cast<const float4*>(parts + (int64_t)s * MN + e), v1 = *reinterpret_cast<const float4*>(parts + (int64_t)s * MN + e + 4);
            a[0] += v0.x; a[1] += v0.y; a[2] += v0.z; a[3] += v0.w; a[4] += v1.x; a[5] += v1.y; a[6] += v1.z; a[7] += v1.w;
        }
        const int64_t m = e / N;
        const int n = (int)(e - m * N);
        if constexpr (std::is_same<T, float>::value) {
            *reinterpret_cast<float4*>(out + m * ldc + n) = make_float4(a[0], a[1], a[2], a[3]);
            *reinterpret_cast<float4*>(out + m * ldc + n + 4) = make_float4(a[4], a[5], a[6], a[7]);
        } else *reinterpret_cast<uint4*>(out + m * ldc + n) = f_to_chunk<bf16>(a);
    }
}
extern "C" int lavt_splitk_reduce(int dtype, const float* parts, int splits, int64_t M, int N, void* out, int64_t ldc, void* stream) {
    LAVT_CHECK_ARG(parts && out && splits > 0 && M > 0 && N > 0 && N % 8 == 0 && ldc % 8 == 0, "lavt_splitk_reduce: bad arguments (N, ldc multiples of 8)");
    const int64_t chunks = M * N / 8;
    if (dtype == LAVT_BF16) hipLaunchKernelGGL(splitk_reduce_kernel<bf16>, dim3(ew_grid(chunks)), dim3(256), 0, ST, parts, splits, M * N, chunks, N, (bf16*)out, ldc);
    else if (dtype == LAVT_F32) hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(ew_grid(chunks)), dim3(256), 0, ST, parts, splits, M * N, chunks, N, (float*)out, ldc);
    else { lavt_set_error("lavt_splitk_reduce: bad dtype %d", dtype); return LAVT_ERR_INVALID; }
    LAVT_CHECK_LAUNCH("lavt_splitk_reduce");
    return LAVT_OK;
}
extern "C" int lavt_cast_multi(const int64_t* desc, int count, int dst_dtype, void* stream) {
    LAVT_CHECK_ARG(desc && count > 0 && (dst_dtype | 1) == 1, "lavt_cast_multi: bad arguments");
    dim3 grid(64, count);
    if (dst_dtype == LAVT_F32) hipLaunchKernelGGL(cast_multi_kernel<float>, grid, dim3(256), 0, ST, desc, count);
    else hipLaunchKernelGGL(cast_multi_kernel<bf16>, grid, dim3(256), 0, ST, desc, count);
    LAVT_CHECK_LAUNCH("lavt_cast_multi");
    return LAVT_OK;
}

extern "C" int lavt_adamw_step(const int64_t* desc, const float* hyper, int count, float* step, float total_steps, float power, void* stream) {
    LAVT_CHECK_ARG(desc && hyper && step && count > 0, "lavt_adamw_step: bad arguments");
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(256, count), dim3(256), 0, ST, desc, hyper, count, step, total_steps, power);      // small tensors: surplus workgroups exit at once
    hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(1), 0, ST, step);
    LAVT_CHECK_LAUNCH("lavt_adamw_step");
    return LAVT_OK;
}

extern "C" int lavt_adamw_chunk_elems(void) { return ADAMW_CE; }
extern "C" int lavt_adamw_step_chunks(const int64_t* desc, const float* hyper, const int32_t* chunks, int nchunks, float* step, float total_steps, float power, void* stream) {
    LAVT_CHECK_ARG(desc && hyper && chunks && step && nchunks > 0, "lavt_adamw_step_chunks: bad arguments");
    hipLaunchKernelGGL(adamw_chunks_kernel, dim3(nchunks), dim3(256), 0, ST, desc, hyper, reinterpret_cast<const int2*>(chunks), step, total_steps, power);
    hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(1), 0, ST, step);
    LAVT_CHECK_LAUNCH("lavt_adamw_step_chunks");
    return LAVT_OK;
}

// ---- fused bilinear upsample + MultiClassDiceLoss (losses.py:38-77 of the reference: the `--loss mc_dice` criterion of the released lavt_one) ----
// per sample b and class c:  I_bc = sum_pix p_c [t == c],  C_bc = sum_pix (p_c^2 + [t == c]),  loss = mean_{b,c} (1 - 2 I_bc / (C_bc + 1e-6)).
// stats = {loss, 0, then per sample {I0, I1, Q0 = sum p0^2, Q1 = sum p1^2, N0 = #[t == 0], N1 = #[t == 1]}}; like the cross-entropy pair above the
// (B, 2, H, W) logits are never written.
template <typename T>
__global__ __launch_bounds__(256) void upsample_dice_fwd_kernel(const T* __restrict__ x, const int64_t* __restrict__ target, float* __restrict__ partial,
                                                                int Hi, int Wi, int Ho, int Wo, float sh, float sw) {
    const int b = blockIdx.y;
    const int64_t n = (int64_t)Ho * Wo;
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int xo = (int)(i % Wo), yo = (int)(i / Wo);
        int y0, y1, x0, x1; float ly, lx;
        bl_coord(yo, sh, Hi, y0, y1, ly);
        bl_coord(xo, sw, Wi, x0, x1, lx);
        const UpCe u = upce_at<T>(x + (int64_t)b * Hi * Wi * 2, Wi, y0, y1, ly, x0, x1, lx);
        const float p0 = expf(u.up0 - u.lse), p1 = expf(u.up1 - u.lse);
        const int64_t t = target[(int64_t)b * n + i];
        if (t == 0) { s[0] += p0; s[4] += 1.f; }
        if (t == 1) { s[1] += p1; s[5] += 1.f; }
        s[2] += p0 * p0; s[3] += p1 * p1;
    }
    __shared__ float red[4][6];
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] = wave_sum(s[k]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 6; ++k) red[threadIdx.x >> 6][k] = s[k];
    __syncthreads();
    if (threadIdx.x < 6) partial[((int64_t)b * gridDim.x + blockIdx.x) * 6 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void upsample_dice_finish_kernel(const float* __restrict__ partial, int nblk, int B, float* __restrict__ stats) {
    __shared__ float red[4][6];
    __shared__ float loss_acc;
    if (threadIdx.x == 0) loss_acc = 0.f;
    for (int b = 0; b < B; ++b) {
        float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int k = threadIdx.x; k < nblk; k += 256)
#pragma unroll
            for (int j = 0; j < 6; ++j) a[j] += partial[((int64_t)b * nblk + k) * 6 + j];
#pragma unroll
        for (int j = 0; j < 6; ++j) a[j] = wave_sum(a[j]);
        __syncthreads();
        if ((threadIdx.x & 63) == 0)
#pragma unroll
            for (int j = 0; j < 6; ++j) red[threadIdx.x >> 6][j] = a[j];
        __syncthreads();
        if (threadIdx.x == 0) {
            float v[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) { v[j] = red[0][j] + red[1][j] + red[2][j] + red[3][j]; stats[2 + b * 6 + j] = v[j]; }
            loss_acc += (1.f - 2.f * v[0] / (v[2] + v[4] + 1e-6f)) + (1.f - 2.f * v[1] / (v[3] + v[5] + 1e-6f));
        }
    }
    if (threadIdx.x == 0) { stats[0] = loss_acc / (2.f * (float)B); stats[1] = 0.f; }
}
// dx[b, yi, xi, c]: with a_bc = -1 / (B (C_bc + eps)) and e_bc = I_bc / (B (C_bc + eps)^2),  d loss / d p_c(pix) = a_bc [t == c] + 2 p_c e_bc,
// through the 2-class softmax (dz1 = p0 p1 (g1 - g0) = -dz0) and the bilinear weights, gathered per low-resolution pixel (no atomics)
template <typename T>
__global__ __launch_bounds__(256) void upsample_dice_bwd_kernel(const T* __restrict__ x, const int64_t* __restrict__ target, const float* __restrict__ stats,
                                                                const float* __restrict__ dloss, T* __restrict__ dx, int B, int Hi, int Wi, int Ho, int Wo,
                                                                float sh, float sw) {
    const int64_t n = (int64_t)B * Hi * Wi;
    const float g = dloss ? dloss[0] : 1.f;
    GRID_STRIDE(i, n) {
        const int xi = (int)(i % Wi), yi = (int)((i / Wi) % Hi), b = (int)(i / Wi / Hi);
        const float* sb = stats + 2 + b * 6;
        const float c0 = sb[2] + sb[4] + 1e-6f, c1 = sb[3] + sb[5] + 1e-6f, invB = 0.5f / (float)B;      // mean over (b, c): 1 / (2B)
        const float A0 = -2.f * invB / c0, A1 = -2.f * invB / c1, E0 = 2.f * invB * sb[0] / (c0 * c0), E1 = 2.f * invB * sb[1] / (c1 * c1);
        int ylo, yhi, xlo, xhi;
        bl_range(yi, sh, Hi, Ho, ylo, yhi);
        bl_range(xi, sw, Wi, Wo, xlo, xhi);
        const T* base = x + (int64_t)b * Hi * Wi * 2;
        float acc = 0.f;
        for (int yo = ylo; yo <= yhi; ++yo) {
            int y0, y1; float ly;
            bl_coord(yo, sh, Hi, y0, y1, ly);
            const float wy = (y0 == yi ? 1.f - ly : 0.f) + (y1 == yi ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int xo = xlo; xo <= xhi; ++xo) {
                int x0, x1; float lx;
                bl_coord(xo, sw, Wi, x0, x1, lx);
                const float wx = (x0 == xi ? 1.f - lx : 0.f) + (x1 == xi ? lx : 0.f);
                if (wx == 0.f) continue;
                const int64_t t = target[((int64_t)b * Ho + yo) * Wo + xo];
                const UpCe u = upce_at<T>(base, Wi, y0, y1, ly, x0, x1, lx);
                const float p0 = expf(u.up0 - u.lse), p1 = expf(u.up1 - u.lse);
                const float g0 = (t == 0 ? A0 : 0.f) + 2.f * p0 * E0, g1 = (t == 1 ? A1 : 0.f) + 2.f * p1 * E1;
                acc += wy * wx * p0 * p1 * (g1 - g0);
            }
        }
        dx[i * 2] = from_f<T>(-acc * g);
        dx[i * 2 + 1] = from_f<T>(acc * g);
    }
}
extern "C" int lavt_upsample_dice_fwd(int dtype, const void* x, const int64_t* target, float* ws, int64_t ws_floats, float* stats,
                                      int B, int Hi, int Wi, int Ho, int Wo, void* stream) {
    LAVT_CHECK_ARG(x && target && ws && stats && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "lavt_upsample_dice_fwd: bad arguments");
    const int64_t n = (int64_t)Ho * Wo;
    int blocks = (int)((n + 1023) / 1024);
    if (blocks > 256) blocks = 256;
    LAVT_CHECK_ARG(ws_floats >= (int64_t)blocks * 6 * B, "lavt_upsample_dice_fwd: scratch of %d floats needed", blocks * 6 * B);
    DISPATCH_T(dtype, "lavt_upsample_dice_fwd", hipLaunchKernelGGL(upsample_dice_fwd_kernel<T>, dim3(blocks, B), dim3(256), 0, ST, (const T*)x, target, ws, Hi, Wi, Ho, Wo, bl_scale(Hi, Ho), bl_scale(Wi, Wo)));
    hipLaunchKernelGGL(upsample_dice_finish_kernel, dim3(1), dim3(256), 0, ST, ws, blocks, B, stats);
    LAVT_CHECK_LAUNCH("lavt_upsample_dice_fwd");
    return LAVT_OK;
}
extern "C" int lavt_upsample_dice_bwd(int dtype, const void* x, const int64_t* target, const float* stats, const float* dloss, void* dx,
                                      int B, int Hi, int Wi, int Ho, int Wo, void* stream) {
    LAVT_CHECK_ARG(x && target && stats && dx && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "lavt_upsample_dice_bwd: bad arguments");
    const int64_t n = (int64_t)B * Hi * Wi;
    DISPATCH_T(dtype, "lavt_upsample_dice_bwd", hipLaunchKernelGGL(upsample_dice_bwd_kernel<T>, dim3(ew_grid(n)), dim3(256), 0, ST, (const T*)x, target, stats, dloss, (T*)dx, B, Hi, Wi, Ho, Wo, bl_scale(Hi, Ho), bl_scale(Wi, Wo)));
    LAVT_CHECK_LAUNCH("lavt_upsample_dice_bwd");
    return LAVT_OK;
}
