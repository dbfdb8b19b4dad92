// 3x3 convolution weight gradient with the nine taps fused (bf16, round 4).
//
//   dW[co][ci][tap] = sum over pixels p of dY[p][co] * X[p + (dy, dx)][ci],   tap = (dy + 1) * 3 + (dx + 1), zero outside the image
//
// (SimpleDecoding's conv1_4 .. conv2_2, lib/mask_predictor.py:60-97; the gradient torch.autograd computes for nn.Conv2d(3x3, pad 1, no bias).)
//
// As a plain TN GEMM over K = pixels with J = 9 * Cin tap-shifted columns (gemm_tn_v2.hip) every (128 x 128) output tile streams its own
// copy of both operands: 144 tiles x 28 800 pixels x 512 B = 2.1 GB of L2 -> LDS fill per launch for the 512 -> 512 convolution at 120 x 120,
// 9.5 TB/s over the launch's 223 us -- the L2 / MALL ceiling, at 0.24 of the MFMA peak (round-3 review).  The nine taps of one input-channel
// block read the SAME pixels of X shifted by one row / one column, so here a workgroup owns (128 output channels) x (64 input channels) x
// (all 9 taps) and walks the image one ROW per K tile:
//   * X rows live in a rolling window in LDS: four slots, each one image row with a zero column on either side (positions 0 and W + 1), the
//     rows above / below the image replaced by a slot of zeros.  A K tile loads ONE new row of X (W x 64 channels) and the row of dY
//     (W x 128 channels): 46 KB per 17.7 MFLOP at W = 120 instead of 32 KB per 2.1 MFLOP -- 6x fewer fill bytes per flop;
//   * the tap shift is an LDS ADDRESS: tap (dy, dx) reads slot (row + dy) at position x + dx + 1 with the transposing LDS read, whose lanes
//     each supply the address of one K row -- no masks, no per-lane coordinates, no zero page in the loop (the halo columns ARE zeros);
//   * the dY fragments of a 32-pixel k-step are read once and reused by all nine taps (36 MFMAs per k-step per wave on 4 A + 9 B fragment
//     reads); 36 accumulator tiles per wave (144 registers: 8 waves per workgroup, one workgroup per CU, 2 waves per SIMD);
//   * the reduction over pixels is cut into `pieces` runs of consecutive image rows so that tiles x pieces ~ 256 workgroups; a piece stores
//     its accumulator registers as they stand (lane-linear float4 records: 16-byte stores, 1 KB per wave-instruction) into the scratch the caller
//     lends and ONE small kernel adds the pieces, un-permutes them through LDS and accumulates contiguous runs into the [Cout][Cin][3][3]
//     gradient: no zero fill of a packed buffer, no atomics, no unpack launch.
// Both sources of a concat convolution (torch.cat([top-down, skip]): Cin = C1 + C2) are read in place: a 64-channel block lies in one source.
#include "gemm_v2_helpers.h"

namespace {

constexpr int CW_BI = 128, CW_BJ = 64, CW_WAVES = 8, CW_THREADS = CW_WAVES * 64;
constexpr int CW_NS = 4;                           // rolling window: rows y - 1, y, y + 1 in use + the row being loaded
constexpr int CW_POS = 130;                        // positions of a slot: 0 = left halo, 1 .. W = pixels, W + 1 .. = zeros
constexpr int CW_SLOT = CW_POS * CW_BJ * 2;        // 16 640 B
constexpr int CW_ASTAGE = 128 * CW_BI * 2;         // dY tile: [128 pixels][128 channels] bf16, k-major
constexpr int CW_LDS = 2 * CW_ASTAGE + (CW_NS + 1) * CW_SLOT;      // 148 736 B

struct CwArgs {
    const bf16* dy; int64_t ldy;
    const bf16* x1; int64_t ldx1;
    const bf16* x2; int64_t ldx2;
    int c1;                                        // channels of x1 (Cin - c1 come from x2)
    int B, H, W, Cout, Cin;
    float* parts;                                  // [pieces][tiles][8 waves][4 fragment rows][9 taps][64 lanes] float4 = pieces x Cout x 9 x Cin floats
    int pieces, rows_per_piece, xcd_order;
    const void* zeros;
};

// 32-byte slot swizzle of a 128-byte row by its POSITION (row bits 1 and 3: tn_swz<8>); any 4 consecutive positions in 4 blocks 8 apart --
// what one transposing read touches, whatever the tap shift -- spread over all bank groups
__device__ __forceinline__ int cw_bswz(int pos) { return (((pos >> 1) & 1) | ((pos >> 2) & 2)) << 1; }

// Fragment reads are ISSUED by one asm statement and WAITED FOR by another (cw_wait), with the MFMAs of the previous fragments in between: the
// LDS latency of item n + 1 hides under the 12 MFMAs of item n inside a wave (the first form waited after every group of reads: the matrix pipe
// was ~50 % busy in the K loop).  The destinations are early-clobber outputs of the issuing statement and in/out operands of the waiting one, so
// no consumer can be scheduled between them; sched_barrier keeps the MFMAs where they are written.
template <int OFF>
__device__ __forceinline__ void cw_issue_b3(unsigned a0, unsigned a1, unsigned a2, unsigned a3, unsigned a4, unsigned a5, u64 (&lo)[3], u64 (&hi)[3]) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %6 offset:%c12\n\tds_read_b64_tr_b16 %1, %7 offset:%c12\n\t"
        "ds_read_b64_tr_b16 %2, %8 offset:%c12\n\tds_read_b64_tr_b16 %3, %9 offset:%c12\n\t"
        "ds_read_b64_tr_b16 %4, %10 offset:%c12\n\tds_read_b64_tr_b16 %5, %11 offset:%c12"
        : "=&v"(lo[0]), "=&v"(hi[0]), "=&v"(lo[1]), "=&v"(hi[1]), "=&v"(lo[2]), "=&v"(hi[2])
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "n"(OFF)
        : "memory");
}
template <int HO, int KOFF>
__device__ __forceinline__ void cw_issue_a4(const unsigned (&a)[4], u64 (&l)[4], u64 (&h)[4]) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8 offset:%c13\n\tds_read_b64_tr_b16 %1, %8 offset:%c13+%c12\n\t"
        "ds_read_b64_tr_b16 %2, %9 offset:%c13\n\tds_read_b64_tr_b16 %3, %9 offset:%c13+%c12\n\t"
        "ds_read_b64_tr_b16 %4, %10 offset:%c13\n\tds_read_b64_tr_b16 %5, %10 offset:%c13+%c12\n\t"
        "ds_read_b64_tr_b16 %6, %11 offset:%c13\n\tds_read_b64_tr_b16 %7, %11 offset:%c13+%c12"
        : "=&v"(l[0]), "=&v"(h[0]), "=&v"(l[1]), "=&v"(h[1]), "=&v"(l[2]), "=&v"(h[2]), "=&v"(l[3]), "=&v"(h[3])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "n"(HO), "n"(KOFF)
        : "memory");
}
__device__ __forceinline__ void cw_wait_b3(u64 (&lo)[3], u64 (&hi)[3]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2])::"memory");
}
__device__ __forceinline__ void cw_wait_a4(u64 (&l)[4], u64 (&h)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(l[0]), "+v"(h[0]), "+v"(l[1]), "+v"(h[1]), "+v"(l[2]), "+v"(h[2]), "+v"(l[3]), "+v"(h[3])::"memory");
}

template <int KS>          // k-steps of 32 pixels per image row: W <= 32 KS
__global__ __launch_bounds__(CW_THREADS) void conv_wgrad3x3_kernel(const CwArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int A_INSTR = KS;                                    // 32 KS pixels x 16 chunks / 512 threads
    constexpr int B_INSTR = (KS * 32 * 8 + CW_THREADS - 1) / CW_THREADS;
    char* const sA = smem;
    char* const sB = smem + 2 * CW_ASTAGE;
    char* const sZ = sB + CW_NS * CW_SLOT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 2, wj = wave & 3;                       // 2 x 4 waves: 64 output channels x 16 input channels (x 9 taps) each
    const int tiles_j = (a.Cin + CW_BJ - 1) / CW_BJ;          // Cin % 8 == 0; a last tile beyond Cin reads zeros and its columns are never reduced
    // Workgroups go round-robin over the 8 XCDs (private L2 each); with gridDim.x a multiple of 8 a tile keeps its XCD in every piece.  Give an XCD a
    // contiguous run of tiles (tile_j fastest): its workgroups of one piece then share the dY panel of (mostly) one tile_i -- measured before: every
    // dY panel crossed the fabric once per tile_j (8 x 29.5 MB of the 343 MB per launch).
    const int tile_lin = ((gridDim.x & 7) == 0 && a.xcd_order) ? xcd_tile_id(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int tile_i = tile_lin / tiles_j, tile_j = tile_lin - tile_i * tiles_j;
    const int i0 = tile_i * CW_BI, j0 = tile_j * CW_BJ;
    const int W = a.W, H = a.H, rows = a.B * a.H;
    const int g0 = blockIdx.y * a.rows_per_piece, g1 = min(rows, g0 + a.rows_per_piece);
    if (g0 >= g1) return;
    const bool second = j0 >= a.c1;
    const bf16* X = second ? a.x2 + (j0 - a.c1) : a.x1 + j0;
    const int64_t ldx = second ? a.ldx2 : a.ldx1;
    const bf16* Z = reinterpret_cast<const bf16*>(a.zeros);

    // zero the window once: halo positions and the zero slot are never written again
    for (int e = tid; e < (CW_NS + 1) * CW_SLOT / 16; e += CW_THREADS) reinterpret_cast<uint4*>(sB)[e] = make_uint4(0, 0, 0, 0);

    // ---- DMA geometry, constant over rows: LDS chunk q = (wave * INSTR + i) * 64 + lane of a tile / slot ----
    const bf16* a_src[A_INSTR];
    bool a_ok[A_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int q = (wave * A_INSTR + i) * 64 + lane, kr = q >> 4, cc = (q & 15) ^ tn_swz<16>(kr);
        a_ok[i] = kr < W;
        a_src[i] = a.dy + (int64_t)kr * a.ldy + i0 + cc * 8;
    }
    const bf16* b_src[B_INSTR];
    bool b_ok[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int q = (wave * B_INSTR + i) * 64 + lane, pos = 1 + (q >> 3), cc = (q & 7) ^ cw_bswz(pos);
        b_ok[i] = pos - 1 < W && j0 + cc * 8 < a.Cin;
        b_src[i] = X + (int64_t)(pos - 1) * ldx + cc * 8;
    }
    const int64_t a_row = (int64_t)W * a.ldy, b_row = (int64_t)W * ldx;
    auto issue_a = [&](int g, int stage) {                         // dY row g -> stage
        const bool ok = g < g1;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) dma16((ok && a_ok[i]) ? a_src[i] + g * a_row : Z, sA + stage * CW_ASTAGE + (wave * A_INSTR + i) * 1024);
    };
    auto issue_b = [&](int g) {                                    // X row g -> slot g % 4 (positions 1 ..)
        const bool ok = g >= 0 && g < rows;
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) dma16((ok && b_ok[i]) ? b_src[i] + g * b_row : Z, sB + (g & (CW_NS - 1)) * CW_SLOT + CW_BJ * 2 + (wave * B_INSTR + i) * 1024);
    };

    // ---- fragment read addresses ----
    const int row_off = 8 * (lane >> 4) + ((lane & 15) >> 2);
    unsigned relA[4];
    {
        const int sAw = tn_swz<16>(row_off);
#pragma unroll
        for (int i = 0; i < 4; ++i) relA[i] = (unsigned)(row_off * CW_BI + ((((wi * 64) / 8 + 2 * i) ^ sAw) + ((lane & 3) >> 1)) * 8 + (lane & 1) * 4) * 2;
    }
    // B: k row `row_off` (+ 4 for the second read) of the k-step sits at position row_off + dx + 1 (+ 4); byte offset inside a slot
    unsigned posoff[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pos = row_off + 4 * h + d;                   // d = dx + 1
            posoff[d][h] = (unsigned)(pos * (CW_BJ * 2) + (((wj * 2) ^ cw_bswz(pos)) + ((lane & 3) >> 1)) * 16 + (lane & 1) * 8);
        }

    f32x4 acc[4][9];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();                                               // the zero fill is complete before any DMA lands
    issue_b(g0 - 1);
    issue_b(g0);
    issue_a(g0, 0);
    issue_b(g0 + 1);
    const unsigned ldsA = lds_addr(sA), ldsB = lds_addr(sB), ldsZ = lds_addr(sZ);
    int y = g0 % H;
    for (int g = g0; g < g1; ++g) {
        const int stage = (g - g0) & 1;
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (g + 1 < g1) {                                          // next row's operands fly under this row's MFMAs
            issue_a(g + 1, stage ^ 1);
            issue_b(g + 2);
        }
        unsigned sb[3];
        sb[0] = y > 0 ? ldsB + ((g - 1) & (CW_NS - 1)) * CW_SLOT : ldsZ;
        sb[1] = ldsB + (g & (CW_NS - 1)) * CW_SLOT;
        sb[2] = y + 1 < H ? ldsB + ((g + 1) & (CW_NS - 1)) * CW_SLOT : ldsZ;
        unsigned aA[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) aA[i] = ldsA + stage * CW_ASTAGE + relA[i];
        // items n = 3 ks + d (d = dy + 1): B fragments of item n + 1 and -- at the last item of a k-step -- the dY fragments of the next k-step are
        // in flight while item n's 12 MFMAs issue
        u64 al[2][4], ah[2][4], bl[2][3], bh[2][3];
        auto issue_b = [&](auto n_tag) {
            constexpr int n = decltype(n_tag)::value, ks = n / 3, d = n % 3;
            cw_issue_b3<ks * 32 * CW_BJ * 2>(sb[d] + posoff[0][0], sb[d] + posoff[0][1], sb[d] + posoff[1][0], sb[d] + posoff[1][1],
                                              sb[d] + posoff[2][0], sb[d] + posoff[2][1], bl[n & 1], bh[n & 1]);
        };
        auto issue_afr = [&](auto ks_tag) {
            constexpr int ks = decltype(ks_tag)::value;
            cw_issue_a4<4 * CW_BI * 2, ks * 32 * CW_BI * 2>(aA, al[ks & 1], ah[ks & 1]);
        };
        issue_afr(std::integral_constant<int, 0>{});
        issue_b(std::integral_constant<int, 0>{});
        cw_wait_a4(al[0], ah[0]);
        cw_wait_b3(bl[0], bh[0]);
        auto item = [&](auto n_tag) {
            constexpr int n = decltype(n_tag)::value, ks = n / 3, d = n % 3;
            if constexpr (n + 1 < 3 * KS) issue_b(std::integral_constant<int, n + 1>{});
            if constexpr (d == 2 && ks + 1 < KS) issue_afr(std::integral_constant<int, ks + 1>{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const bf16x8 fb = frag_from(bl[n & 1][e], bh[n & 1][e]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][d * 3 + e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_from(al[ks & 1][i], ah[ks & 1][i]), fb, acc[i][d * 3 + e], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (n + 1 < 3 * KS) cw_wait_b3(bl[(n + 1) & 1], bh[(n + 1) & 1]);
            if constexpr (d == 2 && ks + 1 < KS) cw_wait_a4(al[(ks + 1) & 1], ah[(ks + 1) & 1]);
        };
        item(std::integral_constant<int, 0>{}); item(std::integral_constant<int, 1>{}); item(std::integral_constant<int, 2>{});
        if constexpr (KS > 1) { item(std::integral_constant<int, 3>{}); item(std::integral_constant<int, 4>{}); item(std::integral_constant<int, 5>{}); }
        if constexpr (KS > 2) { item(std::integral_constant<int, 6>{}); item(std::integral_constant<int, 7>{}); item(std::integral_constant<int, 8>{}); }
        if constexpr (KS > 3) { item(std::integral_constant<int, 9>{}); item(std::integral_constant<int, 10>{}); item(std::integral_constant<int, 11>{}); }
        y = y + 1 == H ? 0 : y + 1;
    }

    // ---- this piece's tile -> parts[piece][tile][wave][i][tap][lane] as float4 (the accumulator registers as they stand: one 16-byte store per
    // lane per fragment, 1 KB contiguous per wave-instruction; conv_wgrad3x3_reduce knows the layout) ----
    float4* out = reinterpret_cast<float4*>(a.parts) + ((((int64_t)blockIdx.y * gridDim.x + tile_lin) * CW_WAVES + wave) * 36) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) out[(i * 9 + t) * 64] = make_float4(acc[i][t][0], acc[i][t][1], acc[i][t][2], acc[i][t][3]);
}

// dW[co][ci][tap] += sum over pieces of the partial tiles.  Block = (tile, wave, fragment row i): 16 output channels x 16 input channels x 9 taps
// (1024 blocks for the 512 -> 512 convolution: the 75 MB of partial tiles are read by the whole chip).  Reads: the pieces' float4 records,
// lane-linear (1 KB per wave-load, four pieces in flight per thread); the sums are scattered into an LDS image of the 16 x 144 output block and
// leave as 16 contiguous runs of 144 floats.
__global__ __launch_bounds__(256) void conv_wgrad3x3_reduce(const float4* __restrict__ parts, int pieces, int tiles, int tiles_j, int Cin, float* __restrict__ dW, int accumulate,
                                                            const float* __restrict__ amax_a, const float* __restrict__ amax_b) {
    __shared__ float t[16][145];
    // e4m3 operands (conv_wgrad3x3_f8_kernel): the partial tiles are sums of products of the QUANTISED values q = x * 448 / |max|
    const float deq = amax_a ? (*amax_a > 0.f ? *amax_a * (1.f / 448.f) : 1.f) * (*amax_b > 0.f ? *amax_b * (1.f / 448.f) : 1.f) : 1.f;
    const int tile = blockIdx.x, wave = blockIdx.y >> 2, i = blockIdx.y & 3, wi = wave >> 2, wj = wave & 3;
    const int tile_i = tile / tiles_j, tile_j = tile - tile_i * tiles_j;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t pstride = (int64_t)tiles * CW_WAVES * 36 * 64;
    for (int tap = grp; tap < 9; tap += 4) {                           // unit = tap: one float4 per lane per piece
        const float4* q = parts + (((int64_t)tile * CW_WAVES + wave) * 36 + (i * 9 + tap)) * 64 + lane;
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
        int p = 0;
        for (; p + 3 < pieces; p += 4) {
            const float4 v0 = q[(int64_t)p * pstride], v1 = q[(int64_t)(p + 1) * pstride], v2 = q[(int64_t)(p + 2) * pstride], v3 = q[(int64_t)(p + 3) * pstride];
            s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
            s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
            s2.x += v2.x; s2.y += v2.y; s2.z += v2.z; s2.w += v2.w;
            s3.x += v3.x; s3.y += v3.y; s3.z += v3.z; s3.w += v3.w;
        }
        for (; p < pieces; ++p) { const float4 v0 = q[(int64_t)p * pstride]; s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w; }
        const int col = (lane & 15) * 9 + tap, r0 = 4 * (lane >> 4);
        t[r0][col] = (s0.x + s1.x) + (s2.x + s3.x); t[r0 + 1][col] = (s0.y + s1.y) + (s2.y + s3.y);
        t[r0 + 2][col] = (s0.z + s1.z) + (s2.z + s3.z); t[r0 + 3][col] = (s0.w + s1.w) + (s2.w + s3.w);
    }
    __syncthreads();
    const int co0 = tile_i * CW_BI + wi * 64 + i * 16;
    const int ci0 = tile_j * CW_BJ + wj * 16, nvalid = min(16, Cin - ci0) * 9;          // (<= 0 for a 16-channel block beyond Cin)
    for (int e = threadIdx.x; e < 16 * 144; e += 256) {
        const int r = e / 144, c = e - r * 144;
        if (c < nvalid) {
            float* dst = dW + ((int64_t)(co0 + r) * Cin + ci0) * 9 + c;
            *dst = accumulate ? *dst + t[r][c] * deq : t[r][c] * deq;          // (a zeroed gradient buffer with one writer: plain stores save the read of dW -- 9-28 MB per launch)
        }
    }
}

// pieces so that tiles x pieces fills the 256 CUs in one round (one workgroup per CU: 145 KB of LDS), at least 8 image rows per piece
int cw_pieces(int B, int H, int Cout, int Cin) {
    const int tiles = (Cout / CW_BI) * ((Cin + CW_BJ - 1) / CW_BJ), rows = B * H;
    int p = 256 / tiles;
    if (p < 1) p = 1;
    const int maxp = rows / 8 > 0 ? rows / 8 : 1;
    return p > maxp ? maxp : p;
}
bool cw_supported(int B, int H, int W, int Cout, int Cin, int c1) {
    return B > 0 && H > 0 && W > 0 && W <= 128 && Cout % CW_BI == 0 && Cin % 8 == 0 && c1 % CW_BJ == 0 && c1 <= Cin && (int64_t)B * H * W < (1LL << 24);
}

// ================================================================================================ e4m3 operands (configs[4])
// The same decomposition on v_mfma_f32_16x16x128_f8f6f4: dY and X arrive as e4m3 bytes ([pixel][channel], the copies the forward convolution and the
// data gradient already contract), a K step is 128 pixels = RPK image rows of up to 128 / RPK pixels (RPK = 1: 64 < W <= 128, RPK = 2: 32 < W <= 64),
// and the pixel-major operands are read with ds_read_b64_tr_b8 -- per 16 lanes a block of 8 rows x 16 bytes, lane l supplying the address of row l / 2,
// half l % 2, lane c receiving column c of the 8 rows (measured on the box, round 5).  Every lane supplies its own row address, so the tap shift and the
// cut of a K step into image rows are address arithmetic as in the bf16 kernel.  A lane's 32 bytes of an operand are the pixels 32 g + 8 r + (0 .. 7)
// (g = lane / 16, r = read) for BOTH operands: which K index the MFMA gives them is immaterial.  Per K step and wave: 16 + 36 transposing reads for 36
// MFMAs of 32 cycles (bf16: 104 reads for 144 MFMAs of 16 cycles over the same pixels) and half the fill bytes.
typedef __attribute__((__vector_size__(8 * sizeof(int)))) int cw_i32x8;
typedef __attribute__((__vector_size__(4 * sizeof(u64)))) u64 cw_u64x4;

template <int RPK> struct Cw8 {
    static constexpr int PW = 128 / RPK;                       // positions of an image row inside a K step
    static constexpr int NS = RPK == 1 ? 4 : 8;                // ring of X rows: y - 1 .. y + RPK in use + RPK rows being loaded
    static constexpr int SLOT = (PW + 2) * 64;                 // one image row of 64 input channels, a zero position on either side
    static constexpr int ASTAGE = 128 * CW_BI;                 // dY tile: [128 pixels][128 channels] bytes
    static constexpr int LDS = 2 * ASTAGE + (NS + 1) * SLOT;   // 74 368 B / 70 784 B
    static constexpr int PRO = 1 + 2 / RPK;                    // row groups loaded ahead of the loop (rows g0 - 1 .. )
};

struct Cw8Args {
    const unsigned char* dy; int64_t ldy;
    const unsigned char* x1; int64_t ldx1;
    const unsigned char* x2; int64_t ldx2;
    int c1;
    int B, H, W, Cout, Cin;
    float* parts;
    int pieces, rows_per_piece, xcd_order;
    const void* zeros;
};

// 16-byte chunk swizzles.  dY tile (128-byte rows, two rows span the banks): the 16 rows of a half-wave's transposing read (8 consecutive rows of two
// blocks 32 apart) land in 16 different 16-byte bank groups; X slot (64-byte positions, four span the banks): any 8 consecutive positions of two
// blocks 32 apart likewise, whatever the tap shift.
__device__ __forceinline__ int cw8_aswz(int kr) { return ((kr >> 1) & 3) | (((kr >> 5) & 1) << 2); }
__device__ __forceinline__ int cw8_bswz(int pos) { return ((pos >> 2) & 1) | (((pos >> 5) & 1) << 1); }

__device__ __forceinline__ void cw8_issue4(unsigned a0, unsigned a1, unsigned a2, unsigned a3, u64 (&d)[4]) {
    asm volatile("ds_read_b64_tr_b8 %0, %4\n\tds_read_b64_tr_b8 %1, %5\n\tds_read_b64_tr_b8 %2, %6\n\tds_read_b64_tr_b8 %3, %7"
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
}
__device__ __forceinline__ void cw8_issue4_rows(unsigned a, u64 (&d)[4]) {          // the four 8-row blocks of one dY fragment: 1 KB apart
    asm volatile("ds_read_b64_tr_b8 %0, %4\n\tds_read_b64_tr_b8 %1, %4 offset:1024\n\tds_read_b64_tr_b8 %2, %4 offset:2048\n\tds_read_b64_tr_b8 %3, %4 offset:3072"
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]) : "v"(a) : "memory");
}
__device__ __forceinline__ void cw8_wait4(u64 (&d)[4]) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])::"memory"); }
__device__ __forceinline__ cw_i32x8 cw8_frag(const u64 (&d)[4]) {
    const cw_u64x4 v = {d[0], d[1], d[2], d[3]};
    return __builtin_bit_cast(cw_i32x8, v);
}

template <int RPK>
__global__ __launch_bounds__(CW_THREADS) void conv_wgrad3x3_f8_kernel(const Cw8Args a) {
    typedef Cw8<RPK> G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sA = smem;
    char* const sB = smem + 2 * G::ASTAGE;
    char* const sZ = sB + G::NS * G::SLOT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 2, wj = wave & 3;
    const int tiles_j = (a.Cin + CW_BJ - 1) / CW_BJ;
    const int tile_lin = ((gridDim.x & 7) == 0 && a.xcd_order) ? xcd_tile_id(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int tile_i = tile_lin / tiles_j, tile_j = tile_lin - tile_i * tiles_j;
    const int i0 = tile_i * CW_BI, j0 = tile_j * CW_BJ;
    const int W = a.W, H = a.H, rows = a.B * a.H;
    const int g0 = blockIdx.y * a.rows_per_piece, g1 = min(rows, g0 + a.rows_per_piece);          // (multiples of RPK: H % RPK == 0)
    if (g0 >= g1) return;
    const bool second = j0 >= a.c1;
    const unsigned char* X = second ? a.x2 + (j0 - a.c1) : a.x1 + j0;
    const int64_t ldx = second ? a.ldx2 : a.ldx1;
    const unsigned char* Z = reinterpret_cast<const unsigned char*>(a.zeros);

    for (int e = tid; e < (G::NS + 1) * G::SLOT / 16; e += CW_THREADS) reinterpret_cast<uint4*>(sB)[e] = make_uint4(0, 0, 0, 0);

    // ---- DMA geometry: LDS chunk q = (wave * INSTR + i) * 64 + lane of the dY tile (8 chunks per pixel) / of a row group of X (4 chunks per position) ----
    const unsigned char* a_src[2];
    bool a_ok[2];
    int a_sub[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = (wave * 2 + i) * 64 + lane, kr = q >> 3, cc = (q & 7) ^ cw8_aswz(kr), sub = kr / G::PW, x = kr % G::PW;
        a_ok[i] = x < W;
        a_sub[i] = sub;
        a_src[i] = a.dy + ((int64_t)sub * W + x) * a.ldy + i0 + cc * 16;
    }
    const int bq = wave * 64 + lane, b_sub = bq / (G::PW * 4), b_x = (bq >> 2) % G::PW, b_cc = (bq & 3) ^ cw8_bswz(b_x + 1);
    const bool b_ok = b_x < W && j0 + b_cc * 16 < a.Cin;
    const unsigned char* const b_src = X + ((int64_t)b_sub * W + b_x) * ldx + b_cc * 16;
    const int b_dst = CW_BJ + ((wave * 16) % G::PW) * CW_BJ;       // wave-uniform: the wave's 16 positions lie in one row of the group
    const int64_t a_row = (int64_t)W * a.ldy, b_row = (int64_t)W * ldx;
    auto issue_a = [&](int r0, int stage) {                        // dY rows r0 .. r0 + RPK - 1 -> stage
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16((a_ok[i] && r0 + a_sub[i] < g1) ? a_src[i] + r0 * a_row : Z, sA + stage * G::ASTAGE + (wave * 2 + i) * 1024);
    };
    auto issue_b = [&](int r0) {                                   // X rows r0 .. r0 + RPK - 1 -> their ring slots (positions 1 ..)
        const int r = r0 + b_sub;
        dma16((b_ok && r >= 0 && r < rows) ? b_src + r0 * b_row : Z, sB + (r & (G::NS - 1)) * G::SLOT + b_dst);
    };

    // ---- fragment read addresses: this lane's pixel rows of a K step are 32 g + 8 r + l / 2, r = 0 .. 3 ----
    const int g = lane >> 4, l = lane & 15, k_base = 32 * g + (l >> 1);
    unsigned relA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) relA[i] = (unsigned)(k_base * CW_BI + (((wi * 4 + i) ^ cw8_aswz(k_base)) * 16) + (l & 1) * 8);      // (the swizzle ignores r)
    const int f_sub = k_base / G::PW;                              // image row of the K step this lane's pixels lie in (8 r never crosses: PW % 32 == 0)
    unsigned posoff[3][4];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int pos = (k_base + 8 * r) % G::PW + d;          // d = dx + 1; pixel x sits at position x + 1
            posoff[d][r] = (unsigned)(pos * CW_BJ + ((wj ^ cw8_bswz(pos)) * 16) + (l & 1) * 8);
        }

    f32x4 acc[4][9];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();                                               // the zero fill is complete before any DMA lands
#pragma unroll
    for (int n = 0; n < G::PRO; ++n) issue_b(g0 - 1 + n * RPK);
    issue_a(g0, 0);
    const unsigned ldsA = lds_addr(sA), ldsB = lds_addr(sB), ldsZ = lds_addr(sZ);
    int y = g0 % H;
    int it = 0;
    for (int r0 = g0; r0 < g1; r0 += RPK, ++it) {
        const int stage = it & 1;
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (r0 + RPK < g1) {                                       // next K step's operands fly under this one's MFMAs
            issue_a(r0 + RPK, stage ^ 1);
            issue_b(g0 - 1 + (it + G::PRO) * RPK);
        }
        unsigned sb[3];
        {
            const int yy = y + f_sub, rr = r0 + f_sub;
            sb[0] = yy > 0 ? ldsB + ((rr - 1) & (G::NS - 1)) * G::SLOT : ldsZ;
            sb[1] = ldsB + (rr & (G::NS - 1)) * G::SLOT;
            sb[2] = yy + 1 < H ? ldsB + ((rr + 1) & (G::NS - 1)) * G::SLOT : ldsZ;
        }
        u64 af[4][4], bf[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) cw8_issue4_rows(ldsA + stage * G::ASTAGE + relA[i], af[i]);
        cw8_issue4(sb[0] + posoff[0][0], sb[0] + posoff[0][1], sb[0] + posoff[0][2], sb[0] + posoff[0][3], bf[0]);
#pragma unroll
        for (int i = 0; i < 4; ++i) cw8_wait4(af[i]);
        cw8_wait4(bf[0]);
        cw_i32x8 fa[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = cw8_frag(af[i]);
#pragma unroll
        for (int n = 0; n < 9; ++n) {                              // tap n = (dy + 1) * 3 + (dx + 1)
            if (n + 1 < 9) {
                const int d = (n + 1) / 3, e = (n + 1) % 3;
                cw8_issue4(sb[d] + posoff[e][0], sb[d] + posoff[e][1], sb[d] + posoff[e][2], sb[d] + posoff[e][3], bf[(n + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            const cw_i32x8 fb = cw8_frag(bf[n & 1]);
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+v"(acc[i][n]) : "v"(fa[i]), "v"(fb));
            __builtin_amdgcn_sched_barrier(0);
            if (n + 1 < 9) cw8_wait4(bf[(n + 1) & 1]);
        }
        y += RPK;
        if (y >= H) y = 0;
    }

    float4* out = reinterpret_cast<float4*>(a.parts) + ((((int64_t)blockIdx.y * gridDim.x + tile_lin) * CW_WAVES + wave) * 36) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) out[(i * 9 + t) * 64] = make_float4(acc[i][t][0], acc[i][t][1], acc[i][t][2], acc[i][t][3]);
}

bool cw8_supported(int B, int H, int W, int Cout, int Cin, int c1) {
    if (!(B > 0 && H > 0 && W > 32 && W <= 128 && Cout % CW_BI == 0 && Cin % 16 == 0 && c1 % CW_BJ == 0 && c1 <= Cin && (int64_t)B * H * W < (1LL << 24))) return false;
    return W > 64 || H % 2 == 0;
}

}  // namespace

/* floats of scratch lavt_conv3x3_wgrad needs for this shape; 0 = shape not covered (the caller uses lavt_gemm_tn's tap-shifted form) */
extern "C" int64_t lavt_conv3x3_wgrad_ws(int B, int H, int W, int Cout, int Cin, int c1) {
    if (!cw_supported(B, H, W, Cout, Cin, c1 > 0 ? c1 : Cin)) return 0;
    return (int64_t)cw_pieces(B, H, Cout, Cin) * Cout * 9 * (((Cin + CW_BJ - 1) / CW_BJ) * CW_BJ);
}

extern "C" int lavt_conv3x3_wgrad(const void* dy, int64_t ldy, const void* x1, int64_t ldx1, const void* x2, int64_t ldx2, int c1, int B, int H, int W, int Cout,
                                  int Cin, float* parts, int64_t parts_floats, float* dW, int accumulate, const void* zeros, void* stream) {
    if (x2 == nullptr) c1 = Cin;
    LAVT_CHECK_ARG(dy && x1 && parts && dW && zeros, "lavt_conv3x3_wgrad: null argument");
    LAVT_CHECK_ARG(cw_supported(B, H, W, Cout, Cin, c1), "lavt_conv3x3_wgrad: needs Cout %% 128 == 0, Cin %% 8 == 0, c1 %% 64 == 0, W <= 128 (ask lavt_conv3x3_wgrad_ws first)");
    LAVT_CHECK_ARG(ldy % 8 == 0 && ldx1 % 8 == 0 && (x2 == nullptr || ldx2 % 8 == 0), "lavt_conv3x3_wgrad: leading dimensions must be multiples of 8 elements");
    int pieces = cw_pieces(B, H, Cout, Cin);
    const int rpp = (B * H + pieces - 1) / pieces;
    pieces = (B * H + rpp - 1) / rpp;                                  // no empty piece: the reduction reads every piece's tile
    LAVT_CHECK_ARG(parts_floats >= (int64_t)pieces * Cout * 9 * (((Cin + CW_BJ - 1) / CW_BJ) * CW_BJ), "lavt_conv3x3_wgrad: scratch too small (lavt_conv3x3_wgrad_ws)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    CwArgs a;
    a.dy = (const bf16*)dy; a.ldy = ldy; a.x1 = (const bf16*)x1; a.ldx1 = ldx1; a.x2 = (const bf16*)x2; a.ldx2 = ldx2; a.c1 = c1;
    a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.Cin = Cin; a.parts = parts; a.pieces = pieces; a.zeros = zeros;
    a.rows_per_piece = rpp;
    a.xcd_order = lavt_tuning().probe[1] ? 0 : 1;
    const int tiles_j = (Cin + CW_BJ - 1) / CW_BJ;
    const dim3 grid((Cout / CW_BI) * tiles_j, pieces);
#define CW_LAUNCH(KS_)                                                                                                                          \
    do {                                                                                                                                        \
        static bool attr = false;                                                                                                               \
        if (!attr) {                                                                                                                            \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad3x3_kernel<KS_>), hipFuncAttributeMaxDynamicSharedMemorySize, CW_LDS) != hipSuccess) { \
                lavt_set_error("lavt_conv3x3_wgrad: cannot reserve %d bytes of LDS", CW_LDS);                                                   \
                return LAVT_ERR_LAUNCH;                                                                                                         \
            }                                                                                                                                   \
            attr = true;                                                                                                                        \
        }                                                                                                                                       \
        hipLaunchKernelGGL(conv_wgrad3x3_kernel<KS_>, grid, dim3(CW_THREADS), CW_LDS, st, a);                                                   \
    } while (0)
    const int ks = (W + 31) / 32;
    if (ks == 1) CW_LAUNCH(1); else if (ks == 2) CW_LAUNCH(2); else if (ks == 3) CW_LAUNCH(3); else CW_LAUNCH(4);
#undef CW_LAUNCH
    hipLaunchKernelGGL(conv_wgrad3x3_reduce, dim3(grid.x, 32), dim3(256), 0, st, reinterpret_cast<const float4*>(parts), pieces, (int)grid.x, tiles_j, Cin, dW, accumulate,
                       (const float*)nullptr, (const float*)nullptr);
    LAVT_CHECK_LAUNCH("lavt_conv3x3_wgrad");
    return LAVT_OK;
}

/* e4m3 form (ABI v7): dy, x1, x2 hold OCP e4m3 bytes q = value * 448 / |max| ([pixels][channels], leading dimensions in bytes, multiples of 16),
 * amax_dy / amax_x point at the |max| each tensor was quantised against (x1 and x2 share one); dW receives the de-quantised fp32 gradient.  Scratch as
 * lavt_conv3x3_wgrad_ws; returns 0 from lavt_conv3x3_wgrad_f8_ok for shapes it does not cover (32 < W <= 128, Cin % 16 == 0, H even when W <= 64). */
extern "C" int lavt_conv3x3_wgrad_f8_ok(int B, int H, int W, int Cout, int Cin, int c1) { return cw8_supported(B, H, W, Cout, Cin, c1 > 0 ? c1 : Cin) ? 1 : 0; }

extern "C" int lavt_conv3x3_wgrad_f8(const void* dy, int64_t ldy, const float* amax_dy, const void* x1, int64_t ldx1, const void* x2, int64_t ldx2, const float* amax_x, int c1,
                                     int B, int H, int W, int Cout, int Cin, float* parts, int64_t parts_floats, float* dW, int accumulate, const void* zeros, void* stream) {
    if (x2 == nullptr) c1 = Cin;
    LAVT_CHECK_ARG(dy && x1 && parts && dW && zeros && amax_dy && amax_x, "lavt_conv3x3_wgrad_f8: null argument");
    LAVT_CHECK_ARG(cw8_supported(B, H, W, Cout, Cin, c1), "lavt_conv3x3_wgrad_f8: needs Cout %% 128 == 0, Cin %% 16 == 0, c1 %% 64 == 0, 32 < W <= 128, H even when W <= 64");
    LAVT_CHECK_ARG(ldy % 16 == 0 && ldx1 % 16 == 0 && (x2 == nullptr || ldx2 % 16 == 0), "lavt_conv3x3_wgrad_f8: leading dimensions must be multiples of 16 bytes");
    const int rpk = W > 64 ? 1 : 2;
    int pieces = cw_pieces(B, H, Cout, Cin);
    int rpp = (B * H + pieces - 1) / pieces;
    rpp = (rpp + rpk - 1) / rpk * rpk;                                 // a K step never straddles two pieces
    pieces = (B * H + rpp - 1) / rpp;
    LAVT_CHECK_ARG(parts_floats >= (int64_t)pieces * Cout * 9 * (((Cin + CW_BJ - 1) / CW_BJ) * CW_BJ), "lavt_conv3x3_wgrad_f8: scratch too small (lavt_conv3x3_wgrad_ws)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    Cw8Args a;
    a.dy = (const unsigned char*)dy; a.ldy = ldy; a.x1 = (const unsigned char*)x1; a.ldx1 = ldx1; a.x2 = (const unsigned char*)x2; a.ldx2 = ldx2; a.c1 = c1;
    a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.Cin = Cin; a.parts = parts; a.pieces = pieces; a.zeros = zeros;
    a.rows_per_piece = rpp;
    a.xcd_order = lavt_tuning().probe[1] ? 0 : 1;
    const int tiles_j = (Cin + CW_BJ - 1) / CW_BJ;
    const dim3 grid((Cout / CW_BI) * tiles_j, pieces);
#define CW8_LAUNCH(RPK_)                                                                                                                        \
    do {                                                                                                                                        \
        static bool attr = false;                                                                                                               \
        if (!attr) {                                                                                                                            \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad3x3_f8_kernel<RPK_>), hipFuncAttributeMaxDynamicSharedMemorySize, Cw8<RPK_>::LDS) != hipSuccess) { \
                lavt_set_error("lavt_conv3x3_wgrad_f8: cannot reserve %d bytes of LDS", Cw8<RPK_>::LDS);                                         \
                return LAVT_ERR_LAUNCH;                                                                                                         \
            }                                                                                                                                   \
            attr = true;                                                                                                                        \
        }                                                                                                                                       \
        hipLaunchKernelGGL(conv_wgrad3x3_f8_kernel<RPK_>, grid, dim3(CW_THREADS), Cw8<RPK_>::LDS, st, a);                                        \
    } while (0)
    if (rpk == 1) CW8_LAUNCH(1); else CW8_LAUNCH(2);
#undef CW8_LAUNCH
    hipLaunchKernelGGL(conv_wgrad3x3_reduce, dim3(grid.x, 32), dim3(256), 0, st, reinterpret_cast<const float4*>(parts), pieces, (int)grid.x, tiles_j, Cin, dW, accumulate,
                       amax_dy, amax_x);
    LAVT_CHECK_LAUNCH("lavt_conv3x3_wgrad_f8");
    return LAVT_OK;
}
