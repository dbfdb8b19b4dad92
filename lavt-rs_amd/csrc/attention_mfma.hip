// bf16 MFMA formulation of the shifted-window attention core for gfx950 (v_mfma_f32_16x16x32_bf16).
// Reference arithmetic: WindowAttention.forward, lib/backbone.py:123-140 (+ shift mask :634-652).
//
// Forward: ONE WAVE per (window, head), no workgroup barrier in the loop.  Scores are computed transposed,
//   S^T = K Q^T, so that each lane owns one query row i (the MFMA column) and 4*NT keys j in registers: the
//   softmax row reductions are in-register + two wave shuffles (lane ^ 16, lane ^ 32), and the un-normalised
//   P^T accumulator is already the B operand of O^T = V^T P^T (no LDS round trip for P).  K fragments come
//   straight from HBM/L2 (they are k-contiguous); V fragments need the transposed layout and are produced once
//   per window with ds_read_b64_tr_b16 from a small LDS image; both stay in registers for all query tiles.
// Backward: one workgroup per (head, chunk of windows).  Per window: Q, K, V, dO in LDS; phase 1 recomputes
//   P = exp(S - lse) and dS = P (dP - delta) tile by tile (un-transposed, so the relative-position-bias gradient
//   of a thread always hits the same (i, j) and is summed over the chunk's windows IN REGISTERS, one atomic per
//   element per workgroup at the end instead of one per window); P and dS go to LDS as bf16; phase 2 forms
//   dV = P^T dO, dK = dS^T Q, dQ = dS K with MFMA, reading the k-major operands with the transposing LDS read.
#include <stdlib.h>

#include <string.h>

#include "common.h"
#include "dtable_body.h"

namespace {

constexpr int HD = 32;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ bf16x8 join4(bf16x4 lo, bf16x4 hi) {
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
// A/B fragment whose 8 k-values are contiguous in LDS: row `row`, elements k0..k0+7
__device__ __forceinline__ bf16x8 lds_row8(const bf16* s, int ld, int row, int k0) {
    return *reinterpret_cast<const bf16x8*>(s + row * ld + k0);
}
__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (bf16)0.f;
    return z;
}
__device__ __forceinline__ bf16x8 ldg8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
// A lane holds two packed quadruples of one (token, head) row: channels 4g .. 4g+3 (p0) and 16+4g .. 16+4g+3 (p1), g = lane / 16.  Lanes g and
// g ^ 1 swap one of them (ds_bpermute, no memory) so that every lane stores 16 contiguous bytes -- 64 contiguous bytes per row and
// wave-instruction instead of 8-byte pieces.  Every lane of the wave must call (the partner of a valid lane is valid: same row).
__device__ __forceinline__ void store_head_row16(bf16* row_head, int g, uint2 p0, uint2 p1, bool valid) {
    const bool odd = g & 1;
    const uint2 send = odd ? p0 : p1;
    const uint2 got = make_uint2((unsigned)__shfl_xor((int)send.x, 16, 64), (unsigned)__shfl_xor((int)send.y, 16, 64));
    const uint4 out = odd ? make_uint4(got.x, got.y, p1.x, p1.y) : make_uint4(p0.x, p0.y, got.x, got.y);
    if (valid) *reinterpret_cast<uint4*>(row_head + (odd ? 16 + 4 * (g - 1) : 4 * g)) = out;
}

// ================================================================================================ forward
// One workgroup (4 waves; 8 for the 25-tile windows) per (window, head).  Q, K, V are staged once into LDS by all threads (one round of 16-byte loads in
// flight instead of the per-wave, per-row latency chain of a one-wave-per-pair kernel: measured 25 us per layer whatever the size);
// the head's column of the relative-position table sits in LDS too (bias[i][j] = tab[base[i] - base[j] + centre]), so the query-tile loop
// touches no global memory except its stores.  A wave owns query tiles it = wave, wave + WAVES, ...; it keeps the K fragments (A operand of
// S^T = K Q^T) and the V^T fragments (A operand of O^T = V^T P^T, transposing LDS read) in registers for all its tiles.
// S^T puts one query per lane column: softmax reductions are in-register + two shuffles, and the un-normalised P^T accumulators of two
// key tiles are directly the B operand of the second MFMA (no LDS round trip for P).
// LDS rows of Q / K / V (/ dO): 64 bytes = four 16-byte chunks, UNPADDED, chunk c of row r stored at chunk c ^ swz(r) (round 6).  With the 80-byte padded rows of
// rounds 2-5 the 16-byte row reads were conflict-free but the transposing 8-byte reads of 8 consecutive rows were 2-way (39 % of the LDS cycles of the 392-token
// backward were bank conflicts).  swz takes bit 2 of the row into bit 1 of the chunk and bit 3 into bit 0: the 16 rows of a row-fragment read (same chunk) land in 16
// different 16-byte bank groups, and the 8 rows x 2 chunks of a transposing read cover the 64 banks once.  Tile offsets are multiples of 16 rows: a lane's swizzle
// is a constant of the lane.
constexpr int F_LD = 32;        // bf16 elements per LDS row
__device__ __forceinline__ int swz(int row) { return (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }

// Arithmetic diet (the 392-token kernels are VALU-issue bound: rocprofv3 counters, profiles/r02_pmc_attention.json, r06_pmc_attn_392_tokens.json): scores live
// in the log2 domain (table column and scale pre-multiplied by log2 e when staged, so the exponential is the bare v_exp_f32), the shift-mask compare is compiled
// out for unshifted blocks (REGION), the padding tile of an odd tile count costs no arithmetic, and padded tokens need no bounds select in any tile: the
// staged constants make their terms vanish (round 6, below).
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

typedef __attribute__((address_space(3))) float lds_f32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float lds_f32_at(uint32_t addr) { return *reinterpret_cast<lds_f32*>(addr); }

// Round 6: the instruction diet of the backward below applied to the forward (before: 26 VALU instructions per MFMA in the 392-token launch, and ONE wave per
// SIMD -- 112 KB of LDS, one 4-wave workgroup per CU; 39.8 -> 27.9 us for 16 x 16 units of 392 tokens, shifted 48.4 -> 31.7; 144- and 49-token windows +-1 us):
//   * WAVES = 8 for the 25-tile windows: two waves per SIMD (the K / V tiles are staged once per workgroup either way; 25 query tiles on 8 waves);
//   * table gather with byte offsets (one subtraction per element); a padded KEY carries an offset that lands every query in a run of -1e30 entries behind
//     the table (its probability is exp2(-1e30 - max) = 0): no bounds select in any tile;
//   * shift mask as one AND + compare + select + add per element on the XOR of the packed region ids; row sum as packed adds.
template <int NT, int WAVES, bool REGION>
__global__ __launch_bounds__(WAVES * 64) void wattn_fwd_mfma(const bf16* __restrict__ qkv, const float* __restrict__ table,
                                                      const int8_t* __restrict__ region, int nw_img, bf16* __restrict__ out,
                                                      float* __restrict__ lse, int wd, int wh, int ww, int nwin, int N, int heads, float scale) {
    constexpr int KS = (NT + 1) / 2, NP = KS * 32, NTHR = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16* Qs = reinterpret_cast<bf16*>(smem_raw);
    bf16* Ks = Qs + NP * F_LD;
    bf16* Vs = Ks + NP * F_LD;
    int* bs = reinterpret_cast<int*>(Vs + NP * F_LD);
    uint8_t* Rs = reinterpret_cast<uint8_t*>(bs + NP);
    float* tab = reinterpret_cast<float*>(Rs + NP);        // [R] table column * log2 e, then [centre + 1] x -1e30
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int w = blockIdx.x / heads, h = blockIdx.x % heads;
    const int C = heads * HD;
    const int R = (2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1);
    const int centre = ((wd - 1) * (2 * wh - 1) + (wh - 1)) * (2 * ww - 1) + (ww - 1);
    const bf16* base = qkv + (int64_t)w * N * 3 * C + h * HD;

    for (int e = tid; e < NP * 4; e += NTHR) {
        const int row = e >> 2, c = e & 3;
        uint4 q = make_uint4(0, 0, 0, 0), k = q, v = q;
        if (row < N) {
            const bf16* r = base + (int64_t)row * 3 * C + c * 8;
            q = *reinterpret_cast<const uint4*>(r);
            k = *reinterpret_cast<const uint4*>(r + C);
            v = *reinterpret_cast<const uint4*>(r + 2 * C);
        }
        const int pc = (c ^ swz(row)) * 8;
        *reinterpret_cast<uint4*>(Qs + row * F_LD + pc) = q;
        *reinterpret_cast<uint4*>(Ks + row * F_LD + pc) = k;
        *reinterpret_cast<uint4*>(Vs + row * F_LD + pc) = v;
    }
    for (int e = tid; e < NP; e += NTHR) {
        const int dz = e / (wh * ww), hy = (e / ww) % wh, wx = e % ww;
        // byte offsets; a padded token: idx_i - idx_j + centre = R + idx_i for every query i, i.e. the -1e30 run behind the table (idx_i <= centre)
        bs[e] = e < N ? 4 * ((dz * (2 * wh - 1) + hy) * (2 * ww - 1) + wx) : -4 * (R - centre);
        Rs[e] = (region && e < N) ? (uint8_t)region[(int64_t)(w % nw_img) * N + e] : 0;
    }
    for (int e = tid; e < R + centre + 1; e += NTHR) tab[e] = e < R ? table[(int64_t)e * heads + h] * LOG2E : -1e30f;
    __syncthreads();

    const int QT = (N + 15) / 16;
    if (wave >= QT) return;
    const float sc2 = scale * LOG2E;
    const f32x4 sc4 = {sc2, sc2, sc2, sc2};
    const uint32_t tab_c = (uint32_t)reinterpret_cast<uintptr_t>((lds_f32*)tab) + 4u * (uint32_t)centre;
    constexpr bool CACHE = NT <= 10;
    constexpr int NC = CACHE ? NT : 1, KC = CACHE ? KS : 1;
    bf16x8 kf[NC], vf[2][KC];
    const int kg = 8 * (g ^ swz(c16));                     // this lane's chunk of a row fragment (rows 16 t + c16)
    const int rr = 4 * g + (c16 >> 2);                     // its row of a transposing read (rows 32 ks + rr, + 16), and the two column groups u = 0, 1
    const int tcol[2] = {((((c16 >> 1) & 1)) ^ swz(rr)) * 8 + 4 * (c16 & 1), ((2 + ((c16 >> 1) & 1)) ^ swz(rr)) * 8 + 4 * (c16 & 1)};
    auto k_frag = [&](int t) { return lds_row8(Ks, F_LD, 16 * t + c16, kg); };
    auto v_frag = [&](int u, int ks) {
        const bf16* p = Vs + (32 * ks + rr) * F_LD + tcol[u];
        return join4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p), __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 16 * F_LD)));
    };
    if constexpr (CACHE) {
#pragma unroll
        for (int t = 0; t < NT; ++t) kf[t] = k_frag(t);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) vf[u][ks] = v_frag(u, ks);
    }

    for (int it = wave; it < QT; it += WAVES) {
        const int i = 16 * it + c16;
        const bool vi = i < N;
        const bf16x8 qf = lds_row8(Qs, F_LD, i, kg);
        const uint32_t ri4 = REGION ? 0x01010101u * Rs[i] : 0u;
        const uint32_t bi = tab_c + (uint32_t)bs[i];           // (a padded query lane gathers from wherever: its column is discarded)
        f32x4 s[2 * KS];
        float mx = -1e30f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(CACHE ? kf[CACHE ? t : 0] : k_frag(t), qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            const int j0 = 16 * t + 4 * g;
            const u32x4 bj = *reinterpret_cast<const u32x4*>(bs + j0);
            const f32x4 bb = {lds_f32_at(bi - bj[0]), lds_f32_at(bi - bj[1]), lds_f32_at(bi - bj[2]), lds_f32_at(bi - bj[3])};
            f32x4 v = __builtin_elementwise_fma(acc, sc4, bb);
            if constexpr (REGION) {
                const uint32_t x = *reinterpret_cast<const uint32_t*>(Rs + j0) ^ ri4;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (x & (0xFFu << (8 * r))) ? -100.0f * LOG2E : 0.f;
            }
            s[t] = v;
            mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
            // (25 tiles: left alone the scheduler hoists every tile's LDS reads to the top -- 434 registers; at the 256 of two waves per SIMD that spilt 76)
            if constexpr (!CACHE) { if (t % 2 == 1) __builtin_amdgcn_sched_barrier(0); }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float nmx = -mx;
        f32x4 sum4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2 * KS; ++t) {
            if (t < NT) {
                const f32x4 d = s[t] + nmx;
                f32x4 p;
#pragma unroll
                for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(d[r]);
                s[t] = p;
                sum4 += p;
            } else s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        if (vi && g == 0) lse[((int64_t)w * heads + h) * N + i] = (mx + __log2f(sum)) * LN2;
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 pw;
            pw[0] = pack_bf16x2(s[2 * ks][0], s[2 * ks][1]); pw[1] = pack_bf16x2(s[2 * ks][2], s[2 * ks][3]);
            pw[2] = pack_bf16x2(s[2 * ks + 1][0], s[2 * ks + 1][1]); pw[3] = pack_bf16x2(s[2 * ks + 1][2], s[2 * ks + 1][3]);
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
            o[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(CACHE ? vf[0][CACHE ? ks : 0] : v_frag(0, ks), pf, o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(CACHE ? vf[1][CACHE ? ks : 0] : v_frag(1, ks), pf, o[1], 0, 0, 0);
        }
        {
            const float inv = 1.f / sum;
            store_head_row16(out + ((int64_t)w * N + (vi ? i : 0)) * C + h * HD, g,
                             make_uint2(pack_bf16x2(o[0][0] * inv, o[0][1] * inv), pack_bf16x2(o[0][2] * inv, o[0][3] * inv)),
                             make_uint2(pack_bf16x2(o[1][0] * inv, o[1][1] * inv), pack_bf16x2(o[1][2] * inv, o[1][3] * inv)), vi);
        }
    }
}

// ================================================================================================ backward
// One workgroup (4 waves) per (head, chunk of windows); per window Q, K, V, dO sit in LDS (rows >= N zero) and everything else stays in
// registers -- no P / dS matrices in LDS, no barrier between the phases:
//   pass 1 (a wave owns KEY tiles jt):   S = Q K^T and dP = dO V^T tile by tile, un-transposed (lane: key j = lane & 15, queries 4g+r);
//            the P / dS accumulators of two consecutive query tiles ARE the A operands (k = 32 queries) of
//            dV[jt] += P^T dO and dK[jt] += dS^T Q, whose B operands come from the transposing LDS read of dO / Q.
//   pass 2 (a wave owns QUERY tiles it): S^T = K Q^T and dP^T = V dO^T (lane: query i = lane & 15, keys 4g+r, as in the forward);
//            dS^T of two key tiles is the B operand of dQ^T[it] += K^T dS^T; the bias is read as float4 along j.
// P / dS are recomputed in the second pass (4 extra MFMAs per tile pair) instead of being exchanged through LDS: the kernel went from
// 159 KB of LDS (one workgroup per CU) to ~55 KB.  The relative-position-bias gradient is binned in an LDS histogram over the table
// index (idx = base[i] - base[j] + centre) and flushed with one global atomic per table entry per workgroup; the dense [heads][N][ld]
// gradient (dbias) is only written when no table pointer is given.
constexpr int R_LD = 32;        // bf16 elements per LDS row of Q / K / V / dO (64-byte rows, chunks swizzled by swz(row): see the forward)

// Table-gradient binning of ONE attention-backward launch (wattn_dtable_kernel's arguments).  The binning kernel (7.5 us x 24 per Swin-B step)
// sits between the attention backward and the qkv data gradient without either needing it: a launch can carry the job of the PREVIOUS
// launch (the layer processed just before in backward, whose dS slabs are still in the Infinity Cache) as extra workgroups that run in the
// CU slots its own 288 workgroups leave free (lavt_window_attn_bwd_chained).
struct DtableJob {
    const bf16* slab;
    float* part;
    int slab_ld, wd, wh, ww, nwin, N, heads, rows_per_block, win_per_group, gx, gz;      // gx, gz: the binning grid (gy = heads)
};

// Round 6: the instruction diet (before: 16.5 VALU instructions per MFMA in the 392-token launch, SGPR pairs of the per-tile select masks spilt into VGPR
// lanes; 69.4 -> 57.2 us for 16 x 16 units of 392 tokens, shifted 81.3 -> 62.4; the 144-token launches are latency-bound and did not move: 17.3 us either way):
//   * table gather: bs[] holds BYTE offsets (4 x index) and the per-task constant carries the table's LDS address and the centre, so an element's address is
//     ONE subtraction (was: subtract, shift-add);
//   * lse and delta are staged NEGATED: bias + (-lse) and dP + (-delta) are packed adds (v_pk_add_f32: two per quadruple instead of four subtractions);
//   * the shift mask is a select AFTER the exponential (masked probabilities are < 2^-144 x e^(s - lse): zero in bf16 either way) -- compare + select per
//     element instead of extract, compare, select, add;
//   * padding selects only in the one tile that has padding (compile-time tile index) and only for the operand that is not discarded with its lane;
//   * the dS slab takes the two packed words the MFMA operand is made of (no second conversion, no re-pack), through a buffer descriptor over the
//     (window, head) slab: a lane without a row carries an offset beyond the descriptor (the hardware drops the store) -- no branch, no 64-bit address per tile.

template <int NT, int WAVES, bool REGION>
__global__ __launch_bounds__(WAVES * 64, (WAVES == 8 && NT <= 10) ? 4 : 2) void wattn_bwd_mfma(const bf16* __restrict__ qkv, const float* __restrict__ table,
                                                      const int8_t* __restrict__ region, int nw_img, const bf16* __restrict__ out,
                                                      const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                      bf16* __restrict__ dqkv, bf16* __restrict__ slab, int slab_ld,
                                                      int wd, int wh, int ww, int nwin, int N, int heads, float scale, int win_per_block,
                                                      int attn_blocks, const DtableJob job, const int split_from, const int split_pieces) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KS = (NT + 1) / 2, NP = KS * 32;
    constexpr int NTHR = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if constexpr (WAVES == 8) {
        if ((int)blockIdx.x >= attn_blocks) {              // riders: the previous launch's binning, two 256-thread binning blocks per workgroup
            // (both halves take the same barriers: a half without a unit runs on unit 0's geometry with its stores masked off)
            const int half = threadIdx.x >> 8, u = 2 * ((int)blockIdx.x - attn_blocks) + half;
            const int units = job.gx * job.heads * job.gz;
            const int uu = u < units ? u : 0;
            const int bx = uu % job.gx, hh = (uu / job.gx) % job.heads, bz = uu / (job.gx * job.heads);
            const int R = (2 * job.wd - 1) * (2 * job.wh - 1) * (2 * job.ww - 1);
            dtable_block(job.slab, u < units ? job.part : nullptr, job.slab_ld, job.wd, job.wh, job.ww, job.nwin, job.N, job.heads, job.rows_per_block,
                         job.win_per_group, bx, hh, bz, job.gx, threadIdx.x & 255, smem_raw + half * (size_t)((R + job.N) * 4 + 16));
            return;
        }
    }
    bf16* Qs = reinterpret_cast<bf16*>(smem_raw);
    bf16* Ks = Qs + NP * R_LD;
    bf16* Vs = Ks + NP * R_LD;
    bf16* Os = Vs + NP * R_LD;                             // dO
    float* dl = reinterpret_cast<float*>(Os + NP * R_LD);  // -delta_i = -sum_d dO*O
    float* ls = dl + NP;                                   // -lse_i * log2 e
    int* bs = reinterpret_cast<int*>(ls + NP);             // 4 x table-index base of token i (a byte offset)
    uint8_t* Rs = reinterpret_cast<uint8_t*>(bs + NP);
    float* tab = reinterpret_cast<float*>(Rs + NP);        // this head's column of the bias table * log2 e

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int kg = 8 * (g ^ swz(c16));                     // this lane's chunk of a row fragment (rows 16 t + c16)
    const int rr = 4 * g + (c16 >> 2);                     // its row of a transposing read (rows 32 ks + rr, + 16) and the two column groups u = 0, 1
    const int tcol[2] = {((((c16 >> 1) & 1)) ^ swz(rr)) * 8 + 4 * (c16 & 1), ((2 + ((c16 >> 1) & 1)) ^ swz(rr)) * 8 + 4 * (c16 & 1)};
    const int QT = (N + 15) / 16;
    // Units beyond a whole number of rounds of the chip (18 windows x 16 heads = 288 on 256 CUs: the 32 CUs that hold two took 21.1 us where one unit per
    // CU takes 14.9) are cut into `split_pieces` workgroups, each with a contiguous run of the unit's 2 QT tasks (a task = one key tile of pass 1 or one query
    // tile of pass 2: whole output rows, nothing to add up): every CU then carries one unit + one small piece.  A piece stages the unit's tensors like a whole unit.
    int unit = blockIdx.x, t_lo = 0, t_hi = 2 * QT;
    if (split_pieces > 1 && (int)blockIdx.x >= split_from) {
        const int k = (int)blockIdx.x - split_from, piece = k % split_pieces;
        unit = split_from + k / split_pieces;
        t_lo = piece * 2 * QT / split_pieces;
        t_hi = (piece + 1) * 2 * QT / split_pieces;
    }
    const int h = unit % heads, chunk = unit / heads;
    const int C = heads * HD;
    const int R = (2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1);
    const int centre = ((wd - 1) * (2 * wh - 1) + (wh - 1)) * (2 * ww - 1) + (ww - 1);
    const float sc2 = scale * LOG2E;
    const f32x4 sc4 = {sc2, sc2, sc2, sc2};
    const uint32_t tab_c = (uint32_t)reinterpret_cast<uintptr_t>((lds_f32*)tab) + 4u * (uint32_t)centre;       // LDS address of tab[centre]
    for (int e = tid; e < R; e += NTHR) tab[e] = table[(int64_t)e * heads + h] * LOG2E;
    for (int e = tid; e < NP; e += NTHR) {
        const int dz = e / (wh * ww), hy = (e / ww) % wh, wx = e % ww;
        bs[e] = e < N ? 4 * ((dz * (2 * wh - 1) + hy) * (2 * ww - 1) + wx) : 0;          // (padded tokens: any index inside the table -- their bias must be FINITE, see below)
    }
    // Task list of a window: t < QT -> pass 1 of key tile t; QT <= t < 2 QT -> pass 2 of query tile t - QT.  A wave takes t = slot, slot + WAVES, ...
    // (9 tiles on 8 waves: 3, 3, 2, ... tasks instead of one wave running 2 + 2).  Waves w, w + 4 share a SIMD, so workgroups that are likely to share a
    // CU (the grid's second round of 256) start the list two slots later: the heavy SIMDs differ.
    // (9 waves of exactly one pass-1 and one pass-2 task -- 96 VGPRs for two workgroups per CU, 34 spills in the shifted variant -- measured
    // slower: 36.7 vs 34.1 us with the table kernels for the stage-2 launch, 10.77 vs 10.50 ms per step.)
    const int slot = (wave + 2 * ((unit >> 8) & 1)) % WAVES;

    const int w_end = min(nwin, (chunk + 1) * win_per_block);
    for (int w = chunk * win_per_block; w < w_end; ++w) {
        __syncthreads();
        const bf16* base = qkv + (int64_t)w * N * 3 * C + h * HD;
        for (int e = tid; e < NP * 4; e += NTHR) {
            const int row = e >> 2, c = e & 3;
            uint4 q = make_uint4(0, 0, 0, 0), k = q, v = q, d = q, o = q;
            if (row < N) {
                const bf16* r = base + (int64_t)row * 3 * C + c * 8;
                q = *reinterpret_cast<const uint4*>(r);
                k = *reinterpret_cast<const uint4*>(r + C);
                v = *reinterpret_cast<const uint4*>(r + 2 * C);
                const int64_t oo = ((int64_t)w * N + row) * C + h * HD + c * 8;
                d = *reinterpret_cast<const uint4*>(dout + oo);
                o = *reinterpret_cast<const uint4*>(out + oo);
            }
            const int pc = (c ^ swz(row)) * 8;
            *reinterpret_cast<uint4*>(Qs + row * R_LD + pc) = q;
            *reinterpret_cast<uint4*>(Ks + row * R_LD + pc) = k;
            *reinterpret_cast<uint4*>(Vs + row * R_LD + pc) = v;
            *reinterpret_cast<uint4*>(Os + row * R_LD + pc) = d;
            float fd[8], fo[8], part = 0.f;
            chunk_to_f<bf16>(d, fd);
            chunk_to_f<bf16>(o, fo);
#pragma unroll
            for (int x = 0; x < 8; ++x) part += fd[x] * fo[x];
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            if (c == 0) { dl[row] = -part; ls[row] = row < N ? -lse[((int64_t)w * heads + h) * N + row] * LOG2E : -1e30f; }
        }
        if constexpr (REGION)
            for (int e = tid; e < NP; e += NTHR) Rs[e] = e < N ? (uint8_t)region[(int64_t)(w % nw_img) * N + e] : 0;
        __syncthreads();
        // this (window, head)'s dS slab [N][slab_ld] behind a raw buffer descriptor (offsets at or beyond its size: the store is dropped)
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(slab + ((int64_t)w * heads + h) * N * slab_ld, 0, N * slab_ld * 2, 0x00020000);

#pragma unroll 1
        for (int t = t_lo + slot; t < t_hi; t += WAVES) {
          if (t < QT) {
            // ---- pass 1: dV, dK of key tile jt ----------------------------------------------------------------------------
            const int jt = t;
            const int j = 16 * jt + c16;
            const bf16x8 kfr = lds_row8(Ks, R_LD, j, kg);
            const bf16x8 vfr = lds_row8(Vs, R_LD, j, kg);
            const uint32_t rj4 = REGION ? 0x01010101u * Rs[j] : 0u;
            f32x4 dv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, dk[2] = {dv[0], dv[0]};
            const uint32_t bj = tab_c - (uint32_t)bs[j];              // + bs[i]: the LDS address of tab[idx_i - idx_j + centre]
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                u32x4 ppw, dsw;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int it = 2 * ks + half;
                    if (it >= NT) {                        // the padding tile of an odd tile count: zero fragments, no arithmetic
                        ppw[2 * half] = 0u; ppw[2 * half + 1] = 0u; dsw[2 * half] = 0u; dsw[2 * half + 1] = 0u;
                        continue;
                    }
                    const bf16x8 qf = lds_row8(Qs, R_LD, 16 * it + c16, kg);
                    const bf16x8 of = lds_row8(Os, R_LD, 16 * it + c16, kg);
                    const f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kfr, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(of, vfr, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    const int i0 = 16 * it + 4 * g;
                    const f32x4 nl4 = *reinterpret_cast<const f32x4*>(ls + i0), nd4 = *reinterpret_cast<const f32x4*>(dl + i0);
                    const u32x4 bi4 = *reinterpret_cast<const u32x4*>(bs + i0);
                    const f32x4 bb = {lds_f32_at(bj + bi4[0]), lds_f32_at(bj + bi4[1]), lds_f32_at(bj + bi4[2]), lds_f32_at(bj + bi4[3])};
                    const f32x4 a = __builtin_elementwise_fma(s, sc4, bb + nl4);
                    f32x4 p;
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(a[r]);
                    if constexpr (REGION) {
                        const uint32_t x = *reinterpret_cast<const uint32_t*>(Rs + i0) ^ rj4;
#pragma unroll
                        for (int r = 0; r < 4; ++r) p[r] = (x & (0xFFu << (8 * r))) ? 0.f : p[r];
                    }
                    // No padding selects: a padded QUERY row carries -lse = -1e30 (P = exp2(-1e30) = 0, dS = 0 x finite); a padded KEY lane holds finite values
                    // (zero K / V rows, a bias from inside the table) in the dV / dK columns that are discarded with it.
                    const f32x4 ds = p * (dp + nd4);
                    ppw[2 * half] = pack_bf16x2(p[0], p[1]); ppw[2 * half + 1] = pack_bf16x2(p[2], p[3]);
                    dsw[2 * half] = pack_bf16x2(ds[0], ds[1]); dsw[2 * half + 1] = pack_bf16x2(ds[2], ds[3]);
                }
                const bf16x8 pp = __builtin_bit_cast(bf16x8, ppw), dp8 = __builtin_bit_cast(bf16x8, dsw);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int off = (32 * ks + rr) * R_LD + tcol[u];
                    const bf16x8 ot = join4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Os + off)),
                                            __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Os + off + 16 * R_LD)));
                    const bf16x8 qt = join4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Qs + off)),
                                            __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Qs + off + 16 * R_LD)));
                    dv[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ot, pp, dv[u], 0, 0, 0);
                    dk[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dp8, dk[u], 0, 0, 0);
                }
            }
            {
                const bool vj = j < N;
                bf16* row = dqkv + ((int64_t)w * N + (vj ? j : 0)) * 3 * C + h * HD;
                store_head_row16(row + C, g, make_uint2(pack_bf16x2(dk[0][0] * scale, dk[0][1] * scale), pack_bf16x2(dk[0][2] * scale, dk[0][3] * scale)),
                                 make_uint2(pack_bf16x2(dk[1][0] * scale, dk[1][1] * scale), pack_bf16x2(dk[1][2] * scale, dk[1][3] * scale)), vj);
                store_head_row16(row + 2 * C, g, make_uint2(pack_bf16x2(dv[0][0], dv[0][1]), pack_bf16x2(dv[0][2], dv[0][3])),
                                 make_uint2(pack_bf16x2(dv[1][0], dv[1][1]), pack_bf16x2(dv[1][2], dv[1][3])), vj);
            }
          } else {
            // ---- pass 2: dQ of query tile it; dS slab for the bias gradient -------------------------------------------------
            const int it = t - QT;
            const int i = 16 * it + c16;
            const bool vi = i < N;
            const bf16x8 qfb = lds_row8(Qs, R_LD, i, kg);
            const bf16x8 ofb = lds_row8(Os, R_LD, i, kg);
            const float nli = ls[i], ndi = dl[i];
            const uint32_t ri4 = REGION ? 0x01010101u * Rs[i] : 0u;
            const uint32_t bi = tab_c + (uint32_t)bs[i];
            const uint32_t srow = vi ? (uint32_t)(i * slab_ld * 2 + 8 * g) : 0x80000000u;        // byte offset of (row i, column 4g) in the slab
            f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                u32x4 dsw;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int jt = 2 * ks + half, j0 = 16 * jt + 4 * g;
                    if (jt >= NT) {
                        dsw[2 * half] = 0u; dsw[2 * half + 1] = 0u;          // (slab columns >= N are never read)
                        continue;
                    }
                    const bf16x8 ka = lds_row8(Ks, R_LD, 16 * jt + c16, kg);
                    const bf16x8 va = lds_row8(Vs, R_LD, 16 * jt + c16, kg);
                    const f32x4 st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qfb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    const f32x4 dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, ofb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    const u32x4 bj4 = *reinterpret_cast<const u32x4*>(bs + j0);
                    const f32x4 bb = {lds_f32_at(bi - bj4[0]), lds_f32_at(bi - bj4[1]), lds_f32_at(bi - bj4[2]), lds_f32_at(bi - bj4[3])};
                    const f32x4 a = __builtin_elementwise_fma(st, sc4, bb + nli);
                    f32x4 p;
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(a[r]);
                    if constexpr (REGION) {
                        const uint32_t x = *reinterpret_cast<const uint32_t*>(Rs + j0) ^ ri4;
#pragma unroll
                        for (int r = 0; r < 4; ++r) p[r] = (x & (0xFFu << (8 * r))) ? 0.f : p[r];
                    }
                    // (a padded KEY column: P finite, dS = P x (0 - delta) finite, times its zero K row = 0 in dQ; its slab columns are never read.
                    //  A padded query lane is discarded with its dQ row and its slab row.)
                    const f32x4 ds = p * (dpt + ndi);
                    const uint32_t w0 = pack_bf16x2(ds[0], ds[1]), w1 = pack_bf16x2(ds[2], ds[3]);
                    dsw[2 * half] = w0; dsw[2 * half + 1] = w1;
                    // dS of this (window, head) goes to its own slab, in bf16 -- the values dQ / dK are computed from -- as plain 8-byte stores;
                    // wattn_dtable_kernel sums the slabs over windows in fp32 and bins them (fp32 slabs: twice the bytes written here and read
                    // there, 9.1 vs ~6 us for the binning kernel of a stage-2 block).
                    // (LDS float atomics for an in-kernel histogram -- ds_add_f32 per element, up to 4 lanes of a wave on one table entry -- were
                    // measured twice: 37 of 57 us per window-head in round 1, 72 vs 34 us for the stage-2 launch in round 2; global atomics as bad.)
                    // (slab_ld >= 16 NT: checked by the host)
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{w0, w1}, srs, (int)(srow + 32u * jt), 0, 0);
                }
                const bf16x8 ds8 = __builtin_bit_cast(bf16x8, dsw);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int off = (32 * ks + rr) * R_LD + tcol[u];
                    const bf16x8 kt = join4(__builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Ks + off)),
                                            __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Ks + off + 16 * R_LD)));
                    dq[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, ds8, dq[u], 0, 0, 0);
                }
            }
            store_head_row16(dqkv + ((int64_t)w * N + (vi ? i : 0)) * 3 * C + h * HD, g,
                             make_uint2(pack_bf16x2(dq[0][0] * scale, dq[0][1] * scale), pack_bf16x2(dq[0][2] * scale, dq[0][3] * scale)),
                             make_uint2(pack_bf16x2(dq[1][0] * scale, dq[1][1] * scale), pack_bf16x2(dq[1][2] * scale, dq[1][3] * scale)), vi);
          }
        }
    }
#endif
}

__global__ __launch_bounds__(256) void wattn_dtable_kernel(const bf16* __restrict__ slab, float* __restrict__ part, int slab_ld, int wd, int wh,
                                                           int ww, int nwin, int N, int heads, int rows_per_block, int win_per_group) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    dtable_block(slab, part, slab_ld, wd, wh, ww, nwin, N, heads, rows_per_block, win_per_group, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, threadIdx.x, smem_raw);
}

__global__ void wattn_dtable_finish(const float* __restrict__ part, float* __restrict__ dtable, int pieces, int per_z, int heads, int R) {
    // blockIdx.z walks a group of per-workgroup histograms (8 loads in flight per thread); the groups meet in dtable through atomics
    const int e = blockIdx.x * blockDim.x + threadIdx.x, h = blockIdx.y;
    if (e >= R) return;
    const int k0 = blockIdx.z * per_z, k1 = min(pieces, k0 + per_z);
    const float* q = part + ((int64_t)k0 * heads + h) * R + e;
    const int64_t st = (int64_t)heads * R;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    int k = k0;
    for (; k + 7 < k1; k += 8, q += 8 * st)
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += q[u * st];
    for (; k < k1; ++k, q += st) acc[0] += *q;
    const float a = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    if (gridDim.z > 1) atomicAdd(dtable + (int64_t)e * heads + h, a);
    else dtable[(int64_t)e * heads + h] += a;
}

template <int NT> size_t bwd_lds_bytes(int R) {
    constexpr int NP = ((NT + 1) / 2) * 32;
    return (size_t)4 * NP * R_LD * 2 + (size_t)3 * NP * 4 + NP + (size_t)R * 4 + 16;
}

}  // namespace

int lavt_window_attn_fwd_mfma(const void* qkv, const float* table, const int8_t* region, int nw_img, void* out, float* lse,
                              int wd, int wh, int ww, int nwin, int N, int heads, float scale, hipStream_t st) {
    if (N > 400 || !table) { lavt_set_error("lavt_window_attn_fwd(mfma): N=%d (<= 400) with the bias table required", N); return LAVT_ERR_INVALID; }
    const int R = (2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1);
    const int centre = ((wd - 1) * (2 * wh - 1) + (wh - 1)) * (2 * ww - 1) + (ww - 1);
    dim3 grid(nwin * heads);
#define LAVT_FWD(NT_)                                                                                                                        \
    do {                                                                                                                                     \
        constexpr int NP = ((NT_ + 1) / 2) * 32, WV = NT_ > 10 ? 8 : 4;          /* 25 query tiles: two waves per SIMD */                      \
        const size_t lds = (size_t)3 * NP * F_LD * 2 + (size_t)NP * 4 + NP + (size_t)(R + centre + 1) * 4 + 16;                               \
        static size_t reserved = 0;                                                                                                          \
        if (lds > 65536 && lds > reserved) {                                                                                                 \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wattn_fwd_mfma<NT_, WV, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess || \
                hipFuncSetAttribute(reinterpret_cast<const void*>(&wattn_fwd_mfma<NT_, WV, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
                lavt_set_error("lavt_window_attn_fwd(mfma): cannot reserve %zu bytes of LDS", lds);                                          \
                return LAVT_ERR_LAUNCH;                                                                                                      \
            }                                                                                                                                \
            reserved = lds;                                                                                                                  \
        }                                                                                                                                    \
        if (region) hipLaunchKernelGGL((wattn_fwd_mfma<NT_, WV, true>), grid, dim3(WV * 64), lds, st, (const bf16*)qkv, table, region, nw_img, (bf16*)out, lse, wd, wh, \
                                       ww, nwin, N, heads, scale);                                                                           \
        else hipLaunchKernelGGL((wattn_fwd_mfma<NT_, WV, false>), grid, dim3(WV * 64), lds, st, (const bf16*)qkv, table, region, nw_img, (bf16*)out, lse, wd, wh, \
                                ww, nwin, N, heads, scale);                                                                                  \
    } while (0)
    if (N <= 64) LAVT_FWD(4);
    else if (N <= 144) LAVT_FWD(9);
    else if (N <= 160) LAVT_FWD(10);
    else LAVT_FWD(25);
#undef LAVT_FWD
    LAVT_CHECK_LAUNCH("lavt_window_attn_fwd(mfma)");
    return LAVT_OK;
}

static void dtable_geometry(int nwin, int N, int heads, int* rpb, int* wgroups, int* wpg) {
    // rows per binning block: <= 512 (row chunk, head) blocks per window group -- as riders of an attention-backward launch they are <= 256 workgroups of two
    // blocks, ONE round in the second workgroup slot of every CU (with 576 blocks = 288 riders the last 32 queued behind the others: round 6)
    *rpb = cdiv(N * heads, 512);
    if (*rpb < 4) *rpb = 4;
    const int g = cdiv(nwin, 32);                             // <= 32 windows summed per workgroup
    *wpg = cdiv(nwin, g);
    *wgroups = cdiv(nwin, *wpg);
}
int64_t lavt_window_attn_bwd_ws_mfma(int nwin, int N, int heads, int bias_ld, int wd, int wh, int ww) {
    int rpb, wgroups, wpg;
    dtable_geometry(nwin, N, heads, &rpb, &wgroups, &wpg);
    const int64_t R = (int64_t)(2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1);
    return (int64_t)nwin * heads * N * bias_ld + (int64_t)wgroups * cdiv(N, rpb) * heads * R;
}

// several layers' per-workgroup table histograms -> their table gradients in ONE launch (deferred form): desc[s] = {part, pieces, heads, R, dtable}
// COMPACT (round 5): blockIdx.y runs over the (layer, head) pairs that exist (the host passes their number) instead of (max heads) x (layers): Swin-B's
// 24 layers have 4 / 8 / 16 / 32 heads, so the (17, 32, 24) grid started 13 056 workgroups of which 6 664 found no head -- an 18 us launch for ~10 MB.
constexpr int DF_ENT = 128, DF_PL = 256 / DF_ENT;
template <bool COMPACT>
__global__ __launch_bounds__(256) void wattn_dtable_finish_multi(const int64_t* __restrict__ desc, int nsets) {
    // DF_ENT table entries x DF_PL piece lanes per workgroup (one thread per entry walking all pieces -- 252 for a stage-0 layer -- was a 38 us launch; 32 x 8: 6 392
    // workgroups for Swin-B, 23 us -- the launch is bound by their number)
    __shared__ float red[DF_PL][DF_ENT];
    int set = blockIdx.z, h = blockIdx.y;
    if constexpr (COMPACT) {
        // lane s reads the head count of set s (nsets <= 64); an inclusive prefix sum over the lanes; the first lane whose sum exceeds blockIdx.y names the set
        const int ln = threadIdx.x & 63;
        const int nh = ln < nsets ? (int)desc[(int64_t)ln * 5 + 2] : 0;
        int pre = nh;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(pre, o, 64);
            if (ln >= o) pre += up;
        }
        const unsigned long long hit = __ballot(pre > (int)blockIdx.y);
        if (!hit) return;
        set = __ffsll((long long)hit) - 1;
        h = (int)blockIdx.y - __shfl(pre - nh, set, 64);
    }
    const int64_t* d = desc + (int64_t)set * 5;
    const float* part = reinterpret_cast<const float*>(d[0]);
    const int pieces = (int)d[1], heads = (int)d[2], R = (int)d[3];
    float* dtable = reinterpret_cast<float*>(d[4]);
    const int el = threadIdx.x % DF_ENT, pl = threadIdx.x / DF_ENT;
    const int e = blockIdx.x * DF_ENT + el;
    if (h >= heads || blockIdx.x * DF_ENT >= R) return;          // uniform per workgroup
    const int64_t st = (int64_t)heads * R;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (e < R) {
        const float* q = part + (int64_t)h * R + e;
        int k = pl;
        for (; k + 15 * DF_PL < pieces; k += 16 * DF_PL) {          // (sixteen pieces in flight per thread: a stage-0 layer's 252 pieces were 8 dependent rounds of 4 -- the launch is latency)
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = q[(int64_t)(k + DF_PL * u) * st];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u & 3] += v[u];
        }
        for (; k + 3 * DF_PL < pieces; k += 4 * DF_PL)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += q[(int64_t)(k + DF_PL * u) * st];
        for (; k < pieces; k += DF_PL) acc[0] += q[(int64_t)k * st];
    }
    red[pl][el] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (pl == 0 && e < R) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < DF_PL; ++k) a += red[k][el];
        dtable[(int64_t)e * heads + h] += a;                 // one writer per entry: the zeroed gradient buffer
    }
}
int lavt_attn_dtable_finish_multi_impl(const int64_t* desc, int n, int max_R, int max_heads, int total_heads, hipStream_t st) {
    if (total_heads > 0 && n <= 64) hipLaunchKernelGGL(wattn_dtable_finish_multi<true>, dim3(cdiv(max_R, DF_ENT), total_heads, 1), dim3(256), 0, st, desc, n);
    else hipLaunchKernelGGL(wattn_dtable_finish_multi<false>, dim3(cdiv(max_R, DF_ENT), max_heads, n), dim3(256), 0, st, desc, n);
    LAVT_CHECK_LAUNCH("lavt_attn_dtable_finish_multi");
    return LAVT_OK;
}
int lavt_window_attn_bwd_pieces_mfma(int nwin, int N, int heads) {
    int rpb, wgroups, wpg;
    dtable_geometry(nwin, N, heads, &rpb, &wgroups, &wpg);
    return wgroups * cdiv(N, rpb);
}

static_assert(sizeof(DtableJob) == sizeof(lavt_dtable_job_t), "lavt_dtable_job_t mirrors DtableJob");
int lavt_attn_dtable_run_mfma(const lavt_dtable_job_t* jb, hipStream_t st) {
    const int R = (2 * jb->wd - 1) * (2 * jb->wh - 1) * (2 * jb->ww - 1);
    hipLaunchKernelGGL(wattn_dtable_kernel, dim3(jb->gx, jb->heads, jb->gz), dim3(256), (size_t)(R + jb->N) * 4, st, reinterpret_cast<const bf16*>(jb->slab), jb->part,
                       jb->slab_ld, jb->wd, jb->wh, jb->ww, jb->nwin, jb->N, jb->heads, jb->rows_per_block, jb->win_per_group);
    LAVT_CHECK_LAUNCH("lavt_attn_dtable_run");
    return LAVT_OK;
}
// prev: the binning job of an EARLIER launch to run inside this one (8-wave variants; launched on its own in front otherwise); mine != NULL: this
// launch's own binning is NOT launched but described in *mine (the caller hands it to the next launch or to lavt_attn_dtable_run)
int lavt_window_attn_bwd_mfma(const void* qkv, const float* table, const int8_t* region, int nw_img, const void* out, const void* dout,
                              const float* lse, void* dqkv, float* dtable, int bias_ld, float* ws, float* parts, int wd, int wh, int ww, int nwin, int N,
                              int heads, float scale, hipStream_t st, const lavt_dtable_job_t* prev, lavt_dtable_job_t* mine) {
    if (N > 400 || !table || !(dtable || parts) || !ws || bias_ld % 4) { lavt_set_error("lavt_window_attn_bwd(mfma): N=%d (<= 400), table, dtable and scratch required", N); return LAVT_ERR_INVALID; }
    const int R = (2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1);
    // >= 2 workgroups per CU when there is that much work
    // (round 6: one window per workgroup up to 2048 units -- 800 units as 400 two-window workgroups took 39.7 us against 36.7 as 800 + pieces, 1152 units 56.5
    // against 49.8: more workgroups than resident slots cost less than a second window behind the first.  LAVT_ATTN_WPB_UNITS = 768 restores the round-5 rule.)
    static const long wpb_units = getenv("LAVT_ATTN_WPB_UNITS") ? atol(getenv("LAVT_ATTN_WPB_UNITS")) : 2048;
    int wpb = (int)(((long)nwin * heads + wpb_units - 1) / wpb_units);
    if (wpb < 1) wpb = 1;
    const int chunks = cdiv(nwin, wpb);
    const int force_waves = lavt_tuning().attn_bwd_waves;
    const int waves = force_waves ? force_waves : 8;
    // riders only on the 8-wave variants that sit two per CU (N <= 160): a rider occupies a whole workgroup slot, and the 392-token video kernel (149 KB of
    // LDS, one workgroup per CU) would run them as an extra round
    const bool eight = !(N <= 64) && !(N <= 144 && waves != 8) && (N <= 160 || lavt_tuning().probe[3] != 0);
    DtableJob job{};
    int riders = 0;
    if (prev != nullptr) {
        const int Rp = (2 * prev->wd - 1) * (2 * prev->wh - 1) * (2 * prev->ww - 1);
        if (eight && 2 * (size_t)((Rp + prev->N) * 4 + 16) <= 32768) {
            memcpy(&job, prev, sizeof(job));
            riders = cdiv(prev->gx * prev->heads * prev->gz, 2);
        } else {
            const int rc = lavt_attn_dtable_run_mfma(prev, st);
            if (rc != LAVT_OK) return rc;
        }
    }
    // split the units beyond whole rounds of 256 (see the kernel): up to 128 of them, each into min(8, 256 / extra) pieces -- at most ONE piece per CU beside
    // its whole units.  (More pieces than CUs queue behind the two resident workgroups of a CU: 400 units = 256 + 144 x 3 pieces measured 26.9 us against
    // 21.0 unsplit, 480 units 28.5 against 23.1; 288 -> 16.4-16.7 against 20.3-20.7, 384 -> 19.7 against 20.9, 576 -> 29.6 against 30.3.)
    const int units = chunks * heads, extra = units % 256;
    int split_from = units, split_pieces = 1;
    if (wpb == 1 && units > 256 && extra > 0 && extra <= 128 && !lavt_tuning().attn_bwd_split_off) {
        split_pieces = 256 / extra < 8 ? 256 / extra : 8;
        split_from = units - extra;
    }
    const int attn_blocks = split_from + (units - split_from) * split_pieces;
    dim3 grid(attn_blocks + riders);
#define LAVT_BWD_K(NT_, WV_, RG_)                                                                                                           \
    do {                                                                                                                                     \
        const size_t lds = bwd_lds_bytes<NT_>(R);                                                                                            \
        if (bias_ld < 16 * NT_) { lavt_set_error("lavt_window_attn_bwd(mfma): bias_ld = %d < %d", bias_ld, 16 * NT_); return LAVT_ERR_INVALID; } \
        static size_t reserved = 0;                                                                                                          \
        if (lds > 65536 && lds > reserved) {                                                                                                 \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wattn_bwd_mfma<NT_, WV_, RG_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
                lavt_set_error("lavt_window_attn_bwd(mfma): cannot reserve %zu bytes of LDS", lds);                                          \
                return LAVT_ERR_LAUNCH;                                                                                                      \
            }                                                                                                                                \
            reserved = lds;                                                                                                                  \
        }                                                                                                                                    \
        hipLaunchKernelGGL((wattn_bwd_mfma<NT_, WV_, RG_>), grid, dim3(WV_ * 64), lds, st, (const bf16*)qkv, table, region, nw_img,          \
                           (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, reinterpret_cast<bf16*>(ws), bias_ld, wd, wh, ww, nwin, N, heads, scale, wpb, attn_blocks, job, \
                           split_from, split_pieces);                                                                                        \
    } while (0)
#define LAVT_BWD(NT_, WV_)                                                                                                                   \
    do {                                                                                                                                     \
        if (region) LAVT_BWD_K(NT_, WV_, true);                                                                                              \
        else LAVT_BWD_K(NT_, WV_, false);                                                                                                    \
    } while (0)
    // 8 waves share one window-head, capped at 128 VGPRs so two workgroups (16 waves) sit on a CU: 5-9% faster than 4 waves x 2 at every
    // stage shape of Swin-B w12 @480 (measured, tools/attn_bench2.py).  LAVT_ATTN_BWD_WAVES=4 keeps the 4-wave variant reachable.
    if (N <= 64) LAVT_BWD(4, 4);
    else if (N <= 144) { if (waves == 8) LAVT_BWD(9, 8); else LAVT_BWD(9, 4); }
    else if (N <= 160) LAVT_BWD(10, 8);
    else LAVT_BWD(25, 8);                  // Video-Swin 8x7x7 windows: 149 KB of LDS, one workgroup per CU
#undef LAVT_BWD
#undef LAVT_BWD_K
    LAVT_CHECK_LAUNCH("lavt_window_attn_bwd(mfma)");
    int rpb, wgroups, wpg;
    dtable_geometry(nwin, N, heads, &rpb, &wgroups, &wpg);
    // per-workgroup histograms [wgroups * chunks][heads][R]: after the slabs, or -- deferred form -- in the caller's persistent buffer, to be
    // summed into the table gradients of all layers by one lavt_attn_dtable_finish_multi launch at the end of backward
    float* part = parts ? parts : ws + (int64_t)nwin * heads * N * bias_ld;
    if (mine != nullptr) {
        mine->slab = ws; mine->part = part; mine->slab_ld = bias_ld; mine->wd = wd; mine->wh = wh; mine->ww = ww; mine->nwin = nwin; mine->N = N; mine->heads = heads;
        mine->rows_per_block = rpb; mine->win_per_group = wpg; mine->gx = cdiv(N, rpb); mine->gz = wgroups;
        return LAVT_OK;
    }
    hipLaunchKernelGGL(wattn_dtable_kernel, dim3(cdiv(N, rpb), heads, wgroups), dim3(256), (size_t)(R + N) * 4, st, reinterpret_cast<const bf16*>(ws), part, bias_ld, wd, wh, ww, nwin,
                       N, heads, rpb, wpg);
    const int pieces = wgroups * cdiv(N, rpb), per_z = 16;
    if (!parts) hipLaunchKernelGGL(wattn_dtable_finish, dim3(cdiv(R, 256), heads, cdiv(pieces, per_z)), dim3(256), 0, st, part, dtable, pieces, per_z, heads, R);
    LAVT_CHECK_LAUNCH("lavt_window_attn_bwd(table gradient)");
    return LAVT_OK;
}
