// Gather-GEMM, NT family, software-pipelined K loop (bf16, and e4m3 operands for BASELINE configs[4]: the F8 variant below): the long-K problems of
// the decoder (3x3 convolutions forward and data gradient as implicit GEMMs, reference lib/mask_predictor.py:60-97) and every other problem that takes
// the 256x256 tile.
//
// Why a third K loop.  gemm_v2.hip leaves the order of fragment reads, MFMAs and DMA issue to hipcc, which schedules for register pressure: per K
// tile a wave issues a few ds_read_b128, waits `lgkmcnt(0)`, runs 4-8 MFMAs, reads again, waits again -- four full LDS round trips per K tile with
// nothing in flight -- and the ~130 instructions that issue the next tile's DMA run as one block behind the barrier, in every wave at the same
// time, with the matrix pipe idle.  Its 256x256 form needed 16 waves (128 registers per lane) and spilled 60-78 registers in the epilogue.
//
// This kernel:
//   * 8 waves (2 per SIMD, 256 registers per lane) as 2 (M) x 4 (N) -- wave tiles of 128 x 64 (256x256 tile) or 64 x 32 (128x128 tile): the
//     256x256 tile moves 192 KB of fragments per K tile from LDS instead of 256 KB; no spills, no scratch;
//   * fragment reads are issued from inline asm one MFMA GROUP ahead (double-buffered registers: A halves x k-steps for the large tile, k-steps for
//     the small one) and waited for with counted `s_waitcnt lgkmcnt(n)`: a group's operands land under the previous group's MFMAs;
//   * ONE barrier per K tile, placed in front of the LAST group: by then every read of the tile has completed (lgkmcnt(0)), so the stage can be
//     refilled at once (DMA of tile kt + STAGES goes into the stage tile kt occupied: a full ring of STAGES tiles in flight or resident) and the
//     first fragments of tile kt + 1 are requested;
//   * the DMA issue is spread over the last group: two MFMAs, one DMA instruction with its address arithmetic, L times.  The K loop is ONE basic
//     block (tiles beyond K are issued against the zero page, cursor updates are selects), so nothing the compiler does can regroup it;
//   * convolutions walk K with the TAPS FASTEST (channel block outer): the 9 (27) shifted reads of a 64-channel block of the activations follow each
//     other, so 8 of 9 come from the XCD's L2; the concat source switches once per launch instead of being selected per DMA instruction.
// Measured (tools/conv_small_probe.py, hipGraph-timed, against the gemm_v2 form): 2x120x120 512->512 forward 134.7 -> 120.5 us (0.45 of the bf16 peak),
// 640->512 158.6 -> 138.3 (0.49), data gradient 127 -> 115 (0.47); 2x60x60 512->512 56.6 -> 42.0, 768->512 80.5 -> 59.7, data gradient 57 -> 44; plain
// 7200x512x4608 47.4 -> 37.0.  What bounds it now (ablation builds, same tool): the DMA stream alone takes 85 us on the
// 125 us convolution (every CU ingests ~52 GB/s whatever the tile size or ring depth: 12 TB/s over the chip, L2-hit traffic), MFMA + fragment
// reads alone 99 us (the chip holds ~1.7 GHz under this load: 0.57 of the nominal peak is what an MFMA-only loop reaches) -- the two overlap to 125.
// DMA geometry (lane-linear LDS images, chunk / slot swizzles applied to the per-lane SOURCE address) and the epilogue are those of gemm_v2.hip; the
// tap-walking mode addresses its operands through buffer descriptors (zeros for halo / padding = an offset beyond the descriptor's range).
#include "gemm_v2_helpers.h"

namespace {
// workgroups of the 128x128 tile from which the 2-stage ring (two workgroups per CU) replaces the 4-stage one (LAVT_PROBE slot 7 >= 100 overrides: experiments)
static inline long s2_min128() { const int v = lavt_tuning().probe[7]; return v >= 100 ? v : 257; }


typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(8 * sizeof(int)))) int i32x8;

// Buffer-descriptor LDS-DMA (buffer_load_dwordx4 ... offen lds).  The resource type and its builtins exist in the device pass only; the host pass, which
// still parses the kernel body to emit its launch stub, sees placeholders.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t buf_make(const void* base) {          // 2 GB window, raw (stride 0) addressing, offsets beyond it read zero
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ buf_rsrc_t buf_make_n(const void* base, unsigned bytes) {          // offsets >= bytes read zero
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void buf_dma16(buf_rsrc_t rs, void* lds_dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_dst, 16, voff, soff, 0, 0);
}
#else
struct buf_rsrc_t { int unused; };
__host__ __device__ inline buf_rsrc_t buf_make(const void*) { return buf_rsrc_t{0}; }
__host__ __device__ inline buf_rsrc_t buf_make_n(const void*, unsigned) { return buf_rsrc_t{0}; }
__host__ __device__ inline void buf_dma16(buf_rsrc_t, void*, unsigned, unsigned) {}
#endif

// N ds_read_b128 at addr + BASE + i * STRIDE, issued only (no wait): outputs are early-clobber so that no destination aliases the address
template <int N, int BASE, int STRIDE> __device__ __forceinline__ void pipe_issue(u32x4 (&f)[N], unsigned addr) {
    static_assert(N == 2 || N == 4, "N");
    if constexpr (N == 4)
        asm volatile("ds_read_b128 %0, %4 offset:%c5\n\tds_read_b128 %1, %4 offset:%c5+%c6\n\tds_read_b128 %2, %4 offset:%c5+%c6*2\n\tds_read_b128 %3, %4 offset:%c5+%c6*3"
                     : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]) : "v"(addr), "n"(BASE), "n"(STRIDE) : "memory");
    else
        asm volatile("ds_read_b128 %0, %2 offset:%c3\n\tds_read_b128 %1, %2 offset:%c3+%c4" : "=&v"(f[0]), "=&v"(f[1]) : "v"(addr), "n"(BASE), "n"(STRIDE) : "memory");
}
// transposing reads of NF fragments of a k-major tile (one address register per fragment: the slot swizzle differs per lane), k-step offset KOFF
template <int NF, int HO, int KOFF> __device__ __forceinline__ void pipe_issue_tr(const unsigned (&a)[NF], u64 (&l)[NF], u64 (&h)[NF]) {
    static_assert(NF == 2 || NF == 4, "NF");
    if constexpr (NF == 4)
        asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%c13\n\tds_read_b64_tr_b16 %1, %8 offset:%c13+%c12\n\t"
                     "ds_read_b64_tr_b16 %2, %9 offset:%c13\n\tds_read_b64_tr_b16 %3, %9 offset:%c13+%c12\n\t"
                     "ds_read_b64_tr_b16 %4, %10 offset:%c13\n\tds_read_b64_tr_b16 %5, %10 offset:%c13+%c12\n\t"
                     "ds_read_b64_tr_b16 %6, %11 offset:%c13\n\tds_read_b64_tr_b16 %7, %11 offset:%c13+%c12"
                     : "=&v"(l[0]), "=&v"(h[0]), "=&v"(l[1]), "=&v"(h[1]), "=&v"(l[2]), "=&v"(h[2]), "=&v"(l[3]), "=&v"(h[3])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "n"(HO), "n"(KOFF) : "memory");
    else
        asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%c7\n\tds_read_b64_tr_b16 %1, %4 offset:%c7+%c6\n\t"
                     "ds_read_b64_tr_b16 %2, %5 offset:%c7\n\tds_read_b64_tr_b16 %3, %5 offset:%c7+%c6"
                     : "=&v"(l[0]), "=&v"(h[0]), "=&v"(l[1]), "=&v"(h[1]) : "v"(a[0]), "v"(a[1]), "n"(HO), "n"(KOFF) : "memory");
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): indices that stay compile-time constants inside a lambda (a `#pragma unroll` loop over a
// lambda's int parameter left the per-instruction pointer arrays dynamically indexed in the prologue: 64 bytes of scratch per lane)
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
__device__ __forceinline__ i32x8 pipe8_pack(u32x4 lo, u32x4 hi) {          // the 32 bytes of a row an fp8 MFMA takes from a lane: chunks g and g + 4
    return i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
}
template <int CNT> __device__ __forceinline__ void pipe_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CNT) : "memory"); }
// ties registers to the wait in front of it: consumers of `f` cannot be scheduled above this (empty) statement, which stays behind the wait
template <int N> __device__ __forceinline__ void pipe_tie(u32x4 (&f)[N]) {
    if constexpr (N == 4) asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
    else asm volatile("" : "+v"(f[0]), "+v"(f[1]));
}
template <int N> __device__ __forceinline__ void pipe_tie(u64 (&l)[N], u64 (&h)[N]) {
    if constexpr (N == 4) asm volatile("" : "+v"(l[0]), "+v"(h[0]), "+v"(l[1]), "+v"(h[1]), "+v"(l[2]), "+v"(h[2]), "+v"(l[3]), "+v"(h[3]));
    else asm volatile("" : "+v"(l[0]), "+v"(h[0]), "+v"(l[1]), "+v"(h[1]));
}

// B fragments of one k-step: k-contiguous tiles are read like A (ds_read_b128), k-major tiles by two transposing reads per fragment
template <bool BKM, int NI> struct BFrags {
    u32x4 kc[BKM ? 1 : NI];
    u64 lo[BKM ? NI : 1], hi[BKM ? NI : 1];
    __device__ __forceinline__ bf16x8 get(int j) const {
        if constexpr (BKM) return frag_from(lo[j], hi[j]);
        else return __builtin_bit_cast(bf16x8, kc[j]);
    }
    __device__ __forceinline__ void tie() {
        if constexpr (BKM) pipe_tie<NI>(lo, hi);
        else pipe_tie<NI>(kc);
    }
};

// sum over the 16 lanes of a DPP row, left in every lane: four v_add_f32 with a DPP operand (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror --
// the mirrors act as xor 4 / xor 8 once the lanes of a quad / half row agree).  (As __shfl_xor = ds_bpermute the 128 exchanges of the statistics
// epilogue cost 4 us on a 123 us convolution.)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// Column statistics of a wave's [MI * 16 rows] x [NI * 16 columns] accumulator tile (lavt_gemm_nt_t.colstats): per column the sum over the
// wave's valid rows and the second moment about THEIR mean -- both passes run on registers, so no E[x^2] - E[x]^2 -- stored as block `blk` of
// the partial table.  C^T accumulator layout: a lane holds 4 consecutive columns of row (lane % 16) of every fragment, the 16 rows of a fragment
// sit in the 16 lanes of a DPP row: 8 (4) in-register adds + 4 DPP adds per column.
template <int MI, int NI>
__device__ __forceinline__ void pipe_colstats(const lavt_gemm_nt_t& p, const f32x4 (&acc)[MI][NI], int m_base, int n_base, int lane, int blk) {
    const int l15 = lane & 15, g = lane >> 4;
    const int nvalid = min(max(p.M - m_base, 0), MI * 16);
    const float inv = nvalid > 0 ? 1.f / (float)nvalid : 0.f;
    float* out = p.colstats + (int64_t)blk * 2 * p.N;
    bool ok[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) ok[i] = m_base + i * 16 + l15 < p.M;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] += ok[i] ? p.alpha * acc[i][j][r] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] = row16_sum(s[r]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = p.alpha * acc[i][j][r] - s[r] * inv;
                q[r] += ok[i] ? d * d : 0.f;
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) q[r] = row16_sum(q[r]);
        const int n = n_base + j * 16 + 4 * g;
        if (l15 == 0 && n + 3 < p.N) {                       // (N % 4 == 0: lavt_gemm_nt_colstats_plan)
            *reinterpret_cast<float4*>(out + n) = make_float4(s[0], s[1], s[2], s[3]);
            *reinterpret_cast<float4*>(out + p.N + n) = make_float4(q[0], q[1], q[2], q[3]);
        }
    }
}

// F8 (BASELINE.json configs[4]): e4m3 operands, k-contiguous.  The operand tiles are the same BYTES as the bf16 ones (rows of 128 B = 128 elements),
// so the DMA geometry, the ring and the fragment read addresses do not change; the two ds_read_b128 a lane issues for the two bf16 k-steps of a row
// (chunks g and g + 4) are the 32 bytes ONE v_mfma_scale_f32_16x16x128_f8f6f4 takes from it.  The MFMA groups are therefore cut over A fragments (pairs of
// fragments x every B fragment x both k-steps) instead of over k-steps: same register budget, same counted waits (see the F8 branch of the K loop).
template <int BM, int BN, bool BKM, int STAGES, int MODE, int LEAN, bool F8 = false>
__global__ __launch_bounds__(512) void gemm_nt_pipe_kernel(const lavt_gemm_nt_t p) {
    // MODE 3 (round 5): the tap-walking issue for a channel count that is not a multiple of 64 (Swin-T's conv1_2: 384 + 96 = 480 input channels,
    // reference lib/mask_predictor.py:30-38): the LAST 64-channel block of the reduction is partial, its missing 16-byte chunks are offsets beyond the
    // descriptors' range (zeros) in both operands -- instead of the general decode (MODE 0: per-instruction divisions, 0.19 of peak on that convolution)
    constexpr bool SIMPLE = MODE == 1, CONVFAST = MODE == 2 || MODE == 3, CTAIL = MODE == 3;
    using T = typename std::conditional<F8, unsigned char, bf16>::type;
    constexpr int WAVES = 8, BK = F8 ? 128 : 64, EPC = F8 ? 16 : 8, ES = (int)sizeof(T);
    static_assert(!(F8 && BKM), "fp8: k-contiguous operands");
    constexpr int WAVES_N = 4;
    constexpr int A_INSTR = BM / (8 * WAVES), B_INSTR = BN / (8 * WAVES);      // DMA instructions per wave per K tile (8 rows x 8 chunks each)
    constexpr int B_CH = BN / EPC;
    constexpr int L = A_INSTR + B_INSTR;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int WM = BM / 2, WN = BN / WAVES_N, MI = WM / 16, NI = WN / 16;
    constexpr bool SPLITA = MI == 8;                                         // large tile: MFMA groups of (half the A fragments) x (all B fragments) of a k-step
    constexpr int MIH = SPLITA ? MI / 2 : MI;
    static_assert(MIH == 4 && (NI == 2 || NI == 4), "wave tiles of 128 x 64 or 64 x 32");
    // lgkmcnt is a 4-bit counter: the largest batch that stays in flight behind a counted wait
    constexpr int NB = BKM ? 2 * NI : NI;                                    // LDS instructions of one k-step of B fragments
    static_assert(MIH + NB <= 15, "counted LDS waits");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // (scalar: LDS destinations of the DMA go through M0)
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile_id = xcd_tile_id(blockIdx.x, gridDim.x);
    const int tile_m = tile_id / tiles_n, tile_n = tile_id % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int bz = blockIdx.y;

    const T* A = reinterpret_cast<const T*>(p.A) + (int64_t)bz * p.strideA;
    const T* A2 = reinterpret_cast<const T*>(p.A2);
    const T* B = reinterpret_cast<const T*>(p.B) + (int64_t)bz * p.strideB;
    const T* Z = reinterpret_cast<const T*>(p.zeros);
    const bool conv = p.conv_kc > 0;
    const ConvGeom cg = conv_geom(p);

    // ---- per-lane DMA geometry (as gemm_v2.hip) --------------------------------------------------------------------
    const int cl = (lane & 7) ^ (lane >> 3);
    int a_src[A_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int m = m0 + (wave * A_INSTR + i) * 8 + (lane >> 3);
        int src = -1;
        if (m < p.M) src = p.a_rowmap ? p.a_rowmap[m] : m;
        a_src[i] = src;
    }
    int b_row[B_INSTR], b_col[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        if constexpr (!BKM) {
            const int n = n0 + (wave * B_INSTR + i) * 8 + (lane >> 3);
            b_row[i] = n < p.N ? n : -1;
            b_col[i] = 0;
        } else {
            const int q = (wave * B_INSTR + i) * 64 + lane;
            const int kr = q / B_CH, cc = (q - kr * B_CH) ^ tn_swz<B_CH>(kr);
            const int n = n0 + cc * EPC;
            b_row[i] = n < p.N ? kr : -1;
            b_col[i] = n;
        }
    }
    const T* a_ptr[A_INSTR];
    const T* b_ptr[B_INSTR];
    int64_t b_step[B_INSTR];
    int a_step[A_INSTR];
    if constexpr (SIMPLE) {
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            a_ptr[i] = a_src[i] >= 0 ? A + (int64_t)a_src[i] * p.lda + cl * EPC : Z;
            a_step[i] = a_src[i] >= 0 ? BK : 0;
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            if constexpr (!BKM) {
                b_ptr[i] = b_row[i] >= 0 ? B + (int64_t)b_row[i] * p.ldb + cl * EPC : Z;
                b_step[i] = b_row[i] >= 0 ? BK : 0;
            } else {
                b_ptr[i] = b_row[i] >= 0 ? B + (int64_t)b_row[i] * p.ldb + b_col[i] : Z;
                b_step[i] = b_row[i] >= 0 ? (int64_t)BK * p.ldb : 0;
            }
        }
    }
    // CONVFAST: a bit per tap "this row's neighbour lies inside the volume" and the row's pointers into the two sources at the lane's channel chunk
    // The operands are addressed through buffer descriptors (buffer_load ... lds): a lane's offset is a 32-bit register that never changes, the (tap,
    // channel block) of a K tile is the scalar offset of the instruction, and a chunk that must read zero -- halo, padding rows, columns beyond N --
    // is an offset beyond the descriptor's range (the hardware writes zeros).  Against per-lane 64-bit pointers + a zero page this is 3 vector
    // instructions per A instruction (mask test + select) and none per B instruction instead of ~7 / ~4.  lavt_gemm_nt_pipe_tile sends operands of
    // 2 GB or more to the general mode.
    constexpr unsigned BUF_OOB = 0x80000000u;
    unsigned a_vmask[A_INSTR], a_vo1[A_INSTR], a_vo2[A_INSTR], b_vo[B_INSTR];
    int c_kin = 0, c_tap = 0, c_tap0 = 0, c_ntap = cg.taps;
    if constexpr (CONVFAST) {
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            unsigned m = 0;
            if (a_src[i] >= 0) {
                int z, y, x;
                conv_coords(cg, a_src[i], z, y, x);
                for (int t = 0; t < cg.taps; ++t) {
                    int dz, dy, dx;
                    conv_tap(cg, t, dz, dy, dx);
                    if (p.conv_flip) { dz = -dz; dy = -dy; dx = -dx; }
                    const bool ok = ((unsigned)(z + dz) < (unsigned)cg.d) & ((unsigned)(y + dy) < (unsigned)cg.h) & ((unsigned)(x + dx) < (unsigned)cg.w);
                    m |= (ok ? 1u : 0u) << t;
                }
            }
            a_vmask[i] = m;
            // byte offset of the lane's chunk in either source, or out of the descriptor's range (a row that does not exist: the load then writes zeros)
            a_vo1[i] = a_src[i] >= 0 ? (unsigned)(((int64_t)a_src[i] * p.lda + cl * EPC) * ES) : BUF_OOB;
            a_vo2[i] = (A2 && a_src[i] >= 0) ? (unsigned)(((int64_t)a_src[i] * p.lda2 + cl * EPC) * ES) : BUF_OOB;
        }
        if (p.conv_tap_split > 0) { c_tap = bz * p.conv_tap_split; c_ntap = p.conv_tap_split; }
        if (p.conv_kc_split > 0) c_kin = bz * p.conv_kc_split;          // split over channel blocks: this entry's first channel
        c_tap0 = c_tap;
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            if constexpr (!BKM) b_vo[i] = b_row[i] >= 0 ? (unsigned)(((int64_t)b_row[i] * p.ldb + cl * EPC) * ES) : BUF_OOB;
            else b_vo[i] = b_row[i] >= 0 ? (unsigned)(((int64_t)b_row[i] * p.ldb + b_col[i]) * ES) : BUF_OOB;
        }
    }
    const bool has_a2 = p.A2 != nullptr;
    const int64_t lda1 = p.lda, lda2 = p.lda2;
    const int a_split = p.a_split, conv_kc = p.conv_kc, flip = p.conv_flip;

    // ---- DMA issue of one K tile, in three parts so that the L instructions can be spread between MFMAs ---------------------------------
    // issue_begin: the wave-uniform part (scalar unit); issue_one(idx): DMA instruction idx (A rows first, then B); issue_end: cursor advance.
    // `past` = the tile lies beyond K (the loop issues STAGES tiles ahead unconditionally, so that it has no branches and the counted waits
    // are constants): such a tile reads the zero page / the start of the weight rows.
    // the concat source the channel blocks currently come from: with taps fastest the channel block only grows, so the rows switch from the first
    // source to the second ONCE (issue_end) instead of being selected per DMA instruction
    unsigned a_vo[A_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) a_vo[i] = CONVFAST ? a_vo1[i] : 0u;
    int64_t cur_lda = lda1;
    int cur_base = 0;
    // descriptor bases sit `bias` bytes below the tensors so that the scalar offset bias + (tap shift) + channel offset is never negative
    const int max_rows = ((cg.kd >> 1) * cg.h + (cg.kh >> 1)) * cg.w + (cg.kw >> 1);
    const unsigned bias1 = (unsigned)((int64_t)max_rows * lda1 * ES), bias2 = (unsigned)((int64_t)max_rows * lda2 * ES);
    buf_rsrc_t rs_a = buf_make(reinterpret_cast<const char*>(A) - bias1);
    const buf_rsrc_t rs_a2 = buf_make(reinterpret_cast<const char*>(A2 ? A2 : A) - bias2);
    // (CTAIL: the weight descriptor ends with the weights -- the partial block of the last row's last tap would otherwise read past the allocation)
    const buf_rsrc_t rs_b = CTAIL ? buf_make_n(B, (unsigned)((int64_t)p.N * p.ldb * ES)) : buf_make(B);
    unsigned cur_bias = bias1;
    if (CONVFAST && has_a2 && c_kin >= a_split) {          // (a channel-split entry that starts inside the second source)
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) a_vo[i] = a_vo2[i];
        cur_lda = lda2; cur_base = a_split; rs_a = rs_a2; cur_bias = bias2;
    }

    // ---- DMA issue of one K tile, in three parts so that the L instructions can be spread between MFMAs ---------------------------------
    // issue_begin: the wave-uniform part (scalar unit); issue_one(idx): DMA instruction idx (A rows first, then B); issue_end: cursor advance.
    // `past` = the tile lies beyond K (the loop issues STAGES tiles ahead unconditionally, so that it has no branches and the counted waits
    // are constants): such a tile reads the zero page / the start of the weight rows.
    unsigned u_soff_a = 0, u_soff_b = 0;
    bool u_past = false;
    unsigned u_bit = 0;
    int u_kt = 0;
    int u_nch = 8;          // CTAIL: 16-byte chunks of the current channel block that hold channels (8 everywhere but in the last, partial block)
    // Row offset of every tap, ((dz h + dy) w + dx) rows of the current source in BYTES, as a table over the lanes (lane t = tap t, taps <= 32): the K loop
    // fetches its tap's entry with one v_readlane.  (Kept as running (dz, dy, dx) + a 64-bit multiply per K tile the tap walk was ~70 scalar
    // instructions per wave per K tile -- 5.5 x the plain GEMM's, on the ONE scalar unit a CU's eight waves share: rocprofv3 SQ_INSTS_SALU 11.8 M vs
    // 2.2 M per launch at 2 x 60 x 60, where the convolution ran 33 % longer than a plain GEMM of its size.)
    int tap_sh = 0;
    auto tap_table = [&](int64_t ld) {
        int dz, dy, dx;
        conv_tap(cg, lane < cg.taps ? lane : 0, dz, dy, dx);
        tap_sh = (int)((int64_t)(((dz * cg.h + dy) * cg.w + dx) * (flip ? -1 : 1)) * ld * ES);
    };
    if constexpr (CONVFAST) tap_table(cur_lda);
    // running position in the weight operand (advanced in issue_end), in bytes from B: k-major B -- (tap, channel block) rows; k-contiguous B -- the columns
    // of (tap, channel block) inside the [Cout][taps][Cin] rows
    unsigned c_bbo = (unsigned)(((int64_t)c_kin * p.ldb + (int64_t)c_tap0 * p.b_tap_stride) * ES);
    const unsigned bb_tap = (unsigned)(p.b_tap_stride * ES), bb_wrap = (unsigned)(((int64_t)BK * p.ldb - (int64_t)(c_ntap - 1) * p.b_tap_stride) * ES);
    int c_boff = c_kin;
    const int boff_wrap = BK - (c_ntap - 1) * conv_kc;
    auto issue_begin = [&](int kt, bool past) {
        u_past = past; u_kt = kt;
        if constexpr (CONVFAST) {
            u_soff_a = cur_bias + (unsigned)(__builtin_amdgcn_readlane(tap_sh, c_tap) + (c_kin - cur_base) * ES);
            u_bit = past ? 0u : 1u << c_tap;                                          // (a tile beyond K: every A lane out of range, B from the start of the rows)
            u_soff_b = past ? 0u : (BKM ? c_bbo : (unsigned)(c_boff * ES));
            if constexpr (CTAIL) u_nch = min(8, (conv_kc - c_kin) >> 3);
        }
    };
    auto issue_one = [&](auto idx_c, char* sbase) {
        constexpr int idx = decltype(idx_c)::value;
        if constexpr (idx < A_INSTR) {
            constexpr int i = idx;
            char* dst = sbase + (wave * A_INSTR + i) * 1024;
            if constexpr (SIMPLE) {
                dma16(u_past ? Z : a_ptr[i], dst);
                a_ptr[i] += a_step[i];
            } else if constexpr (CONVFAST) {
                if constexpr (CTAIL) buf_dma16(rs_a, dst, ((a_vmask[i] & u_bit) && cl < u_nch) ? a_vo[i] : BUF_OOB, u_soff_a);
                else buf_dma16(rs_a, dst, (a_vmask[i] & u_bit) ? a_vo[i] : BUF_OOB, u_soff_a);
            } else {
                const int k = u_kt * BK + cl * EPC;
                int kin = k, dz = 0, dy = 0, dx = 0;
                if (conv) {
                    const int tap = k / p.conv_kc;
                    kin = k - tap * p.conv_kc;
                    conv_tap(cg, tap, dz, dy, dx);
                    if (p.conv_flip) { dz = -dz; dy = -dy; dx = -dx; }
                }
                const bool second = (p.A2 != nullptr) && kin >= p.a_split;
                const T* base = second ? A2 : A;
                const int64_t ld = second ? p.lda2 : p.lda;
                const int kk = second ? kin - p.a_split : kin;
                int src = a_src[i];
                if (conv) src = conv_nbr(cg, src, dz, dy, dx);
                dma16((src >= 0 && k < p.K) ? base + (int64_t)src * ld + kk : Z, dst);
            }
        } else {
            constexpr int i = idx - A_INSTR;
            char* dst = sbase + A_BYTES + (wave * B_INSTR + i) * 1024;
            if constexpr (SIMPLE) {
                dma16(u_past ? Z : b_ptr[i], dst);
                b_ptr[i] += b_step[i];
            } else if constexpr (CONVFAST) {
                if constexpr (CTAIL && !BKM) buf_dma16(rs_b, dst, cl < u_nch ? b_vo[i] : BUF_OOB, u_soff_b);
                else buf_dma16(rs_b, dst, b_vo[i], u_soff_b);
            } else {
                const int k = u_kt * BK + cl * EPC;
                const T* g = Z;
                if constexpr (!BKM) {
                    if (b_row[i] >= 0 && k < p.K) g = B + (int64_t)b_row[i] * p.ldb + k;
                } else {
                    const int kb = u_kt * BK + b_row[i];
                    if (b_row[i] >= 0 && kb < p.K) {
                        int64_t off;
                        if (conv) { const int t2 = kb / p.conv_kc; off = (int64_t)(kb - t2 * p.conv_kc) * p.ldb + (int64_t)t2 * p.b_tap_stride; }
                        else off = (int64_t)kb * p.ldb;
                        g = B + off + b_col[i];
                    }
                }
                dma16(g, dst);
            }
        }
    };
    auto issue_end = [&]() {
        if constexpr (CONVFAST) {
            // taps fastest: the nine (27) shifted reads of a 64-channel block follow each other, so all but the first are served by the XCD's L2.
            // Selects, no branches: the whole K tile is one basic block.
            ++c_tap;
            const bool wt = c_tap == c_tap0 + c_ntap;
            c_tap = wt ? c_tap0 : c_tap;
            c_kin += wt ? BK : 0;
            if (has_a2 && wt && c_kin == a_split) {           // (uniform, taken once per launch)
#pragma unroll
                for (int i = 0; i < A_INSTR; ++i) a_vo[i] = a_vo2[i];
                cur_lda = lda2; cur_base = a_split; rs_a = rs_a2; cur_bias = bias2;
                tap_table(lda2);
            }
            c_bbo += wt ? bb_wrap : bb_tap;
            c_boff += wt ? boff_wrap : conv_kc;
        }
    };

    // ---- fragment read addresses (bytes inside a stage; + the stage base at the read) -----------------------------------
    const int l15 = lane & 15, g4 = lane >> 4;
    const unsigned lds0 = lds_addr(smem);
    unsigned a_rd[2], b_rd[2];                    // k-contiguous tiles: 128-byte rows, chunk index XOR (row & 7); k-step 1 flips chunk bit 2
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_rd[ks] = lds0 + (unsigned)((wm * WM + l15) * 128 + (((ks * 4 + g4) ^ (l15 & 7)) << 4));
        b_rd[ks] = lds0 + (unsigned)(A_BYTES + (wn * WN + l15) * 128 + (((ks * 4 + g4) ^ (l15 & 7)) << 4));
    }
    unsigned b_tr[NI];                            // k-major tiles: swizzled 32-byte slot of column wn*WN + 16 j, row 8 (lane / 16) + (lane % 16) / 4
    {
        const int row_off = 8 * g4 + (l15 >> 2), sw = tn_swz<B_CH>(row_off);
#pragma unroll
        for (int j = 0; j < NI; ++j)
            b_tr[j] = lds0 + (unsigned)(A_BYTES + (row_off * BN + ((((wn * WN) / 8 + 2 * j) ^ sw) + ((lane & 3) >> 1)) * 8 + (lane & 1) * 4) * 2);
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 fa[2][MIH];
    BFrags<BKM, NI> fb[2];

    // reads of k-step KS of B into buffer `dst`, of A fragments [H * MIH, (H + 1) * MIH) of k-step KS into buffer `dst`
#define PIPE_LOAD_B(KS, dst, soff)                                                                                                   \
    do {                                                                                                                              \
        if constexpr (BKM) {                                                                                                          \
            unsigned ab_[NI];                                                                                                         \
            _Pragma("unroll") for (int j_ = 0; j_ < NI; ++j_) ab_[j_] = b_tr[j_] + (soff);                                           \
            pipe_issue_tr<NI, 4 * BN * 2, (KS) * 32 * BN * 2>(ab_, fb[dst].lo, fb[dst].hi);                                          \
        } else {                                                                                                                      \
            pipe_issue<NI, 0, 2048>(fb[dst].kc, b_rd[KS] + (soff));                                                                   \
        }                                                                                                                             \
    } while (0)
#define PIPE_LOAD_A(KS, H, dst, soff) pipe_issue<MIH, (H) * MIH * 2048, 2048>(fa[dst], a_rd[KS] + (soff))
#define PIPE_MFMA(H, abuf, bbuf)                                                                                                      \
    do {                                                                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < MIH; ++i_) {                                                                          \
            const bf16x8 av_ = __builtin_bit_cast(bf16x8, fa[abuf][i_]);                                                              \
            _Pragma("unroll") for (int j_ = 0; j_ < NI; ++j_) acc[(H) * MIH + i_][j_] = mfma16<T>(fb[bbuf].get(j_), av_, acc[(H) * MIH + i_][j_]); \
        }                                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                                            \
    } while (0)

    // MFMAs [C * PER, (C + 1) * PER) of a group (row-major over (A fragment, B fragment)), for the group that shares its time with the DMA issue
#define PIPE_MFMA_PART(H, abuf, bbuf, C, PER)                                                                                         \
    do {                                                                                                                              \
        _Pragma("unroll") for (int m_ = (C) * (PER); m_ < ((C) + 1) * (PER); ++m_) {                                                  \
            const int i_ = m_ / NI, j_ = m_ % NI;                                                                                     \
            acc[(H) * MIH + i_][j_] = mfma16<T>(fb[bbuf].get(j_), __builtin_bit_cast(bf16x8, fa[abuf][i_]), acc[(H) * MIH + i_][j_]); \
        }                                                                                                                             \
    } while (0)
    // last group of a K tile: behind the barrier every read of the tile has completed, so its stage is refilled (tile kt + STAGES) while the
    // group's MFMAs run -- PER MFMAs, then one DMA instruction with its address arithmetic, L times (as one block after the barrier the ~130
    // instructions of the issue kept the matrix pipe of every wave idle at the same time: 0.3-0.5 us of a 1.8 us K tile)
#define PIPE_LAST_GROUP(H, abuf, bbuf)                                                                                                \
    do {                                                                                                                              \
        wait_vmcnt<(STAGES - 2) * L>();                                                                                               \
        __builtin_amdgcn_s_barrier();                                                                                                 \
        PIPE_LOAD_A(0, 0, 0, sn);                                                                                                     \
        PIPE_LOAD_B(0, 0, sn);                                                                                                        \
        issue_begin(kt + STAGES, kt + STAGES >= ktiles);                                                                              \
        static_for<L>([&](auto c_) {                                                                                                  \
            PIPE_MFMA_PART(H, abuf, bbuf, decltype(c_)::value, (MIH * NI) / L);                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                                                        \
            issue_one(c_, smem + (kt % STAGES) * STAGE_BYTES);                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                                        \
        });                                                                                                                           \
        issue_end();                                                                                                                  \
    } while (0)
    static_assert((MIH * NI) % L == 0, "MFMAs of the last group spread evenly over the DMA instructions");

    // (CTAIL: a K tile is a (channel block, tap) pair and the last block is partial: taps x ceil(channels / 64) tiles)
    const int ktiles = CTAIL ? c_ntap * ((conv_kc + BK - 1) / BK) : (p.K + BK - 1) / BK;
#pragma unroll
    for (int t = 0; t < STAGES; ++t) {
        issue_begin(t, t >= ktiles);
        static_for<L>([&](auto c) { issue_one(c, smem + t * STAGE_BYTES); });
        issue_end();
    }
    wait_vmcnt<(STAGES - 1) * L>();                                // tile 0 landed
    __builtin_amdgcn_s_barrier();
    if constexpr (!F8) {
    PIPE_LOAD_A(0, 0, 0, 0u);
    PIPE_LOAD_B(0, 0, 0u);
    for (int kt = 0; kt < ktiles; ++kt) {
        const unsigned so = (unsigned)((kt % STAGES) * STAGE_BYTES);
        const unsigned sn = (unsigned)(((kt + 1) % STAGES) * STAGE_BYTES);
        if constexpr (SPLITA) {
            // group 0: A lower half x k-step 0          (in flight behind it: A upper half of k-step 0, B of k-step 1)
            PIPE_LOAD_A(0, 1, 1, so);
            PIPE_LOAD_B(1, 1, so);
            pipe_wait<MIH + NB>();
            pipe_tie<MIH>(fa[0]); fb[0].tie();
            PIPE_MFMA(0, 0, 0);
            // group 1: A upper half x k-step 0          (A lower half of k-step 1)
            PIPE_LOAD_A(1, 0, 0, so);
            pipe_wait<MIH>();
            pipe_tie<MIH>(fa[1]); fb[1].tie();
            PIPE_MFMA(1, 1, 0);
            // group 2: A lower half x k-step 1          (A upper half of k-step 1)
            PIPE_LOAD_A(1, 1, 1, so);
            pipe_wait<MIH>();
            pipe_tie<MIH>(fa[0]);
            PIPE_MFMA(0, 0, 1);
            // group 3: A upper half x k-step 1
            pipe_wait<0>();
            pipe_tie<MIH>(fa[1]);
            PIPE_LAST_GROUP(1, 1, 1);
        } else {
            // group 0: k-step 0          (in flight behind it: k-step 1)
            PIPE_LOAD_A(1, 0, 1, so);
            PIPE_LOAD_B(1, 1, so);
            pipe_wait<MIH + NB>();
            pipe_tie<MIH>(fa[0]); fb[0].tie();
            PIPE_MFMA(0, 0, 0);
            // group 1: k-step 1
            pipe_wait<0>();
            pipe_tie<MIH>(fa[1]); fb[1].tie();
            PIPE_LAST_GROUP(0, 1, 1);
        }
    }
    } else {
        // ---- fp8 K loop: MFMA groups = (a pair of A fragments) x (every B fragment), both k-steps of the byte tile in one instruction ------------
        // NP groups per K tile (4 for the 256x256 tile, 2 for the 128x128 one).  A pairs are double-buffered (fa8[buf][k-step][fragment of the pair]),
        // the B fragments of a tile (fb8[k-step][fragment]) are used by every group, so the next tile's B reads are issued behind the last group and
        // are the one LDS round trip of a K tile that is waited for in the open (at the top of the loop); the next tile's first A pair is requested
        // right behind the barrier and lands under the last group.  At most 4 + 2 NI reads are in flight (lgkmcnt is a 4-bit counter).
        constexpr int NP = MI / 2;
        static_assert(NP == 2 || NP == 4, "A fragment pairs");
        static_assert((2 * NI) % L == 0, "MFMAs of the last group spread evenly over the DMA instructions");
        u32x4 fa8[2][2][2], fb8[2][NI];
#define PIPE8_LOAD_A(P, dst, soff)                                                                                                   \
    do {                                                                                                                              \
        pipe_issue<2, (P) * 2 * 2048, 2048>(fa8[dst][0], a_rd[0] + (soff));                                                          \
        pipe_issue<2, (P) * 2 * 2048, 2048>(fa8[dst][1], a_rd[1] + (soff));                                                          \
    } while (0)
#define PIPE8_LOAD_B(soff)                                                                                                            \
    do {                                                                                                                              \
        pipe_issue<NI, 0, 2048>(fb8[0], b_rd[0] + (soff));                                                                            \
        pipe_issue<NI, 0, 2048>(fb8[1], b_rd[1] + (soff));                                                                            \
    } while (0)
#define PIPE8_TIE_A(buf) do { pipe_tie<2>(fa8[buf][0]); pipe_tie<2>(fa8[buf][1]); } while (0)
        // MFMA m of group P (row-major over (fragment of the pair, B fragment)); (B, A) operand order as in the bf16 path: the accumulators hold C^T
        // tiles; formats 0 = e4m3, scales 0x7F = 2^0
        // The MFMA is issued from inline asm (unit block scales = the unscaled form; cbsz / blgp 0 = e4m3 x e4m3): as a builtin, hipcc's scheduler sank all the
        // MFMAs of a K tile behind the DMA issue (its high-register-pressure pass drops the sched_barrier edges), with every A pair live at once.
#define PIPE8_MFMA_ONE(P, abuf, f_, j_)                                                                                               \
    do {                                                                                                                              \
        const i32x8 b_ = pipe8_pack(fb8[0][j_], fb8[1][j_]), a_ = pipe8_pack(fa8[abuf][0][f_], fa8[abuf][1][f_]);                     \
        asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+v"(acc[(P) * 2 + (f_)][j_]) : "v"(b_), "v"(a_));               \
    } while (0)
#define PIPE8_MFMA(P, abuf)                                                                                                           \
    do {                                                                                                                              \
        _Pragma("unroll") for (int f_ = 0; f_ < 2; ++f_) {                                                                            \
            _Pragma("unroll") for (int j_ = 0; j_ < NI; ++j_) PIPE8_MFMA_ONE(P, abuf, f_, j_);                                        \
        }                                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                                            \
    } while (0)
        // MFMAs [C * PER, (C + 1) * PER) of the last group (row-major over (fragment of the pair, B fragment))
#define PIPE8_MFMA_PART(P, abuf, C, PER)                                                                                              \
    do {                                                                                                                              \
        _Pragma("unroll") for (int m_ = (C) * (PER); m_ < ((C) + 1) * (PER); ++m_) {                                                  \
            const int f_ = m_ / NI, j_ = m_ % NI;                                                                                     \
            PIPE8_MFMA_ONE(P, abuf, f_, j_);                                                                                          \
        }                                                                                                                             \
    } while (0)
        PIPE8_LOAD_A(0, 0, 0u);
        PIPE8_LOAD_B(0u);
        for (int kt = 0; kt < ktiles; ++kt) {
            const unsigned so = (unsigned)((kt % STAGES) * STAGE_BYTES);
            const unsigned sn = (unsigned)(((kt + 1) % STAGES) * STAGE_BYTES);
            // group 0: pair 0 (requested behind the previous tile's barrier) x B (requested behind the previous tile's last group)
            pipe_wait<0>();
            PIPE8_TIE_A(0); pipe_tie<NI>(fb8[0]); pipe_tie<NI>(fb8[1]);
            PIPE8_LOAD_A(1, 1, so);
            PIPE8_MFMA(0, 0);
            if constexpr (NP == 4) {          // 256x256 tile: pairs 1 and 2, each with the next pair requested into the other buffer
                PIPE8_LOAD_A(2, 0, so);
                pipe_wait<4>();
                PIPE8_TIE_A(1);
                PIPE8_MFMA(1, 1);
                PIPE8_LOAD_A(3, 1, so);
                pipe_wait<4>();
                PIPE8_TIE_A(0);
                PIPE8_MFMA(2, 0);
            }
            // last group: behind the barrier the stage is refilled (tile kt + STAGES), one DMA instruction per (2 NI / L) MFMAs
            pipe_wait<0>();
            PIPE8_TIE_A(1);
            wait_vmcnt<(STAGES - 2) * L>();
            __builtin_amdgcn_s_barrier();
            PIPE8_LOAD_A(0, 0, sn);
            issue_begin(kt + STAGES, kt + STAGES >= ktiles);
            static_for<L>([&](auto c_) {
                PIPE8_MFMA_PART(NP - 1, 1, decltype(c_)::value, (2 * NI) / L);
                __builtin_amdgcn_sched_barrier(0);
                issue_one(c_, smem + (kt % STAGES) * STAGE_BYTES);
                __builtin_amdgcn_sched_barrier(0);
            });
            issue_end();
            PIPE8_LOAD_B(sn);
        }
        pipe_wait<0>();                                            // (the reads requested for a tile beyond K)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");         // the last MFMAs retire before the epilogue reads the accumulators (asm MFMAs are invisible to the hazard recogniser)
#undef PIPE8_MFMA_PART
#undef PIPE8_MFMA
#undef PIPE8_MFMA_ONE
#undef PIPE8_TIE_A
#undef PIPE8_LOAD_B
#undef PIPE8_LOAD_A
    }
    wait_vmcnt<0>();                                               // the tiles issued beyond K
#undef PIPE_LAST_GROUP
#undef PIPE_MFMA_PART
#undef PIPE_LOAD_A
#undef PIPE_LOAD_B
#undef PIPE_MFMA
    if constexpr (F8) {
        // dequantisation: the tensors were quantised as q = e4m3(x * 448 / amax); amax <= 0 stands for "scale 1" (uncalibrated first step)
        lavt_gemm_nt_t q = p;
        const float da = p.deq_a ? *p.deq_a : 0.f, db = p.deq_b ? *p.deq_b : 0.f;
        q.alpha = p.alpha * (da > 0.f ? da * (1.f / 448.f) : 1.f) * (db > 0.f ? db * (1.f / 448.f) : 1.f);
        if (p.colstats) pipe_colstats<MI, NI>(q, acc, m0 + wm * WM, n0 + wn * WN, lane, tile_m * 2 + wm);
        nt_epilogue<bf16, MI, NI, false, false, LEAN>(q, acc, m0 + wm * WM, n0 + wn * WN, lane, bz);
    } else {
        if (p.colstats) pipe_colstats<MI, NI>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, tile_m * 2 + wm);
        nt_epilogue<T, MI, NI, false, false, LEAN>(p, acc, m0 + wm * WM, n0 + wn * WN, lane, bz);
    }
}

static inline int conv_taps_of(const lavt_gemm_nt_t& p) {
    return (p.conv_kd > 0 ? p.conv_kd : 1) * (p.conv_kh > 0 ? p.conv_kh : 3) * (p.conv_kw > 0 ? p.conv_kw : 3);
}
template <int BM, int BN, bool BKM, int STAGES, int MODE, int LEAN, bool F8 = false> int launch_pipe_(const lavt_gemm_nt_t& p, hipStream_t st) {
    constexpr size_t lds = STAGES * (size_t)(BM * 128 + BN * 128);
    static bool attr_set = false;
    if (!attr_set && lds > 65536) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_pipe_kernel<BM, BN, BKM, STAGES, MODE, LEAN, F8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            lavt_set_error("lavt_gemm_nt(pipe): cannot reserve %zu bytes of LDS", lds);
            return LAVT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), p.batch);
    hipLaunchKernelGGL((gemm_nt_pipe_kernel<BM, BN, BKM, STAGES, MODE, LEAN, F8>), grid, dim3(512), lds, st, p);
    LAVT_CHECK_LAUNCH("lavt_gemm_nt(pipe)");
    return LAVT_OK;
}
template <int BM, int BN, bool BKM, int STAGES, int MODE> int launch_pipe_lean(const lavt_gemm_nt_t& p, hipStream_t st) {
    if constexpr (BM == 256 && BKM && MODE == 2) {          // split-output data gradient of a concat convolution: the plain store with the second output kept (LEAN 3)
        if (p.C2 && p.act == 0 && !p.mul && !p.Cpre && !p.bias && !p.R && !p.row_scale && !p.c_rowmap && lavt_tuning().probe[7] != 1)
            return launch_pipe_<BM, BN, BKM, STAGES, MODE, 3>(p, st);
    }
    if (MODE != 0 && p.act == 0 && !p.mul && !p.Cpre && !p.C2) {      // epilogue instantiations without the features a launch does not use (gemm_common.h)
        if (!p.bias && !p.R && !p.row_scale && !p.c_rowmap) return launch_pipe_<BM, BN, BKM, STAGES, MODE, 2>(p, st);
        return launch_pipe_<BM, BN, BKM, STAGES, MODE, 1>(p, st);
    }
    return launch_pipe_<BM, BN, BKM, STAGES, MODE, 0>(p, st);
}
// fp8 operands: the plain and the tap-walking issue only (lavt_gemm_nt_pipe_tile sends everything else to gemm_v2.hip), lean epilogues as above
template <int BM, int BN, int STAGES, int MODE> int launch_pipe_f8_lean(const lavt_gemm_nt_t& p, hipStream_t st) {
    if (p.act == 0 && !p.mul && !p.Cpre && !p.C2) {
        if (!p.bias && !p.R && !p.row_scale && !p.c_rowmap) return launch_pipe_<BM, BN, false, STAGES, MODE, 2, true>(p, st);
        return launch_pipe_<BM, BN, false, STAGES, MODE, 1, true>(p, st);
    }
    return launch_pipe_<BM, BN, false, STAGES, MODE, 0, true>(p, st);
}
static inline bool pipe_f8_simple(const lavt_gemm_nt_t& p) { return p.conv_kc <= 0 && p.A2 == nullptr && p.K % 128 == 0; }
static inline bool pipe_f8_convfast(const lavt_gemm_nt_t& p) {
    const int64_t lim = (1ll << 31) - (1ll << 24);
    const int64_t a_rows = (int64_t)p.M + 2 * ((int64_t)(p.conv_h > 0 ? p.conv_h : 1) * p.conv_w + p.conv_w + 1);
    const bool small = a_rows * p.lda < lim && (p.A2 == nullptr || a_rows * p.lda2 < lim) && (int64_t)p.N * p.ldb + (int64_t)p.batch * p.strideB < lim;
    return p.conv_kc > 0 && p.conv_kc % 128 == 0 && (p.A2 == nullptr || p.a_split % 128 == 0) && conv_taps_of(p) <= 32 && small;
}
template <int BM, int BN, int STAGES> int launch_pipe_f8(const lavt_gemm_nt_t& p, hipStream_t st) {
    if (pipe_f8_simple(p)) return launch_pipe_f8_lean<BM, BN, STAGES, 1>(p, st);
    return launch_pipe_f8_lean<BM, BN, STAGES, 2>(p, st);
}
template <int BM, int BN, bool BKM, int STAGES> int launch_pipe(const lavt_gemm_nt_t& p, hipStream_t st) {
    const bool simple = p.conv_kc <= 0 && p.A2 == nullptr && p.K % 64 == 0;
    // (tap-walking mode: operands through 32-bit buffer offsets -- each of them below 2 GB, the halo margin included)
    const int64_t lim = (1ll << 31) - (1ll << 24);
    const int64_t a_rows = (int64_t)p.M + 2 * ((int64_t)(p.conv_h > 0 ? p.conv_h : 1) * p.conv_w + p.conv_w + 1);
    const int64_t b_rows = p.b_kmajor ? (int64_t)conv_taps_of(p) * p.conv_kc : p.N;
    const bool small = a_rows * p.lda * 2 < lim && (p.A2 == nullptr || a_rows * p.lda2 * 2 < lim) && b_rows * p.ldb * 2 + (int64_t)p.batch * p.strideB * 2 < lim;
    const bool convfast = p.conv_kc > 0 && p.conv_kc % 64 == 0 && (p.A2 == nullptr || p.a_split % 64 == 0) && conv_taps_of(p) <= 32 && small;
    if (simple && !lavt_tuning().gemm_general) return launch_pipe_lean<BM, BN, BKM, STAGES, 1>(p, st);
    if (convfast && !lavt_tuning().gemm_general) return launch_pipe_lean<BM, BN, BKM, STAGES, 2>(p, st);
    if constexpr (!BKM) {          // k-contiguous weights [Cout][taps][Cin] with Cin % 64 != 0 (Cin % 8 == 0; a concat boundary on a 64-channel block): the partial-block form
        const bool convtail = p.conv_kc > 0 && p.conv_kc % 64 != 0 && p.conv_kc % 8 == 0 && (p.A2 == nullptr || p.a_split % 64 == 0) && conv_taps_of(p) <= 32 && small &&
                              p.conv_kc_split <= 0 && p.conv_tap_split <= 0 && p.batch == 1 && p.K == conv_taps_of(p) * p.conv_kc && p.ldb >= (int64_t)conv_taps_of(p) * p.conv_kc &&
                              !lavt_tuning().conv_tail_off;
        if (convtail && !lavt_tuning().gemm_general) return launch_pipe_lean<BM, BN, BKM, STAGES, 3>(p, st);
    }
    return launch_pipe_lean<BM, BN, BKM, STAGES, 0>(p, st);
}

}  // namespace

// Which tile of this kernel family a problem takes (0: none -- it goes to gemm_v2.hip / gemm.hip) and the ring depth.  Measured on MI355X
// (tools/gemm_bench.py, tools/conv_small_probe.py): the 256x256 tile (one workgroup per CU, 128 flop per byte of LDS fill -- every CU ingests ~50 GB/s
// whatever the tile, so the 128x128 tile is fill-bound at half the rate) only when its tiles fill the 256 CUs well (whole rounds at >= 80 %); the 128x128
// tile for long reductions (K >= 1024; LAVT_GEMM_PIPE=3: every 128x128 problem) where gemm_v2.hip would take its 8-wave 128x128 form.
int lavt_gemm_nt_pipe_tile(const lavt_gemm_nt_t& p, int* stages_out) {
    if ((p.dtype != LAVT_BF16 && p.dtype != LAVT_FP8) || p.zeros == nullptr) return 0;
    const lavt_tuning_t& tun = lavt_tuning();
    const int pipe = tun.gemm_pipe;                 // 0: gemm_v2 K loops only; 1: the 256x256 tile; 2: + 128x128 for K >= 1024; 3: + every 128x128 problem
    if (tun.gemm_v2_off || pipe < 1) return 0;
    if (p.dtype == LAVT_FP8) {          // e4m3 operands: k-contiguous, 128-element K tiles, plain or tap-walking issue (anything else: gemm_v2.hip's fp8 form)
        if (tun.gemm_general || tun.fp8_pipe_off || p.b_kmajor || p.c_f32 || p.conv_kc_split > 0 || p.conv_tap_split > 0 || p.lda % 16 || p.ldb % 16 || (p.A2 && p.lda2 % 16)) return 0;
        if (!pipe_f8_simple(p) && !pipe_f8_convfast(p)) return 0;
    }
    if (p.lda % 8 || p.ldb % 8 || (p.A2 && p.lda2 % 8)) return 0;
    if (p.ln_wsum || p.dact_pre) return 0;
    if (p.conv_kc_split > 0) {          // the channel-split reduction exists in this kernel's tap-walking mode only
        if (tun.gemm_general) return 0;          // (LAVT_GEMM_GENERAL forces the general decode: lavt_gemm_nt then refuses the problem)
        *stages_out = 4;
        return 128;
    }
    const int force = tun.gemm_tile;
    const long tiles128 = (long)cdiv(p.M, 128) * cdiv(p.N, 128) * p.batch;
    const long tiles64 = (long)cdiv(p.M, 64) * cdiv(p.N, 64) * p.batch;
    const int big_long = tun.gemm_big_long;
    const bool big = force ? force == 128 : ((tiles128 >= 200 || (big_long > 0 && p.K >= 64 * 64 && tiles128 >= big_long)) && p.N >= 128);
    const long wgs = big ? tiles128 : tiles64;
    const int stages = tun.gemm_stages ? tun.gemm_stages : (wgs >= (big ? s2_min128() : 512) ? 2 : 4);
    const long tiles256 = (long)cdiv(p.M, 128) * cdiv(p.N, 256) * p.batch;
    const bool wide = force ? force == 256 : (tun.gemm_wide && tiles256 >= 256 && p.N % 256 == 0 && p.K >= 1024);
    if (wide) return 0;
    const long tiles256x = (long)cdiv(p.M, 256) * cdiv(p.N, 256) * p.batch;
    const long rounds = (tiles256x + 255) / 256;
    // (round 5: N within 1/16 of a multiple of 256 also takes the large tile -- Swin-T's 480-column data gradient of the concat convolution ran 439 us on 3600
    // 128x128 tiles at 0.35 of peak; the columns beyond N are out-of-range offsets in the weight descriptor and masked stores)
    const bool n_fits = p.N % 256 == 0 || (p.N % 8 == 0 && (long)p.N * 16 >= (long)cdiv(p.N, 256) * 256 * 15);
    const bool huge = force ? force == 512 : (n_fits && p.K >= 1024 && tiles256x >= 128 && tiles256x * 10 >= rounds * 256 * 8);
    if (huge) { *stages_out = 2; return 256; }
    if (big && tun.gemm_waves == 8 && (pipe >= 3 || (pipe == 2 && p.K >= 1024))) { *stages_out = stages == 2 ? 2 : 4; return 128; }
    return 0;
}

// tile: 256 = the 256x256 tile (2-stage ring, 128 KB of LDS), 128 = the 128x128 tile with `stages` (2 or 4) stages
int lavt_gemm_nt_pipe(const lavt_gemm_nt_t& p, int tile, int stages, hipStream_t st) {
    if (p.dtype == LAVT_FP8) {
        if (tile == 256) return launch_pipe_f8<256, 256, 2>(p, st);
        return stages == 2 ? launch_pipe_f8<128, 128, 2>(p, st) : launch_pipe_f8<128, 128, 4>(p, st);
    }
    if (tile == 256) return p.b_kmajor ? launch_pipe<256, 256, true, 2>(p, st) : launch_pipe<256, 256, false, 2>(p, st);
    if (stages == 2) return p.b_kmajor ? launch_pipe<128, 128, true, 2>(p, st) : launch_pipe<128, 128, false, 2>(p, st);
    return p.b_kmajor ? launch_pipe<128, 128, true, 4>(p, st) : launch_pipe<128, 128, false, 4>(p, st);
}

extern "C" int lavt_gemm_nt_colstats_plan(const lavt_gemm_nt_t* pp, int* rows_per_block) {
    if (!pp || !rows_per_block) return 0;
    const lavt_gemm_nt_t& p = *pp;
    *rows_per_block = 0;
    if (p.batch != 1 || p.bias || p.act || p.R || p.row_scale || p.c_rowmap || p.c_f32 || p.C2 || p.Cpre || p.mul || p.N % 4 || p.M <= 0 || p.N <= 0) return 0;
    int stages = 0;
    const int tile = lavt_gemm_nt_pipe_tile(p, &stages);
    if (!tile) return 0;
    *rows_per_block = tile / 2;                      // a wave row of the 2 x 4 wave grid
    return cdiv(p.M, tile) * 2;
}
