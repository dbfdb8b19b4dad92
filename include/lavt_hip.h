/*
 * lavt_hip.h -- C ABI of the MI355X (gfx950) LAVT hot-path library (liblavt_hip.so).
 *
 * Every entry point is `extern "C"`, takes raw DEVICE pointers + sizes + a hipStream_t (as void*),
 * allocates nothing, launches asynchronously on the given stream and returns 0 or a negative LAVT_ERR_*
 * code.  No torch types anywhere.  Global state: the (thread-local) last-error string and one table of
 * tuning switches -- the LAVT_* environment variables listed in lavt-rs_amd/csrc/tuning.hip, which choose
 * between equivalent kernel configurations (tile sizes, ring depths, A/B forms).  They are read once, at
 * the first launch; lavt_tuning_reload() re-reads them.  No launch path calls getenv, and no switch makes
 * a kernel skip work.
 *
 * The reference (Yxxxb/LAVT-RS) is pure PyTorch and has no native layer; each group of functions
 * below names the reference code (file:line under the reference root) whose arithmetic it replaces.
 * The Python host side that binds these with ctypes is lavt-rs_amd/lavt_hip/_capi.py;
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Layout conventions: activations are token-major / NHWC ([rows][channels], channels contiguous);
 * "dtype" selects the storage type of activations and of the compute copies of the weights
 * (LAVT_F32 = exact-fp32 MFMA path used for parity, LAVT_BF16 = bf16 MFMA with fp32 accumulation).
 * Biases, LayerNorm/BatchNorm affine parameters, statistics and every parameter GRADIENT are fp32.
 */
#ifndef LAVT_HIP_H
#define LAVT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LAVT_F32 0
#define LAVT_BF16 1
#define LAVT_FP8 2 /* lavt_gemm_nt only: A and B are OCP e4m3 (1 byte / element, k-contiguous), C / bias / residual bf16, fp32 accumulate */

#define LAVT_OK 0
#define LAVT_ERR_INVALID (-22) /* bad argument (EINVAL) */
#define LAVT_ERR_LAUNCH (-5)   /* kernel launch failed (EIO) */

#define LAVT_ACT_NONE 0
#define LAVT_ACT_GELU 1 /* exact erf form */
#define LAVT_ACT_RELU 2
#define LAVT_ACT_TANH 3
#define LAVT_ACT_GELU_D 4 /* lavt_gemm_nt.act: GELU whose Cpre output holds GELU'(pre) instead of pre (bf16 operands; the derivative shares the
                           * exponential of the activation: +2 fma per element), for a consumer that only needs the derivative (ABI v4).
                           * LayerNorm-folded launches only (ln_wsum): the branch is compiled into that kernel alone */
#define LAVT_ACT_STORED 5 /* lavt_gemm_nt.dact: dact_pre already holds act'(pre) (written by a LAVT_ACT_GELU_D launch): C * dact_pre (ABI v4) */

int lavt_abi_version(void);
const char* lavt_last_error(void);
int lavt_tuning_reload(void); /* ABI v5: re-read the LAVT_* tuning switches from the environment (tests that force a tile configuration) */

/* ---------------------------------------------------------------------------------------------
 * Gather-GEMM, "NT" family:   C[M,N] = epilogue( alpha * A_op[M,K] x B_op[K,N] )
 *
 * Replaces every nn.Linear / 1x1 Conv1d / 3x3 Conv2d forward and data-gradient on the path:
 *   qkv, proj (lib/backbone.py:121,141), Mlp fc1/fc2 (:24-30), PatchMerging.reduction (:286),
 *   PWAM vis_project/f_query/f_key/f_value/W/project_mm (:1244-1327), res_gate (:604-611),
 *   PatchEmbed.proj (:309, after lavt_im2col4), SimpleDecoding conv3x3 (lib/mask_predictor.py:18-38),
 *   and the window partition / roll / pad / reverse copies (lib/backbone.py:33-62, 204-237), which become
 *   row gather (a_rowmap) and row scatter (c_rowmap) of the GEMMs on either side of the attention core.
 *
 * A_op: rows along M with K contiguous.  Row m is read from source row a_rowmap[m] (or m); -1 reads zeros.
 *   conv_cin > 0 makes this an implicit-GEMM 3x3 convolution (pad 1) over a (batch, conv_h, conv_w) pixel grid:
 *   K = 9*conv_kc, k = tap*conv_kc + c, tap = (dy+1)*3+(dx+1); row m = pixel, source row = neighbour pixel
 *   (mirrored neighbour when conv_flip = 1, i.e. the data gradient).  Channels c >= a_split come from A2
 *   (fused torch.cat of [top-down, skip], lib/mask_predictor.py:60,70,81).
 * B_op: b_kmajor = 0: B[N][K] (nn.Linear weight layout);  1: B[K][N] (used for x @ W, i.e. data gradients).
 *   With conv and b_kmajor = 1, k = tap*conv_kc + c addresses B + c*ldb + tap*b_tap_stride + n.
 * Epilogue, in order: *alpha, +bias[n], *row_scale[m], (Cpre = value), act, *mul[out_row][n], +R[out_row][n], store.
 *   Output row = c_rowmap[m] (or m); -1 drops the row.  Columns >= c_split go to C2 (fused split of the
 *   gradient of a concatenation).  c_f32 stores fp32 whatever dtype is.
 * Requirements: K % (16/sizeof(dtype)) == 0; if b_kmajor, N % (16/sizeof(dtype)) == 0; lda/ldb/ldc multiples of
 *   the same; all pointers 16-byte aligned.
 * ------------------------------------------------------------------------------------------- */
typedef struct lavt_gemm_nt {
    int32_t dtype, M, N, K, batch;
    /* A */
    const void* A;
    int64_t lda, strideA;
    const void* A2;
    int64_t lda2;
    int32_t a_split;
    const int32_t* a_rowmap;
    int32_t conv_h, conv_w, conv_kc, conv_flip;
    /* B */
    const void* B;
    int64_t ldb, strideB;
    int32_t b_kmajor;
    int64_t b_tap_stride;
    /* epilogue */
    float alpha;
    const float* bias;
    int64_t strideBias;
    const float* row_scale;
    int64_t strideRowScale;
    int32_t row_scale_div; /* row m uses row_scale[m / row_scale_div] (<=1 means m): per-sample DropPath factors */
    int32_t act;
    void* Cpre;
    int64_t ldcpre;
    const void* R;
    int64_t ldr;
    void* C;
    int64_t ldc, strideC;
    void* C2;
    int64_t ldc2;
    int32_t c_split;
    const int32_t* c_rowmap;
    int32_t c_f32;
    const void* zeros; /* optional: >= 16 bytes of zeros in device memory; enables the LDS-DMA pipeline kernel (bf16) */
    int32_t epi_lds;   /* set by the library (LAVT_GEMM_EPI=lds): stage the C tile through LDS for full-row stores */
    /* 3-D generalisation of the implicit-GEMM convolution (Conv3d of SepTPWAM, lib/video_swin_transformer.py:1327-1460): rows are voxels of a
     * (batch, conv_d, conv_h, conv_w) grid and the taps run over conv_kd x conv_kh x conv_kw (each 1 or 3, 'same' zero padding), kw fastest.
     * All zero = the 2-D default (conv_d = 1, taps 1 x 3 x 3). */
    int32_t conv_d, conv_kd, conv_kh, conv_kw;
    /* Fused activation gradient (ABI v2): when dact_pre != NULL the stored value is C * act'(dact_pre[row][n]) with act = dact (LAVT_ACT_*):
     * the data gradient of the layer AFTER an activation leaves as the gradient w.r.t. the pre-activation (fc2's dgrad of a Swin Mlp applies
     * GELU' of fc1's saved pre-activation, reference lib/backbone.py:24-30 under autograd), so no separate element-wise pass exists.
     * bf16, b_kmajor, no conv taps, K % 64 == 0, lddact % 8 == 0 only (LAVT_ERR_INVALID otherwise: the caller keeps the two-kernel form). */
    const void* dact_pre;
    int64_t lddact;
    int32_t dact;
    /* dtype == LAVT_FP8 (BASELINE.json configs[4]): per-tensor scaling.  deq_a / deq_b point at the device floats holding the |max| the A / B
     * tensors were quantised against (lavt_fp8_quantize*: q = e4m3(x * 448 / amax); a value <= 0 means "scale 1"); the epilogue multiplies the
     * accumulator by (amax_a / 448) * (amax_b / 448) * alpha.  The contraction runs on v_mfma_scale_f32_16x16x128_f8f6f4 with unit block
     * scales (twice the bf16 MFMA rate at half the operand bytes).  Requirements: !b_kmajor, no dact_pre, K % 16 == 0, lda / ldb % 16 == 0,
     * conv_kc (and a_split) % 128 == 0 for the tap-walking fast path. */
    const float* deq_a;
    const float* deq_b;
    int32_t epi_wide; /* set by the library: 16-byte stores from paired accumulator fragments (all row strides / splits 8-element aligned) */
    /* ABI v4, fused PWAM path (reference lib/backbone.py:604-611, 669 and its autograd mirror):
     * mul: element-wise multiplier [out_row][n] applied after the activation and before the residual -- with act = LAVT_ACT_TANH, Cpre, R = x
     *   and mul = r the second gate GEMM writes x + tanh(g) * r directly (the language gate is no separate kernel).
     * res_first (with dact_pre and R): the stored value is (C + R) * act'(dact_pre) instead of C * act'(dact_pre) + R -- a gradient that arrives
     *   beside the GEMM's own contribution joins before the activation gradient. */
    const void* mul;
    int64_t ldmul;
    int32_t res_first;
    /* conv_tap_split > 0 (ABI v4): the reduction of a convolution is cut at tap boundaries over the batch index -- entry bz contracts taps
     * [bz * conv_tap_split, (bz + 1) * conv_tap_split) only (K = conv_tap_split * conv_kc, batch * conv_tap_split = taps; with !b_kmajor strideB
     * advances the packed weight by conv_tap_split * conv_kc elements, with b_kmajor strideB = 0).  Used with c_f32 + strideC as split-K through
     * fp32 partial outputs for the decoder's small-pixel-count convolutions (1 800 rows x K = 13 824: 60-232 tiles with serial chains of 72-216
     * K tiles otherwise); lavt_splitk_reduce adds the partials.  Tap-walking fast path only (conv_kc % 64 == 0). */
    int32_t conv_tap_split;
    /* LayerNorm-folded A operand (ABI v4): ln_wsum != NULL makes this GEMM compute LN(A) B^T + b without a LayerNorm launch: A = the RAW rows of the
     * norm's input, B = the gamma-folded weight and bias = the folded bias of lavt_ln_fold, ln_wsum [N] its row sums; the kernel accumulates the
     * row statistics from the A tiles it streams (K = the normalised width: every tile walks all of it) and applies rstd (acc - mu wsum_n) before the
     * epilogue.  ln_mean / ln_rstd (optional, [M]) receive the statistics for the LayerNorm backward.  bf16, plain k-contiguous problems only
     * (norm2 -> fc1 of a Swin block, lib/backbone.py:243 + :24-30). */
    const float* ln_wsum;
    float* ln_mean;
    float* ln_rstd;
    float ln_eps;
    /* Column statistics of the output from the epilogue (ABI v6): BatchNorm statistics of a bias-free convolution without a pass over its output
     * (reference lib/mask_predictor.py:60-97: conv -> BatchNorm2d -> ReLU).  colstats != NULL: every block of `rows_per_block` output rows (both as
     * reported by lavt_gemm_nt_colstats_plan for this problem) stores, per column n, the sum of its rows and their second moment about the block's
     * own mean -- colstats[(block * 2 + 0) * N + n], colstats[(block * 2 + 1) * N + n] -- taken from the fp32 accumulators (alpha applied, before
     * the rounding to bf16).  lavt_colstats_finish_blocks combines the blocks (parallel-variance form, no E[x^2] - E[x]^2).  Only for launches
     * lavt_gemm_nt_colstats_plan accepts (LAVT_ERR_INVALID otherwise): bf16, batch 1, no bias / activation / residual / row map / fp32 output. */
    float* colstats;
    /* conv_kc_split > 0 (ABI v6): the reduction of a convolution is cut over CHANNEL blocks instead of taps -- entry bz of the batch contracts channels
     * [bz * conv_kc_split, (bz + 1) * conv_kc_split) of every tap (K = taps * conv_kc_split, batch * conv_kc_split = conv_kc, conv_kc_split % 64 == 0,
     * strideB = 0: both weight layouts are offset inside the kernel).  Any divisor of conv_kc / 64 is a split count, where conv_tap_split offers 3 or 9:
     * the decoder's 1 800-row convolutions run as 480 workgroups instead of 180.  With c_f32 + strideC as split-K through fp32 partial outputs
     * (lavt_splitk_reduce).  Pipelined tap-walking kernel only (csrc/gemm_nt_pipe.hip: conv_kc % 64 == 0, a_split % 64 == 0, bf16). */
    int32_t conv_kc_split;
} lavt_gemm_nt_t;

int lavt_gemm_nt(const lavt_gemm_nt_t* p, void* stream);
/* -> number of row blocks whose partial statistics lavt_gemm_nt(p) would store into p->colstats (0: this problem's kernel has no statistics
 * epilogue -- leave colstats NULL and take the statistics from the output with lavt_colstats_meanrstd); *rows_per_block = rows per block */
int lavt_gemm_nt_colstats_plan(const lavt_gemm_nt_t* p, int* rows_per_block);
/* second stage: blocks' (sum, centred second moment) pairs -> mean / rstd (+ running estimates, momentum form of nn.BatchNorm2d) and / or the
 * (sum, M2) pair of all `rows` rows (sum_out / m2_out: what the SyncBatchNorm exchange carries).  Any of the output groups may be NULL. */
int lavt_colstats_finish_blocks(const float* partials, int nblk, int rows_per_block, int rows, int C, float eps, float* mean, float* rstd,
                                float* sum_out, float* m2_out, float* running_mean, float* running_var, float momentum, void* stream);
/* out[m][n] (dtype) = sum_s parts[s][m][n] (fp32): second stage of a split reduction of lavt_gemm_nt (c_f32 partial outputs, one per batch entry) */
int lavt_splitk_reduce(int dtype, const float* parts, int splits, int64_t M, int N, void* out, int64_t ldc, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Gather-GEMM, "TN" family (weight gradients):   C[I,J] += alpha * sum_k A[k][i] * B[k][j]      (fp32 C)
 *
 * Replaces the weight-gradient half of autograd for every layer listed above (the reference gets it from
 * torch.autograd; there is no reference source line).  Rows k of A / B are gathered through a_rowmap / b_rowmap
 * (-1 = zero row); conv_cin > 0: J = 9*conv_kc and column j = tap*conv_kc + c reads B row = neighbour pixel of k
 * for that tap (channels >= b_split from B2).  c_conv_permute stores column (tap,c) at c*9+tap so that C is
 * directly the PyTorch [Cout][Cin][3][3] gradient.  The reduction over k is split across workgroups
 * (split_k, 0 = auto) and accumulated with fp32 atomics: C must hold zeros or a running sum.
 * colsum (optional, fp32 [I]) additionally receives sum_k A[k][i] (the bias gradient).
 * Requirements: I, J (and conv_kc) % (16/sizeof(dtype)) == 0; lda/ldb likewise.
 * ------------------------------------------------------------------------------------------- */
typedef struct lavt_gemm_tn {
    int32_t dtype, I, J, K, batch;
    const void* A;
    int64_t lda, strideA;
    const int32_t* a_rowmap;
    const float* a_rowscale; /* optional fp32 factor of A row k: a_rowscale[k / a_rowscale_div] (mask / DropPath of the forward) */
    int32_t a_rowscale_div;
    const void* B;
    int64_t ldb, strideB;
    const void* B2;
    int64_t ldb2;
    int32_t b_split;
    const int32_t* b_rowmap;
    int32_t conv_h, conv_w, conv_kc;
    float alpha;
    float* C;
    int64_t ldc, strideC;
    int32_t c_conv_permute;
    int32_t split_k;
    float* colsum;
    int64_t strideColsum;
    const void* zeros; /* optional zero page, as in lavt_gemm_nt_t */
    int32_t a_rowscale_binary; /* 1: a_rowscale holds only 0 and ONE non-zero value which the caller folded into alpha (row masks) */
    int32_t accumulate;        /* 1: C += (atomics) even without split-K; 0: C may be overwritten when the reduction is not split */
    int32_t conv_d, conv_kd, conv_kh, conv_kw; /* 3-D taps, as in lavt_gemm_nt_t; c_conv_permute then stores column (tap,c) at c*taps+tap */
    float* partials;           /* optional scratch (ABI v3): with it a split reduction (many K pieces on few output tiles: the long-K weight gradients */
    int64_t partials_floats;   /* of PWAM's 1x1 convolutions, K = 28 800 rows) stores one plain partial tile per piece ([piece][I][J], then [piece][I] */
                               /* for colsum) and a second small kernel adds the pieces into C -- instead of up to ~60 workgroups adding into the same */
                               /* 64x64 tile through fp32 atomics.  Needs pieces*(I*J + I) floats (lavt_gemm_tn_pieces gives an upper bound); too small */
                               /* or NULL selects the atomic form.  The scratch may be shared by consecutive calls on one stream. */
    int32_t colsum_atomic;     /* ABI v4: colsum is ADDED with fp32 atomics even where C is stored plainly -- two problems then share one bias gradient:
                                * a windowed qkv weight gradient taken over the REAL tokens only (token order: the zero rows of padded window
                                * positions are skipped, K = tokens instead of window rows) + a side problem over the padded rows alone, whose
                                * dq / dk / dv still belong to the bias gradient (the reference pads after norm1: lib/backbone.py:205-209) */
    int64_t a_src_rows, b_src_rows; /* ABI v7: rows of the tensors that a_rowmap / b_rowmap index (0 = unknown: at most K).  The pipelined grouped kernel addresses a
                                * mapped operand through ONE 2 GB buffer descriptor with 32-bit byte offsets row * ld * 2: a problem whose mapped source
                                * reaches beyond that falls back to the 64x64 launch instead of reading zeros from beyond the descriptor's range */
} lavt_gemm_tn_t;

int lavt_gemm_tn(const lavt_gemm_tn_t* p, void* stream);
/* upper bound of the K pieces lavt_gemm_tn may cut this problem into (for sizing `partials`) */
int lavt_gemm_tn_pieces(const lavt_gemm_tn_t* p);
/* n (<= 6) independent problems of the family in one call, e.g. the four weight gradients of a Swin block: when they qualify (bf16, no
 * conv taps / concat, batch 1, together >= 256 64x64 output tiles) they run as ONE launch without split-K -- every output element then has
 * a single writer, and with accumulate == 0 it is stored plainly instead of added through fp32 atomics; otherwise they are issued one by
 * one exactly as lavt_gemm_tn would.  A member with split_k < 0 declares that its C (and colsum) hold ZEROS, so that storing and adding
 * give the same result: the grouped launch may then cut its reduction into up to 4 pieces that meet through atomics (the launch lasts as
 * long as its longest serial chain of K tiles). */
int lavt_gemm_tn_grouped(const lavt_gemm_tn_t* probs, int n, void* stream);
/* lavt_gemm_tn_grouped + ONE LayerNorm backward as rider workgroups (ABI v5): the partial-sum form of lavt_layernorm_bwd_partial (bf16, no gather; dy, x,
 * gamma, mean, rstd, dx, dres, rows, C, ws as there) needs nothing from the grouped launch and the launch nothing from it, so when the group runs on the
 * 64x64 launch with column sums (the four weight gradients of a Swin block) the LayerNorm's workgroups are appended to its grid instead of being a launch
 * of their own on the critical chain (norm1's backward of a Swin block: 7.7 us x 24 per Swin-B step).  In every other case the two are issued one after
 * the other: same result. */
int lavt_gemm_tn_grouped_ln(const lavt_gemm_tn_t* probs, int n, const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                            void* dx, float* ws, int64_t ws_floats, const void* dres, int rows, int C, void* stream);
/* The same launch in stream-K form (ABI v5): 128x128 tiles, the K-tile iterations of all members cut into equal runs for ~512 persistent
 * workgroups (every workgroup ingests the same number of bytes); tiles whose reduction is split between runs meet through `scratch` (plain partial
 * tiles + a fixed-order sum: no atomics).  lavt_gemm_tn_grouped_sk_ws returns the floats of scratch the group wants, or 0 when it does not qualify
 * (conv taps, batch > 1, too little work): call lavt_gemm_tn_grouped then.  Members' `partials` fields are ignored. */
int64_t lavt_gemm_tn_grouped_sk_ws(const lavt_gemm_tn_t* probs, int n);
int lavt_gemm_tn_grouped_sk(const lavt_gemm_tn_t* probs, int n, float* scratch, int64_t scratch_floats, void* stream);

/* 3x3 / pad 1 / no-bias convolution weight gradient with the nine taps fused (ABI v5; bf16; csrc/conv_wgrad.hip) -- the gradient autograd computes for
 * SimpleDecoding's conv1_4 .. conv2_2 (lib/mask_predictor.py:60-97):  dW[co][ci][tap] += sum_p dY[p][co] * X[p + (dy, dx)][ci],
 * tap = (dy + 1) * 3 + (dx + 1), pixels p over B images of H x W (NHWC rows), X = the channel concat of x1 (c1 channels) and x2 (Cin - c1; NULL
 * for a single source).  dW is the [Cout][Cin][3][3] parameter gradient itself: accumulate != 0 adds to it, 0 overwrites it (a zeroed buffer with
 * this one writer: saves reading it).  `parts`: caller-lent fp32 scratch of
 * lavt_conv3x3_wgrad_ws(...) floats (partial tiles of the split pixel reduction; no atomics, run-to-run identical); _ws returns 0 for shapes the
 * kernel does not cover (Cout % 128, Cin % 8, c1 % 64, W <= 128) -- use lavt_gemm_tn's tap-shifted form + lavt_unpack_conv_grad then. */
int64_t lavt_conv3x3_wgrad_ws(int B, int H, int W, int Cout, int Cin, int c1);
int lavt_conv3x3_wgrad(const void* dy, int64_t ldy, const void* x1, int64_t ldx1, const void* x2, int64_t ldx2, int c1, int B, int H, int W, int Cout,
                       int Cin, float* parts, int64_t parts_floats, float* dW, int accumulate, const void* zeros, void* stream);
/* The same gradient from OCP e4m3 operands on the fp8 MFMA (round 5; BASELINE.json configs[4]): dy, x1, x2 hold bytes q = value * 448 / |max|
 * ([pixels][channels]; leading dimensions in bytes, multiples of 16) -- the copies lavt_fp8_quantize / lavt_fp8_quantize_current wrote for the forward
 * convolution and the data gradient --, amax_dy / amax_x point at the |max| each was quantised against (x1 and x2 share one scale: they feed one
 * contraction); dW receives the de-quantised fp32 gradient.  Scratch as lavt_conv3x3_wgrad_ws.  lavt_conv3x3_wgrad_f8_ok returns 0 for shapes the
 * kernel does not cover (32 < W <= 128, Cout % 128, Cin % 16, c1 % 64, H even when W <= 64): use the bf16 entry then. */
int lavt_conv3x3_wgrad_f8_ok(int B, int H, int W, int Cout, int Cin, int c1);
int lavt_conv3x3_wgrad_f8(const void* dy, int64_t ldy, const float* amax_dy, const void* x1, int64_t ldx1, const void* x2, int64_t ldx2, const float* amax_x, int c1,
                          int B, int H, int W, int Cout, int Cin, float* parts, int64_t parts_floats, float* dW, int accumulate, const void* zeros, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Shifted-window attention core (WindowAttention.forward, lib/backbone.py:123-140; mask :634-652).
 * qkv: [nwin*N][3C] in windowed row order (columns s*C + h*32 + d, s in {q,k,v}), head_dim = C/heads.
 * out: [nwin*N][C].  bias: dense fp32 [heads][N][bias_ld] (from lavt_relpos_expand; bias_ld = N rounded up to a multiple
 * of 16 -- 64 for 7x7, 160 for 12x12, 416 for the 392-token 8x7x7 video windows -- lets the bf16 MFMA kernels run; padding columns hold -1e30).  region: optional int8
 * [nw_img][N] region ids of the shift mask (window w uses row w % nw_img); unequal ids add -100.
 * lse: fp32 [nwin][heads][N] log-sum-exp of each score row (saved for backward).
 * table = relative_position_bias_table fp32 [(2wd-1)(2wh-1)(2ww-1)][heads] with the FULL window shape (wd, wh, ww) (wd = 1 for the 2-D Swin;
 *   a clipped video window has N < wd*wh*ww tokens and uses the top-left block of the index matrix).  The bf16 MFMA kernels (N <= 400: Q, K, V(, dO) of a whole window in LDS;
 *   WindowAttention3D.forward, lib/video_swin_transformer.py:137-168, with its default 8x7x7 window included) keep the head's table column in LDS and never read the dense bias: `bias` may then be NULL; lavt_attn_uses_table(dtype, N) tells.
 * Backward: dqkv [nwin*N][3C] (every element written); dtable accumulates the table gradient.  bf16 MFMA kernel: every (window, head)
 *   writes its fp32 dS slab into ws (lavt_window_attn_bwd_ws floats; bias_ld = 64 / 160 / 416) with plain stores;
 *   two small kernels sum the slabs over windows, bin them by relative-position index and add the result to dtable (LDS float atomics inside the attention
 *   kernel measured 37 of 57 us per window-head).  Exact-fp32 kernel: global atomics into dtable; needs the dense `bias`.
 * ------------------------------------------------------------------------------------------- */
int lavt_window_attn_fwd(int dtype, const void* qkv, const float* bias, int bias_ld, const int8_t* region, int nw_img, void* out,
                         float* lse, const float* table, int wd, int wh, int ww, int nwin, int N, int heads, int head_dim, float scale,
                         void* stream);
int lavt_window_attn_bwd(int dtype, const void* qkv, const float* bias, int bias_ld, const int8_t* region, int nw_img,
                         const void* out, const void* dout, const float* lse, void* dqkv, const float* table, float* dtable,
                         float* ws, int64_t ws_floats, float* parts, int wd, int wh, int ww, int nwin, int N, int heads, int head_dim, float scale,
                         void* stream);
/* Chained table-gradient binning (ABI v5, bf16 MFMA path, deferred form).  The binning of a launch's dS slabs is needed by nothing before the end of
 * backward, yet as its own launch it sat on the critical chain (7.5 us x 24 per Swin-B step).  lavt_window_attn_bwd_chained does not launch it: it
 * describes it in *mine, and runs the job `prev` describes -- the binning of an EARLIER launch, whose slabs must still be alive -- as extra
 * workgroups of this launch (they fill CU slots the (window, head) workgroups leave free).  lavt_attn_dtable_run launches a job on its own (the last
 * one of a backward pass).  parts as in lavt_window_attn_bwd. */
typedef struct {
    const void* slab;
    float* part;
    int32_t slab_ld, wd, wh, ww, nwin, N, heads, rows_per_block, win_per_group, gx, gz;
} lavt_dtable_job_t;
int lavt_window_attn_bwd_chained(int dtype, const void* qkv, int bias_ld, const int8_t* region, int nw_img, const void* out, const void* dout,
                                 const float* lse, void* dqkv, const float* table, float* ws, int64_t ws_floats, float* parts, int wd, int wh, int ww,
                                 int nwin, int N, int heads, int head_dim, float scale, const lavt_dtable_job_t* prev, lavt_dtable_job_t* mine, void* stream);
int lavt_attn_dtable_run(const lavt_dtable_job_t* job, void* stream);
/* 1 when lavt_window_attn_fwd/bwd take the bias from the table for this (dtype, N) -- no dense bias / lavt_relpos_expand needed */
int lavt_attn_uses_table(int dtype, int N);
/* Deferred table gradient (bf16 MFMA path): with `parts` != NULL (a caller-owned buffer of lavt_window_attn_bwd_pieces(...) * heads * (2wd-1)(2wh-1)(2ww-1)
 * floats that outlives the call) lavt_window_attn_bwd leaves its per-workgroup table histograms there and does not touch dtable; one
 * lavt_attn_dtable_finish_multi launch later adds the histograms of any number of layers into their table gradients:
 * desc = device int64 [n][5] rows {parts, pieces, heads, R, dtable}; max_R / max_heads size the grid. */
int lavt_window_attn_bwd_pieces(int dtype, int nwin, int N, int heads, int bias_ld);
int lavt_attn_dtable_finish_multi(const int64_t* desc, int n, int max_R, int max_heads, void* stream);
/* the same launch sized for the (layer, head) pairs that exist: total_heads = sum over the n <= 64 layers of their head counts */
int lavt_attn_dtable_finish_multi_compact(const int64_t* desc, int n, int max_R, int total_heads, void* stream);
/* floats of scratch (`ws`) lavt_window_attn_bwd wants for these shapes: dS slabs + per-workgroup table histograms (0 for the exact-fp32 kernel) */
int64_t lavt_window_attn_bwd_ws(int dtype, int nwin, int N, int heads, int bias_ld, int wd, int wh, int ww);

/* relative_position_bias_table[(2wd-1)(2wh-1)(2ww-1)][heads] -> dense bias[heads][N][ld]   (wd = 1 for the 2-D Swin; N <= wd*wh*ww tokens) (lib/backbone.py:89-103,125-127)
 * and its transpose (dense gradient -> table gradient, deterministic, accumulates into dtable). */
int lavt_relpos_expand(const float* table, float* dense, int wd, int wh, int ww, int N, int heads, int ld, void* stream);
int lavt_relpos_reduce(const float* ddense, float* dtable, int wd, int wh, int ww, int N, int heads, int ld, void* stream);
/* Row softmax of attention scores for windows too large for the fused kernels (Video-Swin N = 392 / 1152; WindowAttention3D.forward,
 * lib/video_swin_transformer.py:147-161): p[row][j] = softmax_j(s[row][j] + bias[i][j] + mask), i = row % rpw, window = row / rpw (rpw >= N rows
 * per window, rows i >= N are padding and give p = 0); s already holds scale * q k^T (lavt_gemm_nt); padding columns of p (j >= N, up to ld)
 * are written as 0.  heads >= 1: rows come in blocks of rpw ordered (window, head): block b is window b / heads with bias[b % heads] (bias is
 * [heads][N][bias_ld]); heads = 1 for one head at a time.  Backward overwrites dp with ds. */
int lavt_attn_softmax_fwd(int dtype, const void* s, const float* bias, int bias_ld, const int8_t* region, int nw_img, void* p,
                          int64_t rows, int rpw, int N, int ld, int heads, void* stream);
int lavt_attn_softmax_bwd(int dtype, const void* p, void* dp, int64_t rows, int N, int ld, void* stream);
/* dense bias gradient of that path: out[h][i][j] (fp32 [heads][N][ld]) = sum_w ds[w][h][i][j], ds [nwin][heads][rpw][ld] in `dtype` (the `.sum(0)` over
 * windows that autograd performs for the broadcast `attn + relative_position_bias.unsqueeze(0)`, lib/video_swin_transformer.py:151-153);
 * feed the result to lavt_relpos_reduce. */
int lavt_attn_dbias_sum(int dtype, const void* ds, float* out, int nwin, int heads, int N, int rpw, int ld, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm over the channel dimension (nn.LayerNorm, eps 1e-5; lib/backbone.py:201,243,285,328,510).
 * x,y: [rows][C].  mean/rstd: fp32 [rows] saved for backward.  gamma/beta fp32 [C].
 * gather != NULL: row r of the INPUT is assembled from 4 source rows gather[4r..4r+3] of C/4 channels each
 * (-1 = zeros): the 2x2 neighbour concat of PatchMerging (lib/backbone.py:278-283) fused into its LayerNorm.
 * Backward accumulates dgamma/dbeta (fp32 atomics); with gather, dx rows are scattered back (every source
 * row appears exactly once).
 * ------------------------------------------------------------------------------------------- */
int lavt_layernorm_fwd(int dtype, const void* x, const int32_t* gather, const float* gamma, const float* beta, void* y,
                       float* mean, float* rstd, int rows, int C, float eps, void* stream);
int lavt_layernorm_bwd(int dtype, const void* dy, const void* x, const int32_t* gather, const float* gamma,
                       const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta, float* ws, int64_t ws_floats,
                       const void* dres, int rows, int C, void* stream);
/* Deferred form: the weight / bias gradients of a LayerNorm feed nothing but the gradient buffer, so their partial sums need not be reduced
 * inside backward.  lavt_layernorm_bwd_partial writes ONLY the per-workgroup partials (ws: >= lavt_layernorm_bwd_blocks(...) * 2 * C floats,
 * [block][dgamma | dbeta][C]) besides dx; lavt_reduce_partials_multi later adds any number of such partial sets into their gradients in ONE
 * launch: desc = device int64 [n][5] rows {partials, blocks, C, dgamma, dbeta}.  (60 two-kernel reductions per Swin-B step -> 1 launch.) */
int lavt_layernorm_bwd_blocks(int dtype, int rows, int C);
int lavt_layernorm_bwd_partial(int dtype, const void* dy, const void* x, const int32_t* gather, const float* gamma, const float* mean,
                               const float* rstd, void* dx, float* ws, int64_t ws_floats, const void* dres, int rows, int C, void* stream);
/* LayerNorm backward that also writes the LayerNorm OUTPUT xn = xhat * gamma + beta (ABI v4): for a forward that folded the norm into the consumer's GEMM
 * (lavt_gemm_nt.ln_wsum) and never materialised it; the consumer's weight gradient reads xn.  _partial_xn = the deferred-reduction form. */
int lavt_layernorm_bwd_partial_xn(int dtype, const void* dy, const void* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                                  void* dx, void* xn, float* ws, int64_t ws_floats, const void* dres, int rows, int C, void* stream);
/* ABI v7: the same launch + the table-gradient binning job of an earlier lavt_window_attn_bwd_chained launch as rider workgroups (`job` as that call
 * returned it in *mine).  xn == beta == NULL: the plain form (lavt_layernorm_bwd_partial without gather).  Returns 1 and launches NOTHING when this
 * LayerNorm geometry has no rider form: issue the two launches separately. */
int lavt_layernorm_bwd_partial_xn_dtable(int dtype, const void* dy, const void* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                                         void* dx, void* xn, float* ws, int64_t ws_floats, const void* dres, int rows, int C, const lavt_dtable_job_t* job, void* stream);
int lavt_layernorm_bwd_xn(int dtype, const void* dy, const void* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                          void* dx, void* xn, float* dgamma, float* dbeta, float* ws, int64_t ws_floats, const void* dres, int rows, int C, void* stream);
int lavt_reduce_partials_multi(const int64_t* desc, int n, int total_column_blocks, void* stream); /* total_column_blocks = sum over the sets of lavt_reduce_partials_column_blocks(C) (<= 0: unknown; a larger number -- e.g. the ceil(2 C / 32) of the first ABI v6 builds -- only starts idle workgroups) */
int lavt_reduce_partials_column_blocks(int C);
/* dres (optional, [rows][C], not with gather): gradient of the residual stream that bypassed the LayerNorm (x -> LN(x) and x -> + ...):
 * dx = LN'(dy) + dres in the same pass, instead of a separate element-wise add of the two gradients of x */

/* ---------------------------------------------------------------------------------------------
 * Per-channel statistics over rows, and the normalisations built on them:
 *   InstanceNorm1d over all H*W positions (PWAM f_query / W, lib/backbone.py:1311-1327): groups = batch;
 *   BatchNorm2d (+ReLU) of the decoder (lib/mask_predictor.py:19-37; SyncBN = all-reduce of sums between the calls).
 * lavt_colstats: x [groups][rows][C] -> sum = sum_r x and m2 = sum_r (x - mean_group)^2, fp32 [groups][C] (with scratch the call clears them itself; without it pass zeroed buffers).
 *   The second moment is centred (accumulated about the group's first row, then re-centred), never E[x^2]-E[x]^2.
 *   Combining ranks (SyncBN): all-reduce sum -> global mean; m2_r += rows_r*(mean_r - mean)^2; all-reduce m2.
 * lavt_stats_finalize: mean = sum/count, var = m2/count (biased), rstd = 1/sqrt(var+eps); optional running-stat update (unbiased var).
 * ws / ws_floats (here and in lavt_norm_bwd_stats): optional fp32 scratch of >= 1025*groups*2*C floats; with it the per-workgroup partial
 *   sums are written out and reduced by a second tiny kernel instead of contending atomics on the same C addresses.  lavt_layernorm_bwd's
 *   scratch is lavt_layernorm_bwd_blocks(dtype, rows, C) * 2 * C floats (up to 2048 blocks); a smaller one selects the atomic form (slow:
 *   hundreds of workgroups adding to the same C addresses).
 * lavt_norm_apply: y = ((x-mean)*rstd*gamma + beta) (*mul) with optional ReLU; mean/rstd [groups][C]; gamma/beta/mul optional.
 * lavt_norm_bwd_stats: s1 = sum(g), s2 = sum(g*xhat) (cleared by the call when scratch is given, else added to the zeroed buffers passed in) with g = dy (*mul) masked by relu (y>0);  [groups][C].
 *   With relu, gamma / beta given and no mul (BatchNorm + ReLU of the decoder) the mask is recomputed from x with lavt_norm_apply's own expression and y is not read
 *   (lavt_norm_bwd_apply likewise): two of the seven map passes of the BatchNorm backward less.
 * lavt_norm_bwd_apply: dx = gamma*rstd*(g - s1/n - xhat*s2/n); dmul = dy*xhat_affine (optional).
 * ------------------------------------------------------------------------------------------- */
int lavt_colstats(int dtype, const void* x, float* sum, float* m2, float* ws, int64_t ws_floats, int groups, int rows, int C, void* stream);
/* SyncBatchNorm forward after the all-gather of every rank's (sum, m2) pair (allst: [world][2][C], `rows_per_rank` rows each): parallel-variance
 * combination + lavt_stats_finalize in one launch (mean / rstd of the global batch, running estimates with the unbiased variance) */
int lavt_syncbn_combine(const float* allst, int world, float rows_per_rank, float eps, float* mean, float* rstd, float* running_mean,
                        float* running_var, float momentum, int C, void* stream);
/* lavt_colstats + lavt_stats_finalize in two launches instead of four, for the cases where nothing sits between the sums and their use
 * (InstanceNorm; BatchNorm without a cross-rank exchange): mean / rstd [groups][C] directly, optional running-statistics update (groups == 1). */
int lavt_colstats_meanrstd(int dtype, const void* x, float* mean, float* rstd, float* ws, int64_t ws_floats, int groups, int rows, int C, float eps,
                           float* running_mean, float* running_var, float momentum, void* stream);
int lavt_stats_finalize(const float* sum, const float* m2, float count, float eps, float* mean, float* rstd,
                        float* running_mean, float* running_var, float momentum, int n, void* stream);
int lavt_norm_apply(int dtype, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                    const void* mul, int relu, void* y, int groups, int rows, int C, void* stream);
int lavt_norm_bwd_stats(int dtype, const void* dy, const void* x, const void* y, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, const void* mul, int relu, float* s1, float* s2,
                        float* ws, int64_t ws_floats, int groups, int rows, int C, void* stream);
int lavt_norm_bwd_apply(int dtype, const void* dy, const void* x, const void* y, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, const void* mul, int relu, const float* s1, const float* s2,
                        float count, void* dx, void* dmul, int groups, int rows, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Element-wise pieces.
 * ------------------------------------------------------------------------------------------- */
/* dx = dy * act'(pre)   (GELU of Mlp / PWAM projections, ReLU of res_gate) */
int lavt_act_bwd(int dtype, int act, const void* dy, const void* pre, void* dx, int64_t n, void* stream);
/* language gate, lib/backbone.py:669:  xo = x + tanh(gpre) * r ;  backward gives dgpre, dr (+= into dr_acc semantics: written) */
/* O(batch) glue of a forward, one launch each (ABI v5).  lavt_lang_mask: l_mask [B][n_l] (float32, or int64 with is_int64 = 1) -> mask_rows [B * n_l]
 * (float) and maskbias [B][ld] = 1e4 * m - 1e4, -1e4 beyond n_l (lib/backbone.py:1360).  lavt_droppath_factors: f[n][B] = floor(keep[n] + u[n][B]) / keep[n]
 * (timm drop_path, the reference's DropPath: lib/backbone.py:6, 240-245) from one uniform draw u. */
int lavt_lang_mask(const void* l_mask, int is_int64, float* mask_rows, float* maskbias, int B, int n_l, int ld, void* stream);
int lavt_droppath_factors(const float* u, const float* keep, float* f, int n, int B, void* stream);
/* the same factors with the uniform draw made on the device: u[e] = (Philox4x32-10(key = state[0], counter = (state[1], e, 0))[0] >> 8) * 2^-24; the
 *   kernel advances state[1] (two uint64 in device memory: seed, draw counter), so a captured step draws fresh factors on every replay */
int lavt_droppath_draw(void* state, const float* keep, float* f, int n, int B, void* stream);
int lavt_gate_fwd(int dtype, const void* x, const void* gpre, const void* r, void* xo, int64_t n, void* stream);
int lavt_gate_bwd(int dtype, const void* dxo, const void* gpre, const void* r, const void* dr_add, void* dgpre, void* dr, int64_t n, void* stream); /* dr = dxo * tanh(gpre) (+ dr_add) */

/* ---------------------------------------------------------------------------------------------
 * One-kernel W-MSA / SW-MSA forward (ABI v4, bf16): norm1 + pad / roll / window_partition + qkv + relative-position-bias attention
 * (SwinTransformerBlock.forward lib/backbone.py:201-217, WindowAttention.forward :113-140).  One workgroup per (window, head): the window's
 * token rows (row map, -1 = padded token) and the head's 96 weight rows stream through an LDS-DMA ring, LayerNorm is applied algebraically
 * (raw rows x gamma-folded weights, row statistics from the resident tiles, epilogue rstd (acc - mu wsum) + biasp; padded tokens get the plain
 * bias: the reference pads after norm1), q / k / v stay in LDS for the attention core.  Side outputs for backward: qkv [nwin*N][3C], the
 * LayerNorm output xn [tokens][C] and its row statistics.  Needs C = 32 * heads, C % 64 == 0, ws * ws <= 160.
 *   lavt_ln_fold: Wg[n][k] = bf16(gamma_k W[n][k]), wsum[n] = sum_k Wg[n][k], biasp[n] = bias_n + sum_k beta_k W[n][k]   (per weight update)
 * ------------------------------------------------------------------------------------------- */
int lavt_ln_fold(const float* W, const float* gamma, const float* beta, const float* bias, void* Wg, float* wsum, float* biasp, int N, int K, void* stream);
/* every fold of a model in one launch: desc int64 [count][9] = {W, gamma, beta, bias, Wg, wsum, biasp, N, K} in device memory */
int lavt_ln_fold_multi(const int64_t* desc, int count, void* stream);
int lavt_wmsa_fwd(const void* x, const int32_t* wmap, const void* Wg, const float* wsum, const float* biasp, const float* bias, const float* gamma,
                  const float* beta, const float* table, const int8_t* region, int nw_img, void* out, float* lse, void* qkv, void* xn, float* mean,
                  float* rstd, const void* zeros, int ws, int nwin, int N, int heads, int C, float eps, float scale, void* stream);
/* The same launch with a zero-fill RIDER (ABI v5): fill_bytes bytes at fill (16-byte aligned, multiples of 16) are set to zero by extra workgroups appended
 * to the launch's grid.  The step harness zeroes its flat gradient buffer this way, a slice per stage-2 block: the fill is needed by nothing before
 * backward, and as its own launch (58 us at the HBM rate for Swin-B's 475 MB) it headed the captured chain. */
int lavt_wmsa_fwd_rider(const void* x, const int32_t* wmap, const void* Wg, const float* wsum, const float* biasp, const float* bias, const float* gamma,
                        const float* beta, const float* table, const int8_t* region, int nw_img, void* out, float* lse, void* qkv, void* xn, float* mean,
                        float* rstd, const void* zeros, int ws, int nwin, int N, int heads, int C, float eps, float scale, void* fill, int64_t fill_bytes,
                        void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused PWAM (ABI v4, bf16).  Replaces, with the GEMMs above, PWAM.forward (lib/backbone.py:1265-1278),
 * SpatialImageLanguageAttention.forward (:1329-1372) and the language gate (:604-611, :669) and their autograd backward.
 * The instance norm of the query folds into the keys and the W projection collapses onto the <= 32 word probabilities
 * (w = P (V Wo^T) + bo, so IN(w) = (P - Pbar) VW' from the word statistics alone): see csrc/pwam.hip and tools/pwam_algebra_check.py.
 * Word axes are padded to 32 slots; B samples of T pixels each, rows [B*T]; C % 32 == 0.
 *   lavt_pwam_words_fwd: P[row][32] = softmax_{j < n_l}(alpha * IN_T(q) K^T + maskbias)   (q raw, mean / rstd [B][C] of q over the T pixels)
 *   lavt_pwam_lang_fwd:  from V [B][32][ldv], Wo [C][C] (bf16 compute copy), PP = P^T P [B][32][32], sumP [B][32]:
 *                        VW' channel-major VWc [B][C][32] and word-major VWw [B][32][C] (bf16), beta = -Pbar VW' [B][C], rw = rstd of w [B][C],
 *                        Pbar [B][32], Cov [B][32][32]
 *   lavt_pwam_mix mode 0: out0 = GELU(X + xbias) * (Wd Wc^T + v0)      (mm = vis * IN(w);  Wd = P, Wc = VWc, v0 = beta, X = x Wv^T, xbias = bv)
 *                 mode 1: out0 = D * what * GELU'(X), out1 = D * GELU(X)  (d vpre, d what;  D = d mm)
 *                 mode 2: out0 = Wd Wc^T + v0 - X * v1                  (dq;  Wd = dS, Wc = K''^T, X = q)
 *   lavt_pwam_lang_bwd1: HT = dwhat^T P [B][C][32], s = colsum(dwhat) [B][C] -> dVW [B*32][C] (bf16), partial records of Q [32][32] and u [32] in Qp
 *   lavt_pwam_words_bwd: dS[row][32] = P * (dP - sum_j P_j dP_j),  dP = dwhat VW'^T - P Q + (Pbar Q - u)
 *   lavt_pwam_lang_bwd2: G = dS^T q [B][32][C], sdS = colsum(dS) [B][32] -> dK [B*32][lddk] (bf16), K''^T [B][C][32] (bf16), c0, c1 [B][C]
 * ------------------------------------------------------------------------------------------- */
int lavt_pwam_words_fwd(const void* q, int64_t ldq, const void* K, int64_t ldk, const float* mean, const float* rstd, const float* maskbias,
                        void* P, int B, int T, int C, int n_l, float alpha, void* stream);
/* ABI v7: the same launch also leaves the second moments of the word probabilities as per-workgroup records -- rec [B][lavt_pwam_words_records(B, T, C)][1056]
 * floats: [32][32] P^T P | [32] colsum(P) over the workgroup's rows (of the bf16 P it stored) -- which lavt_pwam_lang_fwd_records adds in index order:
 * replaces the P^T P launch and its reduction launch.  rec == NULL: as lavt_pwam_words_fwd. */
int lavt_pwam_words_fwd_moments(const void* q, int64_t ldq, const void* K, int64_t ldk, const float* mean, const float* rstd, const float* maskbias,
                                void* P, float* rec, int B, int T, int C, int n_l, float alpha, void* stream);
int lavt_pwam_words_records(int B, int T, int C);
int lavt_pwam_words_bwd(const void* dwhat, int64_t ldx, const void* VWw, const float* Qp, const float* pbar, const void* P,
                        void* dS, int B, int T, int C, void* stream);
int lavt_pwam_q_parts(int C); /* records per sample in Qp: [B][records][1024 Q | 32 u] floats, written by lavt_pwam_lang_bwd1, summed in fixed order by lavt_pwam_words_bwd */
int lavt_pwam_mix(int mode, const void* Wd, const void* Wc, const float* v0, const float* v1, const float* xbias, const void* X, int64_t ldx, const void* D, int64_t ldd,
                  void* out0, int64_t ld0, void* out1, int64_t ld1, int B, int T, int C, void* stream);
int lavt_pwam_lang_fwd(const void* V, int64_t ldv, const void* Wo, const float* PP, const float* sumP, void* VWc, void* VWw, float* beta, float* rw,
                       float* pbar, float* cov, int B, int T, int C, float eps, void* stream);
/* ABI v7: lavt_pwam_lang_fwd with the moments given as records (PP == NULL): rec [B][nrec][1056] of lavt_pwam_words_fwd_moments */
int lavt_pwam_lang_fwd_records(const void* V, int64_t ldv, const void* Wo, const float* PP, const float* sumP, const float* rec, int nrec, void* VWc, void* VWw,
                               float* beta, float* rw, float* pbar, float* cov, int B, int T, int C, float eps, void* stream);
int lavt_pwam_lang_bwd1(const float* HT, const float* s, const void* VWc, const float* rw, const float* pbar, const float* cov, void* dVW, float* Qp,
                        int B, int T, int C, void* stream);
/* ABI v7: lavt_pwam_mix(1) on workgroups that own one 64-channel group, with H^T = dwhat^T P and colsum(dwhat) as a by-product when rec != NULL:
 * rec [B][lavt_pwam_mix1_records(B, T, C)][C * 33] floats (per record C x 32 H^T, then C sums), added in record order by lavt_pwam_lang_bwd1_records
 * (HT == NULL): replaces the H launch and its reduction launch. */
int lavt_pwam_mix1(const void* P, const void* VWc, const float* beta, const float* xbias, const void* X, int64_t ldx, const void* D, int64_t ldd, void* dvpre,
                   int64_t ld0, void* dwhat, int64_t ld1, float* rec, int B, int T, int C, void* stream);
int lavt_pwam_mix1_records(int B, int T, int C);
int lavt_pwam_lang_bwd1_records(const float* HT, const float* s, const float* rec, int nrec, const void* VWc, const float* rw, const float* pbar, const float* cov,
                                void* dVW, float* Qp, int B, int T, int C, void* stream);
int lavt_pwam_lang_bwd2(const float* G, const float* sdS, const void* K, int64_t ldk, const float* mean, const float* rstd, void* dK, int64_t lddk, void* K2c,
                        float* c0, float* c1, int B, int T, int C, float alpha, void* stream);
/* masked softmax over the (padded) word axis of PWAM scores, lib/backbone.py:1358-1361.
 * s,p: [rows][ld] (only the first n_l columns are real; p's padding columns are written as 0). */
int lavt_rowsoftmax_fwd(int dtype, const void* s, void* p, int64_t rows, int n_l, int ld, void* stream);
int lavt_rowsoftmax_bwd(int dtype, const void* p, const void* dp, void* ds, int64_t rows, int n_l, int ld, void* stream);
/* bilinear resize, align_corners=True, NHWC (F.interpolate in lib/mask_predictor.py:59,69,80) */
int lavt_bilinear_fwd(int dtype, const void* x, void* y, int B, int Hi, int Wi, int Ho, int Wo, int C, void* stream);
/* Producers that write the e4m3 twin of their bf16 output (round 5; BASELINE.json configs[4]): q = e4m3(bf16(y) * 448 / *amax_prev) (scale 1 while
 * *amax_prev <= 0) -- exactly the bytes lavt_fp8_quantize(y) would write -- and |max| of y recorded into *amax_cur by atomic max (delayed scaling):
 * the quantiser launch in front of the consuming fp8 convolution disappears.  lavt_norm_bwd_apply_amax records |max| of the stored dx into *amax
 * (zeroed by the caller once per step, lavt_fp8_advance does): the |max| pass of lavt_fp8_quantize_current disappears. */
int lavt_bilinear_fwd_q8(const void* x, void* y, void* q, const float* amax_prev, float* amax_cur, int B, int Hi, int Wi, int Ho, int Wo, int C, void* stream);
int lavt_norm_apply_q8(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, const void* mul, int relu, void* y,
                       void* q, const float* amax_prev, float* amax_cur, int groups, int rows, int C, void* stream);
int lavt_norm_bwd_apply_amax(const void* dy, const void* x, const void* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                             const void* mul, int relu, const float* s1, const float* s2, float count, void* dx, void* dmul, float* amax, int groups, int rows,
                             int C, void* stream);
int lavt_bilinear_bwd(int dtype, const void* dy, void* dx, int B, int Hi, int Wi, int Ho, int Wo, int C, void* stream);
/* final logits: NHWC [B,Hi,Wi,2] (dtype) -> NCHW fp32 [B,2,Ho,Wo] (lib/_utils.py:21) and its gradient */
int lavt_logits_up_fwd(int dtype, const void* x, float* y, int B, int Hi, int Wi, int Ho, int Wo, void* stream);
int lavt_logits_up_bwd(int dtype, const float* dy, void* dx, int B, int Hi, int Wi, int Ho, int Wo, void* stream);
/* The caller's step right after the path, fused: bilinear upsample (align_corners) of the 2-class low-resolution logits
 * x NHWC [B,Hi,Wi,2] to (Ho, Wo) + class-weighted cross-entropy against target int64 [B,Ho,Wo] (losses.py:7-11, weights (w0, w1);
 * targets other than 0 / 1 are ignored) + the I / U pixel counts of train.py:64-76 (prediction = argmax, class 0 on ties).
 * The [B,2,Ho,Wo] logits are never materialised.  out4 = {loss, sum of weights, I, U} (device, fp32); ws: >= 4*2048 floats of scratch.
 * Backward: dx NHWC [B,Hi,Wi,2] = dloss[0] (device scalar, NULL = 1) * d loss / d x, recomputed from x (gather form, no atomics). */
int lavt_upsample_ce_fwd(int dtype, const void* x, const int64_t* target, float w0, float w1, float* ws, int64_t ws_floats, float* out4,
                         int B, int Hi, int Wi, int Ho, int Wo, void* stream);
int lavt_upsample_ce_bwd(int dtype, const void* x, const int64_t* target, float w0, float w1, const float* out4, const float* dloss,
                         void* dx, int B, int Hi, int Wi, int Ho, int Wo, void* stream);
/* Fused bilinear upsample (align_corners) + MultiClassDiceLoss (reference losses.py:38-77, `--loss mc_dice`, train.py:703-704) on the 2-class
 * low-resolution logits x NHWC [B,Hi,Wi,2]; target int64 [B,Ho,Wo] in {0,1}.  stats (device fp32, 2 + 6*B floats) = {loss, 0, then per sample
 * {I0, I1, sum p0^2, sum p1^2, #[t==0], #[t==1]}}; ws: >= 6*256*B floats of scratch.  Backward: dx = dloss[0] (NULL = 1) * d loss / d x, gather form.
 * With Hi == Ho and Wi == Wo the upsample is the identity: the same entry points serve the un-fused criterion on full-resolution logits. */
int lavt_upsample_dice_fwd(int dtype, const void* x, const int64_t* target, float* ws, int64_t ws_floats, float* stats,
                           int B, int Hi, int Wi, int Ho, int Wo, void* stream);
int lavt_upsample_dice_bwd(int dtype, const void* x, const int64_t* target, const float* stats, const float* dloss, void* dx,
                           int B, int Hi, int Wi, int Ho, int Wo, void* stream);
/* ---- fp8 (OCP e4m3) operand preparation for lavt_gemm_nt(dtype = LAVT_FP8) ----
 * lavt_fp8_quantize: activations, delayed scaling: dst[i] = e4m3(clamp(src[i] * s, +-448)), s = *amax_prev > 0 ? 448 / *amax_prev : 1 (the |max| seen
 *   in the PREVIOUS step; the GEMM reads the same float through deq_a); max |src| of THIS call is folded into *amax_cur (atomic max).
 * lavt_fp8_advance: start of a step, for n slots: prev[i] = cur[i] > 0 ? cur[i] : prev[i]; cur[i] = 0.
 * lavt_fp8_quantize_weight: fp32 parameter -> e4m3 with CURRENT scaling (*amax is computed here, over the whole tensor); taps > 1 re-packs a
 *   [Cout][Cin][taps] convolution weight as [Cout][taps][Cin] (the implicit-GEMM layout) on the way. */
int lavt_fp8_quantize(int src_dtype, const void* src, void* dst, int64_t n, const float* amax_prev, float* amax_cur, void* stream);
int lavt_fp8_advance(float* amax_prev, float* amax_cur, int n, void* stream);
/* lavt_fp8_quantize_current: CURRENT scaling for tensors whose range moves from step to step (the dY operand of the e4m3 data gradients): *amax = max |src|
 *   of THIS tensor (computed here, one extra read pass), dst[i] = e4m3(src[i] * 448 / *amax); the GEMM reads *amax through deq_a.  No calibration step. */
int lavt_fp8_quantize_current(int src_dtype, const void* src, void* dst, int64_t n, float* amax, void* stream);
int lavt_fp8_quantize_weight(const float* src, void* dst, float* amax, int cout, int cin, int taps, void* stream);
/* lavt_fp8_quantize_weight_t: the same scaling, packed TRANSPOSED as [Cin][taps][Cout] -- the k-contiguous weight operand of the e4m3 data gradient
 *   (reference lib/mask_predictor.py:60-97, backward of the 3x3 convolutions: dX[m][ci] = sum over (tap, co) of dY[m - tap][co] * W[co][ci][tap]). */
int lavt_fp8_quantize_weight_t(const float* src, void* dst, float* amax, int cout, int cin, int taps, void* stream);
/* classifier head conv1_1: 1x1 conv hidden->2 with bias (lib/mask_predictor.py:50,99) */
int lavt_cls_head_fwd(int dtype, const void* x, const float* w, const float* b, void* y, int64_t rows, int C, void* stream);
int lavt_cls_head_bwd(int dtype, const void* x, const void* dy, const float* w, void* dx, float* dw, float* db,
                      int64_t rows, int C, void* stream);
/* the same backward with the weight / bias gradient sums left as one record per workgroup (pw [blocks][2 C], pb [blocks][2], blocks =
 *   lavt_cls_head_bwd_blocks) for lavt_reduce_partials_multi: no global atomics, run-to-run identical sums */
int lavt_cls_head_bwd_blocks(int dtype, int64_t rows, int C);
int lavt_cls_head_bwd_partial(int dtype, const void* x, const void* dy, const float* w, void* dx, float* pw, float* pb,
                              int64_t rows, int C, void* stream);
/* PatchEmbed im2col: NCHW fp32 image -> [B*H4*W4][48] patches (zero padded to x4), lib/backbone.py:318-324;
 * col2im scatters the patch gradient back to an NCHW fp32 image gradient. */
int lavt_im2col4(int dtype, const float* img, void* cols, int B, int H, int W, void* stream);
int lavt_col2im4(int dtype, const void* dcols, float* dimg, int B, int H, int W, void* stream);
/* layout / dtype plumbing */
int lavt_cast(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t n, void* stream);
int lavt_nchw_to_nhwc(int src_dtype, const void* src, int dst_dtype, void* dst, int B, int C, int HW, void* stream);
int lavt_nhwc_to_nchw(int src_dtype, const void* src, int dst_dtype, void* dst, int B, int C, int HW, void* stream);
/* conv weight fp32 [Cout][Cin][taps] (taps = 9 for 3x3, 27 for 3x3x3, ...) -> dtype [Cout][taps][Cin] (compute copy used by lavt_gemm_nt) */
int lavt_pack_conv3x3(const float* w, int dtype, void* packed, int Cout, int Cin, int taps, void* stream);
/* gradient counterpart: dw fp32 [Cout][Cin][taps] += packed fp32 [Cout][taps][Cin] (what lavt_gemm_tn writes without c_conv_permute) */
int lavt_unpack_conv_grad(const float* packed, float* dw, int Cout, int Cin, int taps, void* stream);
/* many small fp32 -> dtype casts in one launch: desc = int64 triples (src_ptr, dst_ptr, n) on the DEVICE */
int lavt_cast_multi(const int64_t* desc, int count, int dst_dtype, void* stream);
/* The caller's optimizer step (train.py:688-700: torch.optim.AdamW, amsgrad off, + LambdaLR((1 - it/T)^0.9)) as one
 * multi-tensor launch.  desc: int64 [count][5] = {param, grad, exp_avg, exp_avg_sq (fp32 device pointers), numel}; hyper: fp32 [count][5] =
 * {base lr, weight decay, beta1, beta2, eps} (both tables in device memory); step: device fp32 scalar = optimizer steps taken so far,
 * incremented by the call (so a captured hipGraph keeps advancing its schedule); lr = base lr * (1 - step/total_steps)^power, or the
 * base lr when total_steps <= 0.  Update rule identical to torch.optim.AdamW (decoupled decay, bias-corrected moments). */
int lavt_adamw_step(const int64_t* desc, const float* hyper, int count, float* step, float total_steps, float power, void* stream);
/* The same update driven by a chunk table (ABI v4): desc int64 [count][6] = {param, grad, exp_avg, exp_avg_sq, numel, copy} -- a non-zero `copy` is the
 * parameter's bf16 compute copy in the same layout, written by the same kernel (the mixed-precision trainer's re-cast of the weights the next
 * forward reads, without a second pass over the parameters); chunks int32 [nchunks][2] = {tensor index, chunk index}: workgroup c updates elements
 * [chunk * lavt_adamw_chunk_elems(), ...) of its tensor, so every workgroup has work.  hyper, step, schedule as above. */
int lavt_adamw_chunk_elems(void);
int lavt_adamw_step_chunks(const int64_t* desc, const float* hyper, const int32_t* chunks, int nchunks, float* step, float total_steps, float power, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Text side (lavt_one / lavt_video carry BERT inside the model: lib/_utils.py:38-52; train.py:595-602; the encoder is HF transformers
 * 3.0.2 `BertModel`, absent from the reference tree).  Its Linear / LayerNorm / GELU / attention GEMMs use the entry points above.
 * lavt_bert_embed_fwd: out[r] = word[ids[r]] + type[token_type ? token_type[r] : 0] + pos[r % N]  (BertEmbeddings.forward before LayerNorm;
 *   ids / token_type int64 [rows], tables fp32, out [rows][H] in `dtype`).  lavt_bert_embed_bwd scatter-adds dy into the three fp32 table
 *   gradients (atomics; the buffers must hold the running sums, e.g. zeros).
 * lavt_dropout: y = (keep ? x * scale : 0) + residual (nn.Dropout in training with a caller-drawn uint8 keep mask, scale = 1/(1-p); residual
 *   optional: the `dropout(dense(h)) + input` of BertSelfOutput / BertOutput); without residual it is its own backward. */
int lavt_bert_embed_fwd(int dtype, const int64_t* ids, const int64_t* token_type, const float* word, const float* pos, const float* type,
                        void* out, int rows, int N, int H, void* stream);
int lavt_bert_embed_bwd(int dtype, const void* dy, const int64_t* ids, const int64_t* token_type, float* dword, float* dpos, float* dtype_,
                        int rows, int N, int H, void* stream);
int lavt_dropout(int dtype, const void* x, const uint8_t* keep, float scale, const void* residual, void* y, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LAVT_HIP_H */
