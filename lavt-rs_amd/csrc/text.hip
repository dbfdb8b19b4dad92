// Text side of lavt_one / lavt_video: the BERT-base encoder the reference builds with
// `BertModel.from_pretrained(args.ck_bert)` (lib/_utils.py:38-52; train.py:595-602; HF transformers 3.0.2 `modeling_bert.py`).
// Its Linear / LayerNorm / GELU / attention GEMMs run on the kernels of the visual path; this file adds what only the text side needs:
// the embedding sum (word + position + token type) with its scatter-add backward, and inverted dropout with a caller-drawn keep mask.
// 20-22 tokens per sentence: latency-bound, a wave per token row, 16-byte accesses.
#include "common.h"

namespace {

// out[r][:] = word[ids[r]][:] + pos[r % N][:] + type[tt ? tt[r] : 0][:]          (BertEmbeddings.forward before LayerNorm)
template <typename T>
__global__ __launch_bounds__(256) void bert_embed_fwd_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ tt, const float* __restrict__ word,
                                                             const float* __restrict__ pos, const float* __restrict__ type, T* __restrict__ out,
                                                             int rows, int N, int H) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* w = word + ids[row] * (int64_t)H;
    const float* p = pos + (int64_t)(row % N) * H;
    const float* t = type + (tt ? tt[row] : 0) * (int64_t)H;
    for (int c = lane * 4; c < H; c += 256) {
        const float4 a = *reinterpret_cast<const float4*>(w + c), b = *reinterpret_cast<const float4*>(p + c), d = *reinterpret_cast<const float4*>(t + c);
        T* o = out + (int64_t)row * H + c;
        // (word + type) + position: the order of BertEmbeddings.forward
        o[0] = from_f<T>((a.x + d.x) + b.x); o[1] = from_f<T>((a.y + d.y) + b.y); o[2] = from_f<T>((a.z + d.z) + b.z); o[3] = from_f<T>((a.w + d.w) + b.w);
    }
}

// dword[ids[r]] += dy[r], dpos[r % N] += dy[r], dtype[tt[r]] += dy[r]  (fp32 atomics: a few dozen rows, repeated ids / positions collide)
template <typename T>
__global__ __launch_bounds__(256) void bert_embed_bwd_kernel(const T* __restrict__ dy, const int64_t* __restrict__ ids, const int64_t* __restrict__ tt,
                                                             float* __restrict__ dword, float* __restrict__ dpos, float* __restrict__ dtype_, int rows, int N, int H) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* w = dword + ids[row] * (int64_t)H;
    float* p = dpos + (int64_t)(row % N) * H;
    float* t = dtype_ + (tt ? tt[row] : 0) * (int64_t)H;
    for (int c = lane; c < H; c += 64) {
        const float g = to_f<T>(dy[(int64_t)row * H + c]);
        atomicAdd(w + c, g);
        atomicAdd(p + c, g);
        atomicAdd(t + c, g);
    }
}

// y = x * keep * scale (+ r)   (nn.Dropout in training: keep ~ Bernoulli(1 - p) drawn by the caller, scale = 1 / (1 - p); the optional r is
// the residual BertSelfOutput / BertOutput add before their LayerNorm; backward of x is the same map without r)
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, const uint8_t* __restrict__ keep, float scale, const T* __restrict__ r,
                                                      T* __restrict__ y, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = keep[i] ? to_f<T>(x[i]) * scale : 0.f;
        y[i] = from_f<T>(r ? v + to_f<T>(r[i]) : v);
    }
}

}  // namespace

#define DISPATCH_T(dtype, NAME, ...)                                   \
    if (dtype == LAVT_F32) { using T = float; __VA_ARGS__; }           \
    else if (dtype == LAVT_BF16) { using T = bf16; __VA_ARGS__; }      \
    else { lavt_set_error(NAME ": bad dtype %d", dtype); return LAVT_ERR_INVALID; }
#define ST reinterpret_cast<hipStream_t>(stream)

extern "C" int lavt_bert_embed_fwd(int dtype, const int64_t* ids, const int64_t* token_type, const float* word, const float* pos, const float* type,
                                   void* out, int rows, int N, int H, void* stream) {
    LAVT_CHECK_ARG(ids && word && pos && type && out && rows > 0 && N > 0 && H > 0 && H % 4 == 0, "lavt_bert_embed_fwd: bad arguments");
    DISPATCH_T(dtype, "lavt_bert_embed_fwd", hipLaunchKernelGGL(bert_embed_fwd_kernel<T>, dim3((rows + 3) / 4), dim3(256), 0, ST, ids, token_type, word, pos, type, (T*)out, rows, N, H));
    LAVT_CHECK_LAUNCH("lavt_bert_embed_fwd");
    return LAVT_OK;
}

extern "C" int lavt_bert_embed_bwd(int dtype, const void* dy, const int64_t* ids, const int64_t* token_type, float* dword, float* dpos, float* dtype_,
                                   int rows, int N, int H, void* stream) {
    LAVT_CHECK_ARG(dy && ids && dword && dpos && dtype_ && rows > 0 && N > 0 && H > 0, "lavt_bert_embed_bwd: bad arguments");
    DISPATCH_T(dtype, "lavt_bert_embed_bwd", hipLaunchKernelGGL(bert_embed_bwd_kernel<T>, dim3((rows + 3) / 4), dim3(256), 0, ST, (const T*)dy, ids, token_type, dword, dpos, dtype_, rows, N, H));
    LAVT_CHECK_LAUNCH("lavt_bert_embed_bwd");
    return LAVT_OK;
}

extern "C" int lavt_dropout(int dtype, const void* x, const uint8_t* keep, float scale, const void* residual, void* y, int64_t n, void* stream) {
    LAVT_CHECK_ARG(x && keep && y && n > 0, "lavt_dropout: bad arguments");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    DISPATCH_T(dtype, "lavt_dropout", hipLaunchKernelGGL(dropout_kernel<T>, dim3((int)blocks), dim3(256), 0, ST, (const T*)x, keep, scale, (const T*)residual, (T*)y, n));
    LAVT_CHECK_LAUNCH("lavt_dropout");
    return LAVT_OK;
}
