// Tuning / A-B switches of liblavt_hip.  The LAVT_* environment variables below are read ONCE, at the first launch that asks (or again when the
// host calls lavt_tuning_reload(), which the test-suite does after changing one): no launch path calls getenv.  Every switch selects between
// kernels that compute the same result; none makes a kernel do less work.
#include <atomic>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace {
lavt_tuning_t g_tuning;
std::atomic<bool> g_ready{false};
std::mutex g_mu;

int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
bool env_is(const char* name, char c) {
    const char* e = getenv(name);
    return e && e[0] == c;
}

void read_tuning(lavt_tuning_t& t) {
    t.attn_simple = env_is("LAVT_ATTN_SIMPLE", '1');             // VALU attention kernels for bf16 too
    t.attn_bwd_waves = env_int("LAVT_ATTN_BWD_WAVES", 0);        // 4: the 4-wave attention backward
    t.attn_bwd_split_off = env_is("LAVT_ATTN_BWD_SPLIT", '0');   // A/B switch: the attention backward's surplus units (beyond whole rounds of 256) as whole workgroups, not task pieces
    t.unpack_tiled = !env_is("LAVT_UNPACK_TILED", '0');
    t.gemm_tile = env_int("LAVT_GEMM_TILE", 0);                  // 64 | 128 | 256 | 512: forced tile configuration (tests exercise them)
    t.tn_split = env_int("LAVT_TN_SPLIT", 0);
    t.gemm_epi_lds = env_is("LAVT_GEMM_EPI", 'l');
    t.gemm_epi_narrow = env_is("LAVT_GEMM_EPI", 'n');
    t.gemm_v2_off = env_is("LAVT_GEMM_V2", '0');
    t.tng_tile = 64; t.tng_waves = 4; t.tng_stages = 2;
    if (const char* c = getenv("LAVT_TNG_CFG")) sscanf(c, "%d,%d,%d", &t.tng_tile, &t.tng_waves, &t.tng_stages);
    t.tng_chain = env_int("LAVT_TNG_CHAIN", 48);                 // 32 / 48 / 64 / 128: video step 23.16 / 22.93 / 22.82 / 22.87 ms, image step level
    t.tng_piece = env_int("LAVT_TNG_PIECE", 8);
    t.tn_big = !env_is("LAVT_TN_BIG", '0');
    t.tn_big_min = env_int("LAVT_TN_BIG_MIN", 48);
    t.tn_target = env_int("LAVT_TN_TARGET", 768);
    t.gemm_general = getenv("LAVT_GEMM_GENERAL") != nullptr;
    t.gemm_big_long = env_int("LAVT_GEMM_BIG_LONG", 128);
    t.gemm_stages = env_int("LAVT_GEMM_STAGES", 0);
    t.gemm_waves = env_int("LAVT_GEMM_WAVES", 8);
    t.gemm_wide = getenv("LAVT_GEMM_WIDE") != nullptr;
    t.ln_bwd_waves = env_int("LAVT_LN_BWD_WAVES", 0);
    t.tn_streamk = env_int("LAVT_TNG_STREAMK", 1);
    t.fp8_pipe_off = env_is("LAVT_FP8_PIPE", '0');               // e4m3 problems on gemm_v2.hip's K loop instead of the pipelined one (A/B switch)
    t.gemm_pipe = env_int("LAVT_GEMM_PIPE", 2);                  // gemm_nt_pipe.hip: 0 off, 1 the 256x256 tile, 2 + 128x128 tiles with K >= 1024, 3 + every 128x128 problem
    t.conv_tail_off = env_is("LAVT_CONV_TAIL", '0');             // 480-channel concat convolution (Swin-T) on the general decode instead of the partial-block tap walk (A/B switch)
    t.side_pre_off = env_is("LAVT_SIDE_PRE", '0');               // epilogue side inputs (residual, activation-gradient operand, multiplier) loaded at the head of the epilogue instead of before the K loop
    t.upce_tile_off = env_is("LAVT_UPCE_TILE", '0');             // fused upsample + cross-entropy backward: the wave-per-low-resolution-pixel form instead of the tiled one
    t.tn_pipe = env_int("LAVT_TN_PIPE", 2);                      // gemm_tn_pipe.hip: grouped weight gradients on 128x128 pipelined tiles -- 0: never (gemm_tn_v2.hip's 64x64 launch), 1: uncut groups only, 2: + long reductions cut into K pieces
    t.tn_pipe_min_tiles = env_int("LAVT_TN_PIPE_MIN_TILES", 128);
    t.tn_pipe_min_ktiles = env_int("LAVT_TN_PIPE_MIN_KTILES", 12);   // uncut groups: average K tiles per output tile below which the group stays on the 64x64 launch
    t.tn_pipe_stages = env_int("LAVT_TN_PIPE_STAGES", 4);
    for (int i = 0; i < 8; ++i) t.probe[i] = 0;
    if (const char* c = getenv("LAVT_PROBE")) sscanf(c, "%d,%d,%d,%d,%d,%d,%d,%d", &t.probe[0], &t.probe[1], &t.probe[2], &t.probe[3], &t.probe[4], &t.probe[5], &t.probe[6], &t.probe[7]);
}
}  // namespace

const lavt_tuning_t& lavt_tuning() {
    if (!g_ready.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_ready.load(std::memory_order_relaxed)) {
            read_tuning(g_tuning);
            g_ready.store(true, std::memory_order_release);
        }
    }
    return g_tuning;
}

extern "C" int lavt_tuning_reload(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    read_tuning(g_tuning);
    g_ready.store(true, std::memory_order_release);
    return LAVT_OK;
}
