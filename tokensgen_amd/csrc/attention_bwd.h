// Shared declarations of the attention-backward translation units (attention_bwd.hip: the kernels the entry point launches;
// tests/csrc/attention_bwd_crosscheck.hip: the correct-first cross-check kernels of the test-only library).
#pragma once
#include "common.h"


constexpr int TQ = 64, TK = 64, HD = 64;
constexpr int LDT = 72;                    // LDS row stride in elements (64 + 8: 16-byte aligned rows, staggered banks)
constexpr int TILE_EL = 64 * LDT;

struct BwdParams {
    const bf16_t *q, *k, *v, *o, *dout;
    long q_ld, q_sb, k_ld, k_sb, v_ld, v_sb, o_ld, o_sb, do_ld, do_sb;
    float *dq, *dk, *dv;
    long dq_ld, dq_sb, dk_ld, dk_sb, dv_ld, dv_sb;
    float *lse, *dsum;                     // [batch][heads][nq] fp32 workspaces
    uint4* seed;                           // [batch][heads][nq] 16-byte seed rows of the dK/dV kernel (attn_bwd_stats2_kernel writes them)
    int nq, nk, heads, batch;
    float scale_log2, scale;
    int accumulate;
    int have_lse;                          // lse was written by the forward (tg_attention_fwd_lse): the statistics launch only computes D
    bf16_t* dvb;                           // optional: bf16 of the value dv receives, element (b, key, h, d) at dvb[b*dvb_sb + key*dvb_ld + h*64 + d]; dv itself may then be null
    long dvb_ld, dvb_sb;
};

// accumulator element r of lane (j, hi) sits at row 8*(r/4) + 4*hi + (r%4), column j of the 32 x 32 block
__device__ __forceinline__ int acc_row(int r, int hi) { return 8 * (r >> 2) + 4 * hi + (r & 3); }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

