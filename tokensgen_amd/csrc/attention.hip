// Flash attention forward for the CogVideoX DiT (head_dim 64, no mask), with the To2V processor's
// two independently-normalised key/value segments fused into one pass over the queries:
//     out = softmax(q1 k1^T) v1  +  s * softmax(q2 k2^T) v2
// Replaces the three F.scaled_dot_product_attention calls of
// VideoIPAdapterCogVideoXAttnProcessor2_0 (attention_processor.py:2066-2069, 2117-2135).
//
// gfx950 design:
//   * one workgroup = 4 waves = 128 query rows of one (batch, head); KV tiles of 64 keys are staged once
//     per workgroup into LDS (double-buffered, global->reg early / reg->LDS late so HBM latency hides
//     under the MFMAs) and read by all four waves.
//   * "swapped" products on v_mfma_f32_32x32x16_bf16:  S^T = K Q^T and O^T = V^T P^T.  Each lane then owns
//     ONE query row (lane&31) for both the scores and the output accumulator, so the online-softmax
//     max/sum/rescale are lane-local (one cross-half exchange per tile) and P feeds the second MFMA
//     straight from the accumulator registers — no LDS round trip, no shuffles.
//   * the K rows are fetched from LDS under the bit-2/bit-3 row permutation `pi`, which makes the 8
//     accumulator registers a lane holds for one 16-key step correspond to 8 CONSECUTIVE keys, so the
//     V^T operand is a single 16-byte ds_read_b128 (V is pre-transposed by tg_transpose_v).
//   * LDS tiles are [64 rows][128 B] with the 16-B slot XOR-swizzled by (row>>1)&7: every 16-lane
//     ds_read_b128 group touches 16 distinct bank groups (conflict-free), for K and V^T alike.
//   * blocks are ordered so that the 8 XCDs work on different (batch, head) pairs: one head's K/V
//     (4.5 MB at N=17776) stays resident in that XCD's 4 MiB L2 while its query tiles stream by.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "tokensgen_hip.h"

namespace {

constexpr int KVBLK = 64;
constexpr int TILE_B = 64 * 128;   // one 64x64 bf16 tile

struct Seg {
    const bf16_t* q; long q_ld, q_sb;
    const bf16_t* k; long k_ld, k_sb;
    const bf16_t* vt; long vt_ld;
    int nk;
    const float* kn2;   // FIXEDM kernels: [batch][heads] upper bound on max_j ||k_j||^2 over this segment's keys (tg_attn_segment.k_norm2_max)
};
struct AttnParams {
    Seg s[2];
    int nseg;
    float seg2_scale_b[16];   // weight of segment 2 per batch item (index b & 15): one value for the batch, or the reference's per-item scale list
    bf16_t* out; long o_ld, o_sb;
    int nq, heads, batch;
    float scale_log2;   // softmax scale * log2(e)
    int prescaled;      // 1: K already carries scale*log2(e) (tg_qk_layernorm_rope out_scale): scores are log2-domain as produced
    // "rider": a second, single-segment problem of the same heads/batch whose workgroups are appended to the launch (ping-pong
    // kernel only).  The main attention leaves 3360 - 13*256 = 32 workgroups for its last round of 256 CUs; the To2V block's
    // vip-query attention (96 workgroups of the same length) rides in that round instead of costing a launch of its own.
    Seg r_s;
    bf16_t* r_out; long r_o_ld, r_o_sb;
    int r_nq;
    int main_wgs;       // workgroups of the main problem (rider workgroups follow); 0 rider workgroups when r_nq == 0
    int total_wgs;      // main + rider workgroups (the RETRY launch walks this list)
    // training forward only (tg_attention_fwd_lse; single segment, no rider): per query row the log-sum-exp of the scaled scores in the
    // log2 domain, [batch][heads][lse_rows] fp32 — what tg_attention_bwd otherwise recomputes with a pass of its own
    float* lse; long lse_rows;
    // FIXEDM kernels: retry[0] counts workgroups that were re-run with the running maximum (cumulative), retry[1 + blockIdx.x] is the
    // "this workgroup's constant-shift result had a row sum below 2^-64" flag the verification raises and the RETRY launch consumes
    int* retry;
    // key-axis split of the launch's LAST `nsplit` workgroups (wgids split_first .. total_wgs - 1): each runs as two half-length workgroups
    // that leave (unnormalised O, m, l) partials in split_ws, joined by attn_split_combine_kernel — see attention_launch
    int split_first, nsplit;
    float* split_ws;
    long split_ws_floats;
};
constexpr long SPLIT_HALF_FLOATS = 512L * (64 + 2 + 64);     // per half: O [512][64], m [512], l [512], O2 [512][64] (second half of a 2-segment parent)

// plain fmaxf nests: clang fuses them to v_max3_f32 (built with -fno-honor-nans so MFMA outputs are not canonicalised by an
// extra v_max first); inline-asm versions get an s_nop after every dependent op and measured ~1.5 % slower
__device__ __forceinline__ float vmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float vmax2(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ int pi_row(int i) {   // swap bits 2 and 3
    return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1);
}
__device__ __forceinline__ int swz(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

// QB = 32-row query blocks per wave (1 or 2).  QB=2 shares every K / V^T fragment read between two query
// blocks (half the LDS and L2 traffic per MFMA); QB=1 gives 2x the workgroups for short query ranges.
template <int QB>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_B];   // K[2], Vt[2]
    constexpr int QT = 128 * QB;
    constexpr float RESCALE_THR = 8.0f;   // log2 units: skip the O rescale while the row max grows < 2^8
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int hi = lane >> 5, j = lane & 31;

    const int nqt = (p.nq + QT - 1) / QT;
    const int nhb = p.heads * p.batch;
    int hb, qt;
    if ((nhb & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        hb = xcd + 8 * (slot / nqt);
        qt = slot % nqt;
    } else {
        hb = blockIdx.x / nqt;
        qt = blockIdx.x % nqt;
    }
    const int h = hb % p.heads, b = hb / p.heads;
    const int q0 = qt * QT + wave * (32 * QB);

    // staging map: chunk c in [0,512): row = c>>3, slot = c&7 ; this thread owns chunks tid and tid+256
    const int r0 = tid >> 3, sl = tid & 7;
    const int ldsoff0 = swz(r0, sl), ldsoff1 = swz(r0 + 32, sl);

    // fragment read offsets
    int offK[2][4], offV[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int kd = 0; kd < 4; ++kd) offK[kb][kd] = swz(kb * 32 + pi_row(j), kd * 2 + hi);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) offV[db][ks] = swz(db * 32 + j, ks * 2 + hi);

    for (int sg = 0; sg < p.nseg; ++sg) {
        const Seg& S = p.s[sg];
        bf16x8 qf[QB][4];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int qrow = min(q0 + qb * 32 + j, p.nq - 1);
            const bf16_t* qp = S.q + (long)b * S.q_sb + (long)qrow * S.q_ld + h * 64 + hi * 8;
#pragma unroll
            for (int kd = 0; kd < 4; ++kd) qf[qb][kd] = *(const bf16x8*)(qp + kd * 16);
        }

        const bf16_t* kbase = S.k + (long)b * S.k_sb + h * 64 + sl * 8;
        const bf16_t* vbase = S.vt + ((long)(b * p.heads + h) * 64) * S.vt_ld + sl * 8;
        const int ntiles = (S.nk + KVBLK - 1) / KVBLK;

        uint4 kr0, kr1, vr0, vr1;
        auto gload = [&](int t) {
            const int k0 = min(t * KVBLK + r0, S.nk - 1), k1 = min(t * KVBLK + r0 + 32, S.nk - 1);
            kr0 = *(const uint4*)(kbase + (long)k0 * S.k_ld);
            kr1 = *(const uint4*)(kbase + (long)k1 * S.k_ld);
            vr0 = *(const uint4*)(vbase + (long)r0 * S.vt_ld + t * KVBLK);
            vr1 = *(const uint4*)(vbase + (long)(r0 + 32) * S.vt_ld + t * KVBLK);
        };
        auto lwrite = [&](int buf) {
            char* kb_ = smem + buf * TILE_B;
            char* vb_ = smem + (2 + buf) * TILE_B;
            *(uint4*)(kb_ + ldsoff0) = kr0; *(uint4*)(kb_ + ldsoff1) = kr1;
            *(uint4*)(vb_ + ldsoff0) = vr0; *(uint4*)(vb_ + ldsoff1) = vr1;
        };

        f32x16 acc_o[QB][2];
        float m[QB], l[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            m[qb] = -1e30f; l[qb] = 0.f;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[qb][db][r] = 0.f;
        }

        gload(0);
        lwrite(0);
        __syncthreads();

        for (int t = 0; t < ntiles; ++t) {
            const int cur = t & 1;
            if (t + 1 < ntiles) gload(t + 1);
            const char* tK = smem + cur * TILE_B;
            const char* tV = smem + (2 + cur) * TILE_B;

            // ---- S^T = K Q^T : 2 key blocks x 4 d-steps, every K fragment feeds all QB query blocks ----
            f32x16 sc[QB][2];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[qb][kb][r] = 0.f;
#pragma unroll
            for (int kd = 0; kd < 4; ++kd) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const bf16x8 kf = *(const bf16x8*)(tK + offK[kb][kd]);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb)
                        sc[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][kd], sc[qb][kb], 0, 0, 0);
                }
            }
            // ---- mask the ragged last tile: reg r of block kb is key t*64 + kb*32 + 16*(r>>3) + 8*hi + (r&7)
            if ((t + 1) * KVBLK > S.nk) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = t * KVBLK + kb * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                            if (key >= S.nk) sc[qb][kb][r] = -1e30f;
                        }
            }
            // ---- online softmax (log2 domain), lane-local row; P packed to bf16 MFMA operands in place ----
            bf16x8 pf[QB][4];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float mx = sc[qb][0][0];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[qb][kb][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * p.scale_log2;
                // defer the rescale while the running max grows by less than 2^THR (wave-uniform decision)
                if (__any(mx > m[qb] + RESCALE_THR)) {
                    const float m_new = fmaxf(m[qb], mx);
                    const float alpha = __builtin_amdgcn_exp2f(m[qb] - m_new);
                    m[qb] = m_new;
                    l[qb] *= alpha;
#pragma unroll
                    for (int db = 0; db < 2; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc_o[qb][db][r] *= alpha;
                }
                const float mq = m[qb];
                float ls = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int kb = ks >> 1, rb = (ks & 1) * 8;
                    float e[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        e[i] = __builtin_amdgcn_exp2f(sc[qb][kb][rb + i] * p.scale_log2 - mq);
                        ls += e[i];
                    }
                    union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
                    for (int i = 0; i < 4; ++i) pk.u[i] = pack_bf16x2(e[2 * i], e[2 * i + 1]);
                    pf[qb][ks] = pk.v;
                }
                l[qb] += ls;
            }
            // ---- O^T += V^T P^T : every V^T fragment feeds all QB query blocks ----
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 vf = *(const bf16x8*)(tV + offV[db][ks]);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb)
                        acc_o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb][ks], acc_o[qb][db], 0, 0, 0);
                }
            }
            if (t + 1 < ntiles) lwrite(cur ^ 1);
            __syncthreads();
        }
        // ---- segment epilogue: normalise and write (segment 2 accumulates onto segment 1's bf16 result, like the
        //      reference's `hidden_states + scale * text_video_hidden_states` on bf16 tensors) ----
        //      lane holds O[q0 + qb*32 + j][d = db*32 + 8*(r>>2) + 4*hi + (r&3)]
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const float lt = l[qb] + __shfl_xor(l[qb], 32, 64);
            const float w = (sg == 0 ? 1.f : p.seg2_scale_b[b & 15]) / lt;
            const int q = q0 + qb * 32 + j;
            if (p.lse && hi == 0 && q < p.nq) p.lse[((long)b * p.heads + h) * p.lse_rows + q] = m[qb] + log2f(lt);
            if (q < p.nq) {
                bf16_t* op = p.out + (long)b * p.o_sb + (long)q * p.o_ld + h * 64;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        uint2* dst = (uint2*)(op + db * 32 + g4 * 8 + hi * 4);
                        float v0 = acc_o[qb][db][g4 * 4 + 0] * w, v1 = acc_o[qb][db][g4 * 4 + 1] * w;
                        float v2 = acc_o[qb][db][g4 * 4 + 2] * w, v3 = acc_o[qb][db][g4 * 4 + 3] * w;
                        if (sg > 0) {
                            const uint2 prev = *dst;
                            v0 = bf16lo_to_f32(prev.x) + round_bf16(v0); v1 = bf16hi_to_f32(prev.x) + round_bf16(v1);
                            v2 = bf16lo_to_f32(prev.y) + round_bf16(v2); v3 = bf16hi_to_f32(prev.y) + round_bf16(v3);
                        }
                        uint2 o;
                        o.x = pack_bf16x2(v0, v1);
                        o.y = pack_bf16x2(v2, v3);
                        *dst = o;
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 8-wave "ping-pong" kernel for long query ranges (512 query rows per workgroup, 64 per wave).
//
// Timing ablations of the 4-wave kernel on MI355X: full 9.15 ms = no-MFMA 4.3 ms + MFMA-only ~4 ms — the two
// co-resident waves of a SIMD run the same instruction stream in phase, so their MFMA segments collide and their softmax
// (VALU) segments collide: matrix and vector pipes are used one after the other, not together.  Here the two waves that
// share a SIMD (wave w of group 0, wave w+4 of group 1) are put in ANTI-PHASE by construction: group 1 runs one barrier
// behind group 0, and every wave alternates
//     X(t) = { P(t-1).V(t-1) ; S(t) = K(t).Q^T }   32 MFMAs + 16 ds_read_b128        (matrix segment)
//     Y(t) = { online softmax of S(t) -> P(t), lazy rescale of O }                   (vector segment)
// with one s_barrier after each segment, so while one wave of a SIMD is in X its partner is in Y.
// K/V tiles are staged by LDS-DMA (1 K piece + 1 V^T piece per wave per tile, swizzle on the source address), shared by
// all 512 query rows (half the L2 traffic of the 256-row kernel), double-buffered:
//     pair u = (K(u+1), V(u)) is issued in interval 2u (group 0: start of X(u); group 1: start of Y(u-1)), retired by every
//     wave before the barrier that ends interval 2u+1, first read in interval 2u+2; the buffers it overwrites were last read
//     in interval 2u-1.
//
// Measured on MI355X, N = 17776 (the experiments behind this schedule are written up in DESIGN.md §8): per tile X ~1600 cycles (1152 of
// MFMA), Y ~2000 cycles for ~190 VALU with the running maximum.  What moved it: fragment-granular ds_read pipelining in X, -m seeding
// through the matrix pipe, two v_max3 chains + v_permlane32_swap, four row-sum chains, s_setprio 2 around the matrix segment, and finally
// dropping the maximum altogether (FIXEDM).  Code generation is fragile here: a workgroup-uniform branch around the matrix segment made the
// whole kernel 60 % slower, one copy of the tile loop per wave group 6 % — measure every edit (tools/ab_build.sh, tools/ab_run.sh).
// ------------------------------------------------------------------------------------------------

// FIXEDM 1 — constant-shift softmax, valid for ANY weights.  The softmax is shift invariant, so the running row maximum is only a RANGE
// device.  Here every query row subtracts a constant c_row fixed before the first tile: with B_row = ||q_row|| * max_j ||k_j|| (Cauchy-
// Schwarz; ||q_row|| from the Q fragments this lane already holds, max ||k||^2 per (batch, head) from the K-norm kernel,
// tg_qk_layernorm_rope_pair_kmax) every score lies in [-B_row, B_row], and c_row = max(0, B_row - 64) gives s - c_row <= 64: P <= 2^64 and a
// row sum <= 2^79 cannot overflow.  Underflow: for B_row <= 64 (c_row = 0, the case of LayerNorm gains around 1) the row maximum is >= -64
// and nothing can go wrong by construction; for larger B_row a row whose scores ALL sit far below c_row could lose its sum, which the
// epilogue VERIFIES (row sum >= 2^-64, i.e. the largest weight >= 2^-79: the terms flushed below 2^-126 are then < 2^-32 of the sum) —
// a workgroup with a failing row raises its flag in p.retry and the RETRY launch (running maximum; a one-workgroup-per-CU grid that walks
// the flag list) recomputes exactly those workgroups.  The per-tile max chain / vote / rescale (15 % of the launch) is gone from the hot loop.
// SPLIT: the launch of the half-length workgroups of the split tail (see attention_launch) — its own instantiation so that the partial-result
// epilogue and the runtime tile range stay out of the ordinary kernel's register allocation (as one kernel they cost it ~45 spilled VGPRs)
template <bool PRESCALED, int FIXEDM = 0, bool LSE = false, bool RETRY = false, bool SPLIT = false>
__global__ __launch_bounds__(512) void attn_fwd_pp_kernel(AttnParams p) {
    // K[2], Vt[2] tiles (32 KiB) + the output staging area: 8 waves x 64 query rows x 128 B (64 KiB)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr float RESCALE_THR = 8.0f;
    const int tid = threadIdx.x;
    // one workgroup's work; `wgid` is its index in the (main + rider) workgroup list.  The ordinary launches run it once with
    // wgid = blockIdx.x; the RETRY launch is a small persistent grid that walks the flag list (below)
    // half: -1 = the whole workgroup; 0 / 1 = first / second part of the key range of a split parent (partials go to p.split_ws)
    auto body = [&](const int wgid, const int half_) {
    const int half = SPLIT ? half_ : -1;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int hi = lane >> 5, j = lane & 31;

    const bool rider = p.r_nq > 0 && wgid >= p.main_wgs;      // workgroup-uniform
    const int bid = rider ? wgid - p.main_wgs : wgid;
    const int nq_ = rider ? p.r_nq : p.nq;
    const int nseg_ = rider ? 1 : p.nseg;
    bf16_t* const out_ = rider ? p.r_out : p.out;
    const long o_ld_ = rider ? p.r_o_ld : p.o_ld, o_sb_ = rider ? p.r_o_sb : p.o_sb;
    const int nqt = (nq_ + 511) / 512;
    const int nhb = p.heads * p.batch;
    int hb, qt;
    if ((nhb & 7) == 0) {
        const int xcd = bid & 7, slot = bid >> 3;
        hb = xcd + 8 * (slot / nqt);
        qt = slot % nqt;
    } else {
        hb = bid / nqt;
        qt = bid % nqt;
    }
    const int h = hb % p.heads, b = hb / p.heads;
    const int q0 = qt * 512 + wave * 64;

    // fragment offsets: one base per operand, k-step advances the 16-B slot by 2 (byte ^ (k << 5)), blocks add 4096 B
    const int pr = pi_row(j);
    const int offK0 = pr * 128 + ((hi ^ ((pr >> 1) & 7)) << 4);
    const int offV0 = j * 128 + ((hi ^ ((j >> 1) & 7)) << 4);
    // LDS-DMA piece of this wave: tile rows [wave*8, +8), lane -> row wave*8 + (lane>>3), physical slot lane&7
    const int drow = wave * 8 + (lane >> 3);
    const int dslot = ((lane & 7) ^ ((drow >> 1) & 7)) * 8;     // logical slot (elements) stored at this lane's position

#define PP_BAR()                                         \
    do {                                                 \
        __builtin_amdgcn_sched_barrier(0);               \
        __builtin_amdgcn_s_barrier();                    \
        __builtin_amdgcn_sched_barrier(0);               \
    } while (0)

    for (int sg = 0; sg < nseg_; ++sg) {
        if (SPLIT && half == 0 && sg > 0) break;              // the first half covers segment-1 keys only
        const Seg& S = rider ? p.r_s : p.s[sg];
        bf16x8 qf[2][4];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qrow = min(q0 + qb * 32 + j, nq_ - 1);
            const bf16_t* qp = S.q + (long)b * S.q_sb + (long)qrow * S.q_ld + h * 64 + hi * 8;
#pragma unroll
            for (int kd = 0; kd < 4; ++kd) qf[qb][kd] = *(const bf16x8*)(qp + kd * 16);
        }
        const bf16_t* kbase = S.k + (long)b * S.k_sb + h * 64 + dslot;
        const bf16_t* vsrc = S.vt + ((long)(b * p.heads + h) * 64 + drow) * S.vt_ld + dslot;
        const int ntiles = (S.nk + KVBLK - 1) / KVBLK;
        // tile range of this workgroup: all of the segment, or — split parents, segment 1 — the part that balances the two halves
        // (the second half also owns segment 2)
        int tb = 0, te = ntiles;
        if (SPLIT && half >= 0 && sg == 0) {
            const int nt2 = nseg_ == 2 ? (p.s[1].nk + KVBLK - 1) / KVBLK : 0;
            const int na = max(1, min(ntiles - 1, (ntiles + nt2 + 1) >> 1));
            if (half == 0) te = na; else tb = na;
        }
        // running source pointer (tiles are issued in order tb, tb + 1, ...): only the last tile of the segment can hold rows beyond nk and takes the clamped form
        const bf16_t* kcur = kbase + (long)min(tb * KVBLK + drow, S.nk - 1) * S.k_ld;
        const bf16_t* const klast = kbase + (long)min((ntiles - 1) * KVBLK + drow, S.nk - 1) * S.k_ld;
        const long kstep = (long)KVBLK * S.k_ld;
        auto dmaK = [&](int t) {      // K(t) -> K buffer t&1
            const bf16_t* const ksrc = t == ntiles - 1 ? klast : kcur;
            kcur += kstep;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ksrc,
                                             (__attribute__((address_space(3))) void*)(smem + (t & 1) * TILE_B + wave * 1024), 16, 0, 0);
        };
        auto dmaV = [&](int t) {      // V^T(t) -> V buffer t&1
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsrc + (long)t * KVBLK),
                                             (__attribute__((address_space(3))) void*)(smem + (2 + (t & 1)) * TILE_B + wave * 1024), 16, 0, 0);
        };
        auto dma_pair = [&](int u) {  // (K(u+1), V(u))
            if (u + 1 < te) dmaK(u + 1);
            if (u < te) dmaV(u);
        };
        auto dma_pair_all = [&](int u) { dmaK(u + 1); dmaV(u); };      // steady tiles: both exist

        f32x16 acc_o[2][2], sc[2][2];
        bf16x8 pf[2][4];
        float m[2], l[2];
        // PRESCALED seed operands: ones = 1.0 in k-slot 0 of every key row (lanes 0-31, element 0), negm[qb] = -m of the
        // lane's query row in k-slot 0; m is kept bf16-representable so the seed is exactly -m
        bf16x8 ones = {0, 0, 0, 0, 0, 0, 0, 0}, negm[2] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
        if (hi == 0) {
            ones[0] = (bf16_t)0x3F80;
            if (FIXEDM != 1 && PRESCALED) negm[0][0] = negm[1][0] = (bf16_t)0x4680; // +2^14 = -m
        }
        float cshift[2] = {0.f, 0.f};
        if (FIXEDM == 1) {
            const float kn2 = S.kn2[b * p.heads + h];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                float qn2 = 0.f;
#pragma unroll
                for (int kd = 0; kd < 4; ++kd)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float v = bf16_to_f32((bf16_t)qf[qb][kd][i]);
                        qn2 += v * v;
                    }
                qn2 += __shfl_xor(qn2, 32, 64);                         // the partner lane holds the other 32 head channels of the row
                const float B = sqrtf(qn2 * kn2) * 1.01f;              // 1 %: fp32 accumulation order of the dot products and of the norms
                uint32_t u = __float_as_uint(fmaxf(B - 64.f, 0.f));
                if (u & 0xffffu) u = (u + 0x10000u) & 0xffff0000u;     // round UP to a bf16-representable value: the MFMA seed is exactly -c
                cshift[qb] = __uint_as_float(u);
                if (hi == 0) negm[qb][0] = (bf16_t)((u >> 16) | 0x8000u);
            }
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            // PRESCALED seeds the accumulator with -m, so m must stay finite and bf16-representable.  It starts at -2^14: the first
            // tile then sees every score 2^14 above the reference and takes the ordinary (rare) rescale path, which sets m to the
            // tile's true row max — upward OR downward, so rows whose scores all sit below -126 do not underflow to l = 0 — with no
            // first-tile test in the hot loop (an explicit `t == 0` there cost 1.5-2 %).  Price: the first tile's 64 scores are
            // formed as s + 2^14 in fp32, i.e. to 2^-9 absolute (0.14 % on their weights, below the bf16 rounding of P).
            m[qb] = FIXEDM == 1 ? cshift[qb] : PRESCALED ? -16384.f : -1e30f;
            l[qb] = 0.f;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[qb][db][r] = 0.f;
        }
        // P(t-1).V(t-1): every V^T fragment feeds both query blocks
        auto pv = [&](int tprev) {
            const char* tV = smem + (2 + (tprev & 1)) * TILE_B;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 vf = *(const bf16x8*)(tV + ((offV0 + db * 4096) ^ (ks << 5)));
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb)
                        acc_o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb][ks], acc_o[qb][db], 0, 0, 0);
                }
        };
        // X(t): the 8 fragment groups {V^T(t-1) ks=0..3, K(t) kd=0..3} software-pipelined one group ahead, so the
        // ds_read_b128 latency of group g+1 runs under the 4 MFMAs (128 cycles) of group g instead of after them
        auto xseg = [&](int t, auto guardc, auto seedc) {
            // SEED false (round 6): every row of this WAVE has the constant shift c_row = 0 (B_row <= 64: LayerNorm gains around 1, the shipped weights) — the four seed
            // MFMAs of a tile would write zeros.  The first K k-step then takes the inline constant 0 as its C operand and the tile runs 32 MFMAs instead of 36.
            constexpr bool SEED = decltype(seedc)::value;
            const char* tV = smem + (2 + ((t - 1) & 1)) * TILE_B;
            const char* tK = smem + (t & 1) * TILE_B;
            // fragment i = 2g + xb (g: k-step group, xb: 32-row block of V^T / K); NFR 4-register buffers, fragment i+NFR is
            // fetched into the buffer fragment i just left, so every ds_read_b128 has NFR-1 MFMA pairs (64 cycles each) of cover
            constexpr int NFR = 3;               // fragment buffers in flight (2 / 3 / 4 measured 7.95 / 7.89 / 7.92 ms)
            bf16x8 fr[NFR];
            // The fragment reads are inline asm with hand-counted lgkmcnt waits: for a C++ LDS load the compiler (a) waits vmcnt(0)
            // first whenever an LDS-DMA piece is in flight (it cannot prove the DMA target is another buffer) — group 0 issues its
            // pair at the start of this very segment, so the wait landed in the middle of the MFMA stream — and (b) falls back to
            // lgkmcnt(0) in the first half.  The low 32 bits of a flat LDS address are the LDS offset.
            const uint32_t ldsV = (uint32_t)(uintptr_t)tV, ldsK = (uint32_t)(uintptr_t)tK;
            auto ld = [&](int i) {
                const int g = i >> 1, xb = i & 1;
                const uint32_t a = (g < 4) ? ldsV + (uint32_t)((offV0 + xb * 4096) ^ (g << 5)) : ldsK + (uint32_t)((offK0 + xb * 4096) ^ ((g - 4) << 5));
                asm volatile("ds_read_b128 %0, %1" : "=v"(fr[i % NFR]) : "v"(a));
            };
            // before fragment i is consumed, min(15 - i, NFR - 1) younger reads may still be in flight (LDS returns in order; the
            // "+v" operand ties the MFMAs that read the fragment to this wait)
            auto wait_frag = [&](int i) {
                const int younger = (15 - i) < (NFR - 1) ? (15 - i) : (NFR - 1);
                if (younger >= 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fr[i % NFR]));
                else if (younger == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fr[i % NFR]));
                else if (younger == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fr[i % NFR]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[i % NFR]));
            };
#pragma unroll
            for (int i = 0; i < NFR; ++i) ld(i);
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (PRESCALED && SEED) {
                // seed S with -m through the matrix pipe: ones[key][k=0] x negm[k=0][query] = -m[query] in every register of
                // the lane's row, 4 MFMAs that run under the first fragments' ds_read latency (no 64 v_mov per tile)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) sc[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, negm[qb], z, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int g = i >> 1, xb = i & 1;
                if (i == 8 && !PRESCALED) {
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                            for (int r = 0; r < 16; ++r) sc[qb][kb][r] = 0.f;
                }
                __builtin_amdgcn_sched_barrier(0);
                wait_frag(i);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    if (g < 4) acc_o[qb][xb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % NFR], pf[qb][g], acc_o[qb][xb], 0, 0, 0);
                    else sc[qb][xb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % NFR], qf[qb][g - 4], (PRESCALED && !SEED && g == 4) ? z : sc[qb][xb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (i + NFR < 16) ld(i + NFR);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (decltype(guardc)::value) if ((t + 1) * KVBLK > S.nk) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = t * KVBLK + kb * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                            if (key >= S.nk) sc[qb][kb][r] = -1e30f;
                        }
            }
        };
        // online softmax (log2 domain), lane-local row; P packed to bf16 MFMA operands.
        // PRESCALED: sc holds a = s - m (m = running max when the scores were issued); otherwise sc holds raw q.k
        auto softmax = [&]() {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
              if constexpr (FIXEDM == 0) {
                // row max: two independent v_max3 chains (one per key block), halves joined by v_permlane32_swap (no LDS trip);
                float ma = vmax3(sc[qb][0][0], sc[qb][0][1], sc[qb][0][2]);
                float mb = vmax3(sc[qb][1][0], sc[qb][1][1], sc[qb][1][2]);
#pragma unroll
                for (int r = 3; r < 15; r += 2) {
                    ma = vmax3(ma, sc[qb][0][r], sc[qb][0][r + 1]);
                    mb = vmax3(mb, sc[qb][1][r], sc[qb][1][r + 1]);
                }
                ma = vmax3(ma, mb, sc[qb][0][15]);
                float mx = vmax2(ma, sc[qb][1][15]);
                {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                    mx = vmax2(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
                }
                if (PRESCALED) {
                    if (__any(mx > RESCALE_THR)) {                       // rare: the row max grew by more than 2^THR
                        const float m_new = round_bf16(m[qb] + fmaxf(mx, 0.f));
                        const float delta = m_new - m[qb];
                        const float alpha = __builtin_amdgcn_exp2f(-delta);
                        m[qb] = m_new;
                        if (hi == 0) negm[qb][0] = f32_to_bf16(-m_new);
                        l[qb] *= alpha;
#pragma unroll
                        for (int db = 0; db < 2; ++db)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc_o[qb][db][r] *= alpha;
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                            for (int r = 0; r < 16; ++r) sc[qb][kb][r] -= delta;
                    }
                } else {
                    mx *= p.scale_log2;
                    if (__any(mx > m[qb] + RESCALE_THR)) {
                        const float m_new = fmaxf(m[qb], mx);
                        const float alpha = __builtin_amdgcn_exp2f(m[qb] - m_new);
                        m[qb] = m_new;
                        l[qb] *= alpha;
#pragma unroll
                        for (int db = 0; db < 2; ++db)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc_o[qb][db][r] *= alpha;
                    }
                }
              }
                const float mq = m[qb];
                // row sums as packed f32 adds: plain v_add_f32 on two chains measured slower here (8.94 vs 8.63 ms)
                f32x2 ls2[4];          // one row-sum chain per k-step: no dependent packed adds back to back
                float e[4][8];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int kb = ks >> 1, rb = (ks & 1) * 8;
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        e[ks][i] = PRESCALED ? __builtin_amdgcn_exp2f(sc[qb][kb][rb + i]) : __builtin_amdgcn_exp2f(sc[qb][kb][rb + i] * p.scale_log2 - mq);
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    union { bf16x8 v; uint32_t u[4]; } pk;
#pragma unroll
                    for (int i = 0; i < 4; ++i) pk.u[i] = pack_bf16x2(e[ks][2 * i], e[ks][2 * i + 1]);
                    pf[qb][ks] = pk.v;
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) ls2[ks] = f32x2{e[ks][0], e[ks][1]};
#pragma unroll
                for (int i = 1; i < 4; ++i)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) ls2[ks] += f32x2{e[ks][2 * i], e[ks][2 * i + 1]};
                const f32x2 lsum = (ls2[0] + ls2[1]) + (ls2[2] + ls2[3]);
                l[qb] += lsum[0] + lsum[1];
            }
        };

        // ---- prologue: K(0) resident for everybody; group 1 issues pair 0 and falls one barrier behind ----
        // X(0) runs the P.V half too, on P = 0 against V buffer 1: zero both so that 0 x stale-LDS cannot make a NaN
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) pf[qb][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        *(uint4*)(smem + (2 + ((tb & 1) ^ 1)) * TILE_B + tid * 16) = uint4{0, 0, 0, 0};
        dmaK(tb);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        PP_BAR();
        if (grp == 1) {
            dma_pair(tb);
            PP_BAR();
        }
        // The tile body exists twice: "guarded" (the last two tiles of a range: DMA issue past the end, key mask) and "steady" (everything unconditional) — the end-of-range
        // tests (te twice per DMA issue, nk once per tile) were five of the eleven s_cbranch per tile and wave; without them the launch is 1.0-1.6 % faster (same box,
        // three interleaved rounds).  Specialising the body on the wave group as well (no grp tests left, one back-branch per tile) measured 4-5 % SLOWER: the two
        // groups then run different copies of the loop.
        auto tile = [&](auto guardc, auto seedc, int t) {
            constexpr bool GUARD = decltype(guardc)::value;
            // X(t): matrix segment
            if (grp == 0) { if constexpr (GUARD) dma_pair(t); else dma_pair_all(t); }
            // the matrix segment runs at raised priority: its MFMA / ds_read issue slots are few (one per ~32 cycles) but each one the
            // partner's VALU stream delays idles the matrix pipe; measured -4..6 % (7.52 vs 7.94 ms same box); prio 1: -2 %, prio 3 = 2
            __builtin_amdgcn_s_setprio(2);
            xseg(t, guardc, seedc);
            __builtin_amdgcn_s_setprio(0);                         // (fencing this with sched_barrier(0) measured 2.5 % slower)
            if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // pair t (issued one segment ago) has landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PP_BAR();
            // Y(t): vector segment
            if (grp == 1) { if constexpr (GUARD) dma_pair(t + 1); else dma_pair_all(t + 1); }
            softmax();
            if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // pair t has landed
            PP_BAR();
        };
        // Two copies of the tile loop for the constant-shift kernel, chosen per WAVE (wave-uniform; the barrier structure of the copies is identical, so waves of one
        // workgroup may differ): without the seed MFMAs when every row of the wave has c_row = 0 — with the shipped weights that is every wave of every launch, so
        // the instruction cache sees one copy — and with them otherwise.  (Two copies run by the two wave groups of a workgroup at the same time measured 4-5 % slower,
        // see above: that is not this case.)
        auto run = [&](auto seedc) {
            int t = tb;
            for (; t < te - 2; ++t) tile(std::false_type{}, seedc, t);
            for (; t < te; ++t) tile(std::true_type{}, seedc, t);
        };
        if constexpr (FIXEDM == 1 && PRESCALED) {
            if (!__any(cshift[0] != 0.f || cshift[1] != 0.f)) run(std::false_type{});
            else run(std::true_type{});
        } else {
            run(std::true_type{});
        }
        pv(te - 1);                                                                // X(nt): last P.V
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (grp == 0) PP_BAR();                                                    // barrier counts of the two groups meet again
        PP_BAR();                                                                  // all LDS reads of this segment done

        // ---- segment epilogue: O through LDS so that every global access is a whole 128-byte row of this head.  A lane owns ONE query
        // row in 8-byte pieces (4 d-values per accumulator group), which as direct stores touch 64 different lines per instruction;
        // the wave's staging area [64 rows][128 B] (16-B slots XOR-swizzled by row&7) is written in that shape and read back as
        // 8 lanes x 16 B per row.  With two key segments the normalised result of segment 1 simply WAITS there (no global write +
        // read-modify-write as before): segment 2 adds its `seg2_scale * O2` on top (each lane re-reads exactly what it wrote).
        if (SPLIT && half >= 0) {
            // split parent: leave this half's partial result in the workspace (attn_split_combine_kernel joins the halves).  Segment 1:
            // unnormalised O + (m, l) per row; segment 2 (second half only): the finished bf16(seg2_scale * O2) as floats.
            float* wsb = p.split_ws + (long)(2 * (wgid - p.split_first) + half) * SPLIT_HALF_FLOATS;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const float lt = l[qb] + __shfl_xor(l[qb], 32, 64);
                const int row = wave * 64 + qb * 32 + j;
                float w = 1.f;
                float* dst = wsb + (long)row * 64;
                if (sg == 0) {
                    if (hi == 0) { wsb[512 * 64 + row] = m[qb]; wsb[512 * 65 + row] = lt; }
                } else {
                    w = p.seg2_scale_b[b & 15] / lt;
                    dst += 512 * 66;
                    if (FIXEDM == 1) {
                        if (__any(!(lt >= 5.421011e-20f && lt < 3.0e38f)) && lane == 0) p.retry[1 + wgid] = 1;
                    }
                }
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        f32x4 v = {acc_o[qb][db][g4 * 4 + 0] * w, acc_o[qb][db][g4 * 4 + 1] * w, acc_o[qb][db][g4 * 4 + 2] * w, acc_o[qb][db][g4 * 4 + 3] * w};
                        if (sg > 0) v = f32x4{round_bf16(v[0]), round_bf16(v[1]), round_bf16(v[2]), round_bf16(v[3])};
                        *(f32x4*)(dst + db * 32 + g4 * 8 + hi * 4) = v;
                    }
            }
            continue;
        }
        char* stg = smem + 4 * TILE_B + wave * 8192;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float lt = l[qb] + __shfl_xor(l[qb], 32, 64);
            const float w = (sg == 0 ? 1.f : p.seg2_scale_b[b & 15]) / lt;
            const int row = qb * 32 + j;
            if (LSE && hi == 0 && q0 + row < nq_) p.lse[((long)b * p.heads + h) * p.lse_rows + q0 + row] = m[qb] + log2f(lt);
            if (FIXEDM == 1) {       // verification of the constant shift (see the kernel header); also catches a NaN / inf row sum
                if (__any(!(lt >= 5.421011e-20f && lt < 3.0e38f)) && lane == 0) p.retry[1 + wgid] = 1;
            }
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    uint2* dst = (uint2*)(stg + row * 128 + (((db * 4 + g4) ^ (row & 7)) << 4) + hi * 8);
                    float v0 = acc_o[qb][db][g4 * 4 + 0] * w, v1 = acc_o[qb][db][g4 * 4 + 1] * w;
                    float v2 = acc_o[qb][db][g4 * 4 + 2] * w, v3 = acc_o[qb][db][g4 * 4 + 3] * w;
                    if (sg > 0) {     // `scale * O2` is a bf16 tensor before it is added to O1 (attention_processor.py:2117-2134)
                        const uint2 prev = *dst;
                        v0 = bf16lo_to_f32(prev.x) + round_bf16(v0); v1 = bf16hi_to_f32(prev.x) + round_bf16(v1);
                        v2 = bf16lo_to_f32(prev.y) + round_bf16(v2); v3 = bf16hi_to_f32(prev.y) + round_bf16(v3);
                    }
                    uint2 o;
                    o.x = pack_bf16x2(v0, v1);
                    o.y = pack_bf16x2(v2, v3);
                    *dst = o;
                }
        }
        if (sg + 1 == nseg_) {       // last segment: the staged rows go out, 8 rows (8 lanes x 16 B each) per instruction
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's own LDS writes (no other wave touches its area)
            const int rl = lane >> 3, sl = lane & 7;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 8 + rl;
                const int q = q0 + row;
                const uint4 val = *(const uint4*)(stg + row * 128 + ((sl ^ (row & 7)) << 4));
                if (q < nq_) *(uint4*)(out_ + (long)b * o_sb_ + (long)q * o_ld_ + h * 64 + sl * 8) = val;
            }
        }
    }
#undef PP_BAR
    };
    if constexpr (RETRY) {
        // persistent retry grid (one workgroup per CU): re-run exactly the workgroups whose constant-shift pass raised its flag.  With no flag
        // up — the normal case — this is a 256-workgroup launch that reads the flag list once and ends.
        for (int w = blockIdx.x; w < p.total_wgs; w += gridDim.x) {
            const int flagged = p.retry[1 + w];                // workgroup-uniform
            if (!flagged) continue;
            __syncthreads();                                   // every wave has read the flag before it is lowered
            if (tid == 0) {
                p.retry[1 + w] = 0;
                atomicAdd(&p.retry[0], 1);
            }
            body(w, -1);
            __syncthreads();                                   // LDS is reused by the next flagged workgroup
        }
    } else if constexpr (SPLIT) {
        // two half workgroups per split parent, both on the XCD the parent's (batch, head) lives on (workgroup b runs on XCD b % 8, and the
        // parents' wgid % 8 selects their head's XCD when split_first is a multiple of 8): the halves of one head share its K / V in that L2
        const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
        const int i = 8 * (slot >> 1) + xcd;
        if (i < p.nsplit) body(p.split_first + i, slot & 1);
    } else {
        // whole workgroups: the launch skips the split parents [split_first, split_first + nsplit)
        const int w = (int)blockIdx.x;
        body(w < p.split_first ? w : w + p.nsplit, -1);
    }
}

// Joins the two halves of every split parent: O = (Oa 2^(ma-M) + Ob 2^(mb-M)) / (la 2^(ma-M) + lb 2^(mb-M)), M = max(ma, mb) — with the
// constant shift ma == mb and this is a plain sum — rounds to bf16, adds the second segment's bf16(seg2_scale * O2) like the unsplit epilogue,
// and writes the parent's output rows.  The constant-shift verification of a split parent's segment 1 happens here, on the JOINED row sum.
__global__ __launch_bounds__(256) void attn_split_combine_kernel(AttnParams p, int fixedm) {      // grid: 4 workgroups (128 rows each) per parent
    const int wgid = p.split_first + ((int)blockIdx.x >> 2);
    const bool rider = p.r_nq > 0 && wgid >= p.main_wgs;
    const int bid = rider ? wgid - p.main_wgs : wgid;
    const int nq_ = rider ? p.r_nq : p.nq;
    const int nseg_ = rider ? 1 : p.nseg;
    bf16_t* const out_ = rider ? p.r_out : p.out;
    const long o_ld_ = rider ? p.r_o_ld : p.o_ld, o_sb_ = rider ? p.r_o_sb : p.o_sb;
    const int nqt = (nq_ + 511) / 512;
    const int nhb = p.heads * p.batch;
    int hb, qt;
    if ((nhb & 7) == 0) {
        const int xcd = bid & 7, slot = bid >> 3;
        hb = xcd + 8 * (slot / nqt);
        qt = slot % nqt;
    } else {
        hb = bid / nqt;
        qt = bid % nqt;
    }
    const int h = hb % p.heads, b = hb / p.heads;
    const float* wa = p.split_ws + (long)(2 * ((int)blockIdx.x >> 2)) * SPLIT_HALF_FLOATS;
    const float* wb = wa + SPLIT_HALF_FLOATS;
    bool bad = false;
    for (int item = threadIdx.x; item < 128 * 4; item += 256) {
        const int row = ((int)blockIdx.x & 3) * 128 + (item >> 2), d0 = (item & 3) * 16;
        const int q = qt * 512 + row;
        if (q >= nq_) continue;
        const float ma = wa[512 * 64 + row], la = wa[512 * 65 + row], mb = wb[512 * 64 + row], lb = wb[512 * 65 + row];
        const float M = fmaxf(ma, mb);
        const float fa = __builtin_amdgcn_exp2f(ma - M), fb = __builtin_amdgcn_exp2f(mb - M);
        const float L = la * fa + lb * fb;
        if (fixedm && !(L >= 5.421011e-20f && L < 3.0e38f)) bad = true;
        const float ia = fa / L, ib = fb / L;
        uint32_t o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int d = d0 + 2 * i;
            float v0 = wa[(long)row * 64 + d] * ia + wb[(long)row * 64 + d] * ib;
            float v1 = wa[(long)row * 64 + d + 1] * ia + wb[(long)row * 64 + d + 1] * ib;
            if (nseg_ == 2) {
                v0 = round_bf16(v0) + wb[512 * 66 + (long)row * 64 + d];
                v1 = round_bf16(v1) + wb[512 * 66 + (long)row * 64 + d + 1];
            }
            o[i] = pack_bf16x2(v0, v1);
        }
        bf16_t* dst = out_ + (long)b * o_sb_ + (long)q * o_ld_ + h * 64 + d0;
        *(uint4*)dst = uint4{o[0], o[1], o[2], o[3]};
        *(uint4*)(dst + 8) = uint4{o[4], o[5], o[6], o[7]};
    }
    if (bad) p.retry[1 + wgid] = 1;
}

}  // namespace

static int attention_launch(AttnParams& p, float scale, int k_prescaled, const char* who, hipStream_t stream) {
    const int nq = p.nq, heads = p.heads, batch = p.batch;
    // k_prescaled: K rows already carry scale*log2(e) (written by tg_qk_layernorm_rope with out_scale), `scale` is then ignored
    p.prescaled = k_prescaled ? 1 : 0;
    p.scale_log2 = k_prescaled ? 1.0f : scale * 1.4426950408889634f;
    // kernel choice by query-range length: the 8-wave ping-pong kernel from `pp_min` workgroups of 512 rows (4 per CU) on; below that the
    // 4-wave kernel with 256-row tiles (2 query blocks per wave) while those still give >= 4 workgroups per CU, else 128-row tiles.
    // Measured on MI355X at N = 17776: ping-pong 1.12-1.18 PFLOP/s, 256-row tiles 0.90, 128-row tiles 0.85.
    const long wg256 = (long)((nq + 255) / 256) * heads * batch;
    const long pp_min = tg_knob(TG_KNOB_ATTN_PP_MIN_WG);    // 1024; the cross-check tests lower it (tg_debug_set)
    const long wg512 = (long)((nq + 511) / 512) * heads * batch;
    const bool pp = wg512 >= pp_min;
    if (p.r_nq > 0 && !pp) {
        // the rider needs the ping-pong kernel: run it as a launch of its own through the ordinary dispatch
        AttnParams r{};
        r.s[0] = p.r_s; r.nseg = 1;
        r.out = p.r_out; r.o_ld = p.r_o_ld; r.o_sb = p.r_o_sb;
        r.nq = p.r_nq; r.heads = heads; r.batch = batch;
        p.r_nq = 0;
        const int rc = attention_launch(p, scale, k_prescaled, who, stream);
        return rc ? rc : attention_launch(r, scale, k_prescaled, who, stream);
    }
    const int n_cu = tg_device_cus();
    // ---- ragged last query tile: every ping-pong workgroup takes the same time, so a launch of G workgroups costs ceil(G / CUs) rounds.
    // When dropping the ragged last tile of every (head, batch) saves a round, those rows go to the 4-wave kernel in 128-row workgroups
    // right behind the ping-pong launch (same stream): N = 17776 without a rider 7.69 -> 7.53 ms, the T2To stage's N = 9442 (19 -> 18
    // tiles: 8 -> 7 rounds).  With a rider problem the riders already fill the last round, and the split loses (measured 8.0 vs 7.75 ms).
    const int full_rows = (nq / 512) * 512;
    const long wg_full = (long)(nq / 512) * heads * batch;
    if (pp && p.r_nq == 0 && full_rows < nq && wg_full >= pp_min && (wg_full + n_cu - 1) / n_cu < (wg512 + n_cu - 1) / n_cu) {
        AttnParams m = p;                                   // the full 512-row tiles: ping-pong kernel
        m.nq = full_rows;
        const int rc = attention_launch(m, scale, k_prescaled, who, stream);
        if (rc) return rc;
        AttnParams t = p;                                   // the ragged remainder: 128-row workgroups of the 4-wave kernel
        t.nq = nq - full_rows;
        for (int sg = 0; sg < p.nseg; ++sg) t.s[sg].q = p.s[sg].q + (long)full_rows * p.s[sg].q_ld;
        t.out = p.out + (long)full_rows * p.o_ld;
        if (p.lse) t.lse = p.lse + full_rows;
        const int nqt = (t.nq + 127) / 128;
        hipLaunchKernelGGL(attn_fwd_kernel<1>, dim3((unsigned)(nqt * heads * batch)), dim3(256), 0, stream, t);
        TG_LAUNCH_CHECK(who);
        return TG_OK;
    }
    // constant-shift softmax (attn_fwd_pp_kernel<.., FIXEDM = 1>): every segment of a k_prescaled launch carries the key-norm bound and the
    // caller gave a retry workspace.  The TG_ATTN_FIXEDM knob = 0 (tg_debug_set) forces the running-max kernel.
    const bool fixedm_on = tg_knob(TG_KNOB_ATTN_FIXEDM) != 0;
    bool fixedm = fixedm_on && p.prescaled && pp && p.retry;
    for (int sg = 0; sg < p.nseg && fixedm; ++sg) fixedm = p.s[sg].kn2 != nullptr;
    if (fixedm && p.r_nq > 0) fixedm = p.r_s.kn2 != nullptr;
    p.main_wgs = (int)wg512;
    const long grid512 = wg512 + (p.r_nq > 0 ? (long)((p.r_nq + 511) / 512) * heads * batch : 0);
    p.total_wgs = (int)grid512;
    const unsigned retry_grid = (unsigned)(grid512 < n_cu ? grid512 : n_cu);
    // ---- last-round quantisation: every 512-row workgroup of a launch takes the same time (same key length), so G workgroups cost
    // ceil(G / CUs) rounds — the DiT's 3360 + 96 = 13.5 x 256 pay for 14.  When the remainder R = G mod CUs is at most half a round, R
    // workgroups are split over the KEY axis into 2 R half-length workgroups (their own launch behind the G - R whole ones, partials joined by
    // attn_split_combine_kernel): the tail then costs half a round, 13.5 instead of 14.  Splitting over the queries cannot do that (DESIGN §8:
    // 128-row workgroups are themselves badly quantised).  The split parents are the last R workgroups of the MAIN problem, not the riders:
    // the halves of one head's query tiles share that head's K / V in their XCD's L2, whereas every rider has a (batch, head) of its own and
    // 256 of those streaming at once are HBM-bound.  The TG_ATTN_SPLIT knob = 0 disables; needs the caller's split workspace.
    const bool split_on = tg_knob(TG_KNOB_ATTN_SPLIT) != 0;
    p.nsplit = 0;
    p.split_first = (int)grid512;
    if (pp && split_on && p.split_ws && !p.lse && grid512 > n_cu) {
        const long R = grid512 % n_cu;
        const int nt_main = (p.s[0].nk + KVBLK - 1) / KVBLK;
        if (R > 0 && 2 * R <= n_cu && R <= wg512 && nt_main >= 2 && p.split_ws_floats >= 2 * R * SPLIT_HALF_FLOATS) {
            p.nsplit = (int)R;
            p.split_first = (int)(wg512 - R);
        }
    }
    // whole workgroups | halves of the split parents (rounded up to whole groups of 16 = 8 XCDs x 2 halves)
    const unsigned main_grid = (unsigned)(grid512 - p.nsplit), split_grid = (unsigned)(16 * ((p.nsplit + 7) / 8));
    constexpr size_t PP_LDS = 4 * TILE_B + 8 * 8192;
    // launch one instantiation of the ping-pong kernel (96 KiB of dynamic LDS: the attribute is set once per instantiation AND device)
#define TG_PP(GRID, ...)                                                                                                            \
    do {                                                                                                                            \
        TG_DYN_LDS((attn_fwd_pp_kernel<__VA_ARGS__>), 4 * TILE_B + 8 * 8192);                                                       \
        hipLaunchKernelGGL((attn_fwd_pp_kernel<__VA_ARGS__>), dim3(GRID), dim3(512), PP_LDS, stream, p);                            \
    } while (0)
    if (pp && fixedm) {
        // the verified constant-shift pass, [the split tail and its join,] then the retry launch: a persistent grid that re-runs the flagged
        // workgroups (normally none)
        if (p.lse) {
            TG_PP(main_grid, true, 1, true);
            TG_PP(retry_grid, true, 0, true, true);
        } else {
            TG_PP(main_grid, true, 1);
            if (p.nsplit) {
                TG_PP(split_grid, true, 1, false, false, true);
                hipLaunchKernelGGL(attn_split_combine_kernel, dim3((unsigned)(4 * p.nsplit)), dim3(256), 0, stream, p, 1);
            }
            TG_PP(retry_grid, true, 0, false, true);
        }
    } else if (pp) {
        if (p.lse) {
            TG_PP(main_grid, false, 0, true);
        } else if (p.prescaled) {
            TG_PP(main_grid, true);
            if (p.nsplit) TG_PP(split_grid, true, 0, false, false, true);
        } else {
            TG_PP(main_grid, false);
            if (p.nsplit) TG_PP(split_grid, false, 0, false, false, true);
        }
        if (p.nsplit) hipLaunchKernelGGL(attn_split_combine_kernel, dim3((unsigned)(4 * p.nsplit)), dim3(256), 0, stream, p, 0);
    } else if (wg256 >= 1024) {
        hipLaunchKernelGGL(attn_fwd_kernel<2>, dim3((unsigned)wg256), dim3(256), 0, stream, p);
    } else {
        const int nqt = (nq + 127) / 128;
        hipLaunchKernelGGL(attn_fwd_kernel<1>, dim3((unsigned)(nqt * heads * batch)), dim3(256), 0, stream, p);
    }
#undef TG_PP
    TG_LAUNCH_CHECK(who);
    return TG_OK;
}

static int fill_segment(Seg& S, const tg_attn_segment& g, const char* what) {
    TG_REQUIRE(g.q && g.k && g.vt, TG_ERR_ARG, "tg_attention: %s: null pointer", what);
    TG_REQUIRE(g.nk > 0, TG_ERR_SHAPE, "tg_attention: %s: nk=%d", what, g.nk);
    TG_REQUIRE(g.q_ld % 8 == 0 && g.k_ld % 8 == 0 && g.vt_ld % 64 == 0 && g.q_strideB % 8 == 0 && g.k_strideB % 8 == 0 && tg_aligned16(g.q) &&
               tg_aligned16(g.k) && tg_aligned16(g.vt), TG_ERR_ALIGN, "tg_attention: %s alignment", what);
    TG_REQUIRE(g.vt_ld >= ((g.nk + 63) / 64) * 64, TG_ERR_SHAPE, "tg_attention: %s: vt_ld must cover nk rounded up to 64", what);
    S = Seg{(const bf16_t*)g.q, g.q_ld, g.q_strideB, (const bf16_t*)g.k, g.k_ld, g.k_strideB, (const bf16_t*)g.vt, g.vt_ld, g.nk, g.k_norm2_max};
    return TG_OK;
}

extern "C" long tg_attention_retry_ints(int nq0, int nq1, int heads, int batch) {
    return 1 + (long)((nq0 + 511) / 512 + (nq1 > 0 ? (nq1 + 511) / 512 : 0)) * heads * batch;
}

extern "C" long tg_attention_split_floats(int nq0, int nq1, int heads, int batch) {
    int dev = 0, n_cu = 256;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    const long g = (long)((nq0 + 511) / 512 + (nq1 > 0 ? (nq1 + 511) / 512 : 0)) * heads * batch;
    const long R = g % n_cu;
    return (g > n_cu && R > 0 && 2 * R <= n_cu) ? 2 * R * SPLIT_HALF_FLOATS : 0;
}

extern "C" int tg_attention_fwd_multi(const tg_attn_problem* problems, int nproblems, int heads, int batch, float scale, int k_prescaled,
                                      const tg_attn_workspace* ws, hipStream_t stream) {
    TG_REQUIRE(problems && (nproblems == 1 || nproblems == 2), TG_ERR_ARG, "tg_attention_fwd_multi: 1 or 2 problems");
    TG_REQUIRE(heads > 0 && batch > 0, TG_ERR_SHAPE, "tg_attention_fwd_multi: bad shape");
    const tg_attn_problem& A = problems[0];
    TG_REQUIRE(A.out && A.nq > 0 && (A.nseg == 1 || A.nseg == 2), TG_ERR_ARG, "tg_attention_fwd_multi: problem 0");
    TG_REQUIRE(A.out_ld % 8 == 0 && A.out_strideB % 8 == 0 && tg_aligned16(A.out), TG_ERR_ALIGN, "tg_attention_fwd_multi: output alignment (16 B)");
    AttnParams p{};
    int rc = fill_segment(p.s[0], A.seg[0], "problem 0 segment 1");
    if (rc) return rc;
    p.nseg = A.nseg;
    if (A.nseg == 2 && (rc = fill_segment(p.s[1], A.seg[1], "problem 0 segment 2"))) return rc;
    TG_REQUIRE(!A.seg2_scale_batch || batch <= 16, TG_ERR_SHAPE, "tg_attention_fwd_multi: per-item segment-2 scales for at most 16 batch items (got %d)", batch);
    for (int i = 0; i < 16; ++i) p.seg2_scale_b[i] = A.seg2_scale_batch ? A.seg2_scale_batch[i < batch ? i : batch - 1] : A.seg2_scale;
    p.out = (bf16_t*)A.out; p.o_ld = A.out_ld; p.o_sb = A.out_strideB;
    p.nq = A.nq; p.heads = heads; p.batch = batch;
    if (nproblems == 2) {
        const tg_attn_problem& B = problems[1];
        TG_REQUIRE(B.out && B.nq > 0 && B.nseg == 1, TG_ERR_ARG, "tg_attention_fwd_multi: problem 1 must have one key segment");
        TG_REQUIRE(B.out_ld % 8 == 0 && B.out_strideB % 8 == 0 && tg_aligned16(B.out), TG_ERR_ALIGN, "tg_attention_fwd_multi: output alignment (16 B)");
        if ((rc = fill_segment(p.r_s, B.seg[0], "problem 1"))) return rc;
        p.r_out = (bf16_t*)B.out; p.r_o_ld = B.out_ld; p.r_o_sb = B.out_strideB; p.r_nq = B.nq;
    }
    if (ws && ws->retry) {
        TG_REQUIRE(ws->retry_ints >= tg_attention_retry_ints(p.nq, p.r_nq, heads, batch), TG_ERR_SHAPE,
                   "tg_attention_fwd_multi: retry workspace of %ld ints, need %ld (tg_attention_retry_ints)", ws->retry_ints,
                   tg_attention_retry_ints(p.nq, p.r_nq, heads, batch));
        p.retry = ws->retry;
    }
    if (ws && ws->split) {
        TG_REQUIRE(tg_aligned16(ws->split), TG_ERR_ALIGN, "tg_attention_fwd_multi: split workspace alignment (16 B)");
        p.split_ws = ws->split;
        p.split_ws_floats = ws->split_floats;
    }
    return attention_launch(p, scale, k_prescaled, "tg_attention_fwd_multi", stream);
}

extern "C" int tg_attention_fwd(const void* q1, long q1_ld, long q1_strideB,
                                const void* k1, long k1_ld, long k1_strideB, const void* vt1, long vt1_ld, int nk1,
                                const void* q2, long q2_ld, long q2_strideB,
                                const void* k2, long k2_ld, long k2_strideB, const void* vt2, long vt2_ld, int nk2,
                                float seg2_scale, void* out, long out_ld, long out_strideB,
                                int nq, int heads, int batch, float scale, int k_prescaled, hipStream_t stream) {
    TG_REQUIRE(q1 && k1 && vt1 && out, TG_ERR_ARG, "tg_attention_fwd: null pointer");
    TG_REQUIRE(nq > 0 && nk1 > 0 && heads > 0 && batch > 0, TG_ERR_SHAPE, "tg_attention_fwd: bad shape nq=%d nk1=%d", nq, nk1);
    tg_attn_problem A{};
    A.seg[0] = tg_attn_segment{q1, q1_ld, q1_strideB, k1, k1_ld, k1_strideB, vt1, vt1_ld, nk1, nullptr};
    A.nseg = 1;
    if (q2) {
        TG_REQUIRE(k2 && vt2 && nk2 > 0, TG_ERR_ARG, "tg_attention_fwd: segment 2 incomplete");
        A.seg[1] = tg_attn_segment{q2, q2_ld, q2_strideB, k2, k2_ld, k2_strideB, vt2, vt2_ld, nk2, nullptr};
        A.nseg = 2;
    }
    A.seg2_scale = seg2_scale;
    A.out = out; A.out_ld = out_ld; A.out_strideB = out_strideB; A.nq = nq;
    return tg_attention_fwd_multi(&A, 1, heads, batch, scale, k_prescaled, nullptr, stream);
}

extern "C" int tg_attention_fwd_lse(const void* q, long q_ld, long q_strideB, const void* k, long k_ld, long k_strideB, const void* vt, long vt_ld, int nk,
                                    void* out, long out_ld, long out_strideB, int nq, int heads, int batch, float scale, float* lse, hipStream_t stream) {
    TG_REQUIRE(q && k && vt && out && lse, TG_ERR_ARG, "tg_attention_fwd_lse: null pointer");
    TG_REQUIRE(nq > 0 && nk > 0 && heads > 0 && batch > 0, TG_ERR_SHAPE, "tg_attention_fwd_lse: bad shape nq=%d nk=%d", nq, nk);
    TG_REQUIRE(out_ld % 8 == 0 && out_strideB % 8 == 0 && tg_aligned16(out), TG_ERR_ALIGN, "tg_attention_fwd_lse: output alignment (16 B)");
    AttnParams p{};
    const tg_attn_segment g{q, q_ld, q_strideB, k, k_ld, k_strideB, vt, vt_ld, nk, nullptr};
    int rc = fill_segment(p.s[0], g, "segment");
    if (rc) return rc;
    p.nseg = 1;
    p.out = (bf16_t*)out; p.o_ld = out_ld; p.o_sb = out_strideB;
    p.nq = nq; p.heads = heads; p.batch = batch;
    p.lse = lse; p.lse_rows = nq;
    return attention_launch(p, scale, 0, "tg_attention_fwd_lse", stream);
}

// tg_attention_fwd_lse on the inference path's fast kernels: K rows that already carry scale * log2(e) (k_prescaled) and, with the key-norm bound
// of tg_qk_layernorm_rope_pair_kmax + a retry workspace, the verified constant-shift softmax (attn_fwd_pp_kernel<true, 1, LSE> + its retry launch).
extern "C" int tg_attention_fwd_lse_ex(const void* q, long q_ld, long q_strideB, const void* k, long k_ld, long k_strideB, const void* vt, long vt_ld, int nk,
                                       void* out, long out_ld, long out_strideB, int nq, int heads, int batch, float scale, int k_prescaled,
                                       const float* k_norm2_max, int* retry, long retry_ints, float* lse, hipStream_t stream) {
    TG_REQUIRE(q && k && vt && out && lse, TG_ERR_ARG, "tg_attention_fwd_lse_ex: null pointer");
    TG_REQUIRE(nq > 0 && nk > 0 && heads > 0 && batch > 0, TG_ERR_SHAPE, "tg_attention_fwd_lse_ex: bad shape nq=%d nk=%d", nq, nk);
    TG_REQUIRE(out_ld % 8 == 0 && out_strideB % 8 == 0 && tg_aligned16(out), TG_ERR_ALIGN, "tg_attention_fwd_lse_ex: output alignment (16 B)");
    AttnParams p{};
    const tg_attn_segment g{q, q_ld, q_strideB, k, k_ld, k_strideB, vt, vt_ld, nk, k_norm2_max};
    int rc = fill_segment(p.s[0], g, "segment");
    if (rc) return rc;
    p.nseg = 1;
    p.out = (bf16_t*)out; p.o_ld = out_ld; p.o_sb = out_strideB;
    p.nq = nq; p.heads = heads; p.batch = batch;
    p.lse = lse; p.lse_rows = nq;
    if (retry) {
        TG_REQUIRE(retry_ints >= tg_attention_retry_ints(nq, 0, heads, batch), TG_ERR_SHAPE,
                   "tg_attention_fwd_lse_ex: retry workspace of %ld ints, need %ld (tg_attention_retry_ints)", retry_ints, tg_attention_retry_ints(nq, 0, heads, batch));
        p.retry = retry;
    }
    return attention_launch(p, scale, k_prescaled, "tg_attention_fwd_lse_ex", stream);
}
