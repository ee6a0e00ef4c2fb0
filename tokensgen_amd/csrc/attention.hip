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
};
struct AttnParams {
    Seg s[2];
    int nseg;
    float seg2_scale;
    bf16_t* out; long o_ld, o_sb;
    int nq, heads, batch;
    float scale_log2;   // softmax scale * log2(e)
};

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
            const float w = (sg == 0 ? 1.f : p.seg2_scale) / lt;
            const int q = q0 + qb * 32 + j;
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

}  // namespace

extern "C" int tg_attention_fwd(const void* q1, long q1_ld, long q1_strideB,
                                const void* k1, long k1_ld, long k1_strideB, const void* vt1, long vt1_ld, int nk1,
                                const void* q2, long q2_ld, long q2_strideB,
                                const void* k2, long k2_ld, long k2_strideB, const void* vt2, long vt2_ld, int nk2,
                                float seg2_scale, void* out, long out_ld, long out_strideB,
                                int nq, int heads, int batch, float scale, hipStream_t stream) {
    TG_REQUIRE(q1 && k1 && vt1 && out, TG_ERR_ARG, "tg_attention_fwd: null pointer");
    TG_REQUIRE(nq > 0 && nk1 > 0 && heads > 0 && batch > 0, TG_ERR_SHAPE, "tg_attention_fwd: bad shape nq=%d nk1=%d", nq, nk1);
    TG_REQUIRE(q1_ld % 8 == 0 && k1_ld % 8 == 0 && vt1_ld % 64 == 0 && out_ld % 4 == 0 && q1_strideB % 8 == 0 &&
               k1_strideB % 8 == 0 && out_strideB % 4 == 0 && tg_aligned16(q1) && tg_aligned16(k1) && tg_aligned16(vt1) &&
               (((uintptr_t)out) & 7) == 0, TG_ERR_ALIGN, "tg_attention_fwd: segment 1 alignment");
    TG_REQUIRE(vt1_ld >= ((nk1 + 63) / 64) * 64, TG_ERR_SHAPE, "tg_attention_fwd: vt1_ld must cover nk1 rounded up to 64");
    AttnParams p{};
    p.s[0] = Seg{(const bf16_t*)q1, q1_ld, q1_strideB, (const bf16_t*)k1, k1_ld, k1_strideB, (const bf16_t*)vt1, vt1_ld, nk1};
    p.nseg = 1;
    if (q2) {
        TG_REQUIRE(k2 && vt2 && nk2 > 0, TG_ERR_ARG, "tg_attention_fwd: segment 2 incomplete");
        TG_REQUIRE(q2_ld % 8 == 0 && k2_ld % 8 == 0 && vt2_ld % 64 == 0 && q2_strideB % 8 == 0 && k2_strideB % 8 == 0 &&
                   tg_aligned16(q2) && tg_aligned16(k2) && tg_aligned16(vt2), TG_ERR_ALIGN, "tg_attention_fwd: segment 2 alignment");
        TG_REQUIRE(vt2_ld >= ((nk2 + 63) / 64) * 64, TG_ERR_SHAPE, "tg_attention_fwd: vt2_ld must cover nk2 rounded up to 64");
        p.s[1] = Seg{(const bf16_t*)q2, q2_ld, q2_strideB, (const bf16_t*)k2, k2_ld, k2_strideB, (const bf16_t*)vt2, vt2_ld, nk2};
        p.nseg = 2;
    }
    p.seg2_scale = seg2_scale;
    p.out = (bf16_t*)out; p.o_ld = out_ld; p.o_sb = out_strideB;
    p.nq = nq; p.heads = heads; p.batch = batch;
    p.scale_log2 = scale * 1.4426950408889634f;
    // 256-row query tiles (2 query blocks per wave) once they still give >= 4 workgroups per CU, else 128-row tiles
    const long wg256 = (long)((nq + 255) / 256) * heads * batch;
    // measured on MI355X at N=17776: 256-row tiles 900 TFLOP/s vs 128-row tiles 845; s_setprio around the MFMA clusters and an
    // intra-wave S(t+1)/softmax(t) software pipeline both measured slower (885 / 781) and were dropped
    if (wg256 >= 1024) {
        hipLaunchKernelGGL(attn_fwd_kernel<2>, dim3((unsigned)wg256), dim3(256), 0, stream, p);
    } else {
        const int nqt = (nq + 127) / 128;
        hipLaunchKernelGGL(attn_fwd_kernel<1>, dim3((unsigned)(nqt * heads * batch)), dim3(256), 0, stream, p);
    }
    TG_LAUNCH_CHECK("tg_attention_fwd");
    return TG_OK;
}
