// Flash-attention BACKWARD for head_dim 64 on gfx950 MFMA (the training step's dominant operator; the forward is
// attention.hip).  Replaces what autograd runs for F.scaled_dot_product_attention in the reference's training step
// (attention_processor.py:2066-2125 under train_cogvideo_to2v.py:1995-2010): given Q, K, V, O, dO it returns dQ, dK, dV.
//
//   P = softmax(scale * Q K^T)           recomputed tile by tile from the row log-sum-exp (never materialised)
//   dV = P^T dO        dP = dO V^T        dS = P o (dP - D),  D_i = sum_d dO_id O_id        dQ = scale dS K       dK = scale dS^T Q
//
// Three launches, no atomics (deterministic): (1) statistics — the row log-sum-exp in the log2 domain and D; (2) one workgroup per
// 64-key tile walks the query tiles and accumulates dK, dV; (3) one workgroup per 64-query tile walks the key tiles and accumulates dQ.
// All products run on v_mfma_f32_32x32x16_bf16, which computes X Y^T for two k-contiguous operands: S = Q K^T and dP = dO V^T take
// the tiles as they are stored; dV = P^T dO, dK = dS^T Q, dQ = dS K need P^T / dS^T / dS (written by the lanes straight from their
// accumulators as 8-byte runs, because a lane's 4 consecutive accumulator rows are 4 consecutive positions along the dimension that
// has to become contiguous) and dO^T, Q^T, K^T (transposed while staging into LDS).
// Correct-first kernel: LDS-staged 64 x 64 tiles, 4 waves, one 32 x 32 block per wave; gradients are fp32 (accumulate flag: the To2V
// processor's three attention calls share K/V tensors, so their gradients add up).
#include "common.h"
#include "tokensgen_hip.h"

namespace {

constexpr int TQ = 64, TK = 64, HD = 64;
constexpr int LDT = 72;                    // LDS row stride in elements (64 + 8: 16-byte aligned rows, staggered banks)
constexpr int TILE_EL = 64 * LDT;

struct BwdParams {
    const bf16_t *q, *k, *v, *o, *dout;
    long q_ld, q_sb, k_ld, k_sb, v_ld, v_sb, o_ld, o_sb, do_ld, do_sb;
    float *dq, *dk, *dv;
    long dq_ld, dq_sb, dk_ld, dk_sb, dv_ld, dv_sb;
    float *lse, *dsum;                     // [batch][heads][nq] fp32 workspaces
    int nq, nk, heads, batch;
    float scale_log2, scale;
    int accumulate;
};

// rows [r0, r0 + 64) x 64 head columns of a [n][ld] bf16 matrix -> dst[row][LDT]; rows >= n are zero
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ src, long ld, int r0, int n, bf16_t* __restrict__ dst) {
    for (int v = threadIdx.x; v < 512; v += 256) {
        const int r = v >> 3, c = (v & 7) * 8;
        uint4 val = uint4{0, 0, 0, 0};
        if (r0 + r < n) val = *(const uint4*)(src + (long)(r0 + r) * ld + c);
        *(uint4*)(dst + r * LDT + c) = val;
    }
}
// the same tile transposed: dst[col][row]
__device__ __forceinline__ void stage_tile_t(const bf16_t* __restrict__ src, long ld, int r0, int n, bf16_t* __restrict__ dst) {
    for (int v = threadIdx.x; v < 512; v += 256) {
        const int r = v & 63, c = (v >> 6) * 8;           // consecutive threads -> consecutive rows: conflict-free column writes
        uint4 val = uint4{0, 0, 0, 0};
        if (r0 + r < n) val = *(const uint4*)(src + (long)(r0 + r) * ld + c);
        const uint32_t u[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dst[(c + 2 * i) * LDT + r] = (bf16_t)(u[i] & 0xffffu);
            dst[(c + 2 * i + 1) * LDT + r] = (bf16_t)(u[i] >> 16);
        }
    }
}

// C[32 x 32] += X[xr0 .. +32][0 .. 64) . Y[yr0 .. +32][0 .. 64)^T  over the 64-long contiguous dimension (4 MFMA k-steps of 16)
__device__ __forceinline__ f32x16 mma_nt(const bf16_t* X, int xr0, const bf16_t* Y, int yr0, f32x16 c, int lane) {
    const int j = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 a = *(const bf16x8*)(X + (xr0 + j) * LDT + ks * 16 + hi * 8);
        const bf16x8 b = *(const bf16x8*)(Y + (yr0 + j) * LDT + ks * 16 + hi * 8);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    return c;
}
// accumulator element r of lane (j, hi) sits at row 8*(r/4) + 4*hi + (r%4), column j of the 32 x 32 block
__device__ __forceinline__ int acc_row(int r, int hi) { return 8 * (r >> 2) + 4 * hi + (r & 3); }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// (1) statistics: lse2[i] = log2(sum_j 2^(scale_log2 * q_i.k_j)),  D[i] = sum_d dO[i][d] * O[i][d]
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_stats_kernel(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* sQ = (bf16_t*)smem_raw;
    bf16_t* sK = sQ + TILE_EL;
    float* sM = (float*)(sK + TILE_EL);                 // [64 queries][4 partials] running max
    float* sL = sM + 256;                               // ... and sum
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hi = lane >> 5;
    const int qb = wave >> 1, kb = wave & 1;
    const int h = blockIdx.y % p.heads, b = blockIdx.y / p.heads;
    const int q0 = blockIdx.x * TQ;
    const bf16_t* Q = p.q + (long)b * p.q_sb + h * HD;
    const bf16_t* Kp = p.k + (long)b * p.k_sb + h * HD;
    // D: 4 threads per query row, 16 head columns each
    {
        const int r = tid >> 2, c = (tid & 3) * 16;
        float d = 0.f;
        if (q0 + r < p.nq) {
            const bf16_t* o = p.o + (long)b * p.o_sb + (long)(q0 + r) * p.o_ld + h * HD + c;
            const bf16_t* g = p.dout + (long)b * p.do_sb + (long)(q0 + r) * p.do_ld + h * HD + c;
#pragma unroll
            for (int i = 0; i < 16; ++i) d += bf16_to_f32(o[i]) * bf16_to_f32(g[i]);
        }
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        if ((tid & 3) == 0 && q0 + r < p.nq) p.dsum[((long)b * p.heads + h) * p.nq + q0 + r] = d;
    }
    stage_tile(Q, p.q_ld, q0, p.nq, sQ);
    float m = -1e30f, l = 0.f;                            // this lane's query (column j of block qb) over its share of the keys
    for (int k0 = 0; k0 < p.nk; k0 += TK) {
        __syncthreads();
        stage_tile(Kp, p.k_ld, k0, p.nk, sK);
        __syncthreads();
        const f32x16 st = mma_nt(sK, kb * 32, sQ, qb * 32, zero16(), lane);        // S^T block: rows = keys, column = query j
        float s[16], mx = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + kb * 32 + acc_row(r, hi);
            s[r] = key < p.nk ? st[r] * p.scale_log2 : -1e30f;
            mx = fmaxf(mx, s[r]);
        }
        const float mn = fmaxf(m, mx);
        float add = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) add += exp2f(s[r] - mn);
        l = l * exp2f(m - mn) + add;
        m = mn;
    }
    sM[(qb * 32 + j) * 4 + kb * 2 + hi] = m;
    sL[(qb * 32 + j) * 4 + kb * 2 + hi] = l;
    __syncthreads();
    if (tid < TQ && q0 + tid < p.nq) {
        const float* mm = sM + tid * 4;
        const float* ll = sL + tid * 4;
        const float M = fmaxf(fmaxf(mm[0], mm[1]), fmaxf(mm[2], mm[3]));
        const float L = ll[0] * exp2f(mm[0] - M) + ll[1] * exp2f(mm[1] - M) + ll[2] * exp2f(mm[2] - M) + ll[3] * exp2f(mm[3] - M);
        p.lse[((long)b * p.heads + h) * p.nq + q0 + tid] = M + log2f(L);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// (2) dK, dV: one workgroup per 64-key tile
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* sK = (bf16_t*)smem_raw;          // [key][d]
    bf16_t* sV = sK + TILE_EL;
    bf16_t* sQ = sV + TILE_EL;               // [q][d]
    bf16_t* sdO = sQ + TILE_EL;
    bf16_t* sQt = sdO + TILE_EL;             // [d][q]
    bf16_t* sdOt = sQt + TILE_EL;
    bf16_t* sPt = sdOt + TILE_EL;            // [key][q]
    bf16_t* sdSt = sPt + TILE_EL;
    float* sLse = (float*)(sdSt + TILE_EL);  // [64]
    float* sD = sLse + 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hi = lane >> 5;
    const int b0 = wave >> 1, b1 = wave & 1;             // block coordinates of this wave in every 64 x 64 product
    const int h = blockIdx.y % p.heads, b = blockIdx.y / p.heads;
    const int k0 = blockIdx.x * TK;
    const bf16_t* Q = p.q + (long)b * p.q_sb + h * HD;
    const bf16_t* dO = p.dout + (long)b * p.do_sb + h * HD;
    stage_tile(p.k + (long)b * p.k_sb + h * HD, p.k_ld, k0, p.nk, sK);
    stage_tile(p.v + (long)b * p.v_sb + h * HD, p.v_ld, k0, p.nk, sV);
    f32x16 dk = zero16(), dv = zero16();                 // block (key block b0, d block b1)
    const long stat0 = ((long)b * p.heads + h) * p.nq;
    for (int q0 = 0; q0 < p.nq; q0 += TQ) {
        __syncthreads();                                  // previous iteration's readers of sQ.. / sPt.. are done
        stage_tile(Q, p.q_ld, q0, p.nq, sQ);
        stage_tile(dO, p.do_ld, q0, p.nq, sdO);
        stage_tile_t(Q, p.q_ld, q0, p.nq, sQt);
        stage_tile_t(dO, p.do_ld, q0, p.nq, sdOt);
        if (tid < TQ) {
            const bool ok = q0 + tid < p.nq;
            sLse[tid] = ok ? p.lse[stat0 + q0 + tid] : 0.f;
            sD[tid] = ok ? p.dsum[stat0 + q0 + tid] : 0.f;
        }
        __syncthreads();
        // S and dP blocks: rows = queries (block b0), column = key j of key block b1
        const f32x16 s = mma_nt(sQ, b0 * 32, sK, b1 * 32, zero16(), lane);
        const f32x16 dp = mma_nt(sdO, b0 * 32, sV, b1 * 32, zero16(), lane);
        const bool key_ok = k0 + b1 * 32 + j < p.nk;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int qr = b0 * 32 + 8 * g + 4 * hi;                          // 4 consecutive query rows
            float pv[4], ds[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = key_ok && (q0 + qr + e < p.nq);
                const float pe = ok ? exp2f(s[4 * g + e] * p.scale_log2 - sLse[qr + e]) : 0.f;
                pv[e] = pe;
                ds[e] = pe * (dp[4 * g + e] - sD[qr + e]);
            }
            uint2 a, c;
            a.x = pack_bf16x2(pv[0], pv[1]); a.y = pack_bf16x2(pv[2], pv[3]);
            c.x = pack_bf16x2(ds[0], ds[1]); c.y = pack_bf16x2(ds[2], ds[3]);
            *(uint2*)(sPt + (b1 * 32 + j) * LDT + qr) = a;                   // P^T[key][q .. q+3]
            *(uint2*)(sdSt + (b1 * 32 + j) * LDT + qr) = c;
        }
        __syncthreads();
        dv = mma_nt(sPt, b0 * 32, sdOt, b1 * 32, dv, lane);                   // dV[key][d] += P^T[key][:] . dO^T[d][:]
        dk = mma_nt(sdSt, b0 * 32, sQt, b1 * 32, dk, lane);                   // dK[key][d] += dS^T[key][:] . Q^T[d][:]
    }
    // rows = keys (block b0), column = head dim j of d block b1
    float* DK = p.dk + (long)b * p.dk_sb + h * HD + b1 * 32 + j;
    float* DV = p.dv + (long)b * p.dv_sb + h * HD + b1 * 32 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = k0 + b0 * 32 + acc_row(r, hi);
        if (key >= p.nk) continue;
        float* a = DK + (long)key * p.dk_ld;
        float* c = DV + (long)key * p.dv_ld;
        const float vk = dk[r] * p.scale, vv = dv[r];
        *a = p.accumulate ? *a + vk : vk;
        *c = p.accumulate ? *c + vv : vv;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// (3) dQ: one workgroup per 64-query tile
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* sQ = (bf16_t*)smem_raw;          // [q][d]
    bf16_t* sdO = sQ + TILE_EL;
    bf16_t* sK = sdO + TILE_EL;              // [key][d]
    bf16_t* sV = sK + TILE_EL;
    bf16_t* sKt = sV + TILE_EL;              // [d][key]
    bf16_t* sdS = sKt + TILE_EL;             // [q][key]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hi = lane >> 5;
    const int b0 = wave >> 1, b1 = wave & 1;
    const int h = blockIdx.y % p.heads, b = blockIdx.y / p.heads;
    const int q0 = blockIdx.x * TQ;
    const bf16_t* Kp = p.k + (long)b * p.k_sb + h * HD;
    const bf16_t* Vp = p.v + (long)b * p.v_sb + h * HD;
    stage_tile(p.q + (long)b * p.q_sb + h * HD, p.q_ld, q0, p.nq, sQ);
    stage_tile(p.dout + (long)b * p.do_sb + h * HD, p.do_ld, q0, p.nq, sdO);
    // S^T / dP^T blocks: rows = keys (block b0), column = query j of query block b1: the lane's query is fixed
    const int qrow = q0 + b1 * 32 + j;
    const bool q_ok = qrow < p.nq;
    const long stat = ((long)b * p.heads + h) * p.nq + (q_ok ? qrow : 0);
    const float lse = q_ok ? p.lse[stat] : 0.f, dsum = q_ok ? p.dsum[stat] : 0.f;
    f32x16 dq = zero16();                                 // block (query block b0, d block b1)
    for (int k0 = 0; k0 < p.nk; k0 += TK) {
        __syncthreads();
        stage_tile(Kp, p.k_ld, k0, p.nk, sK);
        stage_tile(Vp, p.v_ld, k0, p.nk, sV);
        stage_tile_t(Kp, p.k_ld, k0, p.nk, sKt);
        __syncthreads();
        const f32x16 st = mma_nt(sK, b0 * 32, sQ, b1 * 32, zero16(), lane);
        const f32x16 dpt = mma_nt(sV, b0 * 32, sdO, b1 * 32, zero16(), lane);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int kr = b0 * 32 + 8 * g + 4 * hi;                          // 4 consecutive keys
            float ds[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = q_ok && (k0 + kr + e < p.nk);
                const float pe = ok ? exp2f(st[4 * g + e] * p.scale_log2 - lse) : 0.f;
                ds[e] = pe * (dpt[4 * g + e] - dsum);
            }
            uint2 c;
            c.x = pack_bf16x2(ds[0], ds[1]); c.y = pack_bf16x2(ds[2], ds[3]);
            *(uint2*)(sdS + (b1 * 32 + j) * LDT + kr) = c;                    // dS[q][key .. key+3]
        }
        __syncthreads();
        dq = mma_nt(sdS, b0 * 32, sKt, b1 * 32, dq, lane);                    // dQ[q][d] += dS[q][:] . K^T[d][:]
    }
    float* DQ = p.dq + (long)b * p.dq_sb + h * HD + b1 * 32 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int q = q0 + b0 * 32 + acc_row(r, hi);
        if (q >= p.nq) continue;
        float* a = DQ + (long)q * p.dq_ld;
        const float vq = dq[r] * p.scale;
        *a = p.accumulate ? *a + vq : vq;
    }
}

}  // namespace

extern "C" long tg_attention_bwd_ws_floats(int nq, int heads, int batch) { return 2L * batch * heads * nq; }

extern "C" int tg_attention_bwd(const void* q, long q_ld, long q_sb, const void* k, long k_ld, long k_sb, const void* v, long v_ld, long v_sb,
                                const void* o, long o_ld, long o_sb, const void* dout, long do_ld, long do_sb,
                                float* dq, long dq_ld, long dq_sb, float* dk, long dk_ld, long dk_sb, float* dv, long dv_ld, long dv_sb,
                                int nq, int nk, int heads, int batch, float scale, int accumulate, float* ws, hipStream_t stream) {
    TG_REQUIRE(q && k && v && o && dout && dq && dk && dv && ws, TG_ERR_ARG, "tg_attention_bwd: null pointer");
    TG_REQUIRE(nq > 0 && nk > 0 && heads > 0 && batch > 0, TG_ERR_SHAPE, "tg_attention_bwd: bad shape nq=%d nk=%d heads=%d batch=%d", nq, nk, heads, batch);
    TG_REQUIRE(tg_aligned16(q) && tg_aligned16(k) && tg_aligned16(v) && tg_aligned16(o) && tg_aligned16(dout) && q_ld % 8 == 0 && k_ld % 8 == 0 &&
               v_ld % 8 == 0 && o_ld % 8 == 0 && do_ld % 8 == 0 && q_sb % 8 == 0 && k_sb % 8 == 0 && v_sb % 8 == 0 && o_sb % 8 == 0 && do_sb % 8 == 0,
               TG_ERR_ALIGN, "tg_attention_bwd: q/k/v/o/dO need 16-byte aligned rows");
    BwdParams p{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)o, (const bf16_t*)dout, q_ld, q_sb, k_ld, k_sb, v_ld, v_sb,
                o_ld, o_sb, do_ld, do_sb, dq, dk, dv, dq_ld, dq_sb, dk_ld, dk_sb, dv_ld, dv_sb, ws, ws + (long)batch * heads * nq, nq, nk, heads, batch,
                scale * 1.4426950408889634f, scale, accumulate};
    constexpr int LDS_STATS = 2 * TILE_EL * 2 + 512 * 4, LDS_KV = 8 * TILE_EL * 2 + 128 * 4, LDS_Q = 6 * TILE_EL * 2;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_KV);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_Q);
        attr = true;
    }
    const dim3 gq((unsigned)((nq + TQ - 1) / TQ), (unsigned)(batch * heads)), gk((unsigned)((nk + TK - 1) / TK), (unsigned)(batch * heads));
    hipLaunchKernelGGL(attn_bwd_stats_kernel, gq, dim3(256), LDS_STATS, stream, p);
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel, gk, dim3(256), LDS_KV, stream, p);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, gq, dim3(256), LDS_Q, stream, p);
    TG_LAUNCH_CHECK("tg_attention_bwd");
    return TG_OK;
}
