// TEST SCAFFOLDING (tests/libtg_crosscheck.so, built by tests/csrc/Makefile; never linked into libtokensgen_hip.so): the correct-first attention
// backward kernels (head_dim 64) — LDS-staged 64 x 64 tiles, 4 waves, one 32 x 32 block per wave, explicit transposes while staging, and a
// one-thread-per-query statistics kernel — kept as an INDEPENDENT implementation of the same mathematics.  tests/test_train_gpu.py runs the autograd
// comparison on them through `tgx_attention_bwd` (same arguments as tg_attention_bwd) and holds the product kernels against them.
//   dV = P^T dO        dP = dO V^T        dS = P o (dP - D)        dQ = scale dS K       dK = scale dS^T Q
#include <stdarg.h>
#include <stdio.h>

#include "attention_bwd.h"

// the product library's error plumbing is not exported: a local one for the TG_REQUIRE / TG_LAUNCH_CHECK macros of common.h
static thread_local char g_xerr[512] = "";
extern "C" __attribute__((visibility("hidden"))) int tg_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_xerr, sizeof(g_xerr), fmt, ap);
    va_end(ap);
    return code;
}
extern "C" const char* tgx_last_error_string(void) { return g_xerr; }

namespace {

// statistics, the slow obvious way: one thread per (batch, head, query) walks all keys twice (row maximum, then the sum) and forms D_i = sum_d dO_id O_id
__global__ void xcheck_stats_kernel(BwdParams p) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)p.batch * p.heads * p.nq) return;
    const int q = (int)(i % p.nq), h = (int)((i / p.nq) % p.heads), b = (int)(i / ((long)p.nq * p.heads));
    const bf16_t* Q = p.q + (long)b * p.q_sb + (long)q * p.q_ld + h * HD;
    const bf16_t* O = p.o + (long)b * p.o_sb + (long)q * p.o_ld + h * HD;
    const bf16_t* G = p.dout + (long)b * p.do_sb + (long)q * p.do_ld + h * HD;
    float qv[HD], d = 0.f;
    for (int c = 0; c < HD; ++c) {
        qv[c] = bf16_to_f32(Q[c]);
        d += bf16_to_f32(O[c]) * bf16_to_f32(G[c]);
    }
    float m = -1e30f;
    for (int pass = 0; pass < 2; ++pass) {
        float l = 0.f;
        for (int k = 0; k < p.nk; ++k) {
            const bf16_t* Kr = p.k + (long)b * p.k_sb + (long)k * p.k_ld + h * HD;
            float s = 0.f;
            for (int c = 0; c < HD; ++c) s += qv[c] * bf16_to_f32(Kr[c]);
            s *= p.scale_log2;
            if (pass == 0) m = fmaxf(m, s); else l += exp2f(s - m);
        }
        if (pass == 1) p.lse[i] = m + log2f(l);
    }
    p.dsum[i] = d;
}

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
        *a = (p.accumulate & 2) ? *a + vk : vk;
        *c = (p.accumulate & 2) ? *c + vv : vv;
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
        *a = (p.accumulate & 1) ? *a + vq : vq;
    }
}


}  // namespace

// same arguments as tg_attention_bwd (include/tokensgen_hip.h) except the workspace: `ws` = 2 * batch * heads * nq floats (log-sum-exp | D); `lse` is ignored
extern "C" int tgx_attention_bwd(const void* q, long q_ld, long q_sb, const void* k, long k_ld, long k_sb, const void* v, long v_ld, long v_sb,
                                  const void* o, long o_ld, long o_sb, const void* dout, long do_ld, long do_sb,
                                  float* dq, long dq_ld, long dq_sb, float* dk, long dk_ld, long dk_sb, float* dv, long dv_ld, long dv_sb,
                                  int nq, int nk, int heads, int batch, float scale, int accumulate, const float* lse, float* ws, hipStream_t stream) {
    (void)lse;
    TG_REQUIRE(q && k && v && o && dout && dq && dk && dv && ws, TG_ERR_ARG, "tgx_attention_bwd: null pointer");
    TG_REQUIRE(nq > 0 && nk > 0 && heads > 0 && batch > 0, TG_ERR_SHAPE, "tgx_attention_bwd: bad shape");
    accumulate = accumulate == 1 ? 3 : (accumulate & 3);
    const long nrow = (long)batch * heads * nq;
    const BwdParams p{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)o, (const bf16_t*)dout, q_ld, q_sb, k_ld, k_sb, v_ld, v_sb,
                      o_ld, o_sb, do_ld, do_sb, dq, dk, dv, dq_ld, dq_sb, dk_ld, dk_sb, dv_ld, dv_sb, ws, ws + nrow, nullptr, nq, nk, heads,
                      batch, scale * 1.4426950408889634f, scale, accumulate, 0};
    hipLaunchKernelGGL(xcheck_stats_kernel, dim3((unsigned)((nrow + 127) / 128)), dim3(128), 0, stream, p);
    constexpr int LDS_KV = 8 * TILE_EL * 2 + 128 * 4, LDS_Q = 6 * TILE_EL * 2;
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_KV);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_Q);
    const dim3 gq((unsigned)((p.nq + TQ - 1) / TQ), (unsigned)(p.batch * p.heads)), gk((unsigned)((p.nk + TK - 1) / TK), (unsigned)(p.batch * p.heads));
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel, gk, dim3(256), LDS_KV, stream, p);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, gq, dim3(256), LDS_Q, stream, p);
    TG_LAUNCH_CHECK("tgx_attention_bwd (cross-check kernels)");
    return TG_OK;
}
