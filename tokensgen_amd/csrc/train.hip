// Backward kernels of the training step around the attention operator (reference loop train_cogvideo_to2v.py:1721-2021,
// trainable parameters = names containing "vip_", :1456-1481): per-head LayerNorm + RoPE backward (attention_processor.py:2031-2056),
// and the two HBM-bound helpers that let the existing MFMA GEMM (tg_gemm_bf16: C = A W^T) compute weight and input gradients —
// a 2-D transpose (dW = dY^T X needs both operands K-contiguous along the token axis) and deterministic column sums (bias gradients).
#include "common.h"
#include "tokensgen_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------------------
// y = rope(bf16(LN64(x) g + b)) * out_scale      (forward: norm.hip qk_norm_rope_slice)
// Backward for one (token, head) row of 64, 8 lanes x 8 elements like the forward:
//   dl = rope^T(dy * out_scale)        (pair (a, b), angle c/s:  da = dy_a c + dy_b s,  db = dy_b c - dy_a s)
//   dg += dl * x_hat   db += dl        (summed over tokens / heads / batch: per-block partial sums, fixed order)
//   dxh = dl * g       dx = rstd (dxh - mean(dxh) - x_hat mean(dxh * x_hat))
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int ROWS_PER_BLOCK = 32;          // (token, head) rows per pass of a 256-thread block
constexpr int PASSES = 16;                  // passes per block: 512 rows per block

__global__ __launch_bounds__(256) void qk_norm_rope_bwd_kernel(const bf16_t* __restrict__ x, long ld, long sb, const float* __restrict__ dy, long dld,
                                                               long dsb, bf16_t* __restrict__ dx, long xld, long xsb, int tokens, int heads, int batch,
                                                               const bf16_t* __restrict__ w, float eps, int start0, int len0,
                                                               const float* __restrict__ cos0, const float* __restrict__ sin0, int start1, int len1,
                                                               const float* __restrict__ cos1, const float* __restrict__ sin1, float out_scale,
                                                               float* __restrict__ partial) {
    __shared__ float red[ROWS_PER_BLOCK][2][64];
    const int tid = threadIdx.x, part = tid & 7, rl = tid >> 3;
    const long total = (long)batch * tokens * heads;
    float g[8], dgs[8], dbs[8];
    {
        const uint4 wv = *(const uint4*)(w + part * 8);
        const uint32_t wu[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { g[2 * i] = bf16lo_to_f32(wu[i]); g[2 * i + 1] = bf16hi_to_f32(wu[i]); }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) dgs[i] = dbs[i] = 0.f;
    for (int ps = 0; ps < PASSES; ++ps) {
        const long rowid = ((long)blockIdx.x * PASSES + ps) * ROWS_PER_BLOCK + rl;
        const bool live = rowid < total;
        const long rid = live ? rowid : total - 1;
        const int h = (int)(rid % heads);
        const long bt = rid / heads;
        const int t = (int)(bt % tokens), b = (int)(bt / tokens);
        const float* cs = nullptr;
        const float* sn = nullptr;
        if (t >= start0 && t < start0 + len0) { cs = cos0 + (long)(t - start0) * 64 + part * 8; sn = sin0 + (long)(t - start0) * 64 + part * 8; }
        else if (t >= start1 && t < start1 + len1) { cs = cos1 + (long)(t - start1) * 64 + part * 8; sn = sin1 + (long)(t - start1) * 64 + part * 8; }
        const uint4 raw = *(const uint4*)(x + (long)b * sb + (long)t * ld + h * 64 + part * 8);
        const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w};
        float v[8], s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = bf16lo_to_f32(u[i]); v[2 * i + 1] = bf16hi_to_f32(u[i]); s += v[2 * i] + v[2 * i + 1]; }
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        const float mean = s * (1.f / 64.f);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] -= mean; q += v[i] * v[i]; }
        q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
        const float rstd = rsqrtf(q * (1.f / 64.f) + eps);
        const float* dp = dy + (long)b * dsb + (long)t * dld + h * 64 + part * 8;
        const float4 d0 = *(const float4*)dp, d1 = *(const float4*)(dp + 4);
        float dl[8] = {d0.x * out_scale, d0.y * out_scale, d0.z * out_scale, d0.w * out_scale,
                       d1.x * out_scale, d1.y * out_scale, d1.z * out_scale, d1.w * out_scale};
        if (cs) {
            const float4 c0 = *(const float4*)cs, c1 = *(const float4*)(cs + 4), s0 = *(const float4*)sn, s1 = *(const float4*)(sn + 4);
            const float c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {      // forward: y_a = a c_a - b s_a,  y_b = b c_b + a s_b
                const float ya = dl[2 * i], yb = dl[2 * i + 1];
                dl[2 * i] = ya * c[2 * i] + yb * sv[2 * i + 1];
                dl[2 * i + 1] = yb * c[2 * i + 1] - ya * sv[2 * i];
            }
        }
        float m1 = 0.f, m2 = 0.f, dxh[8], xh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            xh[i] = v[i] * rstd;
            if (live) { dgs[i] += dl[i] * xh[i]; dbs[i] += dl[i]; }
            dxh[i] = dl[i] * g[i];
            m1 += dxh[i];
            m2 += dxh[i] * xh[i];
        }
        m1 += __shfl_xor(m1, 1, 64); m1 += __shfl_xor(m1, 2, 64); m1 += __shfl_xor(m1, 4, 64);
        m2 += __shfl_xor(m2, 1, 64); m2 += __shfl_xor(m2, 2, 64); m2 += __shfl_xor(m2, 4, 64);
        m1 *= (1.f / 64.f); m2 *= (1.f / 64.f);
        if (live) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = rstd * (dxh[i] - m1 - xh[i] * m2);
            uint4 ov;
            ov.x = pack_bf16x2(o[0], o[1]); ov.y = pack_bf16x2(o[2], o[3]); ov.z = pack_bf16x2(o[4], o[5]); ov.w = pack_bf16x2(o[6], o[7]);
            *(uint4*)(dx + (long)b * xsb + (long)t * xld + h * 64 + part * 8) = ov;
        }
    }
    // block partial of dg / db per channel: row lanes summed in a fixed order
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[rl][0][part * 8 + i] = dgs[i]; red[rl][1][part * 8 + i] = dbs[i]; }
    __syncthreads();
    if (tid < 128) {
        const int st = tid >> 6, c = tid & 63;
        float a = 0.f;
        for (int r = 0; r < ROWS_PER_BLOCK; ++r) a += red[r][st][c];
        partial[(long)blockIdx.x * 128 + tid] = a;
    }
}

// dst[c][r] = src[r][c] for r < rows, 0 for rows <= r < rows_pad (the token axis is zero-padded to the GEMM's K granule)
__global__ __launch_bounds__(256) void transpose_2d_kernel(const bf16_t* __restrict__ src, long ld, int rows, int cols, bf16_t* __restrict__ dst,
                                                           long ldd, int rows_pad) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < rows && c0 + c < cols) ? src[(long)(r0 + r) * ld + c0 + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (c0 + c < cols && r0 + r < rows_pad) dst[(long)(c0 + c) * ldd + r0 + r] = tile[r][c];
    }
}

// partial[blk][c] = sum over the block's rows of src[r][c]  (fp32; summed over blk on the host side in a fixed order)
constexpr int CS_ROWS = 256;
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ src, long ld, int rows, int cols, float* __restrict__ partial) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= cols) return;
    const int r0 = blockIdx.x * CS_ROWS, r1 = min(rows, r0 + CS_ROWS);
    float a = 0.f;
    for (int r = r0; r < r1; ++r) a += bf16_to_f32(src[(long)r * ld + c]);
    partial[(long)blockIdx.x * cols + c] = a;
}

}  // namespace

extern "C" long tg_qk_layernorm_rope_bwd_partial_floats(int tokens, int heads, int batch) {
    const long rows = (long)batch * tokens * heads, per = (long)ROWS_PER_BLOCK * PASSES;
    return ((rows + per - 1) / per) * 128;
}

extern "C" int tg_qk_layernorm_rope_bwd(const void* x, long ld, long strideB, const float* dy, long dy_ld, long dy_strideB, void* dx, long dx_ld,
                                        long dx_strideB, int tokens, int heads, int batch, const void* ln_weight, float eps, int start0, int len0,
                                        const float* cos0, const float* sin0, int start1, int len1, const float* cos1, const float* sin1,
                                        float out_scale, float* partial, hipStream_t stream) {
    TG_REQUIRE(x && dy && dx && ln_weight && partial, TG_ERR_ARG, "tg_qk_layernorm_rope_bwd: null pointer");
    TG_REQUIRE(tokens > 0 && heads > 0 && batch > 0, TG_ERR_SHAPE, "tg_qk_layernorm_rope_bwd: bad shape");
    TG_REQUIRE(ld % 8 == 0 && strideB % 8 == 0 && dx_ld % 8 == 0 && dx_strideB % 8 == 0 && dy_ld % 4 == 0 && dy_strideB % 4 == 0 && tg_aligned16(x) &&
               tg_aligned16(dx) && tg_aligned16(dy) && tg_aligned16(ln_weight), TG_ERR_ALIGN, "tg_qk_layernorm_rope_bwd: rows must be 16-byte aligned");
    TG_REQUIRE((len0 == 0 || (cos0 && sin0 && tg_aligned16(cos0) && tg_aligned16(sin0))) && (len1 == 0 || (cos1 && sin1 && tg_aligned16(cos1) && tg_aligned16(sin1))),
               TG_ERR_ARG, "tg_qk_layernorm_rope_bwd: rope tables missing or unaligned");
    const long rows = (long)batch * tokens * heads, per = (long)ROWS_PER_BLOCK * PASSES;
    hipLaunchKernelGGL(qk_norm_rope_bwd_kernel, dim3((unsigned)((rows + per - 1) / per)), dim3(256), 0, stream, (const bf16_t*)x, ld, strideB, dy, dy_ld,
                       dy_strideB, (bf16_t*)dx, dx_ld, dx_strideB, tokens, heads, batch, (const bf16_t*)ln_weight, eps, start0, len0, cos0, sin0, start1, len1,
                       cos1, sin1, out_scale, partial);
    TG_LAUNCH_CHECK("tg_qk_layernorm_rope_bwd");
    return TG_OK;
}

extern "C" int tg_transpose_2d(const void* src, long ld, int rows, int cols, void* dst, long ld_dst, int rows_pad, hipStream_t stream) {
    TG_REQUIRE(src && dst, TG_ERR_ARG, "tg_transpose_2d: null pointer");
    TG_REQUIRE(rows > 0 && cols > 0 && rows_pad >= rows && ld >= cols && ld_dst >= rows_pad, TG_ERR_SHAPE, "tg_transpose_2d: bad shape");
    hipLaunchKernelGGL(transpose_2d_kernel, dim3((unsigned)((rows_pad + 63) / 64), (unsigned)((cols + 63) / 64)), dim3(256), 0, stream, (const bf16_t*)src, ld,
                       rows, cols, (bf16_t*)dst, ld_dst, rows_pad);
    TG_LAUNCH_CHECK("tg_transpose_2d");
    return TG_OK;
}

extern "C" long tg_colsum_partial_floats(int rows, int cols) { return (long)((rows + CS_ROWS - 1) / CS_ROWS) * cols; }

extern "C" int tg_colsum(const void* src, long ld, int rows, int cols, float* partial, hipStream_t stream) {
    TG_REQUIRE(src && partial, TG_ERR_ARG, "tg_colsum: null pointer");
    TG_REQUIRE(rows > 0 && cols > 0 && ld >= cols, TG_ERR_SHAPE, "tg_colsum: bad shape");
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((rows + CS_ROWS - 1) / CS_ROWS), (unsigned)((cols + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)src, ld,
                       rows, cols, partial);
    TG_LAUNCH_CHECK("tg_colsum");
    return TG_OK;
}
