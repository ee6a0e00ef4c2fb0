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

// 16-byte form of transpose_2d_kernel (cols, row strides and rows_pad multiples of 8, 16-byte aligned bases): 128-byte runs on both the global read and the
// global write side; the tile's row stride of 66 elements keeps the 4-byte LDS stores and the strided 2-byte LDS loads off each other's banks.
__global__ __launch_bounds__(256) void transpose_2d_vec_kernel(const bf16_t* __restrict__ src, long ld, int rows, int cols, bf16_t* __restrict__ dst,
                                                               long ldd, int rows_pad) {
    __shared__ __attribute__((aligned(4))) bf16_t tile[64 * 66];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = threadIdx.x + k * 256, r = i >> 3, cc = (i & 7) * 8;
        uint4 v = uint4{0, 0, 0, 0};
        if (r0 + r < rows && c0 + cc < cols) v = *(const uint4*)(src + (long)(r0 + r) * ld + c0 + cc);
        uint32_t* d = (uint32_t*)(tile + r * 66 + cc);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = threadIdx.x + k * 256, c = i >> 3, rr = (i & 7) * 8;
        if (c0 + c >= cols || r0 + rr >= rows_pad) continue;
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (uint32_t)tile[(rr + 2 * e) * 66 + c] | ((uint32_t)tile[(rr + 2 * e + 1) * 66 + c] << 16);
        *(uint4*)(dst + (long)(c0 + c) * ldd + r0 + rr) = uint4{w[0], w[1], w[2], w[3]};
    }
}

// partial[blk][c] = sum over the block's rows of src[r][c]  (fp32; summed over blk on the host side in a fixed order)
// Rows per block: 256 for the big matrices; fewer for short ones so that the launch still has ~2000 workgroups (a [480 x 3072] product tensor of
// the vip rows was 24 workgroups walking 256 rows each: 75 us per call, 16 calls per layer).  A function of the shape only: the summation order
// stays fixed, and tg_colsum_partial_floats() uses the same value.
static inline int cs_rows(int rows, int cols) {
    const long cb = (cols + 255) / 256;
    long r = ((long)rows * cb + 2047) / 2048;
    r = r < 8 ? 8 : (r > 256 ? 256 : r);
    return (int)r;
}
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ src, long ld, int rows, int cols, float* __restrict__ partial, int CS_ROWS) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= cols) return;
    const int r0 = blockIdx.x * CS_ROWS, r1 = min(rows, r0 + CS_ROWS);
    float a = 0.f;
    for (int r = r0; r < r1; ++r) a += bf16_to_f32(src[(long)r * ld + c]);
    partial[(long)blockIdx.x * cols + c] = a;
}

// 8 columns per thread (16-byte loads; cols and ld multiples of 8): same per-column summation order as colsum_kernel, a quarter of the load instructions
__global__ __launch_bounds__(256) void colsum_vec_kernel(const bf16_t* __restrict__ src, long ld, int rows, int cols, float* __restrict__ partial, int CS_ROWS) {
    const int c = (blockIdx.y * 256 + threadIdx.x) * 8;
    if (c >= cols) return;
    const int r0 = blockIdx.x * CS_ROWS, r1 = min(rows, r0 + CS_ROWS);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < r1; ++r) {
        const uint4 v = *(const uint4*)(src + (long)r * ld + c);
        const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[2 * i] += bf16lo_to_f32(u[i]); a[2 * i + 1] += bf16hi_to_f32(u[i]); }
    }
    float* dst = partial + (long)blockIdx.x * cols + c;
    *(float4*)dst = float4{a[0], a[1], a[2], a[3]};
    *(float4*)(dst + 4) = float4{a[4], a[5], a[6], a[7]};
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward of tg_adaln_modulate:  y = ln * (1 + scale[g]) + shift[g],  ln = bf16(x_hat * gamma + beta),  x_hat = (x - mean) * rstd
// (normalization.py:441-460, 477-488).  One wave per token row.  Per element it also emits the three products whose column sums (over all
// rows, or over the rows of one group) are the parameter gradients:  t_dln = dy (1 + scale)  [-> d beta],  t_dlnx = t_dln * x_hat  [-> d gamma],
// t_dyln = dy * ln  [-> d scale[g]]  (d shift[g] = column sums of dy itself).  fp32 [rows][dim]; summed by tg_colsum_f32 in a fixed order.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adaln_bwd_scalar_kernel(const bf16_t* __restrict__ x, long ldx, long sxb, const bf16_t* __restrict__ dy, long ldd, long sdb,
                                                        bf16_t* __restrict__ dx, long ldo, long sob, const bf16_t* __restrict__ w,
                                                        const bf16_t* __restrict__ bvec, float eps, int tokens, int dim, int batch, int modulate,
                                                        tg_group_table g, float* __restrict__ t_dln, float* __restrict__ t_dlnx,
                                                        float* __restrict__ t_dyln, const bf16_t* __restrict__ add, long lda, long sab) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)tokens * batch) return;
    const int b = (int)(row / tokens), t = (int)(row % tokens);
    const bf16_t* xr = x + (long)b * sxb + (long)t * ldx;
    const bf16_t* dr = dy + (long)b * sdb + (long)t * ldd;
    bf16_t* outr = dx + (long)b * sob + (long)t * ldo;
    float s = 0.f;
    for (int c = lane; c < dim; c += 64) s += bf16_to_f32(xr[c]);
    const float mean = wave_sum(s) / dim;
    float q = 0.f;
    for (int c = lane; c < dim; c += 64) { const float d = bf16_to_f32(xr[c]) - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / dim + eps);
    const bf16_t* scale = nullptr;
    if (modulate) {
        const int gi = g.tok_group[t];
        scale = (const bf16_t*)g.mod + (long)b * g.mod_batch_stride + (long)g.row[gi] * g.mod_ld + g.scale_col[gi];
    }
    float m1 = 0.f, m2 = 0.f;
    for (int c = lane; c < dim; c += 64) {
        const float xh = (bf16_to_f32(xr[c]) - mean) * rstd;
        const float gam = w ? bf16_to_f32(w[c]) : 1.f, bet = bvec ? bf16_to_f32(bvec[c]) : 0.f;
        const float ln = round_bf16(xh * gam + bet);
        const float d = bf16_to_f32(dr[c]);
        const float dln = d * (1.f + (scale ? bf16_to_f32(scale[c]) : 0.f));
        if (t_dln) {                                       // the products are wanted only where the norm's parameters train (the vip rows)
            const long o = row * dim + c;
            t_dln[o] = dln; t_dlnx[o] = dln * xh; t_dyln[o] = d * ln;
        }
        const float dxh = dln * gam;
        m1 += dxh; m2 += dxh * xh;
    }
    m1 = wave_sum(m1) / dim; m2 = wave_sum(m2) / dim;
    const bf16_t* ar = add ? add + (long)b * sab + (long)t * lda : nullptr;
    for (int c = lane; c < dim; c += 64) {
        const float xh = (bf16_to_f32(xr[c]) - mean) * rstd;
        const float dln = bf16_to_f32(dr[c]) * (1.f + (scale ? bf16_to_f32(scale[c]) : 0.f));
        const float dxh = dln * (w ? bf16_to_f32(w[c]) : 1.f);
        float v = rstd * (dxh - m1 - xh * m2);
        if (ar) v = round_bf16(v) + bf16_to_f32(ar[c]);     // + the gradient arriving over the residual connection (both bf16 tensors in the reference)
        outr[c] = f32_to_bf16(v);
    }
}

// The same with 16-byte accesses (8 bf16 per lane per access; needs dim % 8 == 0 and 16-byte aligned rows): the scalar form above moved 2 bytes
// per lane per load and ran at ~1 TB/s.
__device__ __forceinline__ void unpack8(uint4 v, float (&f)[8]) {
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = bf16lo_to_f32(u[i]); f[2 * i + 1] = bf16hi_to_f32(u[i]); }
}
// NC > 0: the row's chunks (dim <= 512 NC) stay in registers as packed bf16 — x, dy and the modulation scale are read from memory ONCE (the NC = 0 form re-reads the row
// for each of its four passes: L1 / L2 hits, but four dependent load -> reduce chains per row; 3.2 TB/s of compulsory traffic at dim 3072).  Same arithmetic in the
// same order: bitwise the same result.
template <int NC>
__global__ __launch_bounds__(256) void adaln_bwd_kernel(const bf16_t* __restrict__ x, long ldx, long sxb, const bf16_t* __restrict__ dy, long ldd, long sdb,
                                                        bf16_t* __restrict__ dx, long ldo, long sob, const bf16_t* __restrict__ w,
                                                        const bf16_t* __restrict__ bvec, float eps, int tokens, int dim, int batch, int modulate,
                                                        tg_group_table g, float* __restrict__ t_dln, float* __restrict__ t_dlnx,
                                                        float* __restrict__ t_dyln, const bf16_t* __restrict__ add, long lda, long sab) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)tokens * batch) return;
    const int b = (int)(row / tokens), t = (int)(row % tokens);
    const bf16_t* xr = x + (long)b * sxb + (long)t * ldx;
    const bf16_t* dr = dy + (long)b * sdb + (long)t * ldd;
    bf16_t* outr = dx + (long)b * sob + (long)t * ldo;
    const int nch = dim >> 3;
    constexpr int NR = NC > 0 ? NC : 1;
    uint4 xq[NR], dq[NR], sq[NR];                      // (the residual rows are read in the last pass: holding them too costs a wave of occupancy and measured slower)
    const bf16_t* scale = nullptr;
    if (modulate) {
        const int gi = g.tok_group[t];
        scale = (const bf16_t*)g.mod + (long)b * g.mod_batch_stride + (long)g.row[gi] * g.mod_ld + g.scale_col[gi];
    }
    const bf16_t* ar = add ? add + (long)b * sab + (long)t * lda : nullptr;
    if (NC > 0) {
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int c = lane + 64 * k;
            const bool in = c < nch;
            xq[k] = in ? *(const uint4*)(xr + c * 8) : uint4{0, 0, 0, 0};
            dq[k] = in ? *(const uint4*)(dr + c * 8) : uint4{0, 0, 0, 0};
            sq[k] = in && scale ? *(const uint4*)(scale + c * 8) : uint4{0, 0, 0, 0};
        }
    }
    auto ld_x = [&](int k, int c, float (&f)[8]) { unpack8(NC > 0 ? xq[k] : *(const uint4*)(xr + c * 8), f); };
    auto ld_d = [&](int k, int c, float (&f)[8]) { unpack8(NC > 0 ? dq[k] : *(const uint4*)(dr + c * 8), f); };
    auto ld_s = [&](int k, int c, float (&f)[8]) { unpack8(NC > 0 ? sq[k] : *(const uint4*)(scale + c * 8), f); };
    float s = 0.f, f[8], e[8];
#pragma unroll NR
    for (int k = 0; k < (NC > 0 ? NC : 1 << 20); ++k) {
        const int c = lane + 64 * k;
        if (c >= nch) break;
        ld_x(k, c, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) s += f[i];
    }
    const float mean = wave_sum(s) / dim;
    float q = 0.f;
#pragma unroll NR
    for (int k = 0; k < (NC > 0 ? NC : 1 << 20); ++k) {
        const int c = lane + 64 * k;
        if (c >= nch) break;
        ld_x(k, c, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = f[i] - mean; q += d * d; }
    }
    const float rstd = rsqrtf(wave_sum(q) / dim + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll NR
    for (int k = 0; k < (NC > 0 ? NC : 1 << 20); ++k) {
        const int c = lane + 64 * k;
        if (c >= nch) break;
        float gam[8], bet[8], sc[8];
        ld_x(k, c, f);
        ld_d(k, c, e);
        if (w) unpack8(*(const uint4*)(w + c * 8), gam);
        if (bvec) unpack8(*(const uint4*)(bvec + c * 8), bet);
        if (scale) ld_s(k, c, sc);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float xh = (f[i] - mean) * rstd;
            const float ga = w ? gam[i] : 1.f;
            const float dln = e[i] * (1.f + (scale ? sc[i] : 0.f));
            if (t_dln) {
                const long o = row * dim + c * 8 + i;
                t_dln[o] = dln; t_dlnx[o] = dln * xh; t_dyln[o] = e[i] * round_bf16(xh * ga + (bvec ? bet[i] : 0.f));
            }
            const float dxh = dln * ga;
            m1 += dxh; m2 += dxh * xh;
        }
    }
    m1 = wave_sum(m1) / dim; m2 = wave_sum(m2) / dim;
#pragma unroll NR
    for (int k = 0; k < (NC > 0 ? NC : 1 << 20); ++k) {
        const int c = lane + 64 * k;
        if (c >= nch) break;
        float gam[8], sc[8], a8[8], v[8];
        ld_x(k, c, f);
        ld_d(k, c, e);
        if (w) unpack8(*(const uint4*)(w + c * 8), gam);
        if (scale) ld_s(k, c, sc);
        if (ar) unpack8(*(const uint4*)(ar + c * 8), a8);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float xh = (f[i] - mean) * rstd;
            const float dxh = e[i] * (1.f + (scale ? sc[i] : 0.f)) * (w ? gam[i] : 1.f);
            v[i] = rstd * (dxh - m1 - xh * m2);
            if (ar) v[i] = round_bf16(v[i]) + a8[i];
        }
        *(uint4*)(outr + c * 8) = uint4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    }
}

// Backward of the gated residual  out = res + gate[g] * y  (cogvideox_transformer_3d.py:290-293, 318-324):  dy = gate[g] * dout (bf16),
// t_dgate = dout * y (fp32; its column sums over the rows of group g are d gate[g]);  d res = dout.
__global__ __launch_bounds__(256) void gate_res_bwd_scalar_kernel(const bf16_t* __restrict__ dout, long ldd, long sdb, const bf16_t* __restrict__ y, long ldy,
                                                           long syb, bf16_t* __restrict__ dyo, long ldo, long sob, int tokens, int dim, int batch,
                                                           tg_group_table g, float* __restrict__ t_dgate, int t_row0) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)batch * tokens * dim;
    if (i >= total) return;
    const int c = (int)(i % dim);
    const long row = i / dim;
    const int b = (int)(row / tokens), t = (int)(row % tokens);
    const int gi = g.tok_group[t];
    const float gate = bf16_to_f32(((const bf16_t*)g.mod)[(long)b * g.mod_batch_stride + (long)g.row[gi] * g.mod_ld + g.gate_col[gi] + c]);
    const float d = bf16_to_f32(dout[(long)b * sdb + (long)t * ldd + c]);
    dyo[(long)b * sob + (long)t * ldo + c] = f32_to_bf16(gate * d);
    if (t >= t_row0) t_dgate[((long)b * (tokens - t_row0) + (t - t_row0)) * dim + c] = d * bf16_to_f32(y[(long)b * syb + (long)t * ldy + c]);
}

__global__ __launch_bounds__(256) void gate_res_bwd_kernel(const bf16_t* __restrict__ dout, long ldd, long sdb, const bf16_t* __restrict__ y, long ldy,
                                                           long syb, bf16_t* __restrict__ dyo, long ldo, long sob, int tokens, int dim, int batch,
                                                           tg_group_table g, float* __restrict__ t_dgate, int t_row0) {       // 8 elements per thread
    const int nch = dim >> 3;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)batch * tokens * nch) return;
    const int c = (int)(i % nch) * 8;
    const long row = i / nch;
    const int b = (int)(row / tokens), t = (int)(row % tokens);
    const int gi = g.tok_group[t];
    float ga[8], d[8], yv[8], o[8];
    unpack8(*(const uint4*)((const bf16_t*)g.mod + (long)b * g.mod_batch_stride + (long)g.row[gi] * g.mod_ld + g.gate_col[gi] + c), ga);
    unpack8(*(const uint4*)(dout + (long)b * sdb + (long)t * ldd + c), d);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = ga[k] * d[k];
    *(uint4*)(dyo + (long)b * sob + (long)t * ldo + c) = uint4{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
    if (t >= t_row0) {
        unpack8(*(const uint4*)(y + (long)b * syb + (long)t * ldy + c), yv);
        float* dst = t_dgate + ((long)b * (tokens - t_row0) + (t - t_row0)) * dim + c;
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k] = d[k] * yv[k];
    }
}

// mode 0: y = silu(x);  mode 1: dx = dy * gelu_tanh'(x)  (F.gelu(approximate="tanh"), diffusers FeedForward);  mode 2: y = gelu_tanh(x), the
// same function as the GEMM's GELU epilogue (the training forward keeps the pre-activation for mode 1 and applies the activation in this pass
// instead of running the FF1 GEMM a second time)
__device__ __forceinline__ float act_one(float v, float dyv, int mode) {
    if (mode == 0) return v / (1.f + __expf(-v));
    if (mode == 2) return gelu_tanh(v);
    return gelu_tanh_bwd(v, dyv);               // (common.h)
}
__global__ __launch_bounds__(256) void act_scalar_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, bf16_t* __restrict__ out, long n, int mode) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = f32_to_bf16(act_one(bf16_to_f32(x[i]), mode == 1 ? bf16_to_f32(dy[i]) : 0.f, mode));      // (the same function as the 16-byte form: same values)
}

__global__ __launch_bounds__(256) void act_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, bf16_t* __restrict__ out, long n8, int mode) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;                  // 8 elements per thread
    if (i >= n8) return;
    float v[8], d[8] = {0, 0, 0, 0, 0, 0, 0, 0}, o[8];
    unpack8(*(const uint4*)(x + i * 8), v);
    if (mode == 1) unpack8(*(const uint4*)(dy + i * 8), d);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = act_one(v[k], d[k], mode);
    *(uint4*)(out + i * 8) = uint4{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
}

// Column sums of SEVERAL matrices in one launch (a vip_norm's gradients need eight: 16 launches with their `partial.sum(0)`s, now 2).  Every item is cut into the
// same number of row blocks NB, so the partials form ONE [NB][total columns] matrix that a single fixed-order sum finishes; the table travels by value.
struct ColsumTable {
    const void* src[TG_COLSUM_MAX];
    long ld[TG_COLSUM_MAX];
    int rows[TG_COLSUM_MAX], cols[TG_COLSUM_MAX], col0[TG_COLSUM_MAX];      // col0: first column of the item in the joint partial matrix
    unsigned first_block[TG_COLSUM_MAX + 1];                                  // blockIdx.y ranges (256 columns per block)
    unsigned f32_mask;
    int count, total_cols;
};
__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumTable t, float* __restrict__ partial) {
    int it = 0;
    while (it + 1 < t.count && blockIdx.y >= t.first_block[it + 1]) ++it;
    const int c = (blockIdx.y - t.first_block[it]) * 256 + threadIdx.x;
    if (c >= t.cols[it]) return;
    const int rows = t.rows[it], per = (rows + (int)gridDim.x - 1) / (int)gridDim.x;
    const int r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
    const long ld = t.ld[it];
    float a = 0.f;
    if ((t.f32_mask >> it) & 1u) for (int r = r0; r < r1; ++r) a += ((const float*)t.src[it])[(long)r * ld + c];
    else for (int r = r0; r < r1; ++r) a += bf16_to_f32(((const bf16_t*)t.src[it])[(long)r * ld + c]);
    partial[(long)blockIdx.x * t.total_cols + t.col0[it] + c] = a;
}

__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ src, long ld, int rows, int cols, float* __restrict__ partial, int CS_ROWS) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= cols) return;
    const int r0 = blockIdx.x * CS_ROWS, r1 = min(rows, r0 + CS_ROWS);
    float a = 0.f;
    for (int r = r0; r < r1; ++r) a += src[(long)r * ld + c];
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
    const bool vec = cols % 8 == 0 && rows_pad % 8 == 0 && ld % 8 == 0 && ld_dst % 8 == 0 && tg_aligned16(src) && tg_aligned16(dst);
    if (vec)
        hipLaunchKernelGGL(transpose_2d_vec_kernel, dim3((unsigned)((rows_pad + 63) / 64), (unsigned)((cols + 63) / 64)), dim3(256), 0, stream, (const bf16_t*)src,
                           ld, rows, cols, (bf16_t*)dst, ld_dst, rows_pad);
    else
        hipLaunchKernelGGL(transpose_2d_kernel, dim3((unsigned)((rows_pad + 63) / 64), (unsigned)((cols + 63) / 64)), dim3(256), 0, stream, (const bf16_t*)src, ld,
                           rows, cols, (bf16_t*)dst, ld_dst, rows_pad);
    TG_LAUNCH_CHECK("tg_transpose_2d");
    return TG_OK;
}

extern "C" long tg_colsum_partial_floats(int rows, int cols) {
    const int CS_ROWS = cs_rows(rows, cols);
    return (long)((rows + CS_ROWS - 1) / CS_ROWS) * cols;
}

extern "C" int tg_colsum_multi(const tg_colsum_item* items, int count, int row_blocks, float* partial, hipStream_t stream) {
    TG_REQUIRE(items && partial && count > 0 && count <= TG_COLSUM_MAX && row_blocks > 0 && row_blocks <= 1024, TG_ERR_ARG, "tg_colsum_multi: 1..%d items, 1..1024 row blocks", TG_COLSUM_MAX);
    ColsumTable t{};
    t.count = count;
    unsigned nb = 0;
    int col0 = 0;
    for (int i = 0; i < count; ++i) {
        const tg_colsum_item& it = items[i];
        TG_REQUIRE(it.src && it.rows > 0 && it.cols > 0 && it.ld >= it.cols, TG_ERR_SHAPE, "tg_colsum_multi: item %d: bad shape", i);
        t.src[i] = it.src; t.ld[i] = it.ld; t.rows[i] = it.rows; t.cols[i] = it.cols; t.col0[i] = col0;
        if (it.src_is_f32) t.f32_mask |= 1u << i;
        t.first_block[i] = nb;
        nb += (unsigned)((it.cols + 255) / 256);
        col0 += it.cols;
    }
    t.first_block[count] = nb;
    t.total_cols = col0;
    hipLaunchKernelGGL(colsum_multi_kernel, dim3((unsigned)row_blocks, nb), dim3(256), 0, stream, t, partial);
    TG_LAUNCH_CHECK("tg_colsum_multi");
    return TG_OK;
}

extern "C" int tg_colsum(const void* src, long ld, int rows, int cols, float* partial, hipStream_t stream) {
    TG_REQUIRE(src && partial, TG_ERR_ARG, "tg_colsum: null pointer");
    TG_REQUIRE(rows > 0 && cols > 0 && ld >= cols, TG_ERR_SHAPE, "tg_colsum: bad shape");
    const int CS_ROWS = cs_rows(rows, cols);
    if (cols % 8 == 0 && ld % 8 == 0 && tg_aligned16(src) && tg_aligned16(partial))
        hipLaunchKernelGGL(colsum_vec_kernel, dim3((unsigned)((rows + CS_ROWS - 1) / CS_ROWS), (unsigned)((cols / 8 + 255) / 256)), dim3(256), 0, stream,
                           (const bf16_t*)src, ld, rows, cols, partial, CS_ROWS);
    else
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((rows + CS_ROWS - 1) / CS_ROWS), (unsigned)((cols + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)src, ld,
                           rows, cols, partial, CS_ROWS);
    TG_LAUNCH_CHECK("tg_colsum");
    return TG_OK;
}

extern "C" int tg_adaln_modulate_bwd(const void* x, long ldx, long strideX, const void* dy, long ld_dy, long stride_dy, void* dx, long ld_dx, long stride_dx,
                                     const void* ln_weight, const void* ln_bias, float eps, int tokens, int dim, int batch, int modulate,
                                     const tg_group_table* g, float* t_dln, float* t_dlnx, float* t_dyln, const void* add, long ld_add, long stride_add,
                                     hipStream_t stream) {
    TG_REQUIRE(x && dy && dx, TG_ERR_ARG, "tg_adaln_modulate_bwd: null pointer");
    TG_REQUIRE((t_dln && t_dlnx && t_dyln) || (!t_dln && !t_dlnx && !t_dyln), TG_ERR_ARG, "tg_adaln_modulate_bwd: give all three product tensors or none");
    TG_REQUIRE(tokens > 0 && dim > 0 && batch > 0 && (!modulate || g), TG_ERR_SHAPE, "tg_adaln_modulate_bwd: bad shape");
    tg_group_table gt{};
    if (g) gt = *g;
    const long rows = (long)tokens * batch;
    // 16-byte form: every row pointer the kernel forms (x, dy, dx, add, the LayerNorm vectors, the scale row of the group table) must be 16-byte aligned
    auto al = [](const void* p_, long ld, long sb) { return !p_ || (tg_aligned16(p_) && ld % 8 == 0 && sb % 8 == 0); };
    bool vec = dim % 8 == 0 && al(x, ldx, strideX) && al(dy, ld_dy, stride_dy) && al(dx, ld_dx, stride_dx) && al(add, ld_add, stride_add) &&
               al(ln_weight, 0, 0) && al(ln_bias, 0, 0);
    if (vec && modulate) {
        vec = tg_aligned16(gt.mod) && gt.mod_ld % 8 == 0 && gt.mod_batch_stride % 8 == 0;
        for (int i = 0; i < TG_MAX_GROUPS; ++i) vec = vec && gt.scale_col[i] % 8 == 0;
    }
    if (vec)
    {
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const bf16_t*)x, ldx, strideX, (const bf16_t*)dy, ld_dy, stride_dy,
                               (bf16_t*)dx, ld_dx, stride_dx, (const bf16_t*)ln_weight, (const bf16_t*)ln_bias, eps, tokens, dim, batch, modulate, gt, t_dln, t_dlnx,
                               t_dyln, (const bf16_t*)add, ld_add, stride_add);
        };
        if (dim <= 512 * 6) go(adaln_bwd_kernel<6>);          // (the 5B model: 3072)
        else if (dim <= 512 * 8) go(adaln_bwd_kernel<8>);
        else go(adaln_bwd_kernel<0>);
    }
    else
        hipLaunchKernelGGL(adaln_bwd_scalar_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const bf16_t*)x, ldx, strideX, (const bf16_t*)dy, ld_dy,
                           stride_dy, (bf16_t*)dx, ld_dx, stride_dx, (const bf16_t*)ln_weight, (const bf16_t*)ln_bias, eps, tokens, dim, batch, modulate, gt, t_dln,
                           t_dlnx, t_dyln, (const bf16_t*)add, ld_add, stride_add);
    TG_LAUNCH_CHECK("tg_adaln_modulate_bwd");
    return TG_OK;
}

extern "C" int tg_gate_residual_bwd(const void* dout, long ld_dout, long stride_dout, const void* y, long ldy, long strideY, void* dy, long ld_dy, long stride_dy,
                                    int tokens, int dim, int batch, const tg_group_table* gate, float* t_dgate, int t_row0, hipStream_t stream) {
    TG_REQUIRE(dout && y && dy && gate && t_dgate, TG_ERR_ARG, "tg_gate_residual_bwd: null pointer");
    TG_REQUIRE(tokens > 0 && dim > 0 && batch > 0 && t_row0 >= 0 && t_row0 < tokens, TG_ERR_SHAPE, "tg_gate_residual_bwd: bad shape");
    const long total = (long)batch * tokens * dim;
    bool vec = dim % 8 == 0 && tg_aligned16(dout) && tg_aligned16(y) && tg_aligned16(dy) && tg_aligned16(gate->mod) && tg_aligned16(t_dgate) && ld_dout % 8 == 0 &&
               stride_dout % 8 == 0 && ldy % 8 == 0 && strideY % 8 == 0 && ld_dy % 8 == 0 && stride_dy % 8 == 0 && gate->mod_ld % 8 == 0 &&
               gate->mod_batch_stride % 8 == 0;
    for (int i = 0; i < TG_MAX_GROUPS; ++i) vec = vec && gate->gate_col[i] % 8 == 0;
    if (vec)
        hipLaunchKernelGGL(gate_res_bwd_kernel, dim3((unsigned)((total / 8 + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)dout, ld_dout, stride_dout,
                           (const bf16_t*)y, ldy, strideY, (bf16_t*)dy, ld_dy, stride_dy, tokens, dim, batch, *gate, t_dgate, t_row0);
    else
        hipLaunchKernelGGL(gate_res_bwd_scalar_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)dout, ld_dout, stride_dout,
                           (const bf16_t*)y, ldy, strideY, (bf16_t*)dy, ld_dy, stride_dy, tokens, dim, batch, *gate, t_dgate, t_row0);
    TG_LAUNCH_CHECK("tg_gate_residual_bwd");
    return TG_OK;
}

extern "C" int tg_act(const void* x, const void* dy, void* out, long n, int mode, hipStream_t stream) {
    TG_REQUIRE(x && out && (mode != 1 || dy), TG_ERR_ARG, "tg_act: null pointer");
    TG_REQUIRE(n > 0 && mode >= 0 && mode <= 2, TG_ERR_SHAPE, "tg_act: bad arguments");
    if (n % 8 == 0 && tg_aligned16(x) && tg_aligned16(out) && (!dy || tg_aligned16(dy)))
        hipLaunchKernelGGL(act_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)out, n / 8, mode);
    else
        hipLaunchKernelGGL(act_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)out, n, mode);
    TG_LAUNCH_CHECK("tg_act");
    return TG_OK;
}

extern "C" int tg_colsum_f32(const float* src, long ld, int rows, int cols, float* partial, hipStream_t stream) {
    TG_REQUIRE(src && partial, TG_ERR_ARG, "tg_colsum_f32: null pointer");
    TG_REQUIRE(rows > 0 && cols > 0 && ld >= cols, TG_ERR_SHAPE, "tg_colsum_f32: bad shape");
    const int CS_ROWS = cs_rows(rows, cols);
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((unsigned)((rows + CS_ROWS - 1) / CS_ROWS), (unsigned)((cols + 255) / 256)), dim3(256), 0, stream, src, ld, rows, cols, partial,
                       CS_ROWS);
    TG_LAUNCH_CHECK("tg_colsum_f32");
    return TG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Optimizer step on a flat arena (train_cogvideo_to2v.py:2012-2021: clip_grad_norm_ on the transformer's parameters, AdamW, zero_grad).
// All HBM-bound streaming passes: 16-byte accesses, grid-stride, no host synchronisation (the clip coefficient stays on the device).
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int OPT_BLOCK = 256, OPT_PER_THREAD = 8, OPT_MAX_BLOCKS = 2048;

// acc[i] += scale * g[i]   (g bf16 or fp32): gradient accumulation over micro-steps (accelerate's `accumulate`: loss / accumulation steps)
template <bool SRC_BF16>
__global__ __launch_bounds__(OPT_BLOCK) void grad_accum_kernel(const void* __restrict__ g, float* __restrict__ acc, long n, float scale, int overwrite) {
    for (long i = (long)blockIdx.x * OPT_BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * OPT_BLOCK) {
        const float v = SRC_BF16 ? bf16_to_f32(((const bf16_t*)g)[i]) : ((const float*)g)[i];
        acc[i] = overwrite ? scale * v : acc[i] + scale * v;
    }
}

// One launch for a whole dictionary of gradients (a block's ~20 parameters were 20 launches of 5 us): the table travels BY VALUE in the kernel arguments (no device
// copy); workgroup w finds its item by a search over the cumulative block counts and covers OPT_BLOCK * 8 elements of it.
struct AccumTable {
    const void* src[TG_ACCUM_MAX];
    float* dst[TG_ACCUM_MAX];
    long n[TG_ACCUM_MAX];
    unsigned first_block[TG_ACCUM_MAX + 1];
    unsigned bf16_mask[(TG_ACCUM_MAX + 31) / 32];
    int count;
};
__global__ __launch_bounds__(OPT_BLOCK) void grad_accum_multi_kernel(AccumTable t, float scale) {
    int lo = 0, hi = t.count;                                  // first_block[lo] <= blockIdx.x < first_block[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (blockIdx.x >= t.first_block[mid]) lo = mid; else hi = mid;
    }
    const long base = (long)(blockIdx.x - t.first_block[lo]) * (OPT_BLOCK * 8);
    const bool bf = (t.bf16_mask[lo >> 5] >> (lo & 31)) & 1u;
    float* acc = t.dst[lo];
    const long n = t.n[lo];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long i = base + (long)k * OPT_BLOCK + threadIdx.x;
        if (i < n) acc[i] += scale * (bf ? bf16_to_f32(((const bf16_t*)t.src[lo])[i]) : ((const float*)t.src[lo])[i]);
    }
}

__global__ __launch_bounds__(OPT_BLOCK) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
    float a = 0.f;
    for (long i = (long)blockIdx.x * OPT_BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * OPT_BLOCK) { const float v = g[i]; a += v * v; }
    a = wave_sum(a);
    __shared__ float red[OPT_BLOCK / 64];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// coef[0] = total norm, coef[1] = min(1, max_norm / (norm + 1e-6))  (torch.nn.utils.clip_grad_norm_); fixed summation order
__global__ void clip_coef_kernel(const float* __restrict__ partial, int n_partial, float max_norm, float* __restrict__ coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < n_partial; ++i) s += (double)partial[i];
    const float norm = (float)sqrt(s);
    coef[0] = norm;
    coef[1] = max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
}

// torch.optim.AdamW (decoupled weight decay, bias correction, eps outside the corrected sqrt) on fp32 moments; parameters are bf16 and are
// read / written once (fp32 arithmetic in between).  clip: optional device pointer to the coefficient of clip_coef_kernel (coef + 1).
// The gradient is read and (zero_grad) zeroed through ONE pointer: two __restrict__ names for the same buffer would be undefined behaviour.
__global__ __launch_bounds__(OPT_BLOCK) void adamw_kernel(bf16_t* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                                                          float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                          const float* __restrict__ clip, int zero_grad) {
    const float cs = clip ? *clip : 1.f;
    for (long i = (long)blockIdx.x * OPT_BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * OPT_BLOCK) {
        const float gi = g[i] * cs;
        float pi = bf16_to_f32(p[i]);
        pi *= 1.f - lr * wd;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = f32_to_bf16(pi);
        if (zero_grad) g[i] = 0.f;
    }
}

inline unsigned opt_blocks(long n) { const long b = (n + OPT_BLOCK - 1) / OPT_BLOCK; return (unsigned)(b < OPT_MAX_BLOCKS ? (b > 0 ? b : 1) : OPT_MAX_BLOCKS); }
}  // namespace

extern "C" int tg_grad_accumulate(const void* grad, int grad_is_bf16, float* acc, long n, float scale, int overwrite, hipStream_t stream) {
    TG_REQUIRE(grad && acc, TG_ERR_ARG, "tg_grad_accumulate: null pointer");
    TG_REQUIRE(n > 0, TG_ERR_SHAPE, "tg_grad_accumulate: n must be positive");
    if (grad_is_bf16) hipLaunchKernelGGL(grad_accum_kernel<true>, dim3(opt_blocks(n)), dim3(OPT_BLOCK), 0, stream, grad, acc, n, scale, overwrite);
    else hipLaunchKernelGGL(grad_accum_kernel<false>, dim3(opt_blocks(n)), dim3(OPT_BLOCK), 0, stream, grad, acc, n, scale, overwrite);
    TG_LAUNCH_CHECK("tg_grad_accumulate");
    return TG_OK;
}

extern "C" int tg_grad_accumulate_multi(const tg_accum_item* items, int count, float scale, hipStream_t stream) {
    TG_REQUIRE(items && count > 0, TG_ERR_ARG, "tg_grad_accumulate_multi: no items");
    for (int i0 = 0; i0 < count; i0 += TG_ACCUM_MAX) {
        AccumTable t{};
        t.count = count - i0 < TG_ACCUM_MAX ? count - i0 : TG_ACCUM_MAX;
        unsigned nb = 0;
        for (int i = 0; i < t.count; ++i) {
            const tg_accum_item& it = items[i0 + i];
            TG_REQUIRE(it.grad && it.acc && it.n > 0, TG_ERR_ARG, "tg_grad_accumulate_multi: item %d: null pointer or empty", i0 + i);
            t.src[i] = it.grad; t.dst[i] = it.acc; t.n[i] = it.n;
            if (it.grad_is_bf16) t.bf16_mask[i >> 5] |= 1u << (i & 31);
            t.first_block[i] = nb;
            const long blocks = (it.n + OPT_BLOCK * 8 - 1) / (OPT_BLOCK * 8);
            TG_REQUIRE(blocks < (1L << 30) && nb + blocks < (1L << 31), TG_ERR_SHAPE, "tg_grad_accumulate_multi: too many elements");
            nb += (unsigned)blocks;
        }
        t.first_block[t.count] = nb;
        hipLaunchKernelGGL(grad_accum_multi_kernel, dim3(nb), dim3(OPT_BLOCK), 0, stream, t, scale);
    }
    TG_LAUNCH_CHECK("tg_grad_accumulate_multi");
    return TG_OK;
}

extern "C" long tg_grad_norm_ws_floats(void) { return OPT_MAX_BLOCKS; }

extern "C" int tg_grad_clip_coef(const float* grad, long n, float max_norm, float* ws, float* coef, hipStream_t stream) {
    TG_REQUIRE(grad && ws && coef, TG_ERR_ARG, "tg_grad_clip_coef: null pointer");
    TG_REQUIRE(n > 0, TG_ERR_SHAPE, "tg_grad_clip_coef: n must be positive");
    const unsigned nb = opt_blocks(n);
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(OPT_BLOCK), 0, stream, grad, n, ws);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, stream, (const float*)ws, (int)nb, max_norm, coef);
    TG_LAUNCH_CHECK("tg_grad_clip_coef");
    return TG_OK;
}

extern "C" int tg_adamw_step(void* param, float* grad, float* exp_avg, float* exp_avg_sq, long n, int step, float lr, float beta1, float beta2, float eps,
                             float weight_decay, const float* clip_coef, int zero_grad, hipStream_t stream) {
    TG_REQUIRE(param && grad && exp_avg && exp_avg_sq, TG_ERR_ARG, "tg_adamw_step: null pointer");
    TG_REQUIRE(n > 0 && step >= 1, TG_ERR_SHAPE, "tg_adamw_step: n and step must be positive");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(opt_blocks(n)), dim3(OPT_BLOCK), 0, stream, (bf16_t*)param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2s, clip_coef, zero_grad);
    TG_LAUNCH_CHECK("tg_adamw_step");
    return TG_OK;
}
