// HBM-bound row kernels of the DiT block: AdaLN-Zero modulation, per-head QK LayerNorm + RoPE,
// V transposition.  One pass over the data each, 16-byte vector accesses, fp32 statistics.
#include "common.h"
#include "tokensgen_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------
// y = LN(x) * (1 + scale[g]) + shift[g]      (normalization.py:441-460, 477-488, 70-92)
// One wave per token row; the row lives in registers between the statistics pass and the apply pass.
// ------------------------------------------------------------------------------------------------
template <int CHUNKS>   // 16-byte chunks per lane: dim = CHUNKS * 64 * 8 at most
__global__ __launch_bounds__(256) void adaln_kernel(const bf16_t* __restrict__ x, long ldx, long sxb,
                                                    bf16_t* __restrict__ y, long ldy, long syb,
                                                    const bf16_t* __restrict__ w, const bf16_t* __restrict__ bvec,
                                                    float eps, int tokens, int dim, int batch, int modulate,
                                                    tg_group_table g) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)tokens * batch) return;
    const int b = (int)(row / tokens), t = (int)(row % tokens);
    const bf16_t* xr = x + (long)b * sxb + (long)t * ldx;
    bf16_t* yr = y + (long)b * syb + (long)t * ldy;
    const int nvec = dim >> 3;

    uint4 raw[CHUNKS];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        const int v = c * 64 + lane;
        raw[c] = (v < nvec) ? *(const uint4*)(xr + v * 8) : uint4{0, 0, 0, 0};
        const uint32_t u[4] = {raw[c].x, raw[c].y, raw[c].z, raw[c].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) s += bf16lo_to_f32(u[i]) + bf16hi_to_f32(u[i]);
    }
    const float mean = wave_sum(s) / dim;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        const int v = c * 64 + lane;
        if (v < nvec) {
            const uint32_t u[4] = {raw[c].x, raw[c].y, raw[c].z, raw[c].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = bf16lo_to_f32(u[i]) - mean, c2 = bf16hi_to_f32(u[i]) - mean;
                q += a * a + c2 * c2;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / dim + eps);

    const bf16_t* shift = nullptr;
    const bf16_t* scale = nullptr;
    if (modulate) {
        const int gi = g.tok_group[t];
        const bf16_t* base = (const bf16_t*)g.mod + (long)b * g.mod_batch_stride + (long)g.row[gi] * g.mod_ld;
        shift = base + g.shift_col[gi];
        scale = base + g.scale_col[gi];
    }
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        const int v = c * 64 + lane;
        if (v >= nvec) continue;
        const uint32_t u[4] = {raw[c].x, raw[c].y, raw[c].z, raw[c].w};
        uint4 wv = w ? *(const uint4*)(w + v * 8) : uint4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        uint4 bv = bvec ? *(const uint4*)(bvec + v * 8) : uint4{0, 0, 0, 0};
        const uint32_t wu[4] = {wv.x, wv.y, wv.z, wv.w}, bu[4] = {bv.x, bv.y, bv.z, bv.w};
        uint32_t su[4] = {0, 0, 0, 0}, hu[4] = {0, 0, 0, 0};
        if (modulate) {
            const uint4 sv = *(const uint4*)(scale + v * 8), hv = *(const uint4*)(shift + v * 8);
            su[0] = sv.x; su[1] = sv.y; su[2] = sv.z; su[3] = sv.w;
            hu[0] = hv.x; hu[1] = hv.y; hu[2] = hv.z; hu[3] = hv.w;
        }
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // LayerNorm output is a bf16 tensor in the reference before it is modulated
            float lo = round_bf16((bf16lo_to_f32(u[i]) - mean) * rstd * bf16lo_to_f32(wu[i]) + bf16lo_to_f32(bu[i]));
            float hi = round_bf16((bf16hi_to_f32(u[i]) - mean) * rstd * bf16hi_to_f32(wu[i]) + bf16hi_to_f32(bu[i]));
            if (modulate) {
                lo = lo * (1.f + bf16lo_to_f32(su[i])) + bf16lo_to_f32(hu[i]);
                hi = hi * (1.f + bf16hi_to_f32(su[i])) + bf16hi_to_f32(hu[i]);
            }
            o[i] = pack_bf16x2(lo, hi);
        }
        *(uint4*)(yr + v * 8) = uint4{o[0], o[1], o[2], o[3]};
    }
}

// ------------------------------------------------------------------------------------------------
// per-head LayerNorm(64) + RoPE, in place  (attention_processor.py:2031-2056, embeddings.py:866-885)
// 8 lanes per (token, head) row, 8 elements (16 B) per lane.
// ------------------------------------------------------------------------------------------------
// one (token, head) slice of 8 elements per lane: LayerNorm over the 8-lane group, affine, bf16 rounding, RoPE, scale, store
// returns the sum of squares of the 8 values as STORED (bf16-rounded, scaled): an 8-lane partial of the row's squared norm
__device__ __forceinline__ float qk_norm_rope_slice(const bf16_t* p, bf16_t* dstp, const bf16_t* __restrict__ w, const bf16_t* __restrict__ bvec, int part,
                                                    float eps, bool rope, const float (&c)[8], const float (&sv)[8], float out_scale, bool live) {
    const uint4 raw = *(const uint4*)p;
    const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w};
    float v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = bf16lo_to_f32(u[i]);
        v[2 * i + 1] = bf16hi_to_f32(u[i]);
        s += v[2 * i] + v[2 * i + 1];
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    const float mean = s * (1.f / 64.f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float d = v[i] - mean; q += d * d; }
    q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
    const float rstd = rsqrtf(q * (1.f / 64.f) + eps);
    const uint4 wv = *(const uint4*)(w + part * 8), bv = *(const uint4*)(bvec + part * 8);
    const uint32_t wu[4] = {wv.x, wv.y, wv.z, wv.w}, bu[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // LN output is bf16 in the reference before RoPE
        v[2 * i] = round_bf16((v[2 * i] - mean) * rstd * bf16lo_to_f32(wu[i]) + bf16lo_to_f32(bu[i]));
        v[2 * i + 1] = round_bf16((v[2 * i + 1] - mean) * rstd * bf16hi_to_f32(wu[i]) + bf16hi_to_f32(bu[i]));
    }
    if (rope) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // out = x*cos + rot(x)*sin, rot(x) = (-x1, x0) per pair
            const float a = v[2 * i], bb = v[2 * i + 1];
            v[2 * i] = a * c[2 * i] - bb * sv[2 * i];
            v[2 * i + 1] = bb * c[2 * i + 1] + a * sv[2 * i + 1];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= out_scale;    // 1, or softmax_scale*log2(e) folded into K before its single bf16 rounding
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
    o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
    if (live) *(uint4*)dstp = o;
    const uint32_t ou[4] = {o.x, o.y, o.z, o.w};
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = bf16lo_to_f32(ou[i]), bq = bf16hi_to_f32(ou[i]);
        ss += a * a + bq * bq;
    }
    return ss;
}

// x2 != nullptr: the k columns of the same rows get the same treatment in the same launch (their own affine and scale): the
// rotary table slice (64 B per lane, 4x the size of the data slice) is fetched once for q and k
// STATS: additionally max_t ||k_t||^2 per (batch, head) of the rows as stored (the constant-shift softmax of tg_attention_fwd_multi needs
// an upper bound on the key norms, tg_attn_segment.k_norm2_max).  Deterministic two-stage reduction, no global atomics: the grid is
// (row blocks, batch), a workgroup folds its <= 32 (token, head group) rows into an LDS table of per-head maxima (ds_max_u32 on the
// float bits — non-negative floats order like unsigned integers) and writes one row of `partial[batch][row block][heads]`;
// qk_kmax_finalize_kernel takes the column maxima.
template <int HPG, bool STATS = false>   // heads per 8-lane group: the token's table slice stays in registers while the group walks HPG heads
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(bf16_t* x, bf16_t* x2, bf16_t* y, bf16_t* y2, long yld, long ysb, long ld, long sb, int tokens,
                                                           int heads, int batch, const bf16_t* __restrict__ w,
                                                           const bf16_t* __restrict__ bvec, const bf16_t* __restrict__ w2,
                                                           const bf16_t* __restrict__ bvec2, float eps,
                                                           int start0, int len0, const float* __restrict__ cos0,
                                                           const float* __restrict__ sin0, int start1, int len1,
                                                           const float* __restrict__ cos1,
                                                           const float* __restrict__ sin1, float out_scale, float out_scale2,
                                                           float* __restrict__ partial) {
    __shared__ unsigned smax[STATS ? 128 : 1];
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int part = (int)(gid & 7);
    const int hgroups = heads / HPG;
    // STATS: blockIdx.y is the batch item and rows are counted inside it; otherwise one flat row index over (b, t, head group)
    const long rowid = gid >> 3;
    const long total = STATS ? (long)tokens * hgroups : (long)batch * tokens * hgroups;
    const bool live = rowid < total;
    const long rid = live ? rowid : total - 1;
    const int h0 = (int)(rid % hgroups) * HPG;
    const long bt = rid / hgroups;
    const int t = (int)(bt % tokens), b = STATS ? (int)blockIdx.y : (int)(bt / tokens);
    if (STATS) {
        if (threadIdx.x < 128) smax[threadIdx.x] = 0u;
        __syncthreads();
    }
    const float* cs = nullptr;
    const float* sn = nullptr;
    if (t >= start0 && t < start0 + len0) {
        cs = cos0 + (long)(t - start0) * 64 + part * 8;
        sn = sin0 + (long)(t - start0) * 64 + part * 8;
    } else if (t >= start1 && t < start1 + len1) {
        cs = cos1 + (long)(t - start1) * 64 + part * 8;
        sn = sin1 + (long)(t - start1) * 64 + part * 8;
    }
    float c[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (cs) {
        const float4 c0 = *(const float4*)cs, c1 = *(const float4*)(cs + 4);
        const float4 s0 = *(const float4*)sn, s1 = *(const float4*)(sn + 4);
        c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w; c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w;
        sv[0] = s0.x; sv[1] = s0.y; sv[2] = s0.z; sv[3] = s0.w; sv[4] = s1.x; sv[5] = s1.y; sv[6] = s1.z; sv[7] = s1.w;
    }
#pragma unroll
    for (int hh = 0; hh < HPG; ++hh) {
        const long off = (long)b * sb + (long)t * ld + (h0 + hh) * 64 + part * 8;
        const long offy = (long)b * ysb + (long)t * yld + (h0 + hh) * 64 + part * 8;       // (y == x, yld == ld, ysb == sb: in place)
        qk_norm_rope_slice(x + off, y + offy, w, bvec, part, eps, cs != nullptr, c, sv, out_scale, live);
        if (x2) {
            float ss = qk_norm_rope_slice(x2 + off, y2 + offy, w2, bvec2, part, eps, cs != nullptr, c, sv, out_scale2, live);
            if (STATS) {
                ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64);
                if (part == 0 && live) atomicMax(&smax[h0 + hh], __float_as_uint(ss));
            }
        }
    }
    if (STATS) {
        __syncthreads();
        if ((int)threadIdx.x < heads)
            partial[((long)b * gridDim.x + blockIdx.x) * heads + threadIdx.x] = __uint_as_float(smax[threadIdx.x]);
    }
}

// kmax[b][h] = max over the row blocks of partial[b][.][h]; one workgroup per (head, batch)
__global__ __launch_bounds__(256) void qk_kmax_finalize_kernel(const float* __restrict__ partial, int nblocks, int heads, float* __restrict__ kmax) {
    __shared__ float red[4];
    const int h = blockIdx.x, b = blockIdx.y;
    const float* src = partial + (long)b * nblocks * heads + h;
    float m = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) m = fmaxf(m, src[(long)i * heads]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) kmax[b * heads + h] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ------------------------------------------------------------------------------------------------
// V [keys][64] per head  ->  V^T [64][ldvt] (zero padded): 64x64 tile through LDS
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_v_kernel(const bf16_t* __restrict__ v, long ld, long sb,
                                                          int key_start, int n_keys, int heads,
                                                          bf16_t* __restrict__ vt, long ldvt) {
    __shared__ bf16_t tile[64][64 + 8];   // +8 keeps 16-byte row alignment and breaks the stride-64 conflict
    const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x;
    // load: 64 keys x 8 chunks of 16 B
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 256;
        const int key = idx >> 3, ch = idx & 7;
        const int gk = kt * 64 + key;
        uint4 val{0, 0, 0, 0};
        if (gk < n_keys) val = *(const uint4*)(v + (long)b * sb + (long)(key_start + gk) * ld + h * 64 + ch * 8);
        *(uint4*)&tile[key][ch * 8] = val;
    }
    __syncthreads();
    // store: 64 d-rows x 8 chunks of 8 keys
    bf16_t* out = vt + ((long)(b * heads + h) * 64) * ldvt + (long)kt * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 256;
        const int d = idx >> 3, ch = idx & 7;
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = (uint32_t)tile[ch * 8 + 2 * j][d] | ((uint32_t)tile[ch * 8 + 2 * j + 1][d] << 16);
        *(uint4*)(out + (long)d * ldvt + ch * 8) = uint4{o[0], o[1], o[2], o[3]};
    }
}

}  // namespace

extern "C" int tg_adaln_modulate(const void* x, long ldx, long strideX, void* y, long ldy, long strideY,
                                 const void* ln_weight, const void* ln_bias, float eps, int tokens, int dim,
                                 int batch, int modulate, const tg_group_table* g, hipStream_t stream) {
    TG_REQUIRE(x && y, TG_ERR_ARG, "tg_adaln_modulate: null pointer");
    TG_REQUIRE(tokens > 0 && batch > 0 && dim > 0 && dim % 8 == 0 && dim <= 8192, TG_ERR_SHAPE,
               "tg_adaln_modulate: bad shape tokens=%d dim=%d batch=%d", tokens, dim, batch);
    TG_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && strideX % 8 == 0 && strideY % 8 == 0 && tg_aligned16(x) && tg_aligned16(y),
               TG_ERR_ALIGN, "tg_adaln_modulate: rows must be 16-byte aligned");
    TG_REQUIRE(!modulate || (g && g->mod && g->tok_group), TG_ERR_ARG, "tg_adaln_modulate: modulate needs a group table");
    tg_group_table gt{};
    if (g) gt = *g;
    const long rows = (long)tokens * batch;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const int chunks = (dim / 8 + 63) / 64;
#define TG_ADALN(C)                                                                                              \
    hipLaunchKernelGGL(adaln_kernel<C>, grid, block, 0, stream, (const bf16_t*)x, ldx, strideX, (bf16_t*)y, ldy,  \
                       strideY, (const bf16_t*)ln_weight, (const bf16_t*)ln_bias, eps, tokens, dim, batch, modulate, gt)
    if (chunks <= 1) TG_ADALN(1);
    else if (chunks <= 2) TG_ADALN(2);
    else if (chunks <= 4) TG_ADALN(4);
    else if (chunks <= 6) TG_ADALN(6);
    else if (chunks <= 8) TG_ADALN(8);
    else TG_ADALN(16);
#undef TG_ADALN
    TG_LAUNCH_CHECK("tg_adaln_modulate");
    return TG_OK;
}

static int qk_norm_rope_launch(void* xq, void* xk, long ld, long strideB, int tokens, int heads, int batch, const void* wq, const void* bq,
                               const void* wk, const void* bk, float eps, int start0, int len0, const float* cos0, const float* sin0,
                               int start1, int len1, const float* cos1, const float* sin1, float q_scale, float k_scale, hipStream_t stream,
                               float* k_norm2_max = nullptr, float* ws = nullptr, void* yq = nullptr, void* yk = nullptr, long yld = 0, long ysb = 0) {
    if (!yq) { yq = xq; yk = xk; yld = ld; ysb = strideB; }             // in place unless a destination is given
    TG_REQUIRE(yld % 8 == 0 && ysb % 8 == 0 && tg_aligned16(yq) && (!xk || (yk && tg_aligned16(yk))), TG_ERR_ALIGN, "tg_qk_layernorm_rope: destination rows must be 16-byte aligned");
    TG_REQUIRE(xq && wq && bq && (!xk || (wk && bk)), TG_ERR_ARG, "tg_qk_layernorm_rope: null pointer");
    TG_REQUIRE(tokens > 0 && heads > 0 && batch > 0, TG_ERR_SHAPE, "tg_qk_layernorm_rope: bad shape");
    TG_REQUIRE(ld % 8 == 0 && strideB % 8 == 0 && tg_aligned16(xq) && (!xk || tg_aligned16(xk)), TG_ERR_ALIGN,
               "tg_qk_layernorm_rope: rows must be 16-byte aligned");
    TG_REQUIRE((len0 == 0 || (cos0 && sin0 && tg_aligned16(cos0) && tg_aligned16(sin0))) &&
               (len1 == 0 || (cos1 && sin1 && tg_aligned16(cos1) && tg_aligned16(sin1))), TG_ERR_ARG,
               "tg_qk_layernorm_rope: rope tables missing or unaligned");
    TG_REQUIRE(len0 >= 0 && len1 >= 0 && start0 >= 0 && start1 >= 0 && start0 + len0 <= tokens && start1 + len1 <= tokens,
               TG_ERR_SHAPE, "tg_qk_layernorm_rope: rope segment outside the token range");
    const int hpg = heads % 4 == 0 ? 4 : heads % 2 == 0 ? 2 : 1;
    if (k_norm2_max) {
        TG_REQUIRE(xk && ws, TG_ERR_ARG, "tg_qk_layernorm_rope_pair_kmax: needs the k columns and a workspace");
        TG_REQUIRE(heads <= 128, TG_ERR_SHAPE, "tg_qk_layernorm_rope_pair_kmax: heads=%d > 128", heads);
        const long rows = (long)tokens * (heads / hpg);            // per batch item
        const unsigned nblk = (unsigned)((rows * 8 + 255) / 256);
#define TG_QK_LAUNCH_S(H_)                                                                                                                \
    hipLaunchKernelGGL((qk_norm_rope_kernel<H_, true>), dim3(nblk, (unsigned)batch), dim3(256), 0, stream, (bf16_t*)xq, (bf16_t*)xk, (bf16_t*)yq, (bf16_t*)yk, yld, ysb, ld,    \
                       strideB, tokens, heads, batch, (const bf16_t*)wq, (const bf16_t*)bq, (const bf16_t*)wk, (const bf16_t*)bk, eps,     \
                       start0, len0, cos0, sin0, start1, len1, cos1, sin1, q_scale, k_scale, ws)
        if (hpg == 4) TG_QK_LAUNCH_S(4); else if (hpg == 2) TG_QK_LAUNCH_S(2); else TG_QK_LAUNCH_S(1);
#undef TG_QK_LAUNCH_S
        TG_LAUNCH_CHECK("tg_qk_layernorm_rope_pair_kmax");
        hipLaunchKernelGGL(qk_kmax_finalize_kernel, dim3((unsigned)heads, (unsigned)batch), dim3(256), 0, stream, (const float*)ws, (int)nblk, heads,
                           k_norm2_max);
        TG_LAUNCH_CHECK("tg_qk_layernorm_rope_pair_kmax (finalize)");
        return TG_OK;
    }
    const long threads = (long)batch * tokens * (heads / hpg) * 8;
#define TG_QK_LAUNCH(H_)                                                                                                                 \
    hipLaunchKernelGGL(qk_norm_rope_kernel<H_>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, (bf16_t*)xq, (bf16_t*)xk, (bf16_t*)yq, (bf16_t*)yk, yld, ysb, ld, \
                       strideB, tokens, heads, batch, (const bf16_t*)wq, (const bf16_t*)bq, (const bf16_t*)wk, (const bf16_t*)bk, eps,     \
                       start0, len0, cos0, sin0, start1, len1, cos1, sin1, q_scale, k_scale, (float*)nullptr)
    if (hpg == 4) TG_QK_LAUNCH(4); else if (hpg == 2) TG_QK_LAUNCH(2); else TG_QK_LAUNCH(1);
#undef TG_QK_LAUNCH
    TG_LAUNCH_CHECK("tg_qk_layernorm_rope");
    return TG_OK;
}

extern "C" int tg_qk_layernorm_rope(void* x, long ld, long strideB, int tokens, int heads, int batch,
                                    const void* ln_weight, const void* ln_bias, float eps,
                                    int start0, int len0, const float* cos0, const float* sin0,
                                    int start1, int len1, const float* cos1, const float* sin1, float out_scale,
                                    hipStream_t stream) {
    return qk_norm_rope_launch(x, nullptr, ld, strideB, tokens, heads, batch, ln_weight, ln_bias, nullptr, nullptr, eps, start0, len0, cos0, sin0,
                               start1, len1, cos1, sin1, out_scale, 1.f, stream);
}

extern "C" int tg_qk_layernorm_rope_pair(void* xq, void* xk, long ld, long strideB, int tokens, int heads, int batch,
                                         const void* q_weight, const void* q_bias, const void* k_weight, const void* k_bias, float eps,
                                         int start0, int len0, const float* cos0, const float* sin0,
                                         int start1, int len1, const float* cos1, const float* sin1, float q_scale, float k_scale,
                                         hipStream_t stream) {
    TG_REQUIRE(xk, TG_ERR_ARG, "tg_qk_layernorm_rope_pair: null pointer");
    return qk_norm_rope_launch(xq, xk, ld, strideB, tokens, heads, batch, q_weight, q_bias, k_weight, k_bias, eps, start0, len0, cos0, sin0,
                               start1, len1, cos1, sin1, q_scale, k_scale, stream);
}

extern "C" long tg_qk_kmax_ws_floats(int tokens, int heads, int batch) {
    const int hpg = heads % 4 == 0 ? 4 : heads % 2 == 0 ? 2 : 1;
    const long rows = (long)tokens * (heads / hpg);
    return ((rows * 8 + 255) / 256) * heads * batch;
}

extern "C" int tg_qk_layernorm_rope_pair_kmax(void* xq, void* xk, long ld, long strideB, int tokens, int heads, int batch,
                                              const void* q_weight, const void* q_bias, const void* k_weight, const void* k_bias, float eps,
                                              int start0, int len0, const float* cos0, const float* sin0,
                                              int start1, int len1, const float* cos1, const float* sin1, float q_scale, float k_scale,
                                              float* k_norm2_max, float* ws, hipStream_t stream) {
    TG_REQUIRE(xk && k_norm2_max && ws, TG_ERR_ARG, "tg_qk_layernorm_rope_pair_kmax: null pointer");
    return qk_norm_rope_launch(xq, xk, ld, strideB, tokens, heads, batch, q_weight, q_bias, k_weight, k_bias, eps, start0, len0, cos0, sin0,
                               start1, len1, cos1, sin1, q_scale, k_scale, stream, k_norm2_max, ws);
}

extern "C" int tg_qk_layernorm_rope_pair_out(const void* xq, const void* xk, long ld, long strideB, void* yq, void* yk, long y_ld, long y_strideB, int tokens, int heads,
                                             int batch, const void* q_weight, const void* q_bias, const void* k_weight, const void* k_bias, float eps,
                                             int start0, int len0, const float* cos0, const float* sin0, int start1, int len1, const float* cos1, const float* sin1,
                                             float q_scale, float k_scale, float* k_norm2_max, float* ws, hipStream_t stream) {
    TG_REQUIRE(xq && xk && yq && yk, TG_ERR_ARG, "tg_qk_layernorm_rope_pair_out: null pointer");
    TG_REQUIRE(!k_norm2_max == !ws, TG_ERR_ARG, "tg_qk_layernorm_rope_pair_out: k_norm2_max and ws go together");
    return qk_norm_rope_launch(const_cast<void*>(xq), const_cast<void*>(xk), ld, strideB, tokens, heads, batch, q_weight, q_bias, k_weight, k_bias, eps, start0, len0, cos0,
                               sin0, start1, len1, cos1, sin1, q_scale, k_scale, stream, k_norm2_max, ws, yq, yk, y_ld, y_strideB);
}

extern "C" int tg_transpose_v(const void* v, long ld, long strideB, int key_start, int n_keys, int heads, int batch,
                              void* vt, long ldvt, hipStream_t stream) {
    TG_REQUIRE(v && vt, TG_ERR_ARG, "tg_transpose_v: null pointer");
    TG_REQUIRE(n_keys > 0 && heads > 0 && batch > 0 && key_start >= 0, TG_ERR_SHAPE, "tg_transpose_v: bad shape");
    TG_REQUIRE(ldvt % 64 == 0 && ldvt >= n_keys, TG_ERR_SHAPE, "tg_transpose_v: ldvt must be a multiple of 64 and >= n_keys");
    TG_REQUIRE(ld % 8 == 0 && strideB % 8 == 0 && tg_aligned16(v) && tg_aligned16(vt), TG_ERR_ALIGN, "tg_transpose_v: alignment");
    hipLaunchKernelGGL(transpose_v_kernel, dim3((unsigned)(ldvt / 64), heads, batch), dim3(256), 0, stream, (const bf16_t*)v,
                       ld, strideB, key_start, n_keys, heads, (bf16_t*)vt, ldvt);
    TG_LAUNCH_CHECK("tg_transpose_v");
    return TG_OK;
}
