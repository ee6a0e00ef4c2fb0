// Small elementwise / gather kernels around the DiT: timestep sinusoid, 2x2 patch gather / scatter,
// and the fused CFG + per-frame DPM-solver++ update of one FIFO window.
#include "common.h"
#include "tokensgen_hip.h"

namespace {

// embeddings.py:28-79 with flip_sin_to_cos=True, downscale_freq_shift=0: emb = [cos(t w) | sin(t w)]
__global__ void timestep_sinusoid_kernel(const int64_t* __restrict__ t, int n, int dim, bf16_t* __restrict__ emb) {
    const int half = dim >> 1;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n * half) return;
    const int i = (int)(idx / half), k = (int)(idx % half);
    const float ex = (-9.210340371976184f * (float)k) / (float)half;   // -ln(1e4) * k / half, fp32 like torch
    const float a = (float)t[i] * expf(ex);
    emb[(long)i * dim + k] = f32_to_bf16(cosf(a));
    emb[(long)i * dim + half + k] = f32_to_bf16(sinf(a));
}

// out[(bf, y, x)][ldo][c*p*p + dy*p + dx] = lat[bf][c][p*y+dy][p*x+dx]   (p = patch size: 2 for To2V, 1 for the T2To model)
__global__ void patchify_kernel(const bf16_t* __restrict__ lat, bf16_t* __restrict__ out, long ldo, int bf, int C, int H, int W, int p) {
    const int hp = H / p, wp = W / p, pp = p * p, K = C * pp;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)bf * hp * wp * K;
    if (idx >= total) return;
    const int kk = (int)(idx % K);
    long tok = idx / K;
    const long row = tok;
    const int x = (int)(tok % wp); tok /= wp;
    const int y = (int)(tok % hp);
    const int f = (int)(tok / hp);
    const int c = kk / pp, dy = (kk / p) % p, dx = kk % p;
    out[row * ldo + kk] = lat[(((long)f * C + c) * H + p * y + dy) * W + p * x + dx];
}

// lat[bf][c][p*y+dy][p*x+dx] = x[(bf, y, x)][ldx][c*p*p + dy*p + dx]
__global__ void unpatchify_kernel(const bf16_t* __restrict__ x, long ldx, bf16_t* __restrict__ lat, int bf, int C, int H, int W, int p) {
    const int hp = H / p, wp = W / p;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)bf * C * H * W;
    if (idx >= total) return;
    const int xx = (int)(idx % W);
    long r = idx / W;
    const int yy = (int)(r % H); r /= H;
    const int c = (int)(r % C);
    const int f = (int)(r / C);
    const long tok = ((long)f * hp + (yy / p)) * wp + (xx / p);
    lat[idx] = x[tok * ldx + c * p * p + (yy % p) * p + (xx % p)];
}

// 3-D rotary tables on the device (embeddings.py:641-707 / 774-828): token (t,h,w), channel segments [t | h | w] of widths
// 2*nt, 2*nh, 2*nw; pair i of a segment uses angle = pos * inv[i] (one fp32 product, as torch.outer on the host), cos/sin repeated
// for the two channels of the pair.  inv_* come from the host (same torch expression as the reference), so only cosf/sinf differ
// from the host tables, by <= 2 ulp.
__global__ void rope_table_3d_kernel(const float* __restrict__ pos_t, int T, const float* __restrict__ pos_h, int H,
                                     const float* __restrict__ pos_w, int W, const float* __restrict__ inv_t, int nt,
                                     const float* __restrict__ inv_h, int nh, const float* __restrict__ inv_w, int nw,
                                     float* __restrict__ cos_out, float* __restrict__ sin_out) {
    const int pairs = nt + nh + nw;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)T * H * W * pairs;
    if (idx >= total) return;
    const int pi = (int)(idx % pairs);
    long tok = idx / pairs;
    const int w = (int)(tok % W); const long q = tok / W;
    const int h = (int)(q % H), t = (int)(q / H);
    float ang;
    if (pi < nt) ang = pos_t[t] * inv_t[pi];
    else if (pi < nt + nh) ang = pos_h[h] * inv_h[pi - nt];
    else ang = pos_w[w] * inv_w[pi - nt - nh];
    const float c = cosf(ang), sn = sinf(ang);
    const long o = tok * (2L * pairs) + 2 * pi;
    *(float2*)(cos_out + o) = float2{c, c};
    *(float2*)(sin_out + o) = float2{sn, sn};
}

// Fused guidance + SDE-DPM-solver++ step for `frames` frames with their own coefficient rows {sa, sb, m1, m2, m3, m4, mn, has_old}.
//   BR = 2: model_out = (uncond, cond)                      v = u + g (c - u)                               (cogvideo_sampling_mp_fifo.py:531-533)
//   BR = 3: model_out = (uncond_txt, uncond_img, txt_img)   v = c + (g - 1)(c - ut) + (gi - 1)(c - ui)      (:528-530, use_separate_guidance)
//   g / gi: per frame from `gpf` [frames][2] when given (use_dynamic_cfg builds an fp32 tensor over the window's timesteps, :519-527), else the
//   two scalars.
//   F32MATH = false — the FIFO worker with static guidance: guidance is Python-float arithmetic on bf16 tensors, every op rounds to bf16;
//                     the solver runs on that bf16 tensor (fp32 inside the kernel, one rounding per output).
//   F32MATH = true  — the model output reaches the solver as an fp32 tensor: the pipelines call `noise_pred.float()` first
//                     (pipeline_cogvideox_mp_fifo.py:1236-1276, pipeline_cogvideox_t2to.py:845-870), and in the worker's dynamic-cfg branch the
//                     fp32 guidance TENSOR promotes the combination.  The sample, the noise and a bf16 old x0 are still bf16 tensors multiplied
//                     by 0-dim fp64 coefficients, which torch evaluates as bf16(coefficient) x tensor -> bf16 before the fp32 additions.
//   F32STATE: old_x0 / x0_out are fp32 (pipelines keep the solver history in fp32) or bf16 (the FIFO queue holds model-dtype x0).
//   pred: 0 v_prediction, 1 epsilon, 2 sample (scheduling_dpm_cogvideox.py:424-436).
template <int BR, bool F32MATH, bool F32STATE>
__global__ void cfg_dpm_step_kernel(const bf16_t* __restrict__ mo, const bf16_t* __restrict__ x, const void* __restrict__ old_x0_,
                                    const bf16_t* __restrict__ noise, const float* __restrict__ coef, float g_s, float gi_s,
                                    const float* __restrict__ gpf, int pred, bf16_t* __restrict__ x_out, void* __restrict__ x0_out_,
                                    int frames, long fe) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)frames * fe;
    if (idx >= total) return;
    const int f = (int)(idx / fe);
    const long e = idx - (long)f * fe;
    const float* c = coef + f * 8;
    const float g = gpf ? gpf[2 * f] : g_s, gi = gpf ? gpf[2 * f + 1] : gi_s;
    float v;
    if (BR == 1) {                                        // no classifier-free guidance (cogvideo_sampling_mp_fifo.py:497-498, 528: the model output is the prediction)
        v = bf16_to_f32(mo[idx]);
    } else if (BR == 2) {
        const float u = bf16_to_f32(mo[idx]), cd = bf16_to_f32(mo[total + idx]);
        if (F32MATH) v = gpf ? u + g * round_bf16(cd - u) : u + g * (cd - u);      // fp32 guidance tensor x the bf16 difference | all fp32 after .float()
        else v = round_bf16(u + round_bf16(g * round_bf16(cd - u)));
    } else {
        const float ut = bf16_to_f32(mo[idx]), ui = bf16_to_f32(mo[total + idx]), cd = bf16_to_f32(mo[2 * total + idx]);
        if (F32MATH) v = gpf ? cd + (g - 1.f) * round_bf16(cd - ut) + (gi - 1.f) * round_bf16(cd - ui) : cd + (g - 1.f) * (cd - ut) + (gi - 1.f) * (cd - ui);
        else v = round_bf16(round_bf16(cd + round_bf16((g - 1.f) * round_bf16(cd - ut))) + round_bf16((gi - 1.f) * round_bf16(cd - ui)));
    }
    const float xs = bf16_to_f32(x[idx]);
    float x0;
    if (F32MATH) x0 = pred == 0 ? round_bf16(round_bf16(c[0]) * xs) - c[1] * v : pred == 1 ? (xs - c[1] * v) / c[0] : v;
    else x0 = pred == 0 ? c[0] * xs - c[1] * v : pred == 1 ? (xs - c[1] * v) / c[0] : v;
    const bool has_old = c[7] != 0.f;
    float d = x0;
    if (has_old) {
        if (F32STATE) d = c[4] * x0 - c[5] * ((const float*)old_x0_)[idx];
        else if (F32MATH) d = c[4] * x0 - round_bf16(round_bf16(c[5]) * bf16_to_f32(((const bf16_t*)old_x0_)[idx]));
        else d = c[4] * x0 - c[5] * bf16_to_f32(((const bf16_t*)old_x0_)[idx]);
    }
    const float nz = bf16_to_f32(noise[((long)f * 2 + (has_old ? 1 : 0)) * fe + e]);
    if (F32MATH) x_out[idx] = f32_to_bf16(round_bf16(round_bf16(c[2]) * xs) - c[3] * d + round_bf16(round_bf16(c[6]) * nz));
    else x_out[idx] = f32_to_bf16(c[2] * xs - c[3] * d + c[6] * nz);
    if (F32STATE) ((float*)x0_out_)[idx] = x0;
    else ((bf16_t*)x0_out_)[idx] = f32_to_bf16(x0);
}

// T2To tail (pipeline_cogvideox_t2to.py:890-899, pca.py:64-66): de-normalise the 16 sampled coefficients and take them back
// through the PCA basis to the condensed-token width, fp32 like the reference's CPU tail, written in the [b f c h w] order
// the To2V stage consumes:  out[bf][c][s] = pmean[c] + sum_j (lat[bf][j][s] * std[j] + mean[j]) * comp[j][c]
template <int NC>
__global__ void pca_inverse_kernel(const bf16_t* __restrict__ lat, const float* __restrict__ std_, const float* __restrict__ mean_,
                                   const float* __restrict__ comp, const float* __restrict__ pmean, bf16_t* __restrict__ out,
                                   int hw, int cout) {
    // block = one frame x 64 output channels; the frame's NC x hw coefficients are de-normalised once into LDS
    extern __shared__ float sl[];                       // [NC][hw]
    const int f = blockIdx.x, c0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < NC * hw; i += blockDim.x) {
        const int j = i / hw;
        sl[i] = bf16_to_f32(lat[(long)f * NC * hw + i]) * std_[j] + mean_[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * hw; i += blockDim.x) {
        const int cl = i / hw, s = i - cl * hw, c = c0 + cl;
        if (c >= cout) break;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j) acc = fmaf(sl[j * hw + s], comp[(long)j * cout + c], acc);
        out[((long)f * cout + c) * hw + s] = f32_to_bf16(acc + pmean[c]);
    }
}


// Resampler's optional PCA low-rank filter (video_ipadapter/resampler.py:230-237 with pca.py:56-66): per token, in fp32 like the reference
// (`latents.to(self.pca.components_.dtype)`):  y_j = sum_c (x_c - mean_c) comp[j][c]  (j < keep; the reference computes every component and
// zeroes y[:, 16:]),  out_c = mean_c + sum_j y_j comp[j][c].  One workgroup per token row.
__global__ __launch_bounds__(256) void pca_filter_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ comp,
                                                         const float* __restrict__ mean, bf16_t* __restrict__ out, long ldo, int D, int keep) {
    __shared__ float red[4][16];
    __shared__ float sy[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16_t* xr = x + (long)blockIdx.x * ldx;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    for (int c = tid; c < D; c += 256) {
        const float v = bf16_to_f32(xr[c]) - mean[c];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < keep) acc[j] = fmaf(v, comp[(long)j * D + c], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float t = wave_sum(acc[j]);
        if (lane == 0) red[wave][j] = t;
    }
    __syncthreads();
    if (tid < 16) sy[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    __syncthreads();
    bf16_t* orow = out + (long)blockIdx.x * ldo;
    for (int c = tid; c < D; c += 256) {
        float o = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < keep) o = fmaf(sy[j], comp[(long)j * D + c], o);
        orow[c] = f32_to_bf16(o + mean[c]);
    }
}

// Training loss of the To2V step and its gradient w.r.t. the model output (train_cogvideo_to2v.py:1995-2004 with scheduling_dpm_cogvideox.py:521-538):
//   pred = sqrt(acp_t) * noisy - sqrt(1 - acp_t) * out     (get_velocity in the sample dtype: every product / difference is a bf16 tensor)
//   loss_b = mean over the batch item of w_t (pred - target)^2,  w_t = 1 / (1 - acp_t) in fp32;   loss = mean_b loss_b
//   dloss/dout = -sqrt(1 - acp_t) * 2 w_t (pred - target) / (elements per item * batch)
// One thread per element; per-block partial loss sums in a fixed order (deterministic), summed per batch item on the host.
__global__ __launch_bounds__(256) void vpred_loss_grad_kernel(const bf16_t* __restrict__ out, const bf16_t* __restrict__ noisy,
                                                              const bf16_t* __restrict__ target, const float* __restrict__ coef, long E,
                                                              float inv_count, bf16_t* __restrict__ grad, float* __restrict__ partial) {
    __shared__ float red[4];
    const int f = blockIdx.y;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const float sa = round_bf16(coef[3 * f]), sb = round_bf16(coef[3 * f + 1]), w = coef[3 * f + 2];
    float term = 0.f;
    if (e < E) {
        const long i = (long)f * E + e;
        const float pred = round_bf16(round_bf16(sa * bf16_to_f32(noisy[i])) - round_bf16(sb * bf16_to_f32(out[i])));
        const float diff = round_bf16(pred - bf16_to_f32(target[i]));
        term = w * round_bf16(diff * diff);
        grad[i] = f32_to_bf16(-sb * (2.f * w * diff * inv_count));
    }
    term = wave_sum(term);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = term;
    __syncthreads();
    if (threadIdx.x == 0) partial[(long)f * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

extern "C" int tg_timestep_sinusoid(const int64_t* t, int n, int dim, void* emb, hipStream_t stream) {
    TG_REQUIRE(t && emb, TG_ERR_ARG, "tg_timestep_sinusoid: null pointer");
    TG_REQUIRE(n > 0 && dim > 0 && dim % 2 == 0, TG_ERR_SHAPE, "tg_timestep_sinusoid: bad shape n=%d dim=%d", n, dim);
    const long total = (long)n * (dim / 2);
    hipLaunchKernelGGL(timestep_sinusoid_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, t, n, dim, (bf16_t*)emb);
    TG_LAUNCH_CHECK("tg_timestep_sinusoid");
    return TG_OK;
}

extern "C" int tg_patchify(const void* lat, void* out, long ldo, int bf, int C, int H, int W, int p, hipStream_t stream) {
    TG_REQUIRE(lat && out, TG_ERR_ARG, "tg_patchify: null pointer");
    TG_REQUIRE(bf > 0 && C > 0 && H > 0 && W > 0 && (p == 1 || p == 2) && H % p == 0 && W % p == 0 && ldo >= (long)C * p * p, TG_ERR_SHAPE,
               "tg_patchify: bad shape (p=%d)", p);
    const long total = (long)bf * (H / p) * (W / p) * C * p * p;
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)lat, (bf16_t*)out, ldo, bf, C, H, W, p);
    TG_LAUNCH_CHECK("tg_patchify");
    return TG_OK;
}

extern "C" int tg_unpatchify(const void* x, long ldx, void* lat, int bf, int C, int H, int W, int p, hipStream_t stream) {
    TG_REQUIRE(x && lat, TG_ERR_ARG, "tg_unpatchify: null pointer");
    TG_REQUIRE(bf > 0 && C > 0 && H > 0 && W > 0 && (p == 1 || p == 2) && H % p == 0 && W % p == 0 && ldx >= (long)C * p * p, TG_ERR_SHAPE,
               "tg_unpatchify: bad shape (p=%d)", p);
    const long total = (long)bf * C * H * W;
    hipLaunchKernelGGL(unpatchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)x, ldx, (bf16_t*)lat, bf, C, H, W, p);
    TG_LAUNCH_CHECK("tg_unpatchify");
    return TG_OK;
}

static int cfg_dpm_launch(const char* who, const void* model_out, int branches, const void* x, const void* old_x0, const void* noise, const float* coef,
                          float guidance, float guidance_img, const float* guidance_per_frame, int f32_math, int f32_state, int prediction_type,
                          void* x_out, void* x0_out, int frames, long frame_elems, hipStream_t stream) {
    TG_REQUIRE(model_out && x && old_x0 && noise && coef && x_out && x0_out, TG_ERR_ARG, "%s: null pointer", who);
    TG_REQUIRE(frames > 0 && frame_elems > 0, TG_ERR_SHAPE, "%s: bad shape", who);
    TG_REQUIRE(branches >= 1 && branches <= 3, TG_ERR_ARG, "%s: branches must be 1 (no guidance), 2 (uncond, cond) or 3 (uncond_txt, uncond_img, txt_img)", who);
    TG_REQUIRE(prediction_type >= 0 && prediction_type <= 2, TG_ERR_ARG, "%s: prediction_type %d", who, prediction_type);
    TG_REQUIRE(f32_math || !f32_state, TG_ERR_ARG, "%s: an fp32 solver state implies fp32 model-output arithmetic", who);
    const long total = (long)frames * frame_elems;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define TG_DPM(BR_, FM_, FS_)                                                                                                                  \
    hipLaunchKernelGGL((cfg_dpm_step_kernel<BR_, FM_, FS_>), grid, block, 0, stream, (const bf16_t*)model_out, (const bf16_t*)x, old_x0,        \
                       (const bf16_t*)noise, coef, guidance, guidance_img, guidance_per_frame, prediction_type, (bf16_t*)x_out, x0_out, frames, \
                       frame_elems)
    if (branches == 1) {
        if (f32_state) TG_DPM(1, true, true); else if (f32_math) TG_DPM(1, true, false); else TG_DPM(1, false, false);
    } else if (branches == 2) {
        if (f32_state) TG_DPM(2, true, true); else if (f32_math) TG_DPM(2, true, false); else TG_DPM(2, false, false);
    } else {
        if (f32_state) TG_DPM(3, true, true); else if (f32_math) TG_DPM(3, true, false); else TG_DPM(3, false, false);
    }
#undef TG_DPM
    TG_LAUNCH_CHECK(who);
    return TG_OK;
}

extern "C" int tg_cfg_dpm_step(const void* model_out, const void* x, const void* old_x0, const void* noise,
                               const float* coef, float guidance, void* x_out, void* x0_out, int frames,
                               long frame_elems, hipStream_t stream) {
    return cfg_dpm_launch("tg_cfg_dpm_step", model_out, 2, x, old_x0, noise, coef, guidance, 0.f, nullptr, 0, 0, 0, x_out, x0_out, frames, frame_elems, stream);
}

extern "C" int tg_cfg_dpm_step_f32(const void* model_out, const void* x, const float* old_x0, const void* noise,
                                   const float* coef, float guidance, void* x_out, float* x0_out, int frames,
                                   long frame_elems, hipStream_t stream) {
    return cfg_dpm_launch("tg_cfg_dpm_step_f32", model_out, 2, x, old_x0, noise, coef, guidance, 0.f, nullptr, 1, 1, 0, x_out, x0_out, frames, frame_elems, stream);
}

extern "C" int tg_cfg_dpm_step_ex(const void* model_out, int branches, const void* x, const void* old_x0, const void* noise, const float* coef,
                                  float guidance, float guidance_img, const float* guidance_per_frame, int f32_math, int f32_state,
                                  int prediction_type, void* x_out, void* x0_out, int frames, long frame_elems, hipStream_t stream) {
    return cfg_dpm_launch("tg_cfg_dpm_step_ex", model_out, branches, x, old_x0, noise, coef, guidance, guidance_img, guidance_per_frame, f32_math,
                          f32_state, prediction_type, x_out, x0_out, frames, frame_elems, stream);
}

extern "C" int tg_pca_inverse(const void* lat, const float* std16, const float* mean16, const float* comp, const float* pmean,
                              void* out, int frames, int ncoef, int hw, int cout, hipStream_t stream) {
    TG_REQUIRE(lat && std16 && mean16 && comp && pmean && out, TG_ERR_ARG, "tg_pca_inverse: null pointer");
    TG_REQUIRE(frames > 0 && hw > 0 && cout > 0, TG_ERR_SHAPE, "tg_pca_inverse: bad shape");
    TG_REQUIRE(ncoef == 16, TG_ERR_SHAPE, "tg_pca_inverse: the T2To model samples 16 PCA coefficients (got %d)", ncoef);
    TG_REQUIRE((long)16 * hw * 4 <= 64 * 1024, TG_ERR_SHAPE, "tg_pca_inverse: hw=%d too large for the LDS stage", hw);
    hipLaunchKernelGGL(pca_inverse_kernel<16>, dim3((unsigned)frames, (unsigned)((cout + 63) / 64)), dim3(256), (size_t)16 * hw * 4, stream,
                       (const bf16_t*)lat, std16, mean16, comp, pmean, (bf16_t*)out, hw, cout);
    TG_LAUNCH_CHECK("tg_pca_inverse");
    return TG_OK;
}

extern "C" int tg_pca_lowrank_filter(const void* x, long ldx, const float* comp, const float* mean, void* out, long ldo, int rows, int D,
                                     int keep, hipStream_t stream) {
    TG_REQUIRE(x && comp && mean && out, TG_ERR_ARG, "tg_pca_lowrank_filter: null pointer");
    TG_REQUIRE(rows > 0 && D > 0 && ldx >= D && ldo >= D, TG_ERR_SHAPE, "tg_pca_lowrank_filter: bad shape rows=%d D=%d", rows, D);
    TG_REQUIRE(keep >= 1 && keep <= 16, TG_ERR_SHAPE, "tg_pca_lowrank_filter: keep=%d (the reference keeps 16 components)", keep);
    hipLaunchKernelGGL(pca_filter_kernel, dim3((unsigned)rows), dim3(256), 0, stream, (const bf16_t*)x, ldx, comp, mean, (bf16_t*)out, ldo, D, keep);
    TG_LAUNCH_CHECK("tg_pca_lowrank_filter");
    return TG_OK;
}

extern "C" long tg_vpred_loss_partial_floats(int frames, long frame_elems) { return (long)frames * ((frame_elems + 255) / 256); }

extern "C" int tg_vpred_loss_grad(const void* model_out, const void* noisy, const void* target, const float* coef, int frames, long frame_elems,
                                  float inv_count, void* grad, float* partial, hipStream_t stream) {
    TG_REQUIRE(model_out && noisy && target && coef && grad && partial, TG_ERR_ARG, "tg_vpred_loss_grad: null pointer");
    TG_REQUIRE(frames > 0 && frames < 65536 && frame_elems > 0, TG_ERR_SHAPE, "tg_vpred_loss_grad: bad shape");
    hipLaunchKernelGGL(vpred_loss_grad_kernel, dim3((unsigned)((frame_elems + 255) / 256), (unsigned)frames), dim3(256), 0, stream,
                       (const bf16_t*)model_out, (const bf16_t*)noisy, (const bf16_t*)target, coef, frame_elems, inv_count, (bf16_t*)grad, partial);
    TG_LAUNCH_CHECK("tg_vpred_loss_grad");
    return TG_OK;
}

extern "C" int tg_rope_table_3d(const float* pos_t, int T, const float* pos_h, int H, const float* pos_w, int W,
                                const float* inv_t, int dim_t, const float* inv_h, int dim_h, const float* inv_w, int dim_w,
                                float* cos_out, float* sin_out, hipStream_t stream) {
    TG_REQUIRE(pos_t && pos_h && pos_w && inv_t && inv_h && inv_w && cos_out && sin_out, TG_ERR_ARG, "tg_rope_table_3d: null pointer");
    TG_REQUIRE(T > 0 && H > 0 && W > 0 && dim_t > 0 && dim_h > 0 && dim_w > 0 && dim_t % 2 == 0 && dim_h % 2 == 0 && dim_w % 2 == 0,
               TG_ERR_SHAPE, "tg_rope_table_3d: bad shape");
    TG_REQUIRE((((uintptr_t)cos_out) & 7) == 0 && (((uintptr_t)sin_out) & 7) == 0, TG_ERR_ALIGN, "tg_rope_table_3d: alignment");
    const long total = (long)T * H * W * ((dim_t + dim_h + dim_w) / 2);
    hipLaunchKernelGGL(rope_table_3d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, pos_t, T, pos_h, H, pos_w, W,
                       inv_t, dim_t / 2, inv_h, dim_h / 2, inv_w, dim_w / 2, cos_out, sin_out);
    TG_LAUNCH_CHECK("tg_rope_table_3d");
    return TG_OK;
}
