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

// out[(bf, y, x)][c*4 + dy*2 + dx] = lat[bf][c][2y+dy][2x+dx]
__global__ void patchify_kernel(const bf16_t* __restrict__ lat, bf16_t* __restrict__ out, int bf, int C, int H, int W) {
    const int hp = H >> 1, wp = W >> 1, K = C * 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)bf * hp * wp * K;
    if (idx >= total) return;
    const int kk = (int)(idx % K);
    long tok = idx / K;
    const int x = (int)(tok % wp); tok /= wp;
    const int y = (int)(tok % hp);
    const int f = (int)(tok / hp);
    const int c = kk >> 2, dy = (kk >> 1) & 1, dx = kk & 1;
    out[idx] = lat[(((long)f * C + c) * H + 2 * y + dy) * W + 2 * x + dx];
}

// lat[bf][c][2y+dy][2x+dx] = x[(bf, y, x)][c*4 + dy*2 + dx]
__global__ void unpatchify_kernel(const bf16_t* __restrict__ x, long ldx, bf16_t* __restrict__ lat, int bf, int C, int H, int W) {
    const int hp = H >> 1, wp = W >> 1;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)bf * C * H * W;
    if (idx >= total) return;
    const int xx = (int)(idx % W);
    long r = idx / W;
    const int yy = (int)(r % H); r /= H;
    const int c = (int)(r % C);
    const int f = (int)(r / C);
    const long tok = ((long)f * hp + (yy >> 1)) * wp + (xx >> 1);
    lat[idx] = x[tok * ldx + c * 4 + (yy & 1) * 2 + (xx & 1)];
}

// coef row: {sa, sb, m1, m2, m3, m4, mn, has_old}
__global__ void cfg_dpm_step_kernel(const bf16_t* __restrict__ mo, const bf16_t* __restrict__ x,
                                    const bf16_t* __restrict__ old_x0, const bf16_t* __restrict__ noise,
                                    const float* __restrict__ coef, float guidance, bf16_t* __restrict__ x_out,
                                    bf16_t* __restrict__ x0_out, int frames, long fe) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)frames * fe;
    if (idx >= total) return;
    const int f = (int)(idx / fe);
    const long e = idx - (long)f * fe;
    const float* c = coef + f * 8;
    const float u = bf16_to_f32(mo[idx]), cd = bf16_to_f32(mo[total + idx]);
    // CFG result is a bf16 tensor in the reference worker (cogvideo_sampling_mp_fifo.py:531-533)
    const float v = round_bf16(u + guidance * (cd - u));
    const float xs = bf16_to_f32(x[idx]);
    const float x0 = c[0] * xs - c[1] * v;
    const bool has_old = c[7] != 0.f;
    float d = x0;
    if (has_old) d = c[4] * x0 - c[5] * bf16_to_f32(old_x0[idx]);
    const float nz = bf16_to_f32(noise[((long)f * 2 + (has_old ? 1 : 0)) * fe + e]);
    x_out[idx] = f32_to_bf16(c[2] * xs - c[3] * d + c[6] * nz);
    x0_out[idx] = f32_to_bf16(x0);
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

extern "C" int tg_patchify(const void* lat, void* out, int bf, int C, int H, int W, hipStream_t stream) {
    TG_REQUIRE(lat && out, TG_ERR_ARG, "tg_patchify: null pointer");
    TG_REQUIRE(bf > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, TG_ERR_SHAPE, "tg_patchify: bad shape");
    const long total = (long)bf * (H / 2) * (W / 2) * C * 4;
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)lat, (bf16_t*)out, bf, C, H, W);
    TG_LAUNCH_CHECK("tg_patchify");
    return TG_OK;
}

extern "C" int tg_unpatchify(const void* x, long ldx, void* lat, int bf, int C, int H, int W, hipStream_t stream) {
    TG_REQUIRE(x && lat, TG_ERR_ARG, "tg_unpatchify: null pointer");
    TG_REQUIRE(bf > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && ldx >= 4L * C, TG_ERR_SHAPE, "tg_unpatchify: bad shape");
    const long total = (long)bf * C * H * W;
    hipLaunchKernelGGL(unpatchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)x, ldx, (bf16_t*)lat, bf, C, H, W);
    TG_LAUNCH_CHECK("tg_unpatchify");
    return TG_OK;
}

extern "C" int tg_cfg_dpm_step(const void* model_out, const void* x, const void* old_x0, const void* noise,
                               const float* coef, float guidance, void* x_out, void* x0_out, int frames,
                               long frame_elems, hipStream_t stream) {
    TG_REQUIRE(model_out && x && old_x0 && noise && coef && x_out && x0_out, TG_ERR_ARG, "tg_cfg_dpm_step: null pointer");
    TG_REQUIRE(frames > 0 && frame_elems > 0, TG_ERR_SHAPE, "tg_cfg_dpm_step: bad shape");
    const long total = (long)frames * frame_elems;
    hipLaunchKernelGGL(cfg_dpm_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)model_out,
                       (const bf16_t*)x, (const bf16_t*)old_x0, (const bf16_t*)noise, coef, guidance, (bf16_t*)x_out, (bf16_t*)x0_out,
                       frames, frame_elems);
    TG_LAUNCH_CHECK("tg_cfg_dpm_step");
    return TG_OK;
}
