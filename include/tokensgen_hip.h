/* libtokensgen_hip.so — C ABI of the MI355X (gfx950) kernels behind the TokensGen denoising hot path.
 *
 * The reference (Vicky0522/TokensGen) is pure Python/PyTorch: it has no native FFI.  Each entry point
 * below replaces a *sequence of PyTorch ops* at the reference's own operator seams; the reference
 * file:line each one stands in for is cited per function.  INTEGRATION.md shows the ctypes binding a
 * maintainer adds on the reference side.
 *
 * Conventions:
 *   - plain pointers + sizes only; all tensors are bf16 (raw uint16 bits) unless stated, fp32 accumulate;
 *   - the caller owns every buffer (outputs and workspace are caller-allocated); nothing is retained;
 *   - every call is asynchronous on the given hipStream_t, never synchronises, never selects a device, reads no environment variable;
 *   - nothing process-wide is kept except what is keyed by the CURRENT device id (CU count, per-kernel launch attributes) and the tg_debug_set knobs;
 *   - returns 0 on success; <0 on error (-1 argument, -2 shape, -3 alignment, <=-100 HIP error);
 *     tg_last_error_string() gives the message for the calling thread.  No exceptions cross the ABI.
 */
#ifndef TOKENSGEN_HIP_H
#define TOKENSGEN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the functions declared in this header are exported (tests/test_host_cpu.py holds `nm -D` to it). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef struct ihipStream_t* hipStream_t;   /* same declaration as <hip/hip_runtime_api.h>; opaque to C callers */

#define TG_MAX_GROUPS 16

/* Token -> modulation-row lookup shared by the AdaLN kernels and the gated-residual GEMM epilogue.
 * A DiT window has G = frames + 2 token groups: video frame f (per-frame modulation), text and
 * condensed ("vip") tokens (frame-0 modulation) — normalization.py:441-460,477-488.
 * `mod` is the output of the modulation linears, [batch][mod_rows][mod_ld] bf16; for a token of group
 * g in batch b the shift/scale/gate vectors start at
 *     mod + b*mod_batch_stride + row[g]*mod_ld + {shift,scale,gate}_col[g]. */
typedef struct tg_group_table {
    const void*    mod;               /* bf16 */
    long           mod_ld;
    long           mod_batch_stride;
    const uint8_t* tok_group;         /* [tokens] group id of each token (same for every batch item) */
    int32_t        row[TG_MAX_GROUPS];
    int32_t        shift_col[TG_MAX_GROUPS];
    int32_t        scale_col[TG_MAX_GROUPS];
    int32_t        gate_col[TG_MAX_GROUPS];
} tg_group_table;

enum {
    TG_EPI_BIAS = 0,          /* C = A W^T + bias                                   (nn.Linear)            */
    TG_EPI_BIAS_GELU = 1,     /* C = gelu_tanh(bf16(A W^T + bias))                  (FeedForward net.0)    */
    TG_EPI_BIAS_SILU = 2,     /* C = silu(bf16(A W^T + bias))                       (TimestepEmbedding act)*/
    TG_EPI_BIAS_GATE_RES = 3, /* C = R + gate[group(m)] * (A W^T + bias)            (gated residual)       */
    /* training step, 4-wave kernel shapes only (M >= 1024, N % 256 == 0, K >= 256; TG_ERR_SHAPE otherwise — the caller then runs tg_gemm_bf16 + tg_act): */
    TG_EPI_BIAS_KEEP_GELU = 4,     /* C = bf16(A W^T + bias) AND R (an OUTPUT here, ldr / strideR) = gelu_tanh(C): FF1 keeping its pre-activation, == TG_EPI_BIAS + tg_act mode 2 */
    TG_EPI_BIAS_MUL_GELU_GRAD = 5  /* C = bf16(A W^T + bias) * gelu_tanh'(R), R = the kept pre-activation: FF2's dgrad, == TG_EPI_BIAS + tg_act mode 1 */
};

const char* tg_version(void);
const char* tg_last_error_string(void);

/* C[b,m,:] = epilogue(A[b,m,:] @ W^T + bias), W is an nn.Linear weight [N,K].  Batched by element strides.
 * Replaces: attention_processor.py:2009-2018 (to_q/k/v, vip_to_q/k/v), :2143 (to_out[0]) fused with the
 * gated residual cogvideox_transformer_3d.py:290-293,318-324; diffusers FeedForward (call sites
 * cogvideox_transformer_3d.py:316,322); the modulation linears normalization.py:447,483,79;
 * embeddings.py:516-536 (patch/text/vip projections), :953-965 (TimestepEmbedding); proj_out dit:748.
 * Requires N%128==0, K%64==0 (pad weights once at load time), any M. */
int tg_gemm_bf16(const void* A, long lda, long strideA, const void* W, long ldw, const void* bias,
                 void* C, long ldc, long strideC, int M, int N, int K, int batch, int epilogue,
                 const void* R, long ldr, long strideR, const tg_group_table* gate, hipStream_t stream);

/* Two GEMMs of the same N, K, batch, leading dimensions and epilogue (bias / GELU / SiLU) in ONE launch of the persistent 256x256
 * kernel (M1, M2 >= 1024, N % 256 == 0): the second problem's tiles are appended to the first one's, so together they leave one
 * partial last round of CUs instead of one each.  Built for the To2V block: to_q|k|v over the text+video rows and vip_to_q|k|v over
 * all rows read the same normalised activations and are independent (attention_processor.py:2009-2018). */
int tg_gemm_bf16_pair(const void* A1, long strideA1, const void* W1, const void* bias1, void* C1, long strideC1, int M1,
                      const void* A2, long strideA2, const void* W2, const void* bias2, void* C2, long strideC2, int M2,
                      long lda, long ldw, long ldc, int N, int K, int batch, int epilogue, hipStream_t stream);

/* The QKV projection(s) of an attention block with the V third delivered TRANSPOSED: like tg_gemm_bf16_pair with the bias epilogue
 * (second problem optional: A2 == NULL), except that output columns n >= v_col0 are not written to C but to
 * Vt[b][n - v_col0][m] (row stride vt_ld elements, a multiple of 64 >= M; columns M..vt_ld-1 are written as zeros; batch stride
 * (N - v_col0)*vt_ld) — the [head][64][keys] image tg_attention_fwd reads, i.e. tg_transpose_v fused into the GEMM epilogue (the
 * tile's MFMA operands are exchanged so that a lane holds consecutive tokens; no extra pass over V).  Values are bitwise those of
 * tg_gemm_bf16 + tg_transpose_v.  Needs the 4-wave kernel's shapes: M >= 1024, N % 256 == 0, K % 64 == 0, K >= 256, v_col0 % 256 == 0.
 * Replaces attention_processor.py:2009-2018 (to_q/k/v, vip_to_q/k/v) + the .view(...).transpose(1, 2) of value at :2024-2029. */
int tg_gemm_bf16_qkv(const void* A1, long strideA1, const void* W1, const void* bias1, void* C1, long strideC1, int M1, void* Vt1, long vt_ld1,
                     const void* A2, long strideA2, const void* W2, const void* bias2, void* C2, long strideC2, int M2, void* Vt2, long vt_ld2,
                     long lda, long ldw, long ldc, int N, int K, int batch, int v_col0, hipStream_t stream);

/* y = LayerNorm(x; w, b, eps) * (1 + scale[g]) + shift[g], g = group of the token.  One pass, fp32 stats.
 * modulate == 0: plain affine LayerNorm (norm_final, cogvideox_transformer_3d.py:741).
 * Replaces normalization.py:441-460 (CogVideoXLayerNormZero), :477-488 (VIP), :70-92 (AdaLayerNorm).
 * x, y: [batch][tokens][dim] with row strides ldx/ldy and batch strides; dim % 8 == 0, dim <= 8192. */
int tg_adaln_modulate(const void* x, long ldx, long strideX, void* y, long ldy, long strideY,
                      const void* ln_weight, const void* ln_bias, float eps, int tokens, int dim, int batch,
                      int modulate, const tg_group_table* g, hipStream_t stream);

/* In-place per-head LayerNorm(64, eps, affine) followed by interleaved-pair RoPE on the rows of up to two
 * token ranges.  x points at the q (or k) columns of a fused QKV buffer: element (b, t, h, d) at
 * x[b*strideB + t*ld + h*64 + d].  seg i rotates tokens [start_i, start_i+len_i) with fp32 tables
 * cos_i/sin_i [len_i][64]; other tokens are normalised only (text rows).  The result is multiplied by out_scale before its
 * single bf16 rounding (1.0, or softmax_scale*log2(e) for K so that tg_attention_fwd can run with k_prescaled=1).
 * Replaces attention_processor.py:2031-2056 + embeddings.py:840-885 (apply_rotary_emb). */
int tg_qk_layernorm_rope(void* x, long ld, long strideB, int tokens, int heads, int batch,
                         const void* ln_weight, const void* ln_bias, float eps,
                         int start0, int len0, const float* cos0, const float* sin0,
                         int start1, int len1, const float* cos1, const float* sin1, float out_scale, hipStream_t stream);

/* The same for the q AND the k columns of the same rows in one launch (both in the same fused buffer: same ld / strideB), each with
 * its own LayerNorm affine (norm_q / norm_k) and output scale: the rotary table slices, four times the size of the data they
 * rotate, are fetched once for the pair. */
int tg_qk_layernorm_rope_pair(void* xq, void* xk, long ld, long strideB, int tokens, int heads, int batch,
                              const void* q_weight, const void* q_bias, const void* k_weight, const void* k_bias, float eps,
                              int start0, int len0, const float* cos0, const float* sin0,
                              int start1, int len1, const float* cos1, const float* sin1, float q_scale, float k_scale,
                              hipStream_t stream);

/* tg_qk_layernorm_rope_pair that also emits k_norm2_max[batch][heads] (fp32) = max over the tokens of the squared norm of the K rows AS
 * STORED (after LayerNorm, rotation, k_scale and the bf16 rounding) — the key-side half of the Cauchy-Schwarz range bound the constant-
 * shift softmax of tg_attention_fwd_multi uses (tg_attn_segment.k_norm2_max; the query-side half is taken from the Q rows inside the
 * attention kernel).  Deterministic two-stage maximum (no atomics in global memory): ws >= tg_qk_kmax_ws_floats(tokens, heads, batch)
 * floats of scratch; heads <= 128. */
long tg_qk_kmax_ws_floats(int tokens, int heads, int batch);
int tg_qk_layernorm_rope_pair_kmax(void* xq, void* xk, long ld, long strideB, int tokens, int heads, int batch,
                                   const void* q_weight, const void* q_bias, const void* k_weight, const void* k_bias, float eps,
                                   int start0, int len0, const float* cos0, const float* sin0,
                                   int start1, int len1, const float* cos1, const float* sin1, float q_scale, float k_scale,
                                   float* k_norm2_max, float* ws, hipStream_t stream);

/* tg_qk_layernorm_rope_pair[_kmax] OUT OF PLACE: reads the q | k slices (xq, xk; rows ld, batch stride strideB) and writes the normalised, rotated, scaled rows to
 * yq, yk (rows y_ld, batch stride y_strideB) — the source stays as the projection left it.  The training forward keeps the pre-norm projection for the backward and the
 * post-norm rows for the attention calls: in place that was a copy pass + the norm pass.  k_norm2_max / ws: both NULL, or as for tg_qk_layernorm_rope_pair_kmax. */
int tg_qk_layernorm_rope_pair_out(const void* xq, const void* xk, long ld, long strideB, void* yq, void* yk, long y_ld, long y_strideB, int tokens, int heads,
                                  int batch, const void* q_weight, const void* q_bias, const void* k_weight, const void* k_bias, float eps,
                                  int start0, int len0, const float* cos0, const float* sin0, int start1, int len1, const float* cos1, const float* sin1,
                                  float q_scale, float k_scale, float* k_norm2_max, float* ws, hipStream_t stream);

/* vt[b][h][d][j] = v[b*strideB + (key_start + j)*ld + h*64 + d] for j < n_keys, zero for n_keys <= j < ldvt.
 * Lays V out key-contiguous so the PV MFMA operands are plain 16-byte LDS reads (the "transpose" that
 * F.scaled_dot_product_attention does internally).  ldvt % 64 == 0, ldvt >= n_keys. */
int tg_transpose_v(const void* v, long ld, long strideB, int key_start, int n_keys, int heads, int batch,
                   void* vt, long ldvt, hipStream_t stream);

/* Flash attention forward, head_dim 64, no mask, softmax scale `scale`, up to two independently
 * normalised key/value segments:   out = softmax(q1 k1^T) v1  +  seg2_scale * softmax(q2 k2^T) v2.
 * q1,q2,k1,k2: element (b, t, h, d) at ptr[b*strideB + t*ld + h*64 + d];  vt1,vt2: from tg_transpose_v
 * ([b][h][64][ldvt]);  out: (b, t, h, d) at out[b*out_strideB + t*out_ld + h*64 + d] (merged heads).
 * Segment 2 is skipped when q2 == NULL.  k_prescaled=1: k1/k2 already carry scale*log2(e) (scale is ignored): the kernel
 * then seeds the score accumulator with -max and needs no per-element scale/subtract in its softmax.
 * Replaces the three F.scaled_dot_product_attention calls + `hs + scale*tv_hs` + head merge,
 * attention_processor.py:2066-2069,2117-2135,2141 (and :1937-1941 for the plain processor).  * Alignment: q/k/vt 16 B, their leading dimensions % 8 (vt_ld % 64); out 16 B with out_ld % 8 and out_strideB % 8 (whole 128-byte
 * head rows are stored 16 B per lane).
 */
int tg_attention_fwd(const void* q1, long q1_ld, long q1_strideB,
                     const void* k1, long k1_ld, long k1_strideB, const void* vt1, long vt1_ld, int nk1,
                     const void* q2, long q2_ld, long q2_strideB,
                     const void* k2, long k2_ld, long k2_strideB, const void* vt2, long vt2_ld, int nk2,
                     float seg2_scale, void* out, long out_ld, long out_strideB,
                     int nq, int heads, int batch, float scale, int k_prescaled, hipStream_t stream);

/* The same operator with its arguments in structs, and room for a second problem in the same launch.
 * problems[0]: as tg_attention_fwd (nseg = 1 or 2).  problems[1] (optional, nseg = 1, same heads/batch/scale): its workgroups are
 * appended to problem 0's in one launch when problem 0 is long enough for the 512-row kernel; otherwise the two run one after the
 * other.  Built for the To2V block: the main attention (text+video queries) leaves most of its last round of CUs idle and the
 * vip-query attention (attention_processor.py:2120-2125), of the same length per workgroup, fits in there. */
typedef struct tg_attn_segment {
    const void* q; long q_ld, q_strideB;
    const void* k; long k_ld, k_strideB;
    const void* vt; long vt_ld;
    int nk;
    /* Optional (NULL = none): [batch][heads] fp32, an upper bound on max_j ||k_j||^2 over this segment's keys in the units the kernel
     * exponentiates (k_prescaled launches: the squared norm of the stored K rows) — tg_qk_layernorm_rope_pair_kmax writes it.  When every
     * segment of a k_prescaled launch carries one and a retry workspace is given, the 512-row kernel runs the CONSTANT-SHIFT softmax: each
     * query row subtracts c = max(0, ||q_row|| sqrt(k_norm2_max) - 64) instead of tracking a running row maximum (the max chain / vote /
     * rescale was 15 % of the launch at the CogVideoX-5B shape).  Valid for any weights: s - c <= 64 holds by Cauchy-Schwarz, so nothing can
     * overflow; a row whose scores all sit far below c is caught by the epilogue's verification (row sum >= 2^-64) and its workgroup is
     * recomputed with the running maximum by the retry launch that follows in the same call (attention_processor.py:2066-2069 semantics
     * either way). */
    const float* k_norm2_max;
} tg_attn_segment;
typedef struct tg_attn_problem {
    tg_attn_segment seg[2];
    int nseg;
    float seg2_scale;
    void* out; long out_ld, out_strideB;
    int nq;
    /* Optional (NULL = seg2_scale for every batch item): HOST array of `batch` (<= 16) weights of segment 2, one per batch item — the
     * reference processor's `scale` list when its length equals the batch size (attention_processor.py:2126-2131). */
    const float* seg2_scale_batch;
} tg_attn_problem;
/* Caller-allocated scratch of tg_attention_fwd_multi (both optional; NULL / a NULL struct = feature off):
 *  retry (with every segment's k_norm2_max: enables the constant-shift path): >= tg_attention_retry_ints(nq0, nq1, heads, batch) ints,
 *    zero-initialised ONCE by the caller and reusable by every later launch on the same stream; retry[0] accumulates the number of
 *    workgroups that failed the verification and were recomputed (a host that sees it grow should stop passing it: the Cauchy-Schwarz
 *    shift is a poor range estimate for those weights and the running maximum is the right tool).
 *  split (>= tg_attention_split_floats(...) floats, 16-B aligned, contents irrelevant): lets a launch whose workgroup count leaves at most
 *    half a round of the device's CUs over (the To2V block: 3360 + 96 = 13.5 x 256) split those last workgroups over the KEY axis into
 *    half-length workgroups joined by a small combine launch — 13.5 rounds instead of 14.  Results are independent of the workspace
 *    contents; a launch shape that does not qualify ignores it. */
typedef struct tg_attn_workspace {
    int* retry; long retry_ints;
    float* split; long split_floats;
} tg_attn_workspace;
long tg_attention_retry_ints(int nq0, int nq1, int heads, int batch);
long tg_attention_split_floats(int nq0, int nq1, int heads, int batch);
int tg_attention_fwd_multi(const tg_attn_problem* problems, int nproblems, int heads, int batch, float scale, int k_prescaled,
                           const tg_attn_workspace* ws, hipStream_t stream);

/* emb[i][:] = bf16( [cos(t_i w_k) | sin(t_i w_k)] ), w_k = exp(-ln(1e4) k / (dim/2)), k < dim/2
 * (flip_sin_to_cos=True, freq_shift=0).  Replaces embeddings.py:28-79 + dit:678.  t: int64[n]. */
int tg_timestep_sinusoid(const int64_t* t, int n, int dim, void* emb, hipStream_t stream);

/* 3-D rotary position tables on the device: cos/sin [T*H*W][dim_t+dim_h+dim_w] fp32, tokens ordered (t,h,w), channels [t|h|w],
 * every pair frequency repeated for its two channels (get_3d_rotary_pos_embed_v2 / get_1d_rotary_pos_embed, embeddings.py:641-707,
 * 774-828, which the reference rebuilds on the CPU and uploads for every window, cogvideo_sampling_mp_fifo.py:478-489).
 * pos_t/h/w: fp32 positions per axis (device); inv_t/h/w: the inverse frequencies theta^(-2i/dim) of each axis, dim/2 of them
 * (device, computed once on the host with the reference's expression).  angle = pos * inv in fp32 like the host, so the tables equal the host ones up to
 * the device cosf/sinf (<= 2 ulp). */
int tg_rope_table_3d(const float* pos_t, int T, const float* pos_h, int H, const float* pos_w, int W,
                     const float* inv_t, int dim_t, const float* inv_h, int dim_h, const float* inv_w, int dim_w,
                     float* cos_out, float* sin_out, hipStream_t stream);

/* Gather p x p patches (p = 2: To2V model; p = 1: the T2To model, train_cogvideo_t2to.py:1277):
 *   out[(b f)(h/p w/p)][ldo][c*p*p + dy*p + dx] = lat[b][f][c][p*y+dy][p*x+dx]
 * (the im2col of Conv2d(k=p,s=p), embeddings.py:516-523) so that patch embedding is one GEMM with K = p*p*C;
 * ldo >= p*p*C lets the caller keep zero pad columns up to the GEMM's K granule. */
int tg_patchify(const void* lat, void* out, long ldo, int bf, int C, int H, int W, int p, hipStream_t stream);

/* Inverse for the output head: lat[b][f][c][p*y+dy][p*x+dx] = x[(b f)(y x)][ldx][c*p*p + dy*p + dx]
 * (cogvideox_transformer_3d.py:754-759). */
int tg_unpatchify(const void* x, long ldx, void* lat, int bf, int C, int H, int W, int p, hipStream_t stream);

/* Fused classifier-free guidance + per-frame SDE-DPM-solver++(2M) update for one FIFO window:
 *   v   = uncond + g*(cond - uncond)                                    (cogvideo_sampling_mp_fifo.py:531-533)
 *   x0  = sa_f*x - sb_f*v ;  d = has_old_f ? m3_f*x0 - m4_f*old_x0 : x0
 *   x'  = m1_f*x - m2_f*d + mn_f*noise[f][has_old_f]                    (scheduling_dpm_cogvideox.py:424-463)
 * for every frame f of the window with its own fp32 coefficient row coef[f] = {sa,sb,m1,m2,m3,m4,mn,has_old}.
 * model_out: [2][frames][frame_elems] bf16 (uncond, cond); x, old_x0, x_out, x0_out: [frames][frame_elems];
 * noise: [frames][2][frame_elems] bf16 (the reference's two randn draws per step). */
int tg_cfg_dpm_step(const void* model_out, const void* x, const void* old_x0, const void* noise,
                    const float* coef, float guidance, void* x_out, void* x0_out,
                    int frames, long frame_elems, hipStream_t stream);

/* The same update as the pipelines' own denoising loops run it (pipeline_cogvideox_t2to.py:845-870 and the To2V base stage,
 * pipeline_cogvideox_mp_fifo.py:1236-1276): `noise_pred.float()` precedes guidance, so CFG is NOT rounded to bf16 and the solver
 * history old_x0 / x0_out is fp32; sample and noise are bf16 tensors there, so `sa*x`, `m1*x`, `mn*noise` are bf16 products with
 * the coefficient itself cast to bf16 first (torch's 0-dim promotion rule); the new sample is cast to bf16.  noise: bf16 [frames][2][E]. */
int tg_cfg_dpm_step_f32(const void* model_out, const void* x, const float* old_x0, const void* noise,
                        const float* coef, float guidance, void* x_out, float* x0_out,
                        int frames, long frame_elems, hipStream_t stream);

/* The general form behind the two above — every guidance / prediction branch of the reference's two sampling loops in the same single launch:
 *   branches = 1: model_out [1][frames][E], no classifier-free guidance:           v = model_out                 (cogvideo_sampling_mp_fifo.py:497-498: guidance_scale <= 1)
 *   branches = 2: model_out [2][frames][E] = (uncond, cond):                       v = u + g (c - u)            (cogvideo_sampling_mp_fifo.py:531-533)
 *   branches = 3: model_out [3][frames][E] = (uncond_txt, uncond_img, txt_img):    v = c + (g - 1)(c - ut) + (gi - 1)(c - ui)
 *                 (`use_separate_guidance`, :528-530 / pipeline_cogvideox_mp_fifo.py:1261-1263; gi = guidance_img)
 *   guidance_per_frame (optional, fp32 [frames][2] = {g, gi} on the device): `use_dynamic_cfg` in the FIFO worker, where the cosine schedule
 *                 is evaluated on the window's per-frame timesteps as an fp32 TENSOR (:519-527) — which also promotes the guided
 *                 prediction to fp32, so pass f32_math = 1 with it; NULL: the two scalars for every frame (the pipelines' dynamic cfg is a
 *                 Python float per step, pipeline_cogvideox_mp_fifo.py:1252-1259).
 *   f32_math:     0 = the worker's static path (guidance on bf16 tensors, every op rounded to bf16); 1 = the solver sees an fp32 model output
 *                 (`noise_pred.float()` in the pipelines; the promoted dynamic-cfg result in the worker) while sample / noise / a bf16 old x0
 *                 are bf16 tensors scaled by 0-dim coefficients (bf16 products).
 *   f32_state:    old_x0 / x0_out are fp32 [frames][E] (pipelines; implies f32_math) instead of bf16 (the FIFO queue).
 *   prediction_type: 0 v_prediction (CogVideoX-5b), 1 epsilon, 2 sample (scheduling_dpm_cogvideox.py:424-436).
 * tg_cfg_dpm_step = (2, scalars, 0, 0, 0); tg_cfg_dpm_step_f32 = (2, scalars, 1, 1, 0). */
int tg_cfg_dpm_step_ex(const void* model_out, int branches, const void* x, const void* old_x0, const void* noise, const float* coef,
                       float guidance, float guidance_img, const float* guidance_per_frame, int f32_math, int f32_state,
                       int prediction_type, void* x_out, void* x0_out, int frames, long frame_elems, hipStream_t stream);

/* T2To tail (pipeline_cogvideox_t2to.py:890-899 with pca.py:64-66): the sampled latents hold `ncoef` (= 16) normalised PCA
 * coefficients per token; de-normalise and project back to the condensed-token width in fp32:
 *   out[f][c][s] = bf16( pmean[c] + sum_j (lat[f][j][s] * std[j] + mean[j]) * comp[j][c] )      f < frames, s < hw, c < cout
 * lat bf16 [frames][ncoef][hw]; std/mean fp32 [ncoef]; comp fp32 [ncoef][cout] (the first ncoef PCA components);
 * pmean fp32 [cout]; out bf16 [frames][cout][hw] (= image_embeddings [b f c h w]). */
int tg_pca_inverse(const void* lat, const float* std16, const float* mean16, const float* comp, const float* pmean,
                   void* out, int frames, int ncoef, int hw, int cout, hipStream_t stream);


/* Training forward of one attention call: tg_attention_fwd with a single key segment that ALSO writes, per query row, the log-sum-exp of the
 * scaled scores in the log2 domain (lse fp32 [batch][heads][nq]): what flash-attention keeps for its backward
 * (F.scaled_dot_product_attention under autograd, attention_processor.py:2066-2125).  vt: the transposed V image as for tg_attention_fwd. */
int tg_attention_fwd_lse(const void* q, long q_ld, long q_strideB, const void* k, long k_ld, long k_strideB, const void* vt, long vt_ld, int nk,
                         void* out, long out_ld, long out_strideB, int nq, int heads, int batch, float scale, float* lse, hipStream_t stream);

/* tg_attention_fwd_lse on the inference path's kernels (the training forward of the 17776^2 call): k_prescaled = 1: the K rows already carry
 * scale * log2(e) (tg_qk_layernorm_rope_pair k_scale; `scale` is then ignored and tg_attention_bwd is called on the same K with scale = ln 2, its
 * dk being the gradient of the SCALED rows); k_norm2_max ([batch][heads] fp32 from tg_qk_layernorm_rope_pair_kmax) + retry (int32 workspace of
 * tg_attention_retry_ints(nq, 0, heads, batch) ints, zeroed once): the verified constant-shift softmax with its retry launch, as in
 * tg_attention_fwd_multi; either NULL: running maximum.  lse as for tg_attention_fwd_lse (log2 domain, of the scores as the kernel sees them). */
int tg_attention_fwd_lse_ex(const void* q, long q_ld, long q_strideB, const void* k, long k_ld, long k_strideB, const void* vt, long vt_ld, int nk,
                            void* out, long out_ld, long out_strideB, int nq, int heads, int batch, float scale, int k_prescaled,
                            const float* k_norm2_max, int* retry, long retry_ints, float* lse, hipStream_t stream);

/* Attention BACKWARD (training step): what autograd runs for F.scaled_dot_product_attention in the reference's training
 * loop (attention_processor.py:2066-2125 under train_cogvideo_to2v.py:1995-2010), head_dim 64, no mask, no dropout.
 *   P = softmax(scale q k^T)   dV = P^T dO   dP = dO v^T   dS = P o (dP - rowsum(dO o O))   dQ = scale dS k   dK = scale dS^T q
 * q, o, dout: bf16 [batch][nq][heads*64] views (row stride *_ld, batch stride *_sb; head h occupies columns 64h..64h+63);
 * k, v: likewise with nk rows (v row-major here, not the transposed image of the forward).  dq / dk / dv: fp32, same indexing;
 * accumulate: 0 overwrite; 1 (= 3) add to what is there in all three; 2 add into dk / dv only, overwrite dq (the To2V processor's three attention
 * calls share K / V tensors — their gradients sum — but not queries).
 * ws: 16-byte aligned fp32 workspace of tg_attention_bwd_ws_floats(nq, nk, heads, batch) floats (row log-sum-exp, rowsum(dO o O), the statistics' seed
 * rows and the one-kernel form's per-(head, query tile) counters, which tg_attention_bwd_ex zeroes itself; the
 * [d][row] operands are read from the row-major tiles with the LDS transpose read, no transposed copies).
 * P is recomputed from the log-sum-exp tile by tile.  Launches: statistics, then dK/dV per 256-key workgroup + dQ per 256-query workgroup (7 executed
 * GEMMs).  No atomics on the data: run-to-run deterministic.  lse: optional [batch][heads][nq] fp32 row log-sum-exp (log2 domain) written by
 * tg_attention_fwd_lse for the same q / k / scale — the statistics launch then only forms rowsum(dO o O); NULL: recomputed here.
 * (An independent correct-first implementation of the same mathematics lives in the TEST-ONLY library tests/libtg_crosscheck.so.) */
int tg_attention_bwd(const void* q, long q_ld, long q_sb, const void* k, long k_ld, long k_sb, const void* v, long v_ld, long v_sb,
                     const void* o, long o_ld, long o_sb, const void* dout, long do_ld, long do_sb,
                     float* dq, long dq_ld, long dq_sb, float* dk, long dk_ld, long dk_sb, float* dv, long dv_ld, long dv_sb,
                     int nq, int nk, int heads, int batch, float scale, int accumulate, const float* lse, float* ws, hipStream_t stream);

/* tg_attention_bwd with the ONE-KERNEL form available (5 executed GEMMs: S and dP are formed once; the key blocks of a head add their dQ
 * contributions to the fp32 dQ tile in key-block order through the XCD's L2 — still no atomics on the data, still bitwise reproducible).
 *   flags & TG_BWD_ONE_KERNEL: the caller has run tg_attention_bwd_probe on THIS device and tg_attention_bwd_probe_verdict returned 1.  The form is
 *     then used for calls with >= 4 query tiles per key block and a multiple of 8 (batch, head) pairs; every other call takes the two launches above.
 *   status: caller-owned DEVICE int32[4], zeroed once by the caller (required with TG_BWD_ONE_KERNEL, may be NULL otherwise):
 *     [0] sticky count of ordered-exchange polls that gave up (a key block never saw its predecessor's signal within the poll limit).  The kernel
 *         does not hang the GPU in that case, it goes on — and the dq of that launch is INVALID.  The caller reads the word whenever it next
 *         synchronises (e.g. once per optimizer micro-step) and must discard the step when it is non-zero.  The launch itself returns 0: the
 *         condition only exists on the device, and the ABI never synchronises.
 *     [1] poll limit override (0: 2^20 polls of ~100 cycles); tests set 1 to force the path.
 *     [2] sticky count of workgroups that found key blocks of their (batch, head) on MORE than one XCD — the placement the ordered exchange rests on
 *         (a head's dQ traffic meets in ONE XCD's L2) did not hold for that launch: dq INVALID, same handling as [0].   [3] reserved (keep zero).
 * Nothing is allocated, freed or synchronised inside; safe under stream capture; no process-wide state. */
enum { TG_BWD_ONE_KERNEL = 1 };
int tg_attention_bwd_ex(const void* q, long q_ld, long q_sb, const void* k, long k_ld, long k_sb, const void* v, long v_ld, long v_sb,
                        const void* o, long o_ld, long o_sb, const void* dout, long do_ld, long do_sb,
                        float* dq, long dq_ld, long dq_sb, float* dk, long dk_ld, long dk_sb, float* dv, long dv_ld, long dv_sb,
                        int nq, int nk, int heads, int batch, float scale, int accumulate, const float* lse, float* ws, int flags, int* status,
                        hipStream_t stream);

/* One or two backward problems of the same heads / batch in one call — identical in effect to calling tg_attention_bwd_ex on each in order (problem 1's accumulate
 * flags see problem 0's results only through tensors the CALLER orders: the two may not write the same dq / dk / dv rows).  When both take the one-kernel form they share ONE launch:
 * the second problem's workgroups are appended behind the first's and fill its last, partial round of CUs (the To2V processor's main call, attention_processor.py:2066-2069, leaves a
 * quarter round; its vip-key call, :2117-2125, has the same number of query tiles and rides there).  Fields as the arguments of tg_attention_bwd_ex. */
typedef struct tg_attn_bwd_problem {
    const void* q; long q_ld, q_sb;  const void* k; long k_ld, k_sb;  const void* v; long v_ld, v_sb;
    const void* o; long o_ld, o_sb;  const void* dout; long do_ld, do_sb;
    float* dq; long dq_ld, dq_sb;  float* dk; long dk_ld, dk_sb;  float* dv; long dv_ld, dv_sb;
    int nq, nk;  float scale;  int accumulate;  const float* lse;  float* ws;
    /* optional (round 6): bf16 of the value dv receives, element (b, key, h, d) at dv_bf16[b*dv_bf16_sb + key*dv_bf16_ld + h*64 + d] — e.g. the V third of the fused
     * projection gradient, so that no conversion pass follows the backward.  With dv_bf16 given, dv may be null (then accumulate bit 1 must be clear). */
    void* dv_bf16; long dv_bf16_ld, dv_bf16_sb;
} tg_attn_bwd_problem;
int tg_attention_bwd_multi(const tg_attn_bwd_problem* problems, int count, int heads, int batch, int flags, int* status, hipStream_t stream);

/* Device probe of what the one-kernel form relies on: (1) the workgroups of one residue class mod 8 of a 1-D launch share an XCD; (2) the exchange protocol itself — a chain
 * of 32 workgroups on one XCD adds to 64 tiles in chain order with exactly the kernel's primitives.  buf: caller-owned device memory of
 * tg_attention_bwd_probe_bytes() bytes (16-byte aligned); the call clears it and launches the probe, asynchronously.  The caller copies buf to the
 * host once the stream has passed it; tg_attention_bwd_probe_verdict(host copy) = 1 when every sum is exact and no poll timed out, else 0. */
long tg_attention_bwd_probe_bytes(void);
int tg_attention_bwd_probe(void* buf, long nbytes, hipStream_t stream);
int tg_attention_bwd_probe_verdict(const void* host_copy, long nbytes);
long tg_attention_bwd_ws_floats(int nq, int nk, int heads, int batch);

/* Backward of tg_qk_layernorm_rope (y = rope(bf16(LN64(x) g + b)) * out_scale; attention_processor.py:2031-2056) for the trainable vip_norm_q /
 * vip_norm_k and the projections behind them.  x: the PRE-norm projection output (bf16, element (b, t, h, d) at x[b*strideB + t*ld + h*64 + d]);
 * dy: gradient w.r.t. y (fp32, same indexing with dy_ld / dy_strideB); dx: gradient w.r.t. x (bf16).  Rotary segments as in the forward.
 * partial: tg_qk_layernorm_rope_bwd_partial_floats floats = [blocks][2][64] per-block sums of (dL/dg, dL/db); the caller adds the blocks. */
int tg_qk_layernorm_rope_bwd(const void* x, long ld, long strideB, const float* dy, long dy_ld, long dy_strideB, void* dx, long dx_ld,
                             long dx_strideB, int tokens, int heads, int batch, const void* ln_weight, float eps, int start0, int len0,
                             const float* cos0, const float* sin0, int start1, int len1, const float* cos1, const float* sin1,
                             float out_scale, float* partial, hipStream_t stream);
long tg_qk_layernorm_rope_bwd_partial_floats(int tokens, int heads, int batch);

/* dst[c][r] = src[r][c] (bf16), r < rows; columns rows..rows_pad-1 of dst are written as zeros.  Weight gradients through tg_gemm_bf16
 * (C = A W^T, both operands K-contiguous): dW[out][in] = dY^T X = tg_gemm(A = dY^T [out][tokens], W = X^T [in][tokens]) with the token axis
 * padded to the GEMM's K granule; input gradients dX = dY W = tg_gemm(A = dY, W = W^T). */
int tg_transpose_2d(const void* src, long ld, int rows, int cols, void* dst, long ld_dst, int rows_pad, hipStream_t stream);

/* Bias gradients: partial[blk][c] = sum of src[r][c] over the block's rows (bf16 in, fp32 out; tg_colsum_partial_floats floats = blocks * cols:
 * 256 rows per block for tall matrices, fewer for short ones — a function of (rows, cols) only, so the summation order is fixed). */
int tg_colsum(const void* src, long ld, int rows, int cols, float* partial, hipStream_t stream);
long tg_colsum_partial_floats(int rows, int cols);

/* Backward of tg_adaln_modulate (y = ln (1 + scale[g]) + shift[g], ln = bf16(x_hat gamma + beta); normalization.py:441-460, 477-488): dx (bf16) and the
 * per-element products whose column sums are the parameter gradients — t_dln = dy (1 + scale) [-> d beta], t_dlnx = t_dln x_hat [-> d gamma],
 * t_dyln = dy ln [-> d scale[g] over the rows of group g]; d shift[g] = column sums of dy.  t_*: fp32 [batch*tokens][dim], caller-allocated, or all
 * three NULL where the norm's parameters are frozen (text / video rows).  add (optional, bf16, laid out like dx): the gradient arriving over the
 * residual connection; dx = bf16(bf16(norm gradient) + add) — the sum autograd forms on bf16 tensors. */
int tg_adaln_modulate_bwd(const void* x, long ldx, long strideX, const void* dy, long ld_dy, long stride_dy, void* dx, long ld_dx, long stride_dx,
                          const void* ln_weight, const void* ln_bias, float eps, int tokens, int dim, int batch, int modulate,
                          const tg_group_table* g, float* t_dln, float* t_dlnx, float* t_dyln, const void* add, long ld_add, long stride_add,
                          hipStream_t stream);

/* Backward of the gated residual out = res + gate[g] y (cogvideox_transformer_3d.py:290-293, 318-324): dy = gate[g] dout (bf16), t_dgate = dout y (fp32
 * [batch][tokens - t_row0][dim], written for the token rows >= t_row0 only — the group whose gate trains; column sums over a group's rows =
 * d gate[g]); d res = dout.  y is read for the token rows >= t_row0 only: a caller that kept just those rows passes the address row 0 WOULD have
 * (first kept row - t_row0 * ldy elements). */
int tg_gate_residual_bwd(const void* dout, long ld_dout, long stride_dout, const void* y, long ldy, long strideY, void* dy, long ld_dy, long stride_dy,
                         int tokens, int dim, int batch, const tg_group_table* gate, float* t_dgate, int t_row0, hipStream_t stream);

/* Elementwise: mode 0 out = silu(x); mode 1 out = dy * gelu_tanh'(x) (the FeedForward activation's derivative); mode 2 out = gelu_tanh(x), bit for
 * bit the TG_EPI_BIAS_GELU epilogue applied to a stored TG_EPI_BIAS output (the training forward keeps the pre-activation). bf16, n elements. */
int tg_act(const void* x, const void* dy, void* out, long n, int mode, hipStream_t stream);

/* tg_colsum for an fp32 matrix (same partial layout). */
int tg_colsum_f32(const float* src, long ld, int rows, int cols, float* partial, hipStream_t stream);
/* Column sums of up to TG_COLSUM_MAX matrices (bf16 or fp32, any row counts) in ONE launch: every item is cut into `row_blocks` row blocks, the partial sums form one
 * fp32 matrix partial[row_blocks][sum of cols] (item i's columns start at the sum of the earlier items' cols); the caller adds the rows in order.  `items`: HOST memory. */
#define TG_COLSUM_MAX 16
typedef struct tg_colsum_item {
    const void* src;
    long        ld;
    int         rows, cols;
    int         src_is_f32;
} tg_colsum_item;
int tg_colsum_multi(const tg_colsum_item* items, int count, int row_blocks, float* partial, hipStream_t stream);

/* Optimizer step on flat arenas (train_cogvideo_to2v.py:2012-2021: accelerator.clip_grad_norm_(transformer.parameters(), max_grad_norm), AdamW
 * (:1091-1098; betas / eps / weight decay of the yaml), zero_grad).  The reference's use_8bit_adam (bitsandbytes block-wise 8-bit moments, not under
 * /root/reference) exists to fit 80 GB parts; with 288 GB the moments stay fp32 (12.7 GB for the 1.6 B trainable parameters), which is torch.optim.AdamW.
 *   tg_grad_accumulate: acc = (overwrite ? 0 : acc) + scale * grad   (grad bf16 or fp32; scale = 1 / gradient_accumulation_steps)
 *   tg_grad_clip_coef:  coef[0] = ||grad||_2 (fixed-order sum), coef[1] = min(1, max_norm / (coef[0] + 1e-6)); ws: tg_grad_norm_ws_floats() floats
 *   tg_adamw_step:      torch.optim.AdamW update of bf16 parameters with fp32 moments, gradient scaled by *clip_coef when given (device pointer,
 *                       no host synchronisation); zero_grad != 0 clears grad in the same pass. */
int tg_grad_accumulate(const void* grad, int grad_is_bf16, float* acc, long n, float scale, int overwrite, hipStream_t stream);
/* acc_i += scale * grad_i for a whole list of (gradient, accumulator) pairs in ONE launch per TG_ACCUM_MAX items (a transformer block hands over ~20
 * gradients at once).  `items` is HOST memory: the table travels in the kernel arguments, nothing is copied to or kept on the device. */
#define TG_ACCUM_MAX 48
typedef struct tg_accum_item {
    const void* grad;            /* bf16 or fp32 device tensor, n elements */
    float*      acc;             /* fp32 device accumulator, n elements */
    long        n;
    int         grad_is_bf16;
} tg_accum_item;
int tg_grad_accumulate_multi(const tg_accum_item* items, int count, float scale, hipStream_t stream);
long tg_grad_norm_ws_floats(void);
int tg_grad_clip_coef(const float* grad, long n, float max_norm, float* ws, float* coef, hipStream_t stream);
int tg_adamw_step(void* param, float* grad, float* exp_avg, float* exp_avg_sq, long n, int step, float lr, float beta1, float beta2, float eps,
                  float weight_decay, const float* clip_coef, int zero_grad, hipStream_t stream);

/* Training loss of the To2V step and its gradient w.r.t. the model output (train_cogvideo_to2v.py:1995-2004; get_velocity
 * scheduling_dpm_cogvideox.py:521-538), per frame f (per-frame timesteps) over frame_elems elements:
 *   pred = bf16(bf16(sa_f * noisy) - bf16(sb_f * out))    sa_f = sqrt(acp_t), sb_f = sqrt(1 - acp_t), both cast to bf16 like the reference
 *   partial[f][block] = sum over the block's 256 elements of w_f * bf16(bf16(pred - target)^2)      w_f = 1 / (1 - acp_t), fp32
 *   grad = bf16( -sb_f * 2 w_f (pred - target) * inv_count )         inv_count = 1 / (elements per batch item * batch size)
 * coef fp32 [frames][3] = (sa, sb, w); model_out / noisy / target / grad bf16 [frames][frame_elems]; partial fp32, tg_vpred_loss_partial_floats
 * floats (summed per batch item on the host: fixed order, deterministic). */
int tg_vpred_loss_grad(const void* model_out, const void* noisy, const void* target, const float* coef, int frames, long frame_elems,
                       float inv_count, void* grad, float* partial, hipStream_t stream);
long tg_vpred_loss_partial_floats(int frames, long frame_elems);

/* Resampler's optional PCA low-rank filter (video_ipadapter/resampler.py:230-237: `pca.transform` -> zero every coefficient from 16 on ->
 * `pca.inverse_transform`, in the PCA's dtype = fp32, pca.py:56-66), per token:
 *   y_j = sum_c (x[r][c] - mean[c]) * comp[j][c]   (j < keep)        out[r][c] = bf16( mean[c] + sum_j y_j * comp[j][c] )
 * x / out bf16 [rows][D] (row strides ldx / ldo, in place allowed); comp fp32 [keep][D] (the first `keep` <= 16 rows of components_);
 * mean fp32 [D]. */
int tg_pca_lowrank_filter(const void* x, long ldx, const float* comp, const float* mean, void* out, long ldo, int rows, int D,
                          int keep, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * 3-D causal VAE (AutoencoderKLCogVideoX).  Activations are channels-last bf16: x[t][h][w][c], C % 64 == 0.
 * --------------------------------------------------------------------------------------------------------- */

/* Implicit-GEMM convolution on MFMA:  y[to][ho][wo][n] = bias[n] + sum_{dt,dh,dw,c} w[n][(dt,dh,dw)][c] * xv(tv,hv,wv)[c]
 *   tv = to + dt - (kt-1)              causal in time: tv < 0 reads `cache` frame (kt-1+tv), or frame 0 when cache == NULL
 *   hv = ho*stride + dh - pad, wv likewise; taps outside [0, H*up) x [0, W*up) contribute zero (F.pad constant 0)
 *   xv(t,h,w) = x[t_map ? t_map[t] : t][h/up][w/up]      nearest-neighbour upsampling folded into the loader
 * Replaces CogVideoXCausalConv3d (cat(cache, x) + F.pad + Conv3d, autoencoder_kl_cogvideox.py:120-145), the Conv2d of
 * CogVideoXUpsample3D after F.interpolate (up=2, kt=1, pad=1) and of CogVideoXDownsample3D (stride=2, pad=0 with the
 * (0,1,0,1) zero pad implied).  w is repacked [Cout_pad][kt*kh*kw][Cin] (Cout_pad % 128 == 0, or Cout_pad in {16, 32, ..., 112} for the
 * narrow output layers — conv_out: 3 / 32 channels — which run 128 x 16 tiles); only columns < cout
 * are stored (row stride ldy).  residual (optional, same layout as y) is added in the epilogue (ResnetBlock3D :309).
 * zeros: >= 2*Cin + 128 bytes of device zeros (source of out-of-range taps).  Cin % 64 == 0, with one exception:
 * Cin == 8 is the encoder's input convolution (CogVideoXEncoder3D.conv_in, :708-712: 3 -> 128 channels, 3x3x3, stride 1): x and cache carry 8 channels per
 * voxel (3 used; HARD PRECONDITION: channels 3..7 of x and cache are finite — the kernel's padded reduction entries multiply them by zero weights, so a NaN / Inf
 * there would reach all 128 outputs; tokensgen_amd.vae zero-fills them), w is packed [128][96] with reduction index k = tap * 3 + channel (81 real entries, the rest zero), cout = cout_pad = 128.
 * The decoder's conv_out (Cin = 128 -> cout <= 4, 3x3x3) takes a halo-tiled kernel with LDS-resident weights at launch scale; same arguments as any narrow layer.
 * gn_partial (optional, tg_conv3d_gn_partial_floats(To,Ho,Wo) floats; needs cout % 128 == 0): per 128-voxel tile row the sums and
 * sums of squares, per GroupNorm(32) group, of the bf16 values this launch stores — summed in a fixed order — so that the
 * GroupNorm / SpatialNorm that reads y next needs no statistics pass of its own: tg_groupnorm_finalize turns them into
 * (mean, rstd) exactly like the second stage of tg_groupnorm_stats.
 * splitk_ws: fp32 workspace of tg_conv3d_splitk_floats(...) floats (may be NULL when that returns 0).  Layers with fewer 128 x 128 tiles
 * than CUs and a long reduction (the 512-channel layers of a 30 x 45 latent tile) cut the (tap, channel) sum into ranges, one workgroup
 * each, and a second launch adds the partial tensors in a fixed order and runs the epilogue — deterministic, like everything else here. */
int tg_conv3d_cl(const void* x, int T, int H, int W, int Cin, const void* cache, const void* w, const void* bias,
                 int cout, int cout_pad, int kt, int kh, int kw, int stride, int pad, int up, const int32_t* t_map,
                 const void* residual, void* y, long ldy, int To, int Ho, int Wo, const void* zeros, float* gn_partial,
                 float* splitk_ws, hipStream_t stream);
long tg_conv3d_splitk_floats(int Cin, int cout, int cout_pad, int kt, int kh, int kw, int To, int Ho, int Wo);
long tg_conv3d_gn_partial_floats(int To, int Ho, int Wo);

/* CogVideoXUpsample3D's spatial part — F.interpolate(scale 2, nearest) followed by Conv2d 3x3, padding 1 (diffusers CogVideoXUpsample3D; call site
 * autoencoder_kl_cogvideox.py:849-883, the decoder's up blocks) — as FOUR 2x2 convolutions on the LOW-resolution input, one per output phase (py, px):
 *   y[t][2 yl + py][2 xl + px][n] = bias[n] + sum_{a,b in {0,1}} sum_c w_phases[py*2+px][n][a*2+b][c] * x[t][yl + a - (1 - py)][xl + b - (1 - px)][c]   (zero outside the image)
 * Of the three upsampled rows a 3x3 tap row reads, two are the same low-resolution row, so their weights can be added beforehand:
 *   py = 0: a = 0 <- dy 0, a = 1 <- dy 1 + dy 2;   py = 1: a = 0 <- dy 0 + dy 1, a = 1 <- dy 2        (columns likewise)
 * 4 taps instead of 9: 44 % of the multiply-adds of tg_conv3d_cl(up = 2) for the same output.  DEVIATION from the reference's arithmetic, stated: the pre-summed weights
 * are rounded to bf16 once (the caller sums in fp32 and rounds), where the reference adds the two or four bf16 products in fp32 — one more bf16 rounding on three layers of
 * the decoder, of the size of the activation roundings between its layers (tests hold it against fp32 torch and against tg_conv3d_cl(up = 2)).
 * time_x2 = 1: the layer also doubles TIME by nearest neighbour (`compress_time`: T frames -> 2T, or 2T - 1 when T is odd > 1 — the first frame is not repeated): the convolution is
 * 2-D per frame, so duplicate input frames give duplicate output frames — each low-resolution frame is convolved ONCE and stored twice (exact), and counted twice in the sums.
 * x [T][H][W][Cin] channels-last bf16; w_phases [4][cout][4][Cin] bf16; y [To][2H][2W] voxels with row stride ldy (To = T, or 2T / 2T - 1 with time_x2); gn_partial (optional):
 * tg_conv3d_up2_subpixel_gn_floats(T, H, W) floats, rows of [2][32] sums as for tg_conv3d_cl (4 x ceil(T H W / 128) rows: each phase owns its rows).
 * Runs on the 256 x 256 convolution kernel: tg_conv3d_up2_subpixel_ok(...) == 1 says the shape qualifies (cout % 256 == 0, Cin % 64 == 0, enough tiles); else TG_ERR_SHAPE and
 * the caller keeps tg_conv3d_cl(up = 2).  zeros as for tg_conv3d_cl. */
long tg_conv3d_up2_subpixel_ok(int T, int H, int W, int Cin, int cout);
long tg_conv3d_up2_subpixel_gn_floats(int T, int H, int W);
int tg_conv3d_up2_subpixel(const void* x, int T, int H, int W, int Cin, const void* w_phases, const void* bias, int cout, void* y, long ldy,
                           int time_x2, const void* zeros, float* gn_partial, hipStream_t stream);
int tg_groupnorm_finalize(const float* partial, long V, int C, float eps, float* stats, hipStream_t stream);

/* GroupNorm(32 groups, eps) statistics of x[V][C] -> stats[32][2] = {mean, rstd} (fp32).  partial: fp32 workspace of
 * tg_groupnorm_partial_floats(V, C) floats.  Deterministic two-stage reduction (fp64 finalisation). */
long tg_groupnorm_partial_floats(long V, int C);
int tg_groupnorm_stats(const void* x, long V, int C, float eps, float* partial, float* stats, hipStream_t stream);

/* y = silu(GroupNorm(x)) with precomputed stats (nn.GroupNorm + SiLU, autoencoder_kl_cogvideox.py:286-303). */
int tg_groupnorm_silu(const void* x, long V, int C, const float* stats, const void* gamma, const void* beta, void* y,
                      int apply_silu, hipStream_t stream);

/* CogVideoXSpatialNorm3D (+ SiLU), :171-188:  y = silu( GN(f) * conv_y(zq) + conv_b(zq) ), zq = the latent tile resized to
 * f's [T][H][W] by nearest neighbour (first frame mapped separately when T is odd > 1).  The two 1x1x1 convs commute with
 * the nearest resize, so the caller evaluates them once per LATENT voxel (one small tg_gemm_bf16 over the latent tile):
 * yz, bz: [Tz*Hz*Wz][ldz] bf16 = conv_y(z), conv_b(z); this kernel gathers them per output voxel. */
int tg_spatialnorm_silu(const void* f, int T, int H, int W, int C, const float* stats, const void* gamma, const void* beta,
                        const void* yz, const void* bz, long ldz, int Tz, int Hz, int Wz, void* y, int apply_silu,
                        hipStream_t stream);

/* The same two norm passes WITHOUT a statistics launch in front of them: instead of (mean, rstd) pairs they take the per-tile sums a convolution's
 * epilogue left (tg_conv3d_cl's gn_partial: rows of [2][32] fp32 — sum and sum of squares per GroupNorm group; sums_f64 = 0) and turn them into
 * (mean, rstd) in their own prologue: every workgroup adds the rows in the same fixed order in fp64 (deterministic, identical in every workgroup), so
 * the ~2 000 tg_groupnorm_finalize launches of a clip disappear.  At most 64 rows are read that way; a longer list (nrows = ceil(V / 128) > 64) is first
 * cut to tg_groupnorm_reduce_rows(nrows) <= 64 rows of fp64 by ONE coalesced pass, tg_groupnorm_reduce (out: that many rows of [2][32] doubles;
 * sums_f64 = 1).  Exactly one of `stats` / `sums` is given; eps is used with `sums` only.  GroupNorm(32, eps) of autoencoder_kl_cogvideox.py:171-188, 286-303. */
long tg_groupnorm_reduce_rows(long nrows);
int tg_groupnorm_reduce(const float* partial, long nrows, double* out, hipStream_t stream);
int tg_groupnorm_silu_ex(const void* x, long V, int C, const float* stats, const void* sums, long nsum, int sums_f64, float eps,
                         const void* gamma, const void* beta, void* y, int apply_silu, hipStream_t stream);
int tg_spatialnorm_silu_ex(const void* f, int T, int H, int W, int C, const float* stats, const void* sums, long nsum, int sums_f64, float eps,
                           const void* gamma, const void* beta, const void* yz, const void* bz, long ldz, int Tz, int Hz, int Wz, void* y,
                           int apply_silu, hipStream_t stream);

/* Temporal average pooling of CogVideoXDownsample3D(compress_time): pairs of frames are averaged; when T is odd the first
 * frame is kept.  x [T][HW][C] -> y [To][HW][C]. */
int tg_avgpool_time(const void* x, int T, long HW, int C, void* y, hipStream_t stream);

/* Layout changes at the VAE boundary: src NCDHW (fp32 if src_fp32 else bf16), one batch item [C][T][H][W], optional
 * crop window (h0,w0,Hc,Wc) and frame range [t0,t0+Tc) -> channels-last bf16 [Tc][Hc][Wc][Cpad] (extra channels zero),
 * scaled by `scale`; and the inverse (channels-last [T][H][W][ld] -> NCDHW window of a larger [C][Tt][Ht][Wt] tensor). */
int tg_ncdhw_to_cl(const void* src, int src_fp32, int C, int Tt, int Ht, int Wt, int t0, int Tc, int h0, int Hc, int w0, int Wc,
                   float scale, void* dst, int Cpad, hipStream_t stream);
int tg_cl_to_ncdhw(const void* src, long ld, int C, int T, int H, int W, void* dst, int dst_fp32, int Tt, int Ht, int Wt,
                   int t0, int h0, int w0, hipStream_t stream);

/* Tile seam blending of tiled_encode/tiled_decode (blend_v / blend_h, :1190-1204): along `axis` (3 = height, 4 = width) of
 * NCDHW tensors a,b [C][T][H][W] (fp32 or bf16), b[..., k, ...] = a[..., -extent+k, ...]*(1-k/extent) + b[..., k, ...]*(k/extent). */
int tg_tile_blend(const void* a, void* b, int is_fp32, int C, int T, int Ha, int Wa, int Hb, int Wb, int axis, int extent,
                  hipStream_t stream);

/* Dispatch overrides for the cross-check tests — NOT part of the operator surface.  The library reads no environment variable and keeps no other
 * process-wide state: the defaults are the shipped path, and these knobs only choose between two PRODUCT kernels of the same operator (each pair is held
 * bitwise or to rounding against each other by tests/).  Knobs: TG_ATTN_PP_MIN_WG (1024: workgroups from which the 8-wave attention kernel is used),
 * TG_ATTN_FIXEDM (1; 0: always the running row maximum), TG_ATTN_SPLIT (1; 0: no key-axis split of the last half round), TG_GEMM_W4 (1; 0: 8-wave GEMM),
 * TG_CONV_SPLITK (1; 0: never), TG_CONV_HALO / TG_CONV_W4 (1; 0: never, 2: whenever legal).  tokensgen_amd.lib forwards same-named environment
 * variables here at load time.  Unknown knob: TG_ERR_ARG.  tg_debug_knob_name(i) enumerates the names (NULL past the end). */
int tg_debug_set(const char* knob, long value);
int tg_debug_get(const char* knob, long* value);
const char* tg_debug_knob_name(int index);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* TOKENSGEN_HIP_H */
