"""Training step pieces on the gfx950 kernels (BASELINE config 5; reference loop train_cogvideo_to2v.py:1721-2021).

Built so far: the loss + its gradient w.r.t. the model output (`vpred_loss_and_grad`, :1995-2010) and the backward of the dominant
operator — the three attention calls of the To2V processor (`to2v_attention_backward`, attention_processor.py:2066-2135) on
tg_attention_bwd.  Not built yet: backward of the projections / norms / RoPE / FeedForward, gradient checkpointing,
the DDP all-reduce and the optimizer — the forward-only product never routes through this module."""
import torch

from . import kernels as K
from . import lib as L

BF16 = torch.bfloat16


@torch.no_grad()
def vpred_loss_and_grad(model_output, noisy_model_input, model_input, timesteps, alphas_cumprod):
    """train_cogvideo_to2v.py:1990-2010.  model_output / noisy_model_input / model_input: bf16 [B, F, C, H, W]; timesteps int64 [B, F] (per-frame)
    or [B]; alphas_cumprod: the scheduler's fp32 table.  Returns (loss scalar fp32 = mean over the batch of the per-item weighted MSE, per-item
    losses [B], d loss / d model_output bf16 like model_output)."""
    for n, t in (("model_output", model_output), ("noisy_model_input", noisy_model_input), ("model_input", model_input)):
        K._chk(t, n)
        assert t.is_contiguous() and t.shape == model_output.shape
    B, F = model_output.shape[:2]
    E = model_output[0, 0].numel()
    ts = timesteps.to(model_output.device).reshape(B, -1)
    if ts.shape[1] == 1:
        ts = ts.expand(B, F)
    acp = alphas_cumprod.to(model_output.device, torch.float32)[ts.reshape(-1)]
    # get_velocity casts the table to the sample dtype before the square roots (scheduling_dpm_cogvideox.py:524-532)
    acp_b = acp.to(BF16)
    coef = torch.stack([(acp_b ** 0.5).float(), ((1 - acp_b) ** 0.5).float(), 1.0 / (1.0 - acp)], dim=1).contiguous()
    grad = torch.empty_like(model_output)
    lib = L.load()
    partial = torch.empty(lib.tg_vpred_loss_partial_floats(B * F, E), dtype=torch.float32, device=model_output.device)
    L.check(lib.tg_vpred_loss_grad(model_output.data_ptr(), noisy_model_input.data_ptr(), model_input.data_ptr(), coef.data_ptr(), B * F, E,
                                   1.0 / (F * E * B), grad.data_ptr(), partial.data_ptr(), K._stream()), "tg_vpred_loss_grad")
    per_item = partial.view(B, -1).sum(dim=1) / (F * E)
    return per_item.mean(), per_item, grad


@torch.no_grad()
def to2v_attention_backward(q, k, v, qx, kx, vx, qv, kv, vv, o1, o2, o3, d_out, heads, sm_scale, vip_scale):
    """Backward of `cat(sdpa(q, k, v) + vip_scale * sdpa(qx, kv, vv), sdpa(qv, cat(kx, kv), cat(vx, vv)))` (attention_processor.py:2066-2135):
    q..vv are the post-norm / post-RoPE projections [B, n, heads*64] (bf16), o1/o2/o3 the three attention outputs saved by the forward,
    d_out [B, N1 + Np, heads*64] the gradient of the concatenated result.  Returns fp32 gradients keyed like the inputs.  kv / vv receive
    the sum of two calls' gradients (accumulate)."""
    N1, Np = q.shape[1], qv.shape[1]
    f32 = torch.float32
    g1 = d_out[:, :N1]
    dq, dk, dv = K.attention_bwd(q, k, v, o1, g1, heads, sm_scale)
    g2 = (g1.float() * float(vip_scale)).to(BF16)          # `scale * O2` is a bf16 tensor in the forward
    dqx, dkv, dvv = K.attention_bwd(qx, kv, vv, o2, g2, heads, sm_scale)
    B, HD = q.shape[0], q.shape[2]
    dkc = torch.zeros(B, N1 + Np, HD, dtype=f32, device=q.device)
    dvc = torch.zeros_like(dkc)
    dkc[:, N1:], dvc[:, N1:] = dkv, dvv
    dqv, _, _ = K.attention_bwd(qv, torch.cat([kx, kv], 1), torch.cat([vx, vv], 1), o3, d_out[:, N1:], heads, sm_scale, dk=dkc, dv=dvc, accumulate=True)
    return dict(q=dq, k=dk, v=dv, qx=dqx, kx=dkc[:, :N1], vx=dvc[:, :N1], qv=dqv, kv=dkc[:, N1:], vv=dvc[:, N1:])


# ---------------------------------------------------------------------------------------------------------------------------------
# projection / norm backward around the attention (the trainable vip_to_{q,k,v}, vip_norm_{q,k}: cogvideox_transformer_3d.py:207-218)
# ---------------------------------------------------------------------------------------------------------------------------------
def _pad_to(n, m):
    return (n + m - 1) // m * m


def transpose_2d(src, rows_pad=None):
    """bf16 [R, C] (row stride may exceed C) -> [C, rows_pad] with zero columns R.. (tg_transpose_2d)."""
    K._chk(src, "src")
    R, C_ = src.shape
    rp = R if rows_pad is None else rows_pad
    dst = torch.empty(C_, rp, dtype=BF16, device=src.device)
    L.check(L.load().tg_transpose_2d(src.data_ptr(), src.stride(0), R, C_, dst.data_ptr(), rp, rp, K._stream()), "tg_transpose_2d")
    return dst


def colsum(src):
    """fp32 [C] column sums of a bf16 [R, C] matrix (tg_colsum + a fixed-order sum of the per-block partials)."""
    K._chk(src, "src")
    R, C_ = src.shape
    lib = L.load()
    part = torch.empty(lib.tg_colsum_partial_floats(R, C_), dtype=torch.float32, device=src.device)
    L.check(lib.tg_colsum(src.data_ptr(), src.stride(0), R, C_, part.data_ptr(), K._stream()), "tg_colsum")
    return part.view(-1, C_).sum(dim=0)


def linear_backward(x2d, dy2d, weight=None, need_dx=False):
    """y = x W^T + b with W [out, in]:  dW = dy^T x (bf16, through the MFMA GEMM with both operands transposed so the token axis is the
    contiguous reduction axis), db = column sums of dy (fp32), dx = dy W (bf16, only when asked).  x2d [M, in], dy2d [M, out] bf16."""
    M, cin = x2d.shape
    cout = dy2d.shape[1]
    Mp, cin_p, cout_p = _pad_to(M, 64), _pad_to(cin, 128), _pad_to(cout, 64)     # the GEMM wants N % 128 == 0 and K % 64 == 0
    dyT = transpose_2d(dy2d, Mp)                                                 # [out, Mp]
    xT = torch.zeros(cin_p, Mp, dtype=BF16, device=x2d.device) if cin_p != cin else None
    if xT is None:
        xT = transpose_2d(x2d, Mp)                                               # [in, Mp]
    else:
        L.check(L.load().tg_transpose_2d(x2d.data_ptr(), x2d.stride(0), M, cin, xT.data_ptr(), Mp, Mp, K._stream()), "tg_transpose_2d")
    dWp = torch.empty(cout, cin_p, dtype=BF16, device=x2d.device)
    K.gemm(dyT, xT, None, dWp, L.EPI_BIAS)                                       # dW[o, i] = sum_m dy[m, o] x[m, i]
    dW = dWp[:, :cin]
    db = colsum(dy2d)
    dx = None
    if need_dx:
        wT = torch.zeros(cin_p, cout_p, dtype=BF16, device=x2d.device)            # W^T [in, out], zero padded
        L.check(L.load().tg_transpose_2d(weight.data_ptr(), weight.stride(0), cout, cin, wT.data_ptr(), cout_p, cout_p, K._stream()), "tg_transpose_2d")
        dyp = dy2d if cout_p == cout else torch.nn.functional.pad(dy2d, (0, cout_p - cout))
        dxp = torch.empty(M, cin_p, dtype=BF16, device=x2d.device)
        K.gemm(dyp, wT, None, dxp, L.EPI_BIAS)                                   # dx[m, i] = sum_o dy[m, o] W[o, i]
        dx = dxp[:, :cin]
    return dW, db, dx


def qk_layernorm_rope_backward(x_pre, dy, heads, ln_weight, eps, seg0=None, seg1=None, out_scale=1.0):
    """Backward of kernels.qk_layernorm_rope.  x_pre: the PRE-norm projection [B, T, heads*64] bf16 (a column slice of the fused QKV buffer
    is fine); dy fp32 [B, T, heads*64].  Returns (dx bf16 contiguous, dgamma fp32 [64], dbeta fp32 [64])."""
    K._chk(x_pre, "x_pre"); K._chk(dy, "dy", torch.float32)
    B, T, HD, ld, sb = K._bmk(x_pre)
    assert HD == heads * 64 and dy.shape == x_pre.shape

    def unpack(seg):
        if seg is None:
            return 0, 0, None, None
        start, (cos, sin) = seg
        return int(start), int(cos.shape[0]), cos, sin
    s0, l0, c0, n0 = unpack(seg0)
    s1, l1, c1, n1 = unpack(seg1)
    dx = torch.empty(B, T, HD, dtype=BF16, device=x_pre.device)
    lib = L.load()
    part = torch.empty(lib.tg_qk_layernorm_rope_bwd_partial_floats(T, heads, B), dtype=torch.float32, device=x_pre.device)
    L.check(lib.tg_qk_layernorm_rope_bwd(x_pre.data_ptr(), ld, sb, dy.data_ptr(), dy.stride(1), dy.stride(0), dx.data_ptr(), dx.stride(1), dx.stride(0),
                                         T, heads, B, ln_weight.data_ptr(), float(eps), s0, l0, K._p(c0), K._p(n0), s1, l1, K._p(c1), K._p(n1),
                                         float(out_scale), part.data_ptr(), K._stream()), "tg_qk_layernorm_rope_bwd")
    sums = part.view(-1, 2, 64).sum(dim=0)
    return dx, sums[0], sums[1]


@torch.no_grad()
def vip_projection_backward(xn_all, qkvv_pre, grads, heads, Nt, N1, vip_norm_q_w, vip_norm_k_w, vip_rope, cond_rope):
    """From the attention gradients of the vip-weight branch to the gradients of the TRAINABLE processor parameters.
    xn_all [B, N, D] bf16: the normalised inputs (text | video | vip rows) the projection read; qkvv_pre [B, N, 3D] bf16: its raw output
    (before vip_norm_q / vip_norm_k and RoPE); grads: to2v_attention_backward(...) (fp32, UNSCALED keys: the training forward keeps the softmax
    scale in the attention call).  Returns dict: vip_to_{q,k,v}.{weight,bias}, vip_norm_{q,k}.{weight,bias}."""
    B, N, D = xn_all.shape
    f32 = torch.float32
    dq = torch.cat([grads["qx"], grads["qv"]], dim=1).contiguous()            # rows: text+video (x-branch) | vip tokens
    dk = torch.cat([grads["kx"], grads["kv"]], dim=1).contiguous()
    dv = torch.cat([grads["vx"], grads["vv"]], dim=1)
    segs = ((Nt, vip_rope), (N1, cond_rope))
    dq_pre, dgq, dbq = qk_layernorm_rope_backward(qkvv_pre[:, :, :D], dq, heads, vip_norm_q_w, 1e-6, *segs)
    dk_pre, dgk, dbk = qk_layernorm_rope_backward(qkvv_pre[:, :, D:2 * D], dk, heads, vip_norm_k_w, 1e-6, *segs)
    d_pre = torch.cat([dq_pre, dk_pre, dv.to(BF16)], dim=2).view(B * N, 3 * D)
    dW, db, _ = linear_backward(xn_all.reshape(B * N, D), d_pre)
    out = {}
    for j, n in enumerate(("q", "k", "v")):
        out[f"vip_to_{n}.weight"], out[f"vip_to_{n}.bias"] = dW[j * D:(j + 1) * D], db[j * D:(j + 1) * D]
    out["vip_norm_q.weight"], out["vip_norm_q.bias"], out["vip_norm_k.weight"], out["vip_norm_k.bias"] = dgq, dbq, dgk, dbk
    return out
