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
