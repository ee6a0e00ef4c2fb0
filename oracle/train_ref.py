"""Oracle: the training step's loss (TEST INFRASTRUCTURE ONLY) — train_cogvideo_to2v.py:1990-2010 with
CogVideoXDPMScheduler.get_velocity (scheduling_dpm_cogvideox.py:521-538), restated with the reference's dtype behaviour: get_velocity works in the
sample dtype (the alphas table is cast to it BEFORE the square roots), the weights keep the TABLE's dtype (fp64 in the reference scheduler, so the
loss is an fp64 scalar).  Pinned (round 3): tests/golden/train_loss.pt holds `add_noise` / `get_velocity` outputs of the reference's scheduler class
and the loss evaluated with the training script's own lines on fixed tensors (tools/make_golden.py::gen_train; fp32 + bf16, [B] and [B, F] timesteps);
tests/test_oracle_golden.py::test_train_loss_oracle_matches_reference holds these functions to it bit for bit.  (The training SCRIPT as a whole needs
accelerate and a dataset and is not run.)"""
import torch


def add_noise(alphas_cumprod, original_samples, noise, timesteps):
    """scheduling_dpm_cogvideox.py:470-495 (samples [N, ...] with timesteps [N]; the training script flattens (b f) around the call for per-frame
    timesteps, train_cogvideo_to2v.py:1789-1795)."""
    acp = alphas_cumprod.to(dtype=original_samples.dtype)
    sa = (acp[timesteps] ** 0.5).flatten()
    sb = ((1 - acp[timesteps]) ** 0.5).flatten()
    while sa.dim() < original_samples.dim():
        sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
    return sa * original_samples + sb * noise


def get_velocity(alphas_cumprod, sample, noise, timesteps):
    """scheduling_dpm_cogvideox.py:521-538."""
    acp = alphas_cumprod.to(dtype=sample.dtype)
    sa = (acp[timesteps] ** 0.5).flatten()
    sb = ((1 - acp[timesteps]) ** 0.5).flatten()
    while sa.dim() < sample.dim():
        sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
    return sa * noise - sb * sample


def vpred_loss(alphas_cumprod, model_output, noisy_model_input, model_input, timesteps):
    """train_cogvideo_to2v.py:1990-2010: per-frame timesteps [B, F] are flattened with the frames; returns (loss, per-item losses)."""
    B = model_output.shape[0]
    if timesteps.dim() > 1:
        timesteps = timesteps.reshape(-1)
        model_output, noisy_model_input, model_input = (t.flatten(0, 1) for t in (model_output, noisy_model_input, model_input))
    pred = get_velocity(alphas_cumprod, model_output, noisy_model_input, timesteps)
    w = 1 / (1 - alphas_cumprod[timesteps])
    while w.dim() < pred.dim():
        w = w.unsqueeze(-1)
    per_item = torch.mean((w * (pred - model_input) ** 2).reshape(B, -1), dim=1)
    return per_item.mean(), per_item
