"""Oracle: CogVideoXDPMScheduler (SDE-DPM-solver++ 2M) — TEST INFRASTRUCTURE ONLY.

Restates longvgen/schedulers/scheduling_dpm_cogvideox.py with the gaussian noise passed in
explicitly (the reference draws it from the global RNG, `:452,460`), so the map is deterministic.
"""
import numpy as np
import torch


def alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.0120, snr_shift_scale=1.0,
                   rescale_betas_zero_snr=True):
    """scheduling_dpm_cogvideox.py:198-221 + rescale_zero_terminal_snr :96-124. Returns (betas, ᾱ) fp64."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float64) ** 2
    ac = torch.cumprod(1.0 - betas, dim=0)
    ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)
    if rescale_betas_zero_snr:
        s = ac.sqrt()
        s0, sT = s[0].clone(), s[-1].clone()
        s = (s - sT) * (s0 / (s0 - sT))
        ac = s ** 2
    return betas, ac


def trailing_timesteps(num_inference_steps, num_train_timesteps=1000):
    """scheduling_dpm_cogvideox.py:321-326 ("trailing")."""
    ratio = num_train_timesteps / num_inference_steps
    return np.round(np.arange(num_train_timesteps, 0, -ratio)).astype(np.int64) - 1


def step_coefficients(ac, t, prev_t, t_back):
    """get_variables/get_mult/mult_noise, scheduling_dpm_cogvideox.py:334-356,424-427. fp64 0-dim tensors.

    Returns dict(sa, sb, m1, m2, m3, m4, mn); m3/m4 None when t_back is None."""
    a_t = ac[t]
    a_p = ac[prev_t] if prev_t >= 0 else torch.tensor(1.0, dtype=ac.dtype)
    lam = ((a_t / (1 - a_t)) ** 0.5).log()
    lam_n = ((a_p / (1 - a_p)) ** 0.5).log()
    h = lam_n - lam
    out = dict(sa=a_t ** 0.5, sb=(1 - a_t) ** 0.5,
               m1=((1 - a_p) / (1 - a_t)) ** 0.5 * (-h).exp(),
               m2=(-2 * h).expm1() * a_p ** 0.5,
               mn=(1 - a_p) ** 0.5 * (1 - (-2 * h).exp()) ** 0.5, m3=None, m4=None)
    if t_back is not None:
        a_b = ac[t_back]
        lam_b = ((a_b / (1 - a_b)) ** 0.5).log()
        r = (lam - lam_b) / h
        out["m3"] = 1 + 1 / (2 * r)
        out["m4"] = 1 / (2 * r)
    return out


def dpm_step(ac, model_output, old_x0, t, prev_t, t_back, sample, draw, prediction_type="v_prediction"):
    """scheduling_dpm_cogvideox.py:358-468 with the noise source injected. Returns (prev_sample, x0).

    `draw()` returns the next gaussian tensor shaped/dtyped like `sample`; it is called once, and a
    second time only on the 2M branch — the same order and count as the reference's `randn_tensor`
    calls (:452,460).  Type promotion follows torch: 0-dim fp64 coefficients do not widen the sample."""
    c = step_coefficients(ac, int(t), int(prev_t), None if t_back is None else int(t_back))
    if prediction_type == "v_prediction":
        x0 = c["sa"] * sample - c["sb"] * model_output
    elif prediction_type == "epsilon":
        x0 = (sample - c["sb"] * model_output) / c["sa"]
    else:
        x0 = model_output
    prev = c["m1"] * sample - c["m2"] * x0 + c["mn"] * draw()
    if old_x0 is None or prev_t < 0:
        return prev, x0
    d = c["m3"] * x0 - c["m4"] * old_x0
    return c["m1"] * sample - c["m2"] * d + c["mn"] * draw(), x0


def add_noise_to_xt(betas, xt_prev, noise, t=999):
    """scheduling_dpm_cogvideox.py:497-518 (un-shifted betas).  The reference indexes betas with a
    1-element tensor, so the coefficients are 1-D fp64 and the result is PROMOTED to fp64 (the caller
    casts it back when writing into the queue, cogvideo_sampling_mp_fifo.py:124-128)."""
    b = betas[torch.tensor([t])]
    while b.ndim < xt_prev.ndim:
        b = b.unsqueeze(-1)
    return (1 - b) ** 0.5 * xt_prev + b ** 0.5 * noise


def cfg_combine(noise_pred, guidance_scale):
    """cogvideo_sampling_mp_fifo.py:531-533."""
    u, c = noise_pred.chunk(2)
    return u + guidance_scale * (c - u)


def cfg_combine_separate(noise_pred, guidance_scale, guidance_scale_img):
    """cogvideo_sampling_mp_fifo.py:528-530 (`use_separate_guidance`): batch rows (uncond_txt, uncond_img, txt_img).  The scales are Python floats
    (static) or fp32 tensors shaped [1, F, 1, 1, 1] (dynamic cfg) — torch's promotion rules then decide the dtype exactly as in the reference."""
    ut, ui, c = noise_pred.chunk(3)
    return c + (guidance_scale - 1) * (c - ut) + (guidance_scale_img - 1) * (c - ui)


def dynamic_guidance(scale, t, num_inference_steps):
    """cogvideo_sampling_mp_fifo.py:519-527: the cosine ramp evaluated on the window's integer timestep TENSOR (fp32 tensor arithmetic),
    shaped for broadcasting over [B, F, C, H, W]."""
    import math
    g = 1 + scale * ((1 - torch.cos(math.pi * ((num_inference_steps - t) / num_inference_steps) ** 5.0)) / 2)
    return g[None, :, None, None, None]
