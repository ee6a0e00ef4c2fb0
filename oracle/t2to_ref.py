"""CPU restatement of the T2To stage (text -> condensed tokens): the reference's LongVGenCogVideoXPipeline.__call__
(longvgen/pipeline/pipeline_cogvideox_t2to.py:768-904) and its PCA tail (pca.py:56-66).

TEST INFRASTRUCTURE ONLY — tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this; the product
(tokensgen_amd/) never does.  Pinned against the reference itself: tests/golden/t2to_tiny.pt is produced by running the
reference pipeline class in this container (tools/make_golden.py t2to) and tests/test_oracle_golden.py replays it here.
"""
import math

import numpy as np
import torch

from . import dit_ref as D
from . import scheduler_ref as S


def dynamic_guidance(guidance_scale, num_inference_steps, t):
    """:852-855 — `1 + g * (1 - cos(pi * ((N - t) / N) ** 5.0)) / 2` in Python floats (t is the integer timestep value,
    NOT the step index, so the base is negative for t > N and the odd power keeps its sign)."""
    n = num_inference_steps
    return 1 + guidance_scale * ((1 - math.cos(math.pi * ((n - t) / n) ** 5.0)) / 2)


def rope_tables(head_dim, num_frames, h, w):
    """:543-564,812-839 — get_3d_rotary_pos_embed_v2 over integer-valued linspace grids with the 52/6/6 split."""
    f32 = np.float32
    return D.rope_3d(head_dim, np.linspace(0, num_frames, num_frames, endpoint=False, dtype=f32),
                     np.linspace(0, h, h, endpoint=False, dtype=f32), np.linspace(0, w, w, endpoint=False, dtype=f32),
                     dim_t=52, dim_h=6, dim_w=6)


def pca_inverse_tail(latents, mean, std, components, pca_mean, width=3072):
    """:890-899 — fp32 on the CPU: (b f h w) c rows, de-normalise the 16 coefficients, zero-pad to `width` columns,
    `Y @ components_ + mean_` (pca.py:64-66), back to [b f c h w] in the model dtype."""
    dtype = latents.dtype
    b, f, c, h, w = latents.shape
    x = latents.to(torch.float32).permute(0, 1, 3, 4, 2).reshape(-1, c)
    x = x * std[:, :16] + mean[:, :16]
    y = torch.zeros(x.shape[0], width, dtype=x.dtype)
    y[:, :16] = x
    out = torch.matmul(y, components) + pca_mean
    return out.reshape(b, f, h, w, width).permute(0, 1, 4, 2, 3).to(dtype)


def sample(denoise, ac, latents, timesteps, guidance_scale, use_dynamic_cfg, draw):
    """The denoising loop :841-888.  denoise(x[2,F,...], t[2]) -> [2,F,...] (uncond first, :795); draw() -> next fp32
    gaussian shaped like the latents, in the reference's call order (one per step, a second on the 2M branch)."""
    dt = latents.dtype
    ts = [int(t) for t in timesteps]
    n = len(ts)
    old = None
    for i, t in enumerate(ts):
        pred = denoise(torch.cat([latents] * 2), torch.tensor([t, t])).float()
        g = dynamic_guidance(guidance_scale, n, t) if use_dynamic_cfg else guidance_scale
        u, c = pred.chunk(2)
        pred = u + g * (c - u)
        prev_t = ts[i + 1] if i + 1 < n else -1
        t_back = ts[i - 1] if i > 0 else None
        latents_f, old = S.dpm_step(ac, pred, old, t, prev_t, t_back, latents, draw)
        latents = latents_f.to(dt)
    return latents


def t2to_pipeline(sd, cfg, latents, prompt_embeds, negative_prompt_embeds, timesteps, ac, guidance_scale, use_dynamic_cfg,
                  draw, mean, std, components, pca_mean, width=3072):
    """Whole stage: plain (no-vip) patch-1 DiT under CFG + SDE-DPM-solver++ + the PCA tail."""
    b, f, c, h, w = latents.shape
    rope = rope_tables(cfg["attention_head_dim"], f, h, w)
    emb = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0)

    def denoise(x, t):
        return D.dit_forward(sd, cfg, x, emb, t, image_rotary_emb=rope)

    lat = sample(denoise, ac, latents, timesteps, guidance_scale, use_dynamic_cfg, draw)
    return pca_inverse_tail(lat, mean, std, components, pca_mean, width), lat
