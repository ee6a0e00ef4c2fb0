"""FIFO diagonal-denoising sampler — host mirror of longvgen/fifo_sampling/cogvideo_sampling_mp_fifo.py.

`FifoWorker.window_step` is the body of the reference's per-GPU worker (`fifo_onestep_per_gpu`, :408-579):
one CFG-batched DiT forward over a 13-latent-frame window, CFG combine and 13 per-frame DPM updates — here
1 DiT forward + ONE fused elementwise launch (tg_cfg_dpm_step) instead of 13 x ~12 tiny torch ops with >= 4
host syncs each.  `cogvideo_fifo_mp_v2` is the driver (:27-395): queue construction, window
geometry, write-back, shift + fresh tail noise.  Multi-GPU: one process per GPU (torch.distributed, RCCL); the
queue is replicated, rank g runs windows {g, g+n, ...} of every iteration and the kept half-windows are
exchanged with one all_gather per iteration (no CUDA-IPC pickling of the pipeline, no per-item respawn).
"""
import math

import numpy as np
import torch

from . import rope as R

BF16 = torch.bfloat16


class FifoWorker:
    """Per-GPU state that the reference ships to each spawned worker (pipe, prompt_embeds, image_rotary_emb, ...)."""

    def __init__(self, transformer, scheduler, prompt_embeds, image_rotary_emb, guidance_scale,
                 vip_grid_h=None, vip_grid_w=None, cond_grid_h=None, cond_grid_w=None):
        self.transformer = transformer
        self.scheduler = scheduler
        self.device = transformer.device
        self.prompt_embeds = prompt_embeds.to(self.device, BF16)
        self.image_rotary_emb = tuple(t.to(self.device, torch.float32).contiguous() for t in image_rotary_emb)
        self.guidance_scale = float(guidance_scale)
        self.vip_grid_h, self.vip_grid_w = vip_grid_h, vip_grid_w
        self.cond_grid_h, self.cond_grid_w = cond_grid_h, cond_grid_w
        self.head_dim = transformer.config.attention_head_dim

    def ropes_for(self, grid_t, cond_grid_t):
        """cogvideo_sampling_mp_fifo.py:478-489 (`_prepare_vip_rotary_positional_embeddings` twice per window)."""
        vr = R.rope_3d(self.head_dim, np.asarray(grid_t, dtype=np.float32), self.vip_grid_h, self.vip_grid_w, device=self.device)
        cr = R.rope_3d(self.head_dim, np.asarray(cond_grid_t, dtype=np.float32), self.cond_grid_h, self.cond_grid_w, device=self.device)
        return vr, cr

    @torch.no_grad()
    def window_step(self, latents, old_x0, has_old, t, prev_t, next_t, noise, grid_t=None, cond_grid_t=None,
                    image_embeddings=None):
        """latents [1,nf,C,H,W] bf16; old_x0 [nf,C,H,W] (rows without an estimate are ignored via has_old);
        t/prev_t/next_t: length-nf integer sequences (next_t <= 0 means "no back step", :544);
        noise [nf,2,C,H,W] bf16.  Returns (latents_out [1,nf,C,H,W], x0 [nf,C,H,W])."""
        nf = latents.shape[1]
        use_vip = image_embeddings is not None
        vr = cr = None
        if use_vip:
            vr, cr = self.ropes_for(grid_t, cond_grid_t)
        x = latents.to(self.device, BF16)
        inp = torch.cat([x, x], dim=0)                                        # :492-497 (CFG batch: uncond, cond)
        tt = torch.as_tensor(np.asarray(t, dtype=np.int64), device=self.device)[None].expand(2, -1)
        pred = self.transformer(hidden_states=inp, encoder_hidden_states=self.prompt_embeds, timestep=tt,
                                image_rotary_emb=self.image_rotary_emb, vip_image_rotary_emb=vr,
                                vip_condition_rotary_emb=cr, vip_encoder_hidden_states=image_embeddings,
                                return_dict=False)[0]
        t_back = [int(v) if int(v) > 0 else None for v in next_t]
        # the 2M branch needs a previous x0 AND a back step; the FIFO driver guarantees they coincide (App. C)
        for j in range(nf):
            if t_back[j] is None and has_old[j] and int(prev_t[j]) >= 0:
                raise IndexError("frame without timestep_back must not carry an old x0 (scheduling_dpm_cogvideox.py:459)")
        x_out, x0 = self.scheduler.window_step(pred, x[0].contiguous(), old_x0.contiguous(), noise.contiguous(),
                                               list(map(int, t)), list(map(int, prev_t)), t_back, list(has_old),
                                               self.guidance_scale)
        return x_out[None], x0
