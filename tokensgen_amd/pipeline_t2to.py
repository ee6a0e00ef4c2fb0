"""T2To stage on the gfx950 kernels: text -> condensed tokens (`image_embeddings`) for the To2V stage.

Host mirror of the reference's `LongVGenCogVideoXPipeline` (longvgen/pipeline/pipeline_cogvideox_t2to.py:265-912):
a plain (no-vip) CogVideoX DiT with patch_size 1 over latents [1, num_chunks*4, 16, 8, 12], RoPE split 52/6/6 over
integer grids (:543-564), classifier-free guidance with the optional cosine "dynamic" schedule (:852-855), the SDE
DPM-solver++ loop with fp32 solver state (:845-870) and the de-normalise + PCA-inverse tail (:890-899) — every device op
goes through libtokensgen_hip.so.  T5 prompt encoding is upstream of the path: pass prompt_embeds / negative_prompt_embeds.
"""
import math
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import cfg_parallel as CP
from . import kernels as K
from . import rope as R

BF16 = torch.bfloat16


class CogVideoXPipelineOutput(SimpleNamespace):
    """:261-263 — `.frames` holds the condensed tokens [b, f, c, h, w]."""


def _load(x):
    from . import compat
    compat.ensure_pca_module()
    return torch.load(x, weights_only=False) if isinstance(x, (str, bytes)) or hasattr(x, "read") else x


class LongVGenCogVideoXPipeline:
    def __init__(self, transformer, scheduler, device=None, **unused):
        self.transformer = transformer
        self.scheduler = scheduler
        self.device = torch.device(device) if device is not None else transformer.device
        self._guidance_scale = 6.0
        self._num_timesteps = 0
        if transformer.use_vip:
            raise ValueError("the T2To model is a plain CogVideoX DiT: do not call set_vip_layers on it")

    @classmethod
    def from_pretrained(cls, path, transformer=None, torch_dtype=BF16, device="cuda", **unused):
        """DiffusionPipeline.from_pretrained as the entry script calls it (infer_cogvideo_mp_fifo.py:225-229): the T2To transformer is handed
        in, the scheduler configuration comes from `<path>/scheduler/scheduler_config.json`; T5 stays on the reference side."""
        from .scheduler import CogVideoXDPMScheduler
        from .transformer import CogVideoXTransformer3DModel
        if transformer is None:
            transformer = CogVideoXTransformer3DModel.from_pretrained(path, subfolder="transformer", torch_dtype=torch_dtype, device=device)
        return cls(transformer, CogVideoXDPMScheduler.from_config(os.path.join(path, "scheduler", "scheduler_config.json")))

    def to(self, device=None, *unused, **kw):
        if device is not None and not isinstance(device, torch.dtype) and torch.device(device) != self.device:
            self.device = torch.device(device)
            self.transformer.to(self.device)
        return self

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def num_timesteps(self):
        return self._num_timesteps

    def prepare_latents(self, batch_size, num_channels_latents, num_chunks, num_frames_per_chunk, height, width, generator=None, latents=None):
        """:436-460.  A CPU generator draws on the CPU and the result is moved (diffusers' randn_tensor rule), so the same
        seed gives the same start as the reference; a device generator draws on the device."""
        shape = (batch_size, num_chunks * num_frames_per_chunk, num_channels_latents, height, width)
        if latents is None:
            gdev = generator.device if generator is not None else self.device
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=BF16)
        elif tuple(latents.shape) != shape:
            raise ValueError(f"latents have shape {tuple(latents.shape)}, expected {shape}")
        return latents.to(self.device, BF16) * self.scheduler.init_noise_sigma

    def _prepare_rotary_positional_embeddings(self, grid_t, grid_h, grid_w, device=None):
        """:543-564 — the 2nd-stage split of the 64 rotary dims: 52 temporal, 6 + 6 spatial."""
        return R.rope_3d(self.transformer.config.attention_head_dim, grid_t, grid_h, grid_w, dim_t=52, dim_h=6, dim_w=6,
                         device=device or self.device)

    @torch.no_grad()
    def __call__(self, prompt=None, negative_prompt=None, height=480, width=720, num_frames_per_chunk=49, num_chunks=1,
                 num_inference_steps=50, timesteps=None, guidance_scale=6, use_dynamic_cfg=False, num_videos_per_prompt=1, eta=0.0,
                 generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None, output_type="pil", return_dict=True,
                 attention_kwargs=None, max_sequence_length=226, longvgen_mean=None, longvgen_std=None, longvgen_pca=None,
                 step_noise=None, cfg_parallel=None, **unused):
        """:566-912.  height/width/num_frames_per_chunk are the condensed-token grid (8, 12, 4 in the shipped configs).
        step_noise: optional callable (i, k) -> bf16 gaussian shaped like the latents (k = 0: first draw of step i, 1: the 2M
        branch's draw); default: `generator`, drawn in the reference's order and dtype (the sample's, scheduling_dpm:452,460).
        cfg_parallel: see tokensgen_amd/cfg_parallel.py (default: split the two CFG halves over ranks 0/1 when >= 2 ranks run)."""
        if prompt is not None or negative_prompt is not None:
            raise NotImplementedError("T5 prompt encoding is upstream of the hot path: pass prompt_embeds and negative_prompt_embeds")
        if prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        if num_frames_per_chunk > 4:
            raise ValueError("The number of frames must equal 4 for now due to static positional embeddings. This will be updated in "
                             "the future to remove this limitation.")
        if longvgen_mean is None or longvgen_std is None or longvgen_pca is None:
            raise ValueError("longvgen_mean, longvgen_std and longvgen_pca are required (:769-771)")
        if timesteps is not None:
            raise NotImplementedError("custom timesteps: CogVideoXDPMScheduler.set_timesteps does not accept them")
        if attention_kwargs:
            raise NotImplementedError("attention_kwargs (LoRA scale / masks) are not on the hot path")
        dev = self.device
        mean, std, pca = _load(longvgen_mean), _load(longvgen_std), _load(longvgen_pca)
        self._guidance_scale = guidance_scale
        do_cfg = guidance_scale > 1.0
        if do_cfg and negative_prompt_embeds is None:
            raise ValueError("negative_prompt_embeds is required when guidance_scale > 1")
        if negative_prompt_embeds is not None and prompt_embeds.shape != negative_prompt_embeds.shape:
            raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed directly, but got: "
                             f"`prompt_embeds` {prompt_embeds.shape} != `negative_prompt_embeds` {negative_prompt_embeds.shape}.")
        batch = prompt_embeds.shape[0]
        if batch != 1:
            raise NotImplementedError("one prompt per call (the reference's entry script does the same)")
        # uncond first (:795); without guidance the kernel still wants both halves: g = 1 reduces to the conditional branch
        neg = negative_prompt_embeds if do_cfg else prompt_embeds
        embeds = torch.cat([neg, prompt_embeds], dim=0).to(dev, BF16)

        self.scheduler.set_timesteps(num_inference_steps, device=None)
        ts = self.scheduler.timesteps.tolist()
        self._num_timesteps = len(ts)
        nfr = num_chunks * num_frames_per_chunk
        latents = self.prepare_latents(batch, 16, num_chunks, num_frames_per_chunk, height, width, generator, latents)
        f32 = np.float32
        rope = self._prepare_rotary_positional_embeddings(
            grid_t=np.linspace(0, nfr, nfr, endpoint=False, dtype=f32), grid_h=np.linspace(0, height, height, endpoint=False, dtype=f32),
            grid_w=np.linspace(0, width, width, endpoint=False, dtype=f32))
        shape = tuple(latents.shape[1:])                        # [F, 16, h, w]

        def draw(i, k):
            if step_noise is not None:
                return step_noise(i, k).to(dev, BF16).reshape(shape)
            gdev = generator.device if generator is not None else dev
            return torch.randn((1,) + shape, generator=generator, device=gdev, dtype=BF16).to(dev)[0]

        old_x0 = None
        cfg_mode = CP.resolve(cfg_parallel)
        zeros = torch.zeros(shape, dtype=torch.float32, device=dev)
        noise = torch.zeros((nfr, 2) + shape[1:], dtype=BF16, device=dev)
        for i, t in enumerate(ts):
            def fwd(lo, hi):
                n = hi - lo
                return self.transformer(hidden_states=torch.cat([latents] * n, dim=0), encoder_hidden_states=embeds[lo:hi],
                                        timestep=torch.full((n,), t, dtype=torch.int64, device=dev), image_rotary_emb=rope, return_dict=False)[0]
            pred = CP.predict(cfg_mode, lambda h: fwd(h, h + 1), lambda: fwd(0, 2))
            if use_dynamic_cfg:                              # :852-855, Python floats like the reference
                self._guidance_scale = 1 + guidance_scale * (
                    (1 - math.cos(math.pi * ((num_inference_steps - t) / num_inference_steps) ** 5.0)) / 2)
            g = self._guidance_scale if do_cfg else 1.0
            prev_t = ts[i + 1] if i + 1 < len(ts) else -1
            t_back = ts[i - 1] if i > 0 else None
            has = old_x0 is not None
            second = has and prev_t >= 0                     # the 2M branch draws a second gaussian and keeps that one (:452-463)
            n0 = draw(i, 0)
            noise[:, 0] = n0
            noise[:, 1] = draw(i, 1) if second else n0
            coef = self.scheduler.coef_table([t] * nfr, [prev_t] * nfr, [t_back] * nfr, [second] * nfr, dev)
            x_out = torch.empty_like(latents[0])
            x0_out = torch.empty(shape, dtype=torch.float32, device=dev)
            K.cfg_dpm_step_f32(pred.reshape(2, nfr, -1), latents[0].reshape(nfr, -1), (old_x0 if has else zeros).reshape(nfr, -1),
                               noise.reshape(nfr, 2, -1), coef, g, x_out.view(nfr, -1), x0_out.view(nfr, -1))
            latents, old_x0 = x_out[None], x0_out

        # :890-899 — de-normalise the 16 coefficients, PCA inverse to the condensed-token width, [b f c h w]
        comp = pca.components_.to(torch.float32)
        width_c = comp.shape[1]
        out = torch.empty(nfr, width_c, height, width, dtype=BF16, device=dev)
        K.pca_inverse(latents[0].contiguous(), std.reshape(-1)[:16].to(dev, torch.float32).contiguous(),
                      mean.reshape(-1)[:16].to(dev, torch.float32).contiguous(), comp[:16].to(dev).contiguous(),
                      pca.mean_.reshape(-1).to(dev, torch.float32).contiguous(), out)
        out = out[None]
        if not return_dict:
            return (out,)
        return CogVideoXPipelineOutput(frames=out)
