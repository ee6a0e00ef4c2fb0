"""MPFIFOVideoIPAdapterCogVideoXPipeline — host mirror of the parts of longvgen/pipeline/pipeline_cogvideox_mp_fifo.py that
are on the hot path: the 52-step base stage on chunk 0 that seeds the FIFO queue (:1186-1307),
`prepare_latents` (:650-674), the RoPE helpers (:769-813), `preprare_for_fifo` (sic, :1491-1514) and `decode_latents`
(:676-684).  Prompt encoding (T5) and the condensed-token encoder (Resampler) are upstream of the path: pass
`prompt_embeds` / `negative_prompt_embeds` / `image_embeddings` tensors (as the reference's own `image_embeddings is not
None` branch does, :611-616); passing raw prompts or frames raises NotImplementedError.
"""
import os
from types import SimpleNamespace

import math

import numpy as np
import torch

from . import cfg_parallel as CP
from . import kernels as K
from . import rope as R
from .fifo import BF16


class MPFIFOVideoIPAdapterCogVideoXPipeline:
    def __init__(self, transformer, scheduler, vae=None, resampler_config=None, device=None, resampler=None):
        self.transformer, self.scheduler, self.vae = transformer, scheduler, vae
        self._resampler = resampler
        self.device = torch.device(device) if device is not None else transformer.device
        self.vae_scale_factor_spatial = 2 ** (len(vae.config.block_out_channels) - 1) if vae is not None else 8
        self.vae_scale_factor_temporal = vae.config.temporal_compression_ratio if vae is not None else 4
        self.vae_scaling_factor_image = vae.config.scaling_factor if vae is not None else 1.15258426
        self.resampler = resampler if resampler is not None else SimpleNamespace(config=SimpleNamespace(**(resampler_config or dict(
            num_temporal_queries=4, num_height_queries=8, num_width_queries=12))))
        self._guidance_scale = 6.0

    @classmethod
    def from_pretrained(cls, path, transformer=None, resampler=None, torch_dtype=BF16, device="cuda", **unused):
        """DiffusionPipeline.from_pretrained as the entry script calls it (infer_cogvideo_mp_fifo.py:122-127, 171-176): the transformer and the
        resampler are handed in; the VAE comes from `<path>/vae`, the scheduler configuration from `<path>/scheduler/scheduler_config.json`.
        The T5 tokenizer / text encoder are NOT loaded: prompt encoding stays on the reference side and `prompt_embeds`
        are passed in."""
        from .scheduler import CogVideoXDPMScheduler
        from .transformer import CogVideoXTransformer3DModel
        from .vae import AutoencoderKLCogVideoX
        dev = transformer.device if transformer is not None else torch.device(device)
        if transformer is None:
            transformer = CogVideoXTransformer3DModel.from_pretrained(path, subfolder="transformer", torch_dtype=torch_dtype, device=dev)
        vae = AutoencoderKLCogVideoX.from_pretrained(path, subfolder="vae", device=dev) if os.path.isdir(os.path.join(path, "vae")) else None
        sched = CogVideoXDPMScheduler.from_config(os.path.join(path, "scheduler", "scheduler_config.json"))
        return cls(transformer, sched, vae=vae, resampler=resampler, device=dev)

    def to(self, device=None, *unused, **kw):
        """`pipe.to(device)`: the modules already live on the device they were built on; a different device moves them."""
        if device is not None and not isinstance(device, torch.dtype) and torch.device(device) != self.device:
            self.device = torch.device(device)
            self.transformer.to(self.device)
            if self.vae is not None:
                self.vae.to(self.device)
            if self._resampler is not None:
                self._resampler.to(self.device)
        return self

    @property
    def guidance_scale(self):
        return self._guidance_scale

    # ---- helpers -------------------------------------------------------------------------------------------
    def prepare_latents(self, batch_size, num_channels_latents, num_frames, height, width, generator=None, latents=None):
        """:650-674 — [B, (F-1)/4+1, C, H/8, W/8] * init_noise_sigma."""
        shape = (batch_size, (num_frames - 1) // self.vae_scale_factor_temporal + 1, num_channels_latents,
                 height // self.vae_scale_factor_spatial, width // self.vae_scale_factor_spatial)
        if latents is None:
            latents = torch.randn(shape, generator=generator, device=self.device, dtype=torch.float32).to(BF16)
        return latents.to(self.device, BF16) * self.scheduler.init_noise_sigma

    def _prepare_rotary_positional_embeddings(self, height, width, num_frames, device=None):
        """:769-795 with the crop region of `get_resize_crop_region_for_grid` (:81-96) against the 720x480 base grid: the full grid at
        720x480 (and at every size of the same 3:2 shape the positions are the base grid's, resampled)."""
        p = self.transformer.config.patch_size
        f = self.vae_scale_factor_spatial * p
        gh, gw = height // f, width // f
        tw, th = 720 // f, 480 // f
        if gh / gw > th / tw:
            rh, rw = th, int(round(th / gh * gw))
        else:
            rw, rh = tw, int(round(tw / gw * gh))
        top, left = int(round((th - rh) / 2.0)), int(round((tw - rw) / 2.0))
        return R.rope_3d_crop(self.transformer.config.attention_head_dim, (0, top, left), (num_frames, top + rh, left + rw), (num_frames, gh, gw))

    def _prepare_vip_rotary_positional_embeddings(self, grid_t, grid_h, grid_w, device=None):
        """:797-813"""
        return R.rope_3d(self.transformer.config.attention_head_dim, grid_t, grid_h, grid_w)

    def decode_latents(self, latents, nf_per_chunk=13):
        """:676-684 — [B,F,C,h,w] -> frames [B,3,T,H,W], one vae.decode per 13-latent-frame chunk."""
        z = (latents.permute(0, 2, 1, 3, 4).float() / self.vae_scaling_factor_image).to(BF16).contiguous()
        frames = [self.vae.decode(z[:, :, c * nf_per_chunk:(c + 1) * nf_per_chunk].contiguous()).sample
                  for c in range(z.shape[2] // nf_per_chunk)]
        return torch.cat(frames, dim=2)

    @torch.no_grad()
    def vae_encode_image(self, frames, nf_per_chunk=49, compressed_nf_per_chunk=13, video_ipadapter_start_frame_idx=1000, generator=None,
                         sample_posterior=True, do_classifier_free_guidance=True, use_separate_guidance=False):
        """Condensed-token encoding of the source video (pipeline_cogvideox_mp_fifo.py:562-648, `use_vae_as_encoder`):
        frames [b, F, 3, H, W] in [-1, 1] -> pad one chunk with the last frame -> per 49-frame chunk: vae.encode, sample, x scaling
        -> `transformer.patch_embed.proj` -> per chunk Resampler -> [2b, 4*(chunks+1), D, 8, 12] (the same tokens in both CFG halves,
        :646; the zero-video "uncond" tokens of the reference are only USED by `use_separate_guidance`, which returns
        [tokens, tokens of an all-zero video, tokens] = [3b, ...], :620-644)."""
        if self._resampler is None or self.vae is None:
            raise RuntimeError("vae_encode_image needs a VAE and a Resampler")
        rq = self.resampler.config
        dev = self.device
        f32 = np.float32
        img = R.rope_3d(rq.dim_head, np.linspace(0, rq.max_temporal_seq_len, rq.max_temporal_seq_len, endpoint=False, dtype=f32),
                        np.linspace(0, rq.max_height_seq_len, rq.max_height_seq_len, endpoint=False, dtype=f32),
                        np.linspace(0, rq.max_width_seq_len, rq.max_width_seq_len, endpoint=False, dtype=f32))             # :1103-1126
        smp = R.rope_3d(rq.dim_head, np.linspace(video_ipadapter_start_frame_idx, video_ipadapter_start_frame_idx + rq.max_temporal_seq_len,
                                                 rq.num_temporal_queries, endpoint=False, dtype=f32),
                        np.linspace(0, rq.max_height_seq_len, rq.num_height_queries, endpoint=False, dtype=f32),
                        np.linspace(0, rq.max_width_seq_len, rq.num_width_queries, endpoint=False, dtype=f32))             # :1127-1149

        def encode(video_bfchw):
            video = video_bfchw.to(dev, BF16).permute(0, 2, 1, 3, 4)
            video = torch.cat([video] + [video[:, :, [-1]]] * nf_per_chunk, dim=2)                      # :581 pad one chunk
            n_chunks = video.shape[2] // nf_per_chunk
            b, _, _, Hs, Ws = video.shape
            sf = self.vae_scale_factor_spatial
            zshape = (b, self.vae.config.latent_channels, compressed_nf_per_chunk, Hs // sf, Ws // sf)
            # The chunks are independent (the reference runs them one after the other on GPU 0, :585-609): chunk c goes to rank c % world, ONE all_gather brings
            # every chunk's tokens to every rank (cfg_parallel.map_chunks_sharded).  The posterior noise of ALL chunks is drawn here, in chunk order, on every rank —
            # the draw `latent_dist.sample(generator)` would make — so the generator ends in the same state and the tokens are bitwise the one-rank run's.
            noises = [torch.randn(zshape, generator=generator, device=dev, dtype=torch.float32).to(BF16) if sample_posterior else None for _ in range(n_chunks)]

            def one(c):
                post = self.vae.encode(video[:, :, c * nf_per_chunk:(c + 1) * nf_per_chunk].contiguous()).latent_dist
                z = (post.mean + post.std * noises[c]) if sample_posterior else post.mode()
                lat = (z.float() * self.vae.config.scaling_factor).to(BF16).permute(0, 2, 1, 3, 4).contiguous()     # b f c h w, one chunk
                tokens = self.transformer.patch_embed_proj(lat)                                          # b f (h w) D
                return self._resampler(tokens.contiguous(), image_rotary_emb=img, sampling_rotary_emb=smp).contiguous()
            return torch.cat(CP.map_chunks_sharded(n_chunks, one, device=dev), dim=1)
        emb = encode(frames)
        if not do_classifier_free_guidance:
            return emb
        if use_separate_guidance:
            # :620-644 — the unconditional-image branch: the same chain on an all-zero video of the same length
            unc = encode(torch.zeros_like(frames))
            if unc.shape[1] != emb.shape[1]:
                emb = torch.cat([emb] + [emb[:, [-1]]] * (unc.shape[1] - emb.shape[1]), dim=1)
            return torch.cat([emb, unc, emb], dim=0)
        return torch.cat([emb, emb], dim=0)

    def preprare_for_fifo(self, num_inference_steps=52, guidance_scale=6.0, video_ipadapter_scale=None, **unused):
        """:1491-1514 — what the non-zero ranks run instead of the base stage: set vip scale + timesteps."""
        self._guidance_scale = guidance_scale
        self._set_vip_scale(video_ipadapter_scale)
        self.scheduler.set_timesteps(num_inference_steps, device=None)

    def _set_vip_scale(self, scale):
        if scale is None:
            return
        for m in self.transformer.modules():          # :981-983 — matched by class NAME in the reference
            if m.__class__.__name__ == "VideoIPAdapterCogVideoXAttnProcessor2_0":
                m.scale = scale

    # ---- base stage --------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, prompt=None, frames=None, prompt_embeds=None, negative_prompt_embeds=None, image_embeddings=None,
                 height=480, width=720, num_frames_per_chunk=49, num_chunks=1, num_inference_steps=52, guidance_scale=6.0,
                 video_ipadapter_scale=None, video_ipadapter_start_frame_idx=1000, latents=None, generator=None, step_noise=None,
                 sampling_params=None, output_type="latent", return_dict=False, cfg_parallel=None, use_separate_guidance=False,
                 guidance_scale_img=None, use_dynamic_cfg=False, uncond_image_embeddings=None, **unused):
        """Base stage (:837-1344): `num_inference_steps` scalar-timestep CFG steps on chunk 0, harvesting
        `latents[:, max(0, 12-i)]` (and the matching x0) into the FIFO seed lists before every step (:1190-1194).
        step_noise: optional callable i -> [nf,2,C,h,w] bf16 (default: seeded device generator).
        cfg_parallel: see tokensgen_amd/cfg_parallel.py (default: split the two CFG halves over ranks 0/1 when >= 2 ranks run).
        use_separate_guidance / guidance_scale_img (:1026-1029, 1197-1200, 1261-1263): 3-way batch (negative prompt + image tokens | prompt +
        zero-video tokens | prompt + image tokens); image_embeddings then has 3 batch rows (vae_encode_image(use_separate_guidance=True)), or
        1 row plus `uncond_image_embeddings` [1, 4*(num_chunks+1), C, h, w].  use_dynamic_cfg (:1252-1259): the cosine guidance ramp, a
        Python float per step."""
        if prompt is not None or frames is not None:
            raise NotImplementedError("T5 prompt encoding and the Resampler are upstream of the hot path: pass prompt_embeds / image_embeddings")
        do_cfg = guidance_scale > 1.0                                                               # :1012
        if prompt_embeds is None or (do_cfg and negative_prompt_embeds is None):
            raise ValueError("prompt_embeds (and, with guidance_scale > 1, negative_prompt_embeds) are required")
        dev = self.device
        self._guidance_scale = guidance_scale
        self._set_vip_scale(video_ipadapter_scale)
        use_vip = image_embeddings is not None
        nb = (3 if use_separate_guidance else 2) if do_cfg else 1       # without guidance: the prompt alone, the model output is the prediction (:1196-1200, 1260)
        g_img = guidance_scale if guidance_scale_img is None else guidance_scale_img                 # infer_cogvideo_mp_fifo.py:313
        embeds = (torch.cat([negative_prompt_embeds] + [prompt_embeds] * (nb - 1), dim=0) if do_cfg else prompt_embeds).to(dev, BF16)   # :1026-1029
        self.scheduler.set_timesteps(num_inference_steps, device=None)
        ts = self.scheduler.timesteps.tolist()
        nf = (num_frames_per_chunk - 1) // self.vae_scale_factor_temporal + 1
        C = self.transformer.config.in_channels
        latents = self.prepare_latents(1, C, num_frames_per_chunk, height, width, generator, latents)
        h, w = latents.shape[-2:]
        p = self.transformer.config.patch_size
        rope = self._prepare_rotary_positional_embeddings(height, width, nf)
        f32 = np.float32
        vr = cr = emb0 = None
        grids = None
        if use_vip:
            rq = self.resampler.config
            gh, gw = np.linspace(0, h // p, h // p, endpoint=False, dtype=f32), np.linspace(0, w // p, w // p, endpoint=False, dtype=f32)
            gt = np.linspace(0, num_chunks * nf, num_chunks * nf, endpoint=False, dtype=f32)
            vnf = rq.num_temporal_queries
            ch = np.linspace(0, h // p, rq.num_height_queries, endpoint=False, dtype=f32)
            cw = np.linspace(0, w // p, rq.num_width_queries, endpoint=False, dtype=f32)
            ct = np.concatenate([np.linspace(video_ipadapter_start_frame_idx + i * nf, video_ipadapter_start_frame_idx + (i + 1) * nf, vnf,
                                             endpoint=False, dtype=f32) for i in range(num_chunks + 1)])
            grids = ([gt, gh, gw], [ct, ch, cw])
            n_c = min(vnf + 1, nf)
            vr = R.rope_3d(64, gt[:nf], gh, gw, device=dev)
            cr = R.rope_3d(64, ct[:n_c], ch, cw, device=dev)
            image_embeddings = image_embeddings.to(dev, BF16)
            if image_embeddings.shape[0] == 1:
                # tokens straight from the T2To stage / vae_encode_image(do_classifier_free_guidance=False), as gen.yaml passes
                # them (infer:262-300): pad one chunk's worth of tokens with the last group and repeat for the two CFG halves
                # (pipeline_cogvideox_mp_fifo.py:611-646; the zero-video "uncond" tokens are computed there but not used)
                per_chunk = image_embeddings.shape[1] // num_chunks
                image_embeddings = torch.cat([image_embeddings] + [image_embeddings[:, [-1]]] * per_chunk, dim=1)
                if not do_cfg:
                    pass                                                                             # :618: no CFG rows are added
                elif use_separate_guidance:
                    if uncond_image_embeddings is None:
                        raise ValueError("use_separate_guidance with single-row image_embeddings needs uncond_image_embeddings (the tokens of an "
                                         "all-zero video: vae_encode_image(zeros, do_classifier_free_guidance=False))")
                    unc = uncond_image_embeddings.to(dev, BF16)
                    if unc.shape[1] != image_embeddings.shape[1]:                                    # :636-640
                        image_embeddings = torch.cat([image_embeddings] + [image_embeddings[:, [-1]]] * (unc.shape[1] - image_embeddings.shape[1]), dim=1)
                    image_embeddings = torch.cat([image_embeddings, unc, image_embeddings], dim=0)
                else:
                    image_embeddings = torch.cat([image_embeddings, image_embeddings], dim=0)
            if image_embeddings.shape[0] != nb or image_embeddings.shape[1] < n_c:
                raise ValueError(f"image_embeddings must be [1, 4*num_chunks, C, h, w] (reference input) or the prepared [{nb}, 4*(num_chunks+1), "
                                 f"C, h, w]; got {tuple(image_embeddings.shape)}")
            emb0 = image_embeddings[:, :n_c].contiguous()
        rope_d = tuple(t.to(dev) for t in rope)
        gen = generator if generator is not None else torch.Generator(device=dev).manual_seed(0)
        fifo_latents, fifo_old = [], []
        old_x0 = None
        cfg_mode = CP.resolve(cfg_parallel)
        for i, t in enumerate(ts):
            k = max(0, nf - 1 - i)
            fifo_latents.insert(0, latents[:, [k]].clone())
            fifo_old.insert(0, None if old_x0 is None else old_x0[[k]].to(BF16)[None])    # the queue holds model-dtype x0
            def fwd(lo, hi):
                n = hi - lo
                return self.transformer(hidden_states=torch.cat([latents] * n, dim=0), encoder_hidden_states=embeds[lo:hi],
                                        timestep=torch.full((n,), t, dtype=torch.int64, device=dev), image_rotary_emb=rope_d,
                                        vip_image_rotary_emb=vr, vip_condition_rotary_emb=cr,
                                        vip_encoder_hidden_states=None if emb0 is None else emb0[lo:hi].contiguous(), return_dict=False)[0]
            pred = CP.predict(cfg_mode, lambda h: fwd(h, h + 1), lambda: fwd(0, nb), n=nb)
            g_txt = guidance_scale
            if use_dynamic_cfg:                                  # :1252-1259 — Python floats, the timestep VALUE against num_inference_steps as written there
                ramp = (1 - math.cos(math.pi * ((num_inference_steps - t) / num_inference_steps) ** 5.0)) / 2
                # the text weight is recomputed from the argument (it lives in self._guidance_scale, :1254); the image weight is the reference's
                # LOCAL variable re-assigned in place (:1257): it compounds from step to step, and the compounded value is what is exported (:1336)
                g_txt, g_img = 1 + guidance_scale * ramp, 1 + g_img * ramp
            prev_t = ts[i + 1] if i + 1 < len(ts) else -1
            t_back = ts[i - 1] if i > 0 else None
            nz = step_noise(i) if step_noise is not None else torch.randn((nf, 2) + tuple(latents.shape[2:]), generator=gen, device=dev,
                                                                          dtype=torch.float32)
            has = old_x0 is not None
            second = has and prev_t >= 0
            # the pipeline loop keeps the solver state in fp32 (`noise_pred.float()`, :1236-1276): f32 variant of the fused step
            coef = self.scheduler.coef_table([t] * nf, [prev_t] * nf, [t_back] * nf, [second] * nf, dev)
            x = torch.empty_like(latents[0])
            x0 = torch.empty(latents.shape[1:], dtype=torch.float32, device=dev)
            K.cfg_dpm_step_ex(pred.reshape(nb, nf, -1), latents[0].reshape(nf, -1),
                              (old_x0 if has else torch.zeros_like(x0)).reshape(nf, -1), nz.to(dev, BF16).contiguous().reshape(nf, 2, -1),
                              coef, g_txt, x.view(nf, -1), x0.view(nf, -1), guidance_img=g_img, prediction_type=self.scheduler.config.prediction_type)
            latents, old_x0 = x[None], x0
        return SimpleNamespace(
            fifo_latents=torch.cat(fifo_latents, dim=1), fifo_old_pred_original_sample=fifo_old, orig_latents=latents.clone(),
            nf_per_chunk=nf, vip_nf_per_chunk=self.resampler.config.num_temporal_queries if use_vip else None,
            num_frames=num_chunks * nf, image_embeddings=image_embeddings, timesteps=self.scheduler.timesteps,
            num_inference_steps=num_inference_steps, do_classifier_free_guidance=bool(do_cfg), use_separate_guidance=bool(use_separate_guidance),
            use_dynamic_cfg=bool(use_dynamic_cfg),
            prompt_embeds=embeds, image_rotary_emb=rope, vip_image_rotary_grid=grids[0] if use_vip else None,
            vip_condition_rotary_grid=grids[1] if use_vip else None, attention_kwargs=None, guidance_scale=guidance_scale,
            guidance_scale_img=g_img, extra_step_kwargs={}, cache_idx=[], condition_frames=None,
            video_ipadapter_start_frame_idx=video_ipadapter_start_frame_idx, sampling_params=sampling_params or dict(use_adaptive_padding=True, num_partitions=4),
            output_type=output_type, return_dict=return_dict)
