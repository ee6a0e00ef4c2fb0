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
from .runtime import RankGuard

BF16 = torch.bfloat16


class FifoWorker:
    """Per-GPU state that the reference ships to each spawned worker (pipe, prompt_embeds, image_rotary_emb, ...)."""

    def __init__(self, transformer, scheduler, prompt_embeds, image_rotary_emb, guidance_scale,
                 vip_grid_h=None, vip_grid_w=None, cond_grid_h=None, cond_grid_w=None, use_separate_guidance=False, guidance_scale_img=None,
                 use_dynamic_cfg=False, num_inference_steps=None, do_classifier_free_guidance=True):
        self.transformer = transformer
        self.scheduler = scheduler
        self.device = transformer.device
        self.prompt_embeds = prompt_embeds.to(self.device, BF16)
        self.image_rotary_emb = tuple(t.to(self.device, torch.float32).contiguous() for t in image_rotary_emb)
        self.guidance_scale = float(guidance_scale)
        # cogvideo_sampling_mp_fifo.py:493-533: 3-way batch (uncond_txt, uncond_img, txt_img) with its own image guidance weight, and the
        # per-frame cosine guidance schedule
        self.use_separate_guidance = bool(use_separate_guidance)
        self.guidance_scale_img = float(guidance_scale if guidance_scale_img is None else guidance_scale_img)
        self.use_dynamic_cfg = bool(use_dynamic_cfg)
        self.num_inference_steps = num_inference_steps
        # :491-498, 528: without classifier-free guidance (guidance_scale <= 1) the batch is the latents alone and the model output IS the prediction
        self.do_cfg = bool(do_classifier_free_guidance)
        nb = self.branches = (3 if self.use_separate_guidance else 2) if self.do_cfg else 1
        if self.prompt_embeds.shape[0] != nb:
            raise ValueError(f"prompt_embeds must hold {nb} rows ({('cond', 'uncond, cond', 'uncond, cond, cond')[nb - 1]}); got {self.prompt_embeds.shape[0]}")
        if self.use_dynamic_cfg and not num_inference_steps:
            raise ValueError("use_dynamic_cfg needs num_inference_steps")
        self.vip_grid_h, self.vip_grid_w = vip_grid_h, vip_grid_w
        self.cond_grid_h, self.cond_grid_w = cond_grid_h, cond_grid_w
        self.head_dim = transformer.config.attention_head_dim

    def ropes_for(self, grid_t, cond_grid_t):
        """cogvideo_sampling_mp_fifo.py:478-489 (`_prepare_vip_rotary_positional_embeddings` twice per window)."""
        vr = R.rope_3d(self.head_dim, np.asarray(grid_t, dtype=np.float32), self.vip_grid_h, self.vip_grid_w, device=self.device)
        cr = R.rope_3d(self.head_dim, np.asarray(cond_grid_t, dtype=np.float32), self.cond_grid_h, self.cond_grid_w, device=self.device)
        return vr, cr

    @torch.no_grad()
    def predict(self, branch, latents, t, grid_t=None, cond_grid_t=None, image_embeddings=None, **unused):
        """The DiT forward of one window (:491-518): branch None = the whole guidance batch [nb, nf, C, H, W] (uncond, cond | uncond_txt, uncond_img, txt_img);
        branch h = row h of it alone, as a batch-1 forward [1, nf, C, H, W] — what ONE rank computes when an iteration has so few windows that the ranks split
        them by guidance branch (cogvideo_fifo_mp_v2, round 6).  The kernels are per-sample: row h of the batched forward is the batch-1 forward of row h."""
        nf = latents.shape[1]
        use_vip = image_embeddings is not None
        vr = cr = None
        if use_vip:
            vr, cr = self.ropes_for(grid_t, cond_grid_t)
        x = latents.to(self.device, BF16)
        nb = self.branches
        if use_vip and image_embeddings.shape[0] != nb:
            raise ValueError(f"image_embeddings must hold {nb} batch rows for this guidance mode; got {image_embeddings.shape[0]}")
        rows = slice(0, nb) if branch is None else slice(int(branch), int(branch) + 1)
        n = rows.stop - rows.start
        inp = torch.cat([x] * n, dim=0)                                       # :492-497 (CFG batch: uncond, cond | uncond_txt, uncond_img, txt_img)
        tt = torch.as_tensor(np.asarray(t, dtype=np.int64), device=self.device)[None].expand(n, -1)
        return self.transformer(hidden_states=inp, encoder_hidden_states=self.prompt_embeds[rows], timestep=tt,
                                image_rotary_emb=self.image_rotary_emb, vip_image_rotary_emb=vr,
                                vip_condition_rotary_emb=cr, vip_encoder_hidden_states=image_embeddings[rows].contiguous() if use_vip else None,
                                return_dict=False)[0]

    @torch.no_grad()
    def finish(self, pred, latents, old_x0, has_old, t, prev_t, next_t, noise, **unused):
        """CFG combine + the window's 13 per-frame DPM updates (:519-550) on the model output `pred` [nb, nf, C, H, W] -> (latents_out [1, nf, C, H, W], x0 [nf, C, H, W])."""
        nf = latents.shape[1]
        x = latents.to(self.device, BF16)
        t_back = [int(v) if int(v) > 0 else None for v in next_t]
        # the 2M branch needs a previous x0 AND a back step; the FIFO driver guarantees they coincide (App. C)
        for j in range(nf):
            if t_back[j] is None and has_old[j] and int(prev_t[j]) >= 0:
                raise IndexError("frame without timestep_back must not carry an old x0 (scheduling_dpm_cogvideox.py:459)")
        gpf = None
        if self.use_dynamic_cfg and self.do_cfg:            # (without CFG the reference still evaluates the schedule, :519-527, and never uses it)
            # :519-527, evaluated like the reference: fp32 tensor arithmetic on the window's integer timesteps (the expression uses the timestep
            # VALUE against num_inference_steps, as the reference does); one small H2D copy per window
            tv = torch.as_tensor(np.asarray(t, dtype=np.int64))
            ramp = (1 - torch.cos(math.pi * ((self.num_inference_steps - tv) / self.num_inference_steps) ** 5.0)) / 2
            gpf = torch.stack([1 + self.guidance_scale * ramp, 1 + self.guidance_scale_img * ramp], dim=1).to(torch.float32).contiguous().to(self.device)
        x_out, x0 = self.scheduler.window_step(pred.contiguous(), x[0].contiguous(), old_x0.contiguous(), noise.contiguous(),
                                               list(map(int, t)), list(map(int, prev_t)), t_back, list(has_old),
                                               self.guidance_scale, self.guidance_scale_img, gpf)
        return x_out[None], x0

    @torch.no_grad()
    def window_step(self, latents, old_x0, has_old, t, prev_t, next_t, noise, grid_t=None, cond_grid_t=None,
                    image_embeddings=None, split_branches=False):
        """latents [1,nf,C,H,W] bf16; old_x0 [nf,C,H,W] (rows without an estimate are ignored via has_old);
        t/prev_t/next_t: length-nf integer sequences (next_t <= 0 means "no back step", :544);
        noise [nf,2,C,H,W] bf16.  Returns (latents_out [1,nf,C,H,W], x0 [nf,C,H,W]).
        split_branches: run the guidance branches as separate batch-1 forwards, one after the other (what the ranks of a split iteration do, on one GPU)."""
        kw = dict(latents=latents, t=t, grid_t=grid_t, cond_grid_t=cond_grid_t, image_embeddings=image_embeddings)
        pred = torch.cat([self.predict(h, **kw) for h in range(self.branches)], dim=0) if split_branches else self.predict(None, **kw)
        return self.finish(pred, latents, old_x0, has_old, t, prev_t, next_t, noise)


# --------------------------------------------------------------------------------------------------
# driver
# --------------------------------------------------------------------------------------------------
def window_plan(queue_start, nf=13, num_partitions=4):
    """Windows of one iteration (cogvideo_sampling_mp_fifo.py:235-253): list of dict(rank,start,mid,end,real_end)."""
    r, l = nf // 2, nf - nf // 2
    plan = []
    for rank in range(2 * num_partitions):
        start = nf * (rank // 2) + r * (rank % 2)
        following = nf * ((rank + 1) // 2) + r * ((rank + 1) % 2)
        if following <= queue_start:        # adaptive padding: window still entirely inside the padding
            continue
        mid = start + (l if rank % 2 == 1 else r)
        real_end = start + nf
        start = max(start, queue_start)
        plan.append(dict(rank=rank, start=start, mid=mid, end=start + nf, real_end=real_end))
    return plan


def keep_slice(w, queue_start, nf=13):
    """(queue_lo, queue_hi, local_lo) of the frames a window writes back (:322-329)."""
    r = nf // 2
    if w["start"] > queue_start:
        return w["mid"], w["end"], w["mid"] - w["start"]
    if w["start"] == queue_start:
        lo = max(r, w["start"])
        return lo, w["real_end"], max(r - w["start"], 0)
    raise NotImplementedError


def _seed_for(seed, i, tag):
    return (int(seed) * 1000003 + int(i) * 8191 + int(tag) * 131 + 12345) % (2 ** 63 - 1)


class _SeededNoise:
    """Noise keyed by (seed, iteration, window) — identical on every rank and for every GPU count, unlike the
    reference's per-process global RNG (:452,460 of the scheduler; SURVEY §7 'stochastic scheduler')."""

    def __init__(self, seed, device):
        self.seed, self.device = seed, device

    def step(self, i, rank, shape):
        g = torch.Generator(device=self.device).manual_seed(_seed_for(self.seed, i, rank))
        return torch.randn(shape, generator=g, device=self.device, dtype=torch.float32).to(BF16)

    def tail(self, i, shape):
        g = torch.Generator(device=self.device).manual_seed(_seed_for(self.seed, i, 97))
        return torch.randn(shape, generator=g, device=self.device, dtype=torch.float32).to(BF16)


@torch.no_grad()
def decode_chunks_sharded(pipe, latents, nf, decode_chunk_fn=None):
    """VAE decode of a [1, chunks*nf, C, h, w] latent video, one `vae.decode` per nf-frame chunk like the reference's
    `decode_latents` (pipeline_cogvideox_mp_fifo.py:676-684) — but the chunks, which are independent, are dealt round-robin to the
    ranks and ONE all_gather brings every chunk's frames to every rank (the reference decodes the whole video on GPU 0 while the
    others idle, cogvideo_sampling_mp_fifo.py:373-376).  decode_chunk_fn(z [1,nf,C,h,w]) -> frames [1,3,F,H,W]; default: the
    pipeline's own decode_latents."""
    import torch.distributed as dist
    fn = decode_chunk_fn or (lambda z: pipe.decode_latents(z, nf_per_chunk=nf))
    chunks = latents.shape[1] // nf
    dist_on = dist.is_available() and dist.is_initialized()       # a 1-rank group runs the same collectives (tests/test_rccl_gpu.py)
    world = dist.get_world_size() if dist_on else 1
    me = dist.get_rank() if dist_on else 0
    mine = [c for c in range(chunks) if c % world == me]
    dec = [fn(latents[:, c * nf:(c + 1) * nf].contiguous()) for c in mine]
    if not dist_on:
        return torch.cat(dec, dim=2)
    meta = [None] * world
    dist.all_gather_object(meta, (tuple(dec[0].shape), dec[0].dtype) if dec else None)      # a rank may hold no chunk
    shape, dtype = next(m for m in meta if m is not None)
    per = (chunks + world - 1) // world
    buf = torch.zeros((per,) + shape, dtype=dtype, device=latents.device)
    for slot, d in enumerate(dec):
        buf[slot] = d
    allbuf = torch.empty((world * per,) + shape, dtype=dtype, device=latents.device)       # rank-major concat
    dist.all_gather_into_tensor(allbuf, buf)
    return torch.cat([allbuf[(c % world) * per + c // world] for c in range(chunks)], dim=2)


def cogvideo_fifo_mp_v2(pipe_list, base_output, noise_seed=0, step_noise_fn=None, tail_noise_fn=None, trace=None,
                        window_fn=None, decode_chunk_fn=None, iteration_hook=None, predict_fn=None, finish_fn=None, split_small_iterations=True, **kwargs):
    """Mirror of cogvideo_sampling_mp_fifo.py:27-395.

    `pipe_list` holds this process's pipeline(s); with torch.distributed initialised (one process per GPU) the
    windows of every iteration are split rank-round-robin and the results all_gathered; the queue, the index
    bookkeeping and the tail noise are replicated.  Returns (orig_video, video, cache_video) like the reference
    (latents when output_type == "latent").  `window_fn(worker, **window_inputs)` lets tests substitute the
    denoiser (CPU/gloo tests of the exchange logic); the default is FifoWorker.window_step (HIP).  With an output_type other
    than "latent" the final VAE decode is sharded by chunk over the ranks (decode_chunks_sharded).  `iteration_hook(i, n_iter, latents, x0q)` is called
    on every rank after iteration i's queue shift (measurement only: bench.py records memory and queue checksums there; it must not modify the queue).
    Round 6 — iterations with FEW windows (the ramp at the head of every run: 1, 2, 3, ... windows while the queue fills; :235-253 runs them on as many workers and
    leaves the others idle): when the ranks number at least (guidance branches) x (windows), every window is split by guidance branch — rank nb k + h runs branch h of
    window k as a batch-1 forward (`predict_fn(h, **window_inputs)`, default FifoWorker.predict), ONE all_gather brings every branch's model output to every rank, and every
    rank applies the identical CFG + solver step to every window (`finish_fn(preds, **window_inputs)`, default FifoWorker.finish: one small elementwise launch per window).  Such an
    iteration costs a batch-1 forward (0.51 of a window step) instead of a whole one: on 8 GPUs the 20 iterations with <= 4 windows.  Needs predict_fn AND finish_fn when
    window_fn is substituted (else such iterations run whole windows as before); split_small_iterations=False turns it off."""
    import torch.distributed as dist
    bo = base_output
    sp = bo.sampling_params
    if sp.get("use_sliding_window_embedding"):
        # (the reference's multi-process driver cannot run this branch either: it calls prepare_fifo_cond_frames() / shift_cond_frames(), which exist only in the
        #  single-process fifo_sampling/sampling.py:76,109 — cogvideo_sampling_mp_fifo.py:150,347 raise NameError as soon as the flag is set)
        raise NotImplementedError("use_sliding_window_embedding: no shipped config sets it, and the reference's cogvideo_fifo_mp_v2 itself raises NameError on it "
                                  "(cogvideo_sampling_mp_fifo.py:150: prepare_fifo_cond_frames is not defined in that module)")
    if len(getattr(bo, "cache_idx", []) or []):
        # (the reference itself cannot run this branch: its merge loop indexes a list with a list — `for cid in cache_latents: cache_latents[cid] = ...`,
        #  cogvideo_sampling_mp_fifo.py:330-331 — and raises TypeError as soon as cache_idx is non-empty)
        raise NotImplementedError("cache_idx capture is a debugging feature the reference itself cannot complete (TypeError at "
                                  "cogvideo_sampling_mp_fifo.py:330-331 for any non-empty cache_idx); it is not mirrored")
    num_partitions = sp.get("num_partitions", 4)
    adaptive = sp.get("use_adaptive_padding", True)
    pipe = pipe_list[0]
    dev = pipe.device
    nf, vnf, T = bo.nf_per_chunk, bo.vip_nf_per_chunk, bo.num_inference_steps
    r, l = nf // 2, nf - nf // 2
    dist_on = dist.is_available() and dist.is_initialized()       # with a process group — of ANY size — the exchange below runs (1 rank: RCCL to itself)
    world = dist.get_world_size() if dist_on else 1
    me = dist.get_rank() if dist_on else 0
    use_vip = bo.image_embeddings is not None

    lat = bo.fifo_latents.to(dev, BF16)
    latents = torch.cat([lat[:, [0]]] * r + [lat], dim=1).contiguous()          # :72-82
    Q = latents.shape[1]
    C, H, W = latents.shape[2:]
    x0q = torch.zeros(Q, C, H, W, dtype=BF16, device=dev)
    has_old = [False] * Q
    fo = [bo.fifo_old_pred_original_sample[0]] * r + list(bo.fifo_old_pred_original_sample)   # :145-146
    for q, t_ in enumerate(fo):
        if t_ is not None:
            x0q[q] = t_.to(dev, BF16).reshape(C, H, W)
            has_old[q] = True

    timesteps = np.asarray(bo.timesteps.cpu() if torch.is_tensor(bo.timesteps) else bo.timesteps, dtype=np.int64)
    t_tab = np.concatenate([timesteps, np.full(r, timesteps[-1])])[::-1].copy()                       # :182-185, flipped
    p_tab = np.concatenate([timesteps[1:], np.full(r + 1, -1)])[::-1].copy()
    n_tab = np.concatenate([[-1], timesteps[:-1], np.full(r, timesteps[-2])])[::-1].copy()

    worker = None
    if use_vip:
        g_t, g_h, g_w = bo.vip_image_rotary_grid
        c_t, c_h, c_w = bo.vip_condition_rotary_grid
        g_t = np.asarray(g_t, dtype=np.float32)
        q_grid_t = np.concatenate([g_t[[0]]] * (r + T - nf) + [g_t[:nf]])                           # :84-93
        feed = np.concatenate([g_t[nf:], np.linspace(g_t[-1] + 1, g_t[-1] + 1 + T, T, endpoint=False, dtype=np.float32)])
        cond = [np.asarray(c_t)] + [np.asarray(c_t)[-vnf:] + (k + 1) * nf for k in range(T // nf + 1)]   # :95-99
        cond_t = np.concatenate(cond)
        emb = bo.image_embeddings.to(dev, BF16)
        emb = torch.cat([emb] + [emb[:, -vnf:]] * (T // nf + 1), dim=1)                            # :101-108
        n_c = min(vnf + 1, nf)
    nb_split = 0                                        # guidance branches a window can be split into (0: windows are never split)
    do_cfg = bool(bo.do_classifier_free_guidance)
    n_branches = (3 if getattr(bo, "use_separate_guidance", False) else 2) if do_cfg else 1
    if window_fn is None:
        worker = FifoWorker(pipe.transformer, pipe.scheduler, bo.prompt_embeds, bo.image_rotary_emb, bo.guidance_scale,
                            *((g_h, g_w, c_h, c_w) if use_vip else (None,) * 4), use_separate_guidance=getattr(bo, "use_separate_guidance", False),
                            guidance_scale_img=getattr(bo, "guidance_scale_img", None), use_dynamic_cfg=getattr(bo, "use_dynamic_cfg", False),
                            num_inference_steps=T, do_classifier_free_guidance=do_cfg)
        window_fn = lambda **kw: worker.window_step(**kw)
        predict_fn = predict_fn or (lambda h, **kw: worker.predict(h, **kw)[0])
        finish_fn = finish_fn or (lambda preds, **kw: worker.finish(preds, **kw))
    if split_small_iterations and predict_fn is not None and finish_fn is not None and n_branches > 1:
        nb_split = n_branches
    noise = _SeededNoise(noise_seed, dev)
    step_noise_fn = step_noise_fn or noise.step
    tail_noise_fn = tail_noise_fn or noise.tail

    queue_start = T - l if adaptive else 0
    outs = []
    n_iter = bo.num_frames + T - nf
    for i in range(n_iter):
        plan = window_plan(queue_start, nf, num_partitions)

        def window_inputs(w, traced):
            s, e = w["start"], w["end"]
            kw = dict(latents=latents[:, s:e].clone(), old_x0=x0q[s:e].clone(), has_old=has_old[s:e], t=t_tab[s:e],
                      prev_t=p_tab[s:e], next_t=n_tab[s:e], noise=step_noise_fn(i, w["rank"], (nf, 2, C, H, W)))
            if use_vip:
                vs = int(np.searchsorted(cond_t, q_grid_t[s] + bo.video_ipadapter_start_frame_idx, side="right") - 1)   # :110-115
                kw.update(grid_t=q_grid_t[s:e].copy(), cond_grid_t=cond_t[vs:vs + n_c].copy(), image_embeddings=emb[:, vs:vs + n_c].contiguous())
                if traced and trace is not None:
                    trace.append((i, w["rank"], s, w["mid"], e, w["real_end"], vs))
            return kw
        guard = RankGuard(f"FIFO iteration {i}")
        if dist_on and nb_split and world >= nb_split * len(plan):
            # few windows, many ranks: rank nb k + h computes guidance branch h of window k (batch 1); the model outputs are exchanged, the cheap solver step is replicated
            k_mine, h_mine = divmod(me, nb_split)
            pel = nf * C * H * W
            pbuf = torch.zeros(pel + 8, dtype=BF16, device=dev)
            with guard:
                if k_mine < len(plan):
                    pbuf[:pel] = predict_fn(h_mine, **window_inputs(plan[k_mine], h_mine == 0)).reshape(-1)
            pbuf[pel:pel + 1] = guard.flag(dev)
            flat = torch.empty(world * (pel + 8), dtype=BF16, device=dev)
            dist.all_gather_into_tensor(flat, pbuf)
            allp = flat.view(world, pel + 8)
            guard.check(allp[:, pel])
            allbuf = torch.empty(len(plan), 2, nf, C, H, W, dtype=BF16, device=dev)
            for k, w in enumerate(plan):
                x_out, x0_out = finish_fn(allp[k * nb_split:(k + 1) * nb_split, :pel].reshape(nb_split, nf, C, H, W), **window_inputs(w, False))
                allbuf[k, 0], allbuf[k, 1] = x_out[0], x0_out
            src_of = lambda k: allbuf[k]
        else:
            mine = [(k, w) for k, w in enumerate(plan) if k % world == me]
            per_rank = (len(plan) + world - 1) // world
            n_el = per_rank * 2 * nf * C * H * W
            xbuf = torch.zeros(n_el + 8, dtype=BF16, device=dev)                   # payload + this rank's failure flag (runtime.RankGuard)
            buf = xbuf[:n_el].view(per_rank, 2, nf, C, H, W)
            with guard:
                for slot, (k, w) in enumerate(mine):
                    x_out, x0_out = window_fn(**window_inputs(w, True))
                    buf[slot, 0], buf[slot, 1] = x_out[0], x0_out
            src_of = None
        if src_of is not None:
            pass
        elif dist_on:                                   # the path's one exchange: kept windows of every rank (+ the failure flags)
            xbuf[n_el:n_el + 1] = guard.flag(dev)
            flat = torch.empty(world * (n_el + 8), dtype=BF16, device=dev)     # rank-major concat
            dist.all_gather_into_tensor(flat, xbuf)
            allx = flat.view(world, n_el + 8)
            guard.check(allx[:, n_el])                  # a rank that raised above makes EVERY rank raise here, this iteration
            allbuf = allx[:, :n_el].reshape(world * per_rank, 2, nf, C, H, W)
        else:
            guard.check(None)
            allbuf = buf
        new_lat, new_x0, new_has = latents.clone(), x0q.clone(), list(has_old)
        for k, w in enumerate(plan):                    # identical write-back on every rank (:308-334)
            src = src_of(k) if src_of is not None else allbuf[(k % world) * per_rank + k // world]
            lo, hi, loc = keep_slice(w, queue_start, nf)
            new_lat[0, lo:hi] = src[0, loc:loc + (hi - lo)]
            new_x0[lo:hi] = src[1, loc:loc + (hi - lo)]
            for q in range(lo, hi):
                new_has[q] = True
        latents, x0q, has_old = new_lat, new_x0, new_has
        outs.append(latents[:, [r]].clone())                                                      # :336-342
        latents[:, :-1] = latents[:, 1:].clone()
        x0q[:-1] = x0q[1:].clone()
        has_old = has_old[1:] + [False]
        latents[:, -1] = pipe.scheduler.add_noise_to_xt(latents[:, -1], tail_noise_fn(i, (1, C, H, W)), torch.tensor([999]))
        if use_vip:
            q_grid_t[:-1] = q_grid_t[1:].copy()
            q_grid_t[-1] = feed[0]
            feed = feed[1:]
        queue_start = max(0, queue_start - 1)
        if iteration_hook is not None:
            iteration_hook(i, n_iter, latents, x0q)

    video_latents = torch.cat(outs[T - nf:], dim=1)                                              # :367
    if getattr(bo, "output_type", "latent") == "latent":
        result = (bo.orig_latents, video_latents, [])
    else:
        video = decode_chunks_sharded(pipe, video_latents, nf, decode_chunk_fn)           # chunk-sharded over the ranks
        orig = decode_chunks_sharded(pipe, bo.orig_latents, nf, decode_chunk_fn)
        vp = getattr(pipe, "video_processor", None)
        if vp is not None:
            video = vp.postprocess_video(video=video, output_type=bo.output_type)
            orig = vp.postprocess_video(video=orig, output_type=bo.output_type)
        result = (orig, video, [])
    if getattr(bo, "return_dict", False):
        from types import SimpleNamespace
        return SimpleNamespace(frames=result[1], orig_frames=result[0], cache_frames=result[2])
    return result
