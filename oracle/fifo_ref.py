"""Oracle: FIFO diagonal-denoising queue (host algorithm) — TEST INFRASTRUCTURE ONLY.

Restates longvgen/fifo_sampling/cogvideo_sampling_mp_fifo.py:27-395 (driver) and :408-579 (worker
body) as single-process functions with the denoiser and the noise source injected.
"""
import numpy as np
import torch

from . import scheduler_ref as S


def window_plan(queue_start, nf=13, num_partitions=4):
    """Window geometry for one FIFO iteration, cogvideo_sampling_mp_fifo.py:235-253.

    Returns list of dicts(rank,start,mid,end,real_end) for the windows that run."""
    r = nf // 2
    l = nf - r
    out = []
    for rank in range(2 * num_partitions):
        start = nf * (rank // 2) + r * (rank % 2)
        nxt = nf * ((rank + 1) // 2) + r * ((rank + 1) % 2)
        if nxt <= queue_start:
            continue
        mid = start + (l if rank % 2 == 1 else r)
        real_end = start + nf
        start = max(start, queue_start)
        out.append(dict(rank=rank, start=start, mid=mid, end=start + nf, real_end=real_end))
    return out


def keep_slice(w, queue_start, nf=13):
    """Which queue positions a finished window writes back, :322-329. Returns (q_lo, q_hi, local_lo)."""
    r = nf // 2
    if w["start"] > queue_start:
        return w["mid"], w["end"], w["mid"] - w["start"]
    lo = max(r, w["start"])
    return lo, w["real_end"], max(r - w["start"], 0)


def timestep_tables(timesteps, nf=13):
    """Per-queue-position (t, prev_t, next_t), already flipped to ascending-noise order, :182-185,255-257."""
    r = nf // 2
    ts = np.asarray(timesteps, dtype=np.int64)
    t = np.concatenate([ts, np.full(r, ts[-1])])[::-1].copy()
    prev_t = np.concatenate([ts[1:], np.full(r + 1, -1)])[::-1].copy()
    next_t = np.concatenate([[-1], ts[:-1], np.full(r, ts[-2])])[::-1].copy()
    return t, prev_t, next_t


def grid_t_tables(grid_t, cond_t, T, nf=13, vip_nf=4):
    """vip RoPE time bookkeeping, :84-99. Returns (queue_grid_t[T+r], grid_t_feed, cond_t_extended)."""
    r = nf // 2
    grid_t = np.asarray(grid_t, dtype=np.float32)
    init = np.concatenate([grid_t[[0]]] * (r + T - nf) + [grid_t[:nf]])
    feed = np.concatenate([grid_t[nf:], np.linspace(grid_t[-1] + 1, grid_t[-1] + 1 + T, T, endpoint=False,
                                                   dtype=np.float32)])
    cond = [np.asarray(cond_t)]
    for i in range(T // nf + 1):
        cond.append(np.asarray(cond_t)[-vip_nf:] + (i + 1) * nf)
    return init, feed, np.concatenate(cond)


def find_embed_index(cond_t, pos, start_frame_idx):
    """:110-115."""
    return int(np.searchsorted(cond_t, pos + start_frame_idx, side="right") - 1)


def window_step(denoise, ac, guidance_scale, latents, old_x0, t, prev_t, next_t, draw, out_dtype, use_separate_guidance=False,
                guidance_scale_img=None, use_dynamic_cfg=False, num_inference_steps=None, prediction_type="v_prediction",
                do_classifier_free_guidance=True):
    """Worker body :491-550: CFG-batched denoiser call, CFG combine, 13 per-frame DPM steps.  do_classifier_free_guidance=False (:497-498, 528: the
    batch is the latents alone, the model output is the prediction; a dynamic-cfg schedule is still evaluated by the reference and never used).

    denoise(latent_in [nb,nf,C,H,W], timesteps [nb,nf]) -> [nb,nf,C,H,W] with nb = 3 for `use_separate_guidance` (uncond_txt, uncond_img,
    txt_img; :493-497, 528-530) else 2; `use_dynamic_cfg`: per-frame cosine guidance as an fp32 tensor (:519-527), which promotes the guided
    prediction to fp32;  draw() -> next gaussian [1,1,C,H,W] (see scheduler_ref.dpm_step).  Returns (latents_out, x0 list)."""
    nf = latents.shape[1]
    nb = (3 if use_separate_guidance else 2) if do_classifier_free_guidance else 1
    inp = torch.cat([latents] * nb)
    tv = torch.as_tensor(t)
    tt = tv[None].expand(nb, -1)
    g, gi = guidance_scale, (guidance_scale if guidance_scale_img is None else guidance_scale_img)
    if use_dynamic_cfg:
        g, gi = S.dynamic_guidance(g, tv, num_inference_steps), S.dynamic_guidance(gi, tv, num_inference_steps)
    pred = denoise(inp, tt)
    if do_classifier_free_guidance:
        pred = S.cfg_combine_separate(pred, g, gi) if use_separate_guidance else S.cfg_combine(pred, g)
    out = latents.clone()
    x0s = []
    for j in range(nf):
        nxt = int(next_t[j]) if next_t[j] > 0 else None
        x, x0 = S.dpm_step(ac, pred[:, [j]], old_x0[j], int(t[j]), int(prev_t[j]), nxt, latents[:, [j]], draw, prediction_type=prediction_type)
        out[:, [j]] = x.to(out_dtype)
        x0s.append(x0.to(out_dtype))
    return out, x0s


def run_fifo(denoise_window, betas, ac, fifo_latents, fifo_old_x0, timesteps, num_frames, guidance_scale,
             grid_t, cond_t, start_frame_idx, step_noise, tail_noise, nf=13, vip_nf=4, num_partitions=4,
             trace=None):
    """Driver :27-367 (latent output).  denoise_window(latents[2,nf,...], timesteps[2,nf], grid_t_window,
    cond_t_window, vip_start) -> [2,nf,...]; step_noise() -> next worker-side gaussian [1,1,C,H,W];
    tail_noise() -> next driver-side gaussian [1,C,H,W].  Returns latents [1, num_frames, C,H,W] (first T-nf outputs discarded, :367)."""
    T = len(timesteps)
    r = nf // 2
    l = nf - r
    latents = torch.cat([fifo_latents[:, [0]]] * r + [fifo_latents], dim=1)
    old_x0 = [fifo_old_x0[0]] * r + list(fifo_old_x0)
    q_grid_t, feed, cond_ext = grid_t_tables(grid_t, cond_t, T, nf, vip_nf)
    t_tab, p_tab, n_tab = timestep_tables(timesteps, nf)
    queue_start = T - l
    outs = []
    for i in range(num_frames + T - nf):
        new_lat = latents.clone()
        new_x0 = list(old_x0)
        for w in window_plan(queue_start, nf, num_partitions):
            s, e = w["start"], w["end"]
            vs = find_embed_index(cond_ext, q_grid_t[s], start_frame_idx)
            n_c = min(vip_nf + 1, nf)
            dn = lambda x, tt, _s=s, _e=e, _vs=vs: denoise_window(x, tt, q_grid_t[_s:_e].copy(),
                                                                   cond_ext[_vs:_vs + n_c].copy(), _vs)
            o_lat, o_x0 = window_step(dn, ac, guidance_scale, latents[:, s:e].clone(), old_x0[s:e],
                                      t_tab[s:e], p_tab[s:e], n_tab[s:e], step_noise,
                                      latents.dtype)
            lo, hi, loc = keep_slice(w, queue_start, nf)
            new_lat[:, lo:hi] = o_lat[:, loc:loc + (hi - lo)]
            new_x0[lo:hi] = o_x0[loc:loc + (hi - lo)]
            if trace is not None:
                trace.append((i, w["rank"], s, w["mid"], e, w["real_end"], vs, lo, hi, loc))
        latents, old_x0 = new_lat, new_x0
        outs.append(latents[:, [r]].clone())
        latents[:, :-1] = latents[:, 1:].clone()
        old_x0 = old_x0[1:] + [None]
        latents[:, -1] = S.add_noise_to_xt(betas, latents[:, -1], tail_noise()).to(latents.dtype)
        q_grid_t[:-1] = q_grid_t[1:].copy()
        q_grid_t[-1] = feed[0]
        feed = feed[1:]
        queue_start = max(0, queue_start - 1)
    return torch.cat(outs[T - nf:], dim=1)


def run_fifo_prenoise(denoise_window, betas, ac, fifo_latents, fifo_old_x0, timesteps, num_frames, guidance_scale,
                      grid_t, cond_t, start_frame_idx, noise_fn, nf=13, vip_nf=4, num_partitions=4, trace=None):
    """Same algorithm as run_fifo, with the gaussian noise PRE-DRAWN per window: noise_fn(i, rank, (nf,2,C,H,W)) gives both
    candidate draws of every frame (index 1 is used on the 2M branch, exactly the draw the reference keeps, :460-461) and
    noise_fn(i, 97, (1,C,H,W)) the tail noise.  This is the keyed-noise form the product uses, so both can share noise."""
    T = len(timesteps)
    r, l = nf // 2, nf - nf // 2
    dt = fifo_latents.dtype
    latents = torch.cat([fifo_latents[:, [0]]] * r + [fifo_latents], dim=1)
    old = [fifo_old_x0[0]] * r + list(fifo_old_x0)
    q_grid_t, feed, cond_ext = grid_t_tables(grid_t, cond_t, T, nf, vip_nf)
    t_tab, p_tab, n_tab = timestep_tables(timesteps, nf)
    C, H, W = latents.shape[2:]
    qs = T - l
    outs = []
    for i in range(num_frames + T - nf):
        nl, no = latents.clone(), list(old)
        for w in window_plan(qs, nf, num_partitions):
            s, e = w["start"], w["end"]
            vs = find_embed_index(cond_ext, q_grid_t[s], start_frame_idx)
            nz = noise_fn(i, w["rank"], (nf, 2, C, H, W))
            x_in = latents[:, s:e].clone()
            tt = torch.as_tensor(t_tab[s:e])[None].expand(2, -1)
            pred = S.cfg_combine(denoise_window(torch.cat([x_in] * 2), tt, q_grid_t[s:e].copy(),
                                                cond_ext[vs:vs + min(vip_nf + 1, nf)].copy(), vs), guidance_scale)
            o_lat, o_x0 = x_in.clone(), []
            for j in range(nf):
                nxt = int(n_tab[s + j]) if n_tab[s + j] > 0 else None
                seq = iter([nz[j, 0][None, None], nz[j, 1][None, None]])
                x, x0 = S.dpm_step(ac, pred[:, [j]].float(), None if old[s + j] is None else old[s + j].float(),
                                   int(t_tab[s + j]), int(p_tab[s + j]), nxt, x_in[:, [j]].float(), lambda: next(seq).float())
                o_lat[:, [j]] = x.to(dt)
                o_x0.append(x0.to(dt))
            lo, hi, loc = keep_slice(w, qs, nf)
            nl[:, lo:hi] = o_lat[:, loc:loc + (hi - lo)]
            no[lo:hi] = o_x0[loc:loc + (hi - lo)]
            if trace is not None:
                trace.append((i, w["rank"], s, w["mid"], e, w["real_end"], vs))
        latents, old = nl, no
        outs.append(latents[:, [r]].clone())
        latents[:, :-1] = latents[:, 1:].clone()
        old = old[1:] + [None]
        latents[:, -1] = S.add_noise_to_xt(betas, latents[:, -1], noise_fn(i, 97, (1, C, H, W))).to(dt)
        q_grid_t[:-1] = q_grid_t[1:].copy()
        q_grid_t[-1] = feed[0]
        feed = feed[1:]
        qs = max(0, qs - 1)
    return torch.cat(outs[T - nf:], dim=1)


def base_stage(denoise, ac, latents, timesteps, guidance_scale, noise_fn, nf=13, use_separate_guidance=False, guidance_scale_img=None,
               use_dynamic_cfg=False, export=None, do_classifier_free_guidance=True):
    """Base stage of the pipeline, pipeline_cogvideox_mp_fifo.py:1186-1307 (chunk 0, scalar timestep, CFG in fp32,
    whole-chunk scheduler step, latents cast back to the model dtype).  denoise(x[nb,nf,...], t[nb]) -> [nb,nf,...] (nb = 3 with
    use_separate_guidance, :1197-1200); noise_fn(i) -> [nf,2,C,H,W] (draw 1 of every frame is the one the 2M branch keeps).  Returns
    (fifo_latents [1,T,C,H,W], fifo_old list, final latents).  The default branch is pinned bit-exact to a run of the reference pipeline
    (tests/golden/base_stage_tiny.pt); the separate-guidance + dynamic-cfg branch (:1252-1263) to a second run (base_stage_dyn_sep.pt): there the
    reference re-assigns its LOCAL `guidance_scale_img = 1 + guidance_scale_img * ramp` every step (:1257), so the image weight COMPOUNDS from
    step to step, while the text weight is recomputed from the unchanged argument (:1254, it goes to self._guidance_scale); `export` (a dict,
    optional) receives what the stage hands to the FIFO driver: guidance_scale (the argument, :1335) and guidance_scale_img (the compounded
    local, :1336)."""
    import math
    nb = (3 if use_separate_guidance else 2) if do_classifier_free_guidance else 1        # :1012 (guidance_scale > 1), :1196-1200
    gi = guidance_scale if guidance_scale_img is None else guidance_scale_img      # :1026-1029
    T = len(timesteps)
    dt = latents.dtype
    fifo_lat, fifo_old, old = [], [], None
    ts = [int(t) for t in timesteps]
    for i, t in enumerate(ts):
        k = max(0, nf - 1 - i)
        fifo_lat.insert(0, latents[:, [k]])
        fifo_old.insert(0, None if old is None else old[:, [k]])
        pred = denoise(torch.cat([latents] * nb), torch.tensor([t] * nb)).float()
        g = guidance_scale
        if use_dynamic_cfg:                                                     # :1252-1259
            g = 1 + guidance_scale * ((1 - math.cos(math.pi * ((T - t) / T) ** 5.0)) / 2)
            gi = 1 + gi * ((1 - math.cos(math.pi * ((T - t) / T) ** 5.0)) / 2)      # (the local is overwritten: compounds)
        if not do_classifier_free_guidance:                                     # :1260: the model output is the prediction
            pass
        elif use_separate_guidance:                                             # :1261-1263
            ut, ui, c = pred.chunk(3)
            pred = c + (g - 1) * (c - ut) + (gi - 1) * (c - ui)
        else:
            u, c = pred.chunk(2)
            pred = u + g * (c - u)
        prev_t = ts[i + 1] if i + 1 < len(ts) else -1
        t_back = ts[i - 1] if i > 0 else None
        nz = noise_fn(i)
        second = old is not None and prev_t >= 0
        n = nz[:, 1 if second else 0].to(dt)[None]            # [1,nf,C,H,W]; randn_tensor(dtype=sample.dtype), scheduling_dpm:452,460
        seq = iter([n, n])
        latents_f, old = S.dpm_step(ac, pred, old, t, prev_t, t_back, latents, lambda: next(seq))
        latents = latents_f.to(dt)
    if export is not None:
        export.update(guidance_scale=guidance_scale, guidance_scale_img=gi)
    return torch.cat(fifo_lat, dim=1), fifo_old, latents
