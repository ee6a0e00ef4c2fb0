"""GPU: whole FIFO queue evolution on the HIP path (tiny DiT, 1 clip = 52 iterations / 274 window steps, keyed noise)
against the oracle driving the oracle DiT with the SAME noise.  bf16 on both sides; the tolerance covers the drift of
52 stochastic denoising steps per frame (end-to-end 3e-2 per step)."""
import os

import numpy as np
import pytest
import torch
from conftest import measured
from types import SimpleNamespace

from oracle import dit_ref as O
from oracle import fifo_ref as Fq
from oracle import scheduler_ref as S

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _noise(i, tag, shape):
    g = torch.Generator().manual_seed(1000 * i + tag)
    return torch.randn(shape, generator=g).to(BF)


@pytest.mark.timeout(900)
def test_fifo_queue_evolution_vs_oracle(golden_dir):
    from tokensgen_amd import fifo
    from tokensgen_amd.scheduler import CogVideoXDPMScheduler
    from tokensgen_amd.transformer import CogVideoXTransformer3DModel
    gt = torch.load(os.path.join(golden_dir, "dit_tiny.pt"), weights_only=False)
    g = torch.load(os.path.join(golden_dir, "fifo_tiny.pt"), weights_only=False)
    cfg, vipcfg = gt["cfg"], gt["vip"]
    sd = {k: v.to(BF) for k, v in O.make_state_dict(cfg, 128, seed=g["weight_seed"]).items()}
    H, W, nf, T = g["H"], g["W"], 13, 52
    num_frames = 13
    lat = g["fifo_latents"].to(BF)
    old = [None if t is None else t.to(BF) for t in g["fifo_old"]]
    prompt, emb = g["prompt"].to(BF), g["image_embeddings"].to(BF)[:, :8]
    grid_t = g["grid_t"][:13].copy()
    cond_t = g["cond_t"][:8].copy()
    rope = O.rope_3d_crop(64, (0, 0, 0), (nf, H // 2, W // 2), (nf, H // 2, W // 2))
    betas, ac = S.alphas_cumprod()
    ts = S.trailing_timesteps(T)
    emb_ext = torch.cat([emb] + [emb[:, -4:]] * (T // nf + 1), dim=1)

    def denoise(x, tt, gt_, ct_, vs):
        return O.dit_forward(sd, cfg, x, prompt, tt, emb_ext[:, vs:vs + 5], rope, O.rope_3d(64, gt_, g["grid_h"], g["grid_w"]),
                             O.rope_3d(64, ct_, g["cond_h"], g["cond_w"]), vip_scale=[0.6])
    ref_trace = []
    ref = Fq.run_fifo_prenoise(denoise, betas, ac, lat, old, ts, num_frames, 6.0, grid_t, cond_t, 1000, _noise, trace=ref_trace)

    m = CogVideoXTransformer3DModel(num_attention_heads=2, attention_head_dim=64, num_layers=2, time_embed_dim=cfg["time_embed_dim"],
                                    text_embed_dim=cfg["text_embed_dim"], use_rotary_positional_embeddings=True, device=DEV)
    m.set_vip_layers(None, **vipcfg)
    m.load_state_dict(sd, strict=True)
    sched = CogVideoXDPMScheduler(prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=1.0, timestep_spacing="trailing")
    sched.set_timesteps(T)
    pipe = SimpleNamespace(device=torch.device(DEV), scheduler=sched, transformer=m, guidance_scale=6.0)
    bo = SimpleNamespace(sampling_params=dict(use_adaptive_padding=True, num_partitions=4), fifo_latents=lat, fifo_old_pred_original_sample=old,
                         nf_per_chunk=nf, vip_nf_per_chunk=4, num_frames=num_frames, image_embeddings=emb, timesteps=sched.timesteps,
                         num_inference_steps=T, do_classifier_free_guidance=True, use_separate_guidance=False, use_dynamic_cfg=False,
                         prompt_embeds=prompt, image_rotary_emb=rope, vip_image_rotary_grid=[grid_t, g["grid_h"], g["grid_w"]],
                         vip_condition_rotary_grid=[cond_t, g["cond_h"], g["cond_w"]], guidance_scale=6.0, cache_idx=[],
                         video_ipadapter_start_frame_idx=1000, output_type="latent", return_dict=False, orig_latents=lat[:, :nf])
    trace = []
    out = fifo.cogvideo_fifo_mp_v2([pipe], bo, step_noise_fn=lambda i, r, s: _noise(i, r, s).to(DEV),
                                   tail_noise_fn=lambda i, s: _noise(i, 97, s).to(DEV), trace=trace)[1]
    assert trace == ref_trace and len(trace) == 274
    assert out.shape == ref.shape == (1, num_frames, 16, H, W) and torch.isfinite(out).all()
    rel = measured(((out.float().cpu() - ref.float()).norm() / ref.float().norm()).item())
    assert rel < 1.5e-2, rel          # 274 window steps of a stochastic bf16 solver, identical keyed noise; measured 6.5e-3


@pytest.mark.timeout(900)
def test_base_stage_seeds_fifo_like_reference(golden_dir):
    """Pipeline base stage (52 scalar-timestep CFG steps on chunk 0) on the HIP path vs the oracle restatement of
    pipeline_cogvideox_mp_fifo.py:1186-1307 with the same keyed noise; then decode the result with the HIP VAE."""
    from tokensgen_amd.pipeline import MPFIFOVideoIPAdapterCogVideoXPipeline
    from tokensgen_amd.scheduler import CogVideoXDPMScheduler
    from tokensgen_amd.transformer import CogVideoXTransformer3DModel
    gt = torch.load(os.path.join(golden_dir, "dit_tiny.pt"), weights_only=False)
    g = torch.load(os.path.join(golden_dir, "fifo_tiny.pt"), weights_only=False)
    cfg, vipcfg = gt["cfg"], gt["vip"]
    sd = {k: v.to(BF) for k, v in O.make_state_dict(cfg, 128, seed=g["weight_seed"]).items()}
    H, W, nf, T = g["H"], g["W"], 13, 52
    gen = torch.Generator().manual_seed(77)
    lat0 = torch.randn(1, nf, 16, H, W, generator=gen).to(BF)
    prompt, emb = g["prompt"].to(BF), g["image_embeddings"].to(BF)[:, :8]
    rope = O.rope_3d_crop(64, (0, 0, 0), (nf, H // 2, W // 2), (nf, H // 2, W // 2))
    _, ac = S.alphas_cumprod()
    ts = S.trailing_timesteps(T)
    vr = O.rope_3d(64, g["grid_t"][:13], g["grid_h"], g["grid_w"])
    cr = O.rope_3d(64, g["cond_t"][:5], g["cond_h"], g["cond_w"])
    den = lambda x, t: O.dit_forward(sd, cfg, x, prompt, t, emb[:, :5], rope, vr, cr, vip_scale=[0.6])
    ref_lat, ref_old, ref_final = Fq.base_stage(den, ac, lat0, ts, 6.0, lambda i: _noise(i, 5, (nf, 2, 16, H, W)))

    m = CogVideoXTransformer3DModel(num_attention_heads=2, attention_head_dim=64, num_layers=2, time_embed_dim=cfg["time_embed_dim"],
                                    text_embed_dim=cfg["text_embed_dim"], use_rotary_positional_embeddings=True, device=DEV)
    m.set_vip_layers(None, **vipcfg)
    m.load_state_dict(sd, strict=True)
    sched = CogVideoXDPMScheduler(prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=1.0, timestep_spacing="trailing")
    pipe = MPFIFOVideoIPAdapterCogVideoXPipeline(m, sched, resampler_config=dict(num_temporal_queries=4, num_height_queries=2, num_width_queries=3))
    out = pipe(prompt_embeds=prompt[1:], negative_prompt_embeds=prompt[:1], image_embeddings=emb, height=H * 8, width=W * 8,
               latents=lat0, step_noise=lambda i: _noise(i, 5, (nf, 2, 16, H, W)))
    assert out.fifo_latents.shape == ref_lat.shape == (1, T, 16, H, W)
    rel = lambda a, b: measured(((a.float().cpu() - b.float()).norm() / b.float().norm()).item())
    assert rel(out.fifo_latents, ref_lat) < 2e-2 and rel(out.orig_latents, ref_final) < 2e-2      # measured 4.3e-3 / 8.5e-3
    assert [o is None for o in out.fifo_old_pred_original_sample] == [o is None for o in ref_old]
    assert np.array_equal(out.vip_condition_rotary_grid[0], g["cond_t"][:8]) and np.array_equal(out.vip_image_rotary_grid[0], g["grid_t"][:13])
    with pytest.raises(NotImplementedError):
        pipe(prompt="a cat")


@pytest.mark.timeout(900)
def test_base_stage_vs_reference_pipeline_golden(golden_dir, parity):
    """SURVEY §8c G11 on the HIP path: the product's base stage (tokensgen_amd.pipeline.__call__) against the reference's OWN pipeline run
    stored in tests/golden/base_stage_tiny.pt (bf16 case), with the reference's gaussian draws replayed in order: FIFO seed latents, the
    None pattern of the x0 list, final chunk-0 latents, position grids and token padding.  52 stochastic bf16 steps: rel-L2 <= 2e-2 (measured 4.5e-3 / 8.3e-3 / 7.7e-3)."""
    from tokensgen_amd.pipeline import MPFIFOVideoIPAdapterCogVideoXPipeline
    from tokensgen_amd.scheduler import CogVideoXDPMScheduler
    from tokensgen_amd.transformer import CogVideoXTransformer3DModel
    gt = torch.load(os.path.join(golden_dir, "dit_tiny.pt"), weights_only=False)
    g = torch.load(os.path.join(golden_dir, "base_stage_tiny.pt"), weights_only=False)
    c = g["cases"]["torch.bfloat16"]
    cfg, vipcfg, H, W, T, nf = gt["cfg"], gt["vip"], g["H"], g["W"], g["steps"], 13
    sd = {k: v.to(BF) for k, v in O.make_state_dict(cfg, 128, seed=g["weight_seed"]).items()}
    m = CogVideoXTransformer3DModel(num_attention_heads=2, attention_head_dim=64, num_layers=2, time_embed_dim=cfg["time_embed_dim"],
                                    text_embed_dim=cfg["text_embed_dim"], use_rotary_positional_embeddings=True, device=DEV)
    m.set_vip_layers(None, **vipcfg)
    m.load_state_dict(sd, strict=True)
    sched = CogVideoXDPMScheduler(prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=1.0, timestep_spacing="trailing")
    pipe = MPFIFOVideoIPAdapterCogVideoXPipeline(m, sched, resampler_config=dict(num_temporal_queries=4, num_height_queries=2, num_width_queries=3))
    draws, per_step, k = list(c["step_draws"]), [], 0
    for i in range(T):
        n = 1 if (i == 0 or i == T - 1) else 2
        per_step.append(draws[k:k + n]); k += n
    # the tiny grid's RoPE crop region is NOT the full grid (get_resize_crop_region_for_grid against the 720x480 base, :81-96): the
    # product must build the same table the reference pipeline built
    for a, b in zip(pipe._prepare_rotary_positional_embeddings(H * 8, W * 8, nf), g["image_rotary_emb"]):
        assert torch.allclose(a.cpu(), b, atol=2e-6)
    out = pipe(prompt_embeds=c["prompt"], negative_prompt_embeds=c["negative"], image_embeddings=c["emb_in"], height=H * 8, width=W * 8,
               num_chunks=g["chunks"], latents=c["init_latents"], video_ipadapter_scale=g["vip_scale"],
               step_noise=lambda i: torch.stack([per_step[i][0][0], per_step[i][-1][0]], dim=1))
    assert torch.equal(out.image_embeddings.cpu(), c["image_embeddings"]) and torch.equal(out.prompt_embeds.cpu(), c["prompt_embeds"])
    assert np.array_equal(out.vip_image_rotary_grid[0], g["vip_image_rotary_grid"][0]) and np.array_equal(out.vip_condition_rotary_grid[0], g["vip_condition_rotary_grid"][0])
    assert [int(t) for t in out.timesteps] == [int(t) for t in g["timesteps"]] and out.num_frames == g["chunks"] * nf
    assert [o is None for o in out.fifo_old_pred_original_sample] == [o is None for o in c["fifo_old"]]
    rel = lambda a, b: ((a.float().cpu() - b.float()).norm() / b.float().norm()).item()
    parity(rel(out.fifo_latents, c["fifo_latents"]), 2e-2, "base stage FIFO seed latents vs reference pipeline run (bf16, 52 SDE steps)")
    parity(rel(out.orig_latents, c["orig_latents"]), 2e-2, "base stage final latents vs reference pipeline run")
    x0 = torch.cat([o for o in out.fifo_old_pred_original_sample if o is not None], dim=1)
    x0r = torch.cat([o for o in c["fifo_old"] if o is not None], dim=1)
    parity(rel(x0, x0r), 2e-2, "base stage x0 seed list vs reference pipeline run")


@pytest.mark.timeout(900)
def test_base_stage_dynamic_separate_guidance_vs_reference_pipeline_golden(golden_dir, parity):
    """The base stage with use_dynamic_cfg + use_separate_guidance against the reference's OWN pipeline run in that mode
    (tests/golden/base_stage_dyn_sep.pt, bf16 case, draws replayed): the image guidance weight is the reference's local variable re-assigned every
    step (pipeline_cogvideox_mp_fifo.py:1257) — it compounds — and the compounded value is what the stage exports to the FIFO driver (:1336)."""
    from tokensgen_amd.pipeline import MPFIFOVideoIPAdapterCogVideoXPipeline
    from tokensgen_amd.scheduler import CogVideoXDPMScheduler
    from tokensgen_amd.transformer import CogVideoXTransformer3DModel
    gt = torch.load(os.path.join(golden_dir, "dit_tiny.pt"), weights_only=False)
    g = torch.load(os.path.join(golden_dir, "base_stage_dyn_sep.pt"), weights_only=False)
    c = g["cases"]["torch.bfloat16"]
    cfg, vipcfg, H, W, T, nf = gt["cfg"], gt["vip"], g["H"], g["W"], g["steps"], 13
    sd = {k: v.to(BF) for k, v in O.make_state_dict(cfg, 128, seed=g["weight_seed"]).items()}
    m = CogVideoXTransformer3DModel(num_attention_heads=2, attention_head_dim=64, num_layers=2, time_embed_dim=cfg["time_embed_dim"],
                                    text_embed_dim=cfg["text_embed_dim"], use_rotary_positional_embeddings=True, device=DEV)
    m.set_vip_layers(None, **vipcfg)
    m.load_state_dict(sd, strict=True)
    sched = CogVideoXDPMScheduler(prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=1.0, timestep_spacing="trailing")
    pipe = MPFIFOVideoIPAdapterCogVideoXPipeline(m, sched, resampler_config=dict(num_temporal_queries=4, num_height_queries=2, num_width_queries=3))
    draws, per_step, k = list(c["step_draws"]), [], 0
    for i in range(T):
        n = 1 if (i == 0 or i == T - 1) else 2
        per_step.append(draws[k:k + n]); k += n
    unc = torch.cat([g["unc_tok"].to(BF)] * (g["chunks"] + 1), dim=1)
    out = pipe(prompt_embeds=c["prompt"], negative_prompt_embeds=c["negative"], image_embeddings=c["emb_in"], uncond_image_embeddings=unc,
               height=H * 8, width=W * 8, num_chunks=g["chunks"], num_inference_steps=T, latents=c["init_latents"], video_ipadapter_scale=g["vip_scale"],
               guidance_scale=g["guidance_scale"], guidance_scale_img=g["guidance_scale_img"], use_separate_guidance=True, use_dynamic_cfg=True,
               step_noise=lambda i: torch.stack([per_step[i][0][0], per_step[i][-1][0]], dim=1))
    assert torch.equal(out.image_embeddings.cpu(), c["image_embeddings"]) and torch.equal(out.prompt_embeds.cpu(), c["prompt_embeds"])
    assert out.guidance_scale == c["guidance_scale_out"]
    assert out.guidance_scale_img == c["guidance_scale_img_out"] != g["guidance_scale_img"]        # Python-float arithmetic: exact
    rel = lambda a, b: ((a.float().cpu() - b.float()).norm() / b.float().norm()).item()
    parity(rel(out.fifo_latents, c["fifo_latents"]), 2e-2, "dynamic + 3-way base stage FIFO seed latents vs reference pipeline run (bf16, 16 SDE steps)")
    parity(rel(out.orig_latents, c["orig_latents"]), 2e-2, "dynamic + 3-way base stage final latents vs reference pipeline run")


@pytest.mark.timeout(900)
def test_flow_without_classifier_free_guidance_vs_oracle(golden_dir, parity):
    """guidance_scale <= 1 switches classifier-free guidance off in the reference (pipeline_cogvideox_mp_fifo.py:1012, 1196-1200, 1260; worker
    cogvideo_sampling_mp_fifo.py:497-498, 528): batch of ONE through the DiT, the model output is the prediction.  Base stage on the HIP path (B = 1
    forwards, tg_cfg_dpm_step_ex branches = 1) against the oracle's base stage with the same keyed noise; its output then drives the FIFO stage in the
    same mode (the worker's no-guidance arithmetic itself is pinned to runs of the reference worker: fifo_worker_variants.pt "no_cfg")."""
    from tokensgen_amd import fifo
    from tokensgen_amd.pipeline import MPFIFOVideoIPAdapterCogVideoXPipeline
    from tokensgen_amd.scheduler import CogVideoXDPMScheduler
    from tokensgen_amd.transformer import CogVideoXTransformer3DModel
    gt = torch.load(os.path.join(golden_dir, "dit_tiny.pt"), weights_only=False)
    g = torch.load(os.path.join(golden_dir, "fifo_tiny.pt"), weights_only=False)
    cfg, vipcfg = gt["cfg"], gt["vip"]
    sd = {k: v.to(BF) for k, v in O.make_state_dict(cfg, 128, seed=g["weight_seed"]).items()}
    H, W, nf, T = g["H"], g["W"], 13, 52
    gen = torch.Generator().manual_seed(79)
    lat0 = torch.randn(1, nf, 16, H, W, generator=gen).to(BF)
    prompt = g["prompt"].to(BF)[1:]
    emb1 = g["image_embeddings"].to(BF)[:1, :8]
    rope = O.rope_3d_crop(64, (0, 0, 0), (nf, H // 2, W // 2), (nf, H // 2, W // 2))
    _, ac = S.alphas_cumprod()
    ts = S.trailing_timesteps(T)
    vr = O.rope_3d(64, g["grid_t"][:13], g["grid_h"], g["grid_w"])
    cr = O.rope_3d(64, g["cond_t"][:5], g["cond_h"], g["cond_w"])
    den = lambda x, t: O.dit_forward(sd, cfg, x, prompt, t, emb1[:, :5], rope, vr, cr, vip_scale=[0.6])
    ref_lat, ref_old, ref_final = Fq.base_stage(den, ac, lat0, ts, 1.0, lambda i: _noise(i, 5, (nf, 2, 16, H, W)), do_classifier_free_guidance=False)
    m = CogVideoXTransformer3DModel(num_attention_heads=2, attention_head_dim=64, num_layers=2, time_embed_dim=cfg["time_embed_dim"],
                                    text_embed_dim=cfg["text_embed_dim"], use_rotary_positional_embeddings=True, device=DEV)
    m.set_vip_layers(None, **vipcfg)
    m.load_state_dict(sd, strict=True)
    sched = CogVideoXDPMScheduler(prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=1.0, timestep_spacing="trailing")
    pipe = MPFIFOVideoIPAdapterCogVideoXPipeline(m, sched, resampler_config=dict(num_temporal_queries=4, num_height_queries=2, num_width_queries=3))
    out = pipe(prompt_embeds=prompt, negative_prompt_embeds=None, image_embeddings=emb1, height=H * 8, width=W * 8, latents=lat0, guidance_scale=1.0,
               step_noise=lambda i: _noise(i, 5, (nf, 2, 16, H, W)))
    assert not out.do_classifier_free_guidance and out.prompt_embeds.shape[0] == 1 and out.image_embeddings.shape[0] == 1
    rel = lambda a, b: ((a.float().cpu() - b.float()).norm() / b.float().norm()).item()
    parity(rel(out.fifo_latents, ref_lat), 2e-2, "no-guidance base stage, FIFO seed latents vs oracle")
    parity(rel(out.orig_latents, ref_final), 2e-2, "no-guidance base stage, final latents vs oracle")
    out.num_frames = 1
    lat = fifo.cogvideo_fifo_mp_v2([pipe], out, noise_seed=3)[1]
    assert lat.shape[1] == 1 and bool(torch.isfinite(lat).all())


def test_fifo_worker_guidance_variants_vs_reference_runs(golden_dir, parity):
    """FifoWorker.window_step on the HIP path against RUNS OF THE REFERENCE WORKER BODY (`fifo_onestep_per_gpu`,
    cogvideo_sampling_mp_fifo.py:408-579; tests/golden/fifo_worker_variants.pt, bf16 cases) for the branches the shipped configs leave off:
    no classifier-free guidance at all (B = 1 forward, branches = 1: the model output is the prediction, :497-498),
    3-way `use_separate_guidance` with guidance_scale_img (B = 3 forward, tg_cfg_dpm_step_ex branches = 3), `use_dynamic_cfg` (per-frame fp32
    guidance -> fp32 solver arithmetic), both together, and an epsilon-prediction scheduler — head window (prev_t = -1) and tail window
    (t = 999 / no back step), with the reference's gaussian draws replayed in order."""
    from tokensgen_amd.fifo import FifoWorker
    from tokensgen_amd.scheduler import CogVideoXDPMScheduler
    from tokensgen_amd.transformer import CogVideoXTransformer3DModel
    gt = torch.load(os.path.join(golden_dir, "dit_tiny.pt"), weights_only=False)
    g = torch.load(os.path.join(golden_dir, "fifo_worker_variants.pt"), weights_only=False)
    cfg, vipcfg, H, W, nf = gt["cfg"], gt["vip"], g["H"], g["W"], 13
    sd = {k: v.to(BF) for k, v in O.make_state_dict(cfg, 128, seed=g["weight_seed"]).items()}
    m = CogVideoXTransformer3DModel(num_attention_heads=2, attention_head_dim=64, num_layers=2, time_embed_dim=cfg["time_embed_dim"],
                                    text_embed_dim=cfg["text_embed_dim"], use_rotary_positional_embeddings=True, device=DEV)
    m.set_vip_layers(None, **vipcfg)
    m.load_state_dict(sd, strict=True)
    rope = O.rope_3d_crop(64, (0, 0, 0), (nf, H // 2, W // 2), (nf, H // 2, W // 2))
    seen = set()
    for c in g["cases"]:
        if "bfloat16" not in c["dtype"]:
            continue
        sched = CogVideoXDPMScheduler(prediction_type=c["prediction_type"], rescale_betas_zero_snr=True, snr_shift_scale=1.0, timestep_spacing="trailing")
        sched.set_timesteps(52)
        w = FifoWorker(m, sched, c["prompt"], rope, c["guidance_scale"], c["grid_h"], c["grid_w"], c["cond_h"], c["cond_w"],
                       use_separate_guidance=c["separate"], guidance_scale_img=c["guidance_scale_img"], use_dynamic_cfg=c["dynamic"],
                       num_inference_steps=52, do_classifier_free_guidance=c.get("cfg", True))
        t, prev_t, next_t = c["t"].tolist(), c["prev_t"].tolist(), c["next_t"].tolist()
        has_old = [o is not None for o in c["old"]]
        old = torch.stack([(o if o is not None else torch.zeros(1, 1, 16, H, W, dtype=BF))[0, 0] for o in c["old"]]).to(DEV)
        gen = torch.Generator().manual_seed(c["rng_seed"])                     # the reference's draws, in its order: one per frame, a second on the 2M branch
        noise = torch.zeros(nf, 2, 16, H, W, dtype=BF)
        for j in range(nf):
            noise[j, 0] = torch.randn(1, 1, 16, H, W, generator=gen, dtype=BF)[0, 0]
            if has_old[j] and prev_t[j] >= 0:
                noise[j, 1] = torch.randn(1, 1, 16, H, W, generator=gen, dtype=BF)[0, 0]
        x, x0 = w.window_step(c["latents"].to(DEV), old, has_old, t, prev_t, next_t, noise.to(DEV), c["grid_t"], c["cond_t"], c["image_embeddings"].to(DEV))
        # the guidance branches as separate batch-1 forwards (what the ranks of a split FIFO iteration compute, round 6): per-sample kernels — bitwise the batched step
        xs, x0s = w.window_step(c["latents"].to(DEV), old, has_old, t, prev_t, next_t, noise.to(DEV), c["grid_t"], c["cond_t"], c["image_embeddings"].to(DEV),
                                split_branches=True)
        assert torch.equal(xs, x) and torch.equal(x0s, x0), c["name"]
        rel = lambda a, b: ((a.float().cpu() - b.float()).norm() / b.float().norm()).item()
        tag = f"{c['name']} window@{c['start']}"
        # the reference's own rounding noise on this case: its bf16 run (the fixture) against the oracle in fp32 on the same bf16 weights / inputs /
        # draws.  Guidance amplifies the DiT's bf16 error — v = 9 c - 5 u_txt - 3 u_img for (g, g_img) = (6, 4) — so the 3-way cases sit higher
        sd32 = {k: v.float() for k, v in sd.items()}
        vr, cr = O.rope_3d(64, c["grid_t"], c["grid_h"], c["grid_w"]), O.rope_3d(64, c["cond_t"], c["cond_h"], c["cond_w"])
        den = lambda xx, tt: O.dit_forward(sd32, cfg, xx, c["prompt"].float(), tt, c["image_embeddings"].float(), rope, vr, cr, vip_scale=[0.6])
        gen2 = torch.Generator().manual_seed(c["rng_seed"])
        _, ac = S.alphas_cumprod()
        o32, x032 = Fq.window_step(den, ac, c["guidance_scale"], c["latents"].float(), [None if o is None else o.float() for o in c["old"]], c["t"].numpy(),
                                   c["prev_t"].numpy(), c["next_t"].numpy(), lambda: torch.randn(1, 1, 16, H, W, generator=gen2, dtype=BF).float(),
                                   torch.float32, use_separate_guidance=c["separate"], guidance_scale_img=c["guidance_scale_img"],
                                   use_dynamic_cfg=c["dynamic"], num_inference_steps=52, prediction_type=c["prediction_type"],
                                   do_classifier_free_guidance=c.get("cfg", True))
        want_x0 = torch.cat(c["out_x0"], dim=1)[0]
        floor = max(rel(c["out_latents"], o32), rel(want_x0, torch.cat(x032, dim=1)[0]))
        parity(floor, 1.0, f"noise floor, {tag}: the reference's bf16 worker run vs the fp32 oracle (informative)")
        tol = max(2e-2, 1.5 * floor)
        parity(rel(x, c["out_latents"]), tol, f"worker latents, {tag}, HIP vs reference worker run (bf16)")
        parity(rel(x0, want_x0), tol, f"worker x0, {tag}")
        seen.add(c["name"])
    assert seen == {"separate", "dynamic", "separate_dynamic", "epsilon_static", "no_cfg", "no_cfg_dynamic"}


@pytest.mark.timeout(900)
def test_base_stage_separate_guidance_and_dynamic_cfg_vs_oracle(golden_dir, parity):
    """The pipeline base stage with `use_separate_guidance` (3-way batch, guidance_scale_img) + `use_dynamic_cfg`
    (pipeline_cogvideox_mp_fifo.py:1026-1029, 1197-1200, 1252-1263) on the HIP path (B = 3 forwards, tg_cfg_dpm_step_ex with three branches and
    the fp32 solver state) against the oracle's base stage with the same keyed noise; the result then seeds a FIFO run in the same mode."""
    from tokensgen_amd import fifo
    from tokensgen_amd.pipeline import MPFIFOVideoIPAdapterCogVideoXPipeline
    from tokensgen_amd.scheduler import CogVideoXDPMScheduler
    from tokensgen_amd.transformer import CogVideoXTransformer3DModel
    gt = torch.load(os.path.join(golden_dir, "dit_tiny.pt"), weights_only=False)
    g = torch.load(os.path.join(golden_dir, "fifo_tiny.pt"), weights_only=False)
    cfg, vipcfg = gt["cfg"], gt["vip"]
    sd = {k: v.to(BF) for k, v in O.make_state_dict(cfg, 128, seed=g["weight_seed"]).items()}
    H, W, nf, T = g["H"], g["W"], 13, 52
    gen = torch.Generator().manual_seed(78)
    lat0 = torch.randn(1, nf, 16, H, W, generator=gen).to(BF)
    prompt = g["prompt"].to(BF)
    emb1 = g["image_embeddings"].to(BF)[:1, :8]
    unc = torch.randn(1, 8, 128, 2, 3, generator=gen).to(BF)                    # stands for the tokens of an all-zero video
    emb3 = torch.cat([emb1, unc, emb1], dim=0)
    prompt3 = torch.cat([prompt[:1], prompt[1:], prompt[1:]], dim=0)
    rope = O.rope_3d_crop(64, (0, 0, 0), (nf, H // 2, W // 2), (nf, H // 2, W // 2))
    _, ac = S.alphas_cumprod()
    ts = S.trailing_timesteps(T)
    vr = O.rope_3d(64, g["grid_t"][:13], g["grid_h"], g["grid_w"])
    cr = O.rope_3d(64, g["cond_t"][:5], g["cond_h"], g["cond_w"])
    den = lambda x, t: O.dit_forward(sd, cfg, x, prompt3, t, emb3[:, :5], rope, vr, cr, vip_scale=[0.6])
    ref_lat, ref_old, ref_final = Fq.base_stage(den, ac, lat0, ts, 6.0, lambda i: _noise(i, 5, (nf, 2, 16, H, W)), use_separate_guidance=True,
                                                guidance_scale_img=4.0, use_dynamic_cfg=True)
    m = CogVideoXTransformer3DModel(num_attention_heads=2, attention_head_dim=64, num_layers=2, time_embed_dim=cfg["time_embed_dim"],
                                    text_embed_dim=cfg["text_embed_dim"], use_rotary_positional_embeddings=True, device=DEV)
    m.set_vip_layers(None, **vipcfg)
    m.load_state_dict(sd, strict=True)
    sched = CogVideoXDPMScheduler(prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=1.0, timestep_spacing="trailing")
    pipe = MPFIFOVideoIPAdapterCogVideoXPipeline(m, sched, resampler_config=dict(num_temporal_queries=4, num_height_queries=2, num_width_queries=3))
    out = pipe(prompt_embeds=prompt[1:], negative_prompt_embeds=prompt[:1], image_embeddings=emb3, height=H * 8, width=W * 8, latents=lat0,
               step_noise=lambda i: _noise(i, 5, (nf, 2, 16, H, W)), use_separate_guidance=True, guidance_scale_img=4.0, use_dynamic_cfg=True)
    rel = lambda a, b: ((a.float().cpu() - b.float()).norm() / b.float().norm()).item()
    parity(rel(out.fifo_latents, ref_lat), 2e-2, "3-way + dynamic base stage, FIFO seed latents vs oracle")
    parity(rel(out.orig_latents, ref_final), 2e-2, "3-way + dynamic base stage, final latents vs oracle")
    assert out.use_separate_guidance and out.use_dynamic_cfg and out.prompt_embeds.shape[0] == 3 and out.image_embeddings.shape[0] == 3
    # the FIFO stage accepts that output (3-way windows, per-frame dynamic guidance) and stays finite for a few iterations
    out.num_frames = 1                                       # 40 iterations: enough to leave the ramp
    lat = fifo.cogvideo_fifo_mp_v2([pipe], out, noise_seed=3)[1]
    assert lat.shape[1] == 1 and bool(torch.isfinite(lat).all())
