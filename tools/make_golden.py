#!/usr/bin/env python3
"""Generate tests/golden/*.pt by running the REFERENCE (/root/reference, imported through
tools/ref_shim) on CPU with seeded inputs.  Run in the build container only:

    python tools/make_golden.py [--full-block]

Fixtures hold inputs/outputs (data), never reference source.  Weights are regenerated from seeds
by oracle.dit_ref.make_state_dict (same torch build here and on the GPU box); a checksum of the
weights is stored to detect RNG drift.
"""
import argparse
import os
import queue
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402

ref_shim.install()
from longvgen.models.cogvideox_transformer_3d import CogVideoXBlock, CogVideoXTransformer3DModel  # noqa: E402
from longvgen.models.embeddings import get_3d_rotary_pos_embed, get_3d_rotary_pos_embed_v2  # noqa: E402
from longvgen.schedulers.scheduling_dpm_cogvideox import CogVideoXDPMScheduler  # noqa: E402
import importlib.util  # noqa: E402

from oracle import dit_ref as O  # noqa: E402  (only for make_state_dict: seeded weights + names)

GOLD = os.path.join(ROOT, "tests", "golden")
# GEMM-tile friendly tiny config (every N % 128 == 0, K % 64 == 0 except proj_out, which the product pads)
TINY = dict(num_attention_heads=2, attention_head_dim=64, num_layers=2, patch_size=2, time_embed_dim=128,
            text_embed_dim=64, in_channels=16, out_channels=16)
TINY_VIP = dict(length=5 * 2 * 3, func_type="1", scale=[0.6],
                resampler_params=dict(output_dim=128, num_height_queries=2, num_width_queries=3,
                                      num_temporal_queries=4))


def load_ref_module(rel, name):
    """Import one reference file without triggering its package __init__ (which needs more deps)."""
    spec = importlib.util.spec_from_file_location(name, os.path.join(ref_shim.REFERENCE_ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sd_checksum(sd):
    return float(sum(v.double().abs().sum() for v in sd.values()))


def tiny_model(seed, vip=True, H=4, W=6):
    m = CogVideoXTransformer3DModel(num_attention_heads=2, attention_head_dim=64, num_layers=2, in_channels=16,
                                    out_channels=16, text_embed_dim=TINY["text_embed_dim"],
                                    time_embed_dim=TINY["time_embed_dim"], sample_width=W,
                                    sample_height=H, sample_frames=49, use_rotary_positional_embeddings=True,
                                    max_text_seq_length=8)
    if vip:
        m.set_vip_layers(None, **TINY_VIP)
    sd = O.make_state_dict(TINY, n_vip_dim=128 if vip else None, seed=seed)
    m.load_state_dict(sd, strict=True)
    return m.eval(), sd


def tiny_inputs(seed, B=2, H=4, W=6):
    g = torch.Generator().manual_seed(seed)
    return dict(hs=torch.randn(B, 13, 16, H, W, generator=g), enc=torch.randn(B, 8, TINY["text_embed_dim"], generator=g),
                vip=torch.randn(B, 5, 128, 2, 3, generator=g), ts=torch.randint(0, 1000, (B, 13), generator=g))


def tiny_ropes(H=4, W=6, t0=0.0):
    f32 = np.float32
    rope = get_3d_rotary_pos_embed(64, ((0, 0, 0), (13, H // 2, W // 2)), (13, H // 2, W // 2))
    vrope = get_3d_rotary_pos_embed_v2(64, np.arange(13, dtype=f32) + f32(t0), np.arange(H // 2, dtype=f32),
                                       np.arange(W // 2, dtype=f32))
    crope = get_3d_rotary_pos_embed_v2(64, np.linspace(1000, 1016.25, 5, dtype=f32),
                                       np.linspace(0, H // 2, 2, endpoint=False, dtype=f32),
                                       np.linspace(0, W // 2, 3, endpoint=False, dtype=f32))
    return rope, vrope, crope


@torch.no_grad()
def gen_dit_tiny():
    out = {"cfg": TINY, "vip": TINY_VIP, "cases": []}
    for seed in (0, 1):
        m, sd = tiny_model(100 + seed)
        inp = tiny_inputs(200 + seed)
        rope, vrope, crope = tiny_ropes(t0=3.0 * seed)
        for dt in (torch.float32, torch.bfloat16):
            mm, _ = tiny_model(100 + seed)
            mm = mm.to(dt)
            for ts in (inp["ts"], inp["ts"][:, 0].clone()):
                taps = {}
                hooks = [blk.register_forward_hook(lambda mod, a, o, i=i: taps.__setitem__(f"block{i}.hidden", o[0].clone()))
                         for i, blk in enumerate(mm.transformer_blocks) if i == 0]
                y = mm(inp["hs"].to(dt), inp["enc"].to(dt), ts, vip_encoder_hidden_states=inp["vip"].to(dt),
                       image_rotary_emb=rope, vip_image_rotary_emb=vrope, vip_condition_rotary_emb=crope,
                       return_dict=False)[0]
                for h in hooks:
                    h.remove()
                out["cases"].append(dict(weight_seed=100 + seed, input_seed=200 + seed, t0=3.0 * seed, dtype=str(dt),
                                         ts=ts, out=y.clone(), taps={k: v.clone() for k, v in taps.items()},
                                         sd_checksum=sd_checksum(sd)))
        if seed == 0:
            out["inputs0"] = inp
            out["ropes0"] = dict(rope=rope, vrope=vrope, crope=crope)
    # plain (no-vip) model: the T2To / base-model processor
    m, sd = tiny_model(300, vip=False)
    inp = tiny_inputs(301)
    rope, _, _ = tiny_ropes()
    y = m(inp["hs"], inp["enc"], inp["ts"][:, 0], image_rotary_emb=rope, return_dict=False)[0]
    out["plain"] = dict(weight_seed=300, input_seed=301, out=y, sd_checksum=sd_checksum(sd))
    torch.save(out, os.path.join(GOLD, "dit_tiny.pt"))
    print("dit_tiny.pt", len(out["cases"]), "cases")


@torch.no_grad()
def gen_vip_processor():
    """VIP attention processor alone (head_dim 64, 2 heads), with intermediate q/k/v and the three SDPAs."""
    ap = sys.modules["longvgen.models.attention_processor"]
    torch.manual_seed(7)
    D, heads = 128, 2
    attn = ap.Attention(query_dim=D, dim_head=64, heads=heads, qk_norm="layer_norm", eps=1e-6, bias=True,
                        out_bias=True, processor=ap.CogVideoXAttnProcessor2_0())
    proc = ap.VideoIPAdapterCogVideoXAttnProcessor2_0(heads=heads, cross_attention_dim=D, dim_head=64, eps=1e-6,
                                                      scale=[0.6], qk_norm="layer_norm", bias=True, num_tokens=30)
    attn.set_processor(proc)
    for p in attn.parameters():
        p.data = torch.randn_like(p) * (0.3 if p.ndim == 1 else 0.08)
    Nt, Nv, Np = 8, 13 * 2 * 3, 30
    hid = torch.randn(1, Nv, D)
    enc = torch.randn(1, Nt + Np, D)
    rope, vrope, crope = tiny_ropes()
    captured = []
    orig = ap.F.scaled_dot_product_attention

    def spy(q, k, v, **kw):
        o = orig(q, k, v, **kw)
        captured.append(dict(q=q.clone(), k=k.clone(), v=v.clone(), o=o.clone()))
        return o
    ap.F.scaled_dot_product_attention = spy
    try:
        oh, oe = attn(hid, encoder_hidden_states=enc, image_rotary_emb=rope, vip_image_rotary_emb=vrope,
                      vip_condition_rotary_emb=crope)
    finally:
        ap.F.scaled_dot_product_attention = orig
    sd = {("attn1." + k): v.clone() for k, v in attn.state_dict().items()}
    torch.save(dict(sd=sd, hid=hid, enc=enc, rope=rope, vrope=vrope, crope=crope, sdpa=captured, out_hidden=oh,
                    out_enc=oe, scale=[0.6], heads=heads, n_vip=Np), os.path.join(GOLD, "vip_processor.pt"))
    print("vip_processor.pt sdpa calls:", len(captured))


def make_sched():
    s = CogVideoXDPMScheduler(prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=1.0,
                              timestep_spacing="trailing")
    s.set_timesteps(52)
    return s


@torch.no_grad()
def gen_scheduler():
    s = make_sched()
    ts = s.timesteps.clone()
    out = dict(alphas_cumprod=s.alphas_cumprod.clone(), betas=s.betas.clone(), timesteps=ts, steps=[])
    g = torch.Generator().manual_seed(5)
    shape = (1, 1, 2, 2, 3)
    triples = [(999, 980, None, False), (980, 961, 999, True), (500 - 1, 480, 518, True), (37, 18, 57, True),
               (18, -1, 37, True), (18, -1, 37, False), (961, 941, 980, True), (999, 980, None, True)]
    # every (t, prev_t, t_back) the FIFO uses is a consecutive triple of the 52 trailing steps (+ the edges above)
    for i in range(1, 51):
        triples.append((int(ts[i]), int(ts[i + 1]), int(ts[i - 1]), True))
    for (t, pt, tb, has_old) in triples:
        if t not in ts.tolist():
            t = int(ts[(ts - t).abs().argmin()])
        for dt in (torch.float32, torch.bfloat16):
            mo = torch.randn(shape, generator=g).to(dt)
            x = torch.randn(shape, generator=g).to(dt)
            old = torch.randn(shape, generator=g).to(dt) if (has_old and tb is not None) else None
            torch.manual_seed(1000 + t)
            prev, x0 = s.step(mo, old, torch.tensor(t), torch.tensor(pt), None if tb is None else torch.tensor(tb), x,
                              return_dict=False)
            out["steps"].append(dict(t=t, prev_t=pt, t_back=tb, dtype=str(dt), model_output=mo, sample=x, old=old,
                                     noise_seed=1000 + t, prev_sample=prev.clone(), x0=x0.clone()))
    x = torch.randn(1, 2, 2, 3, generator=g)
    n = torch.randn(1, 2, 2, 3, generator=g)
    out["add_noise_to_xt"] = dict(x=x, noise=n, out=s.add_noise_to_xt(x, n, torch.Tensor([999]).long()))
    torch.save(out, os.path.join(GOLD, "scheduler.pt"))
    print("scheduler.pt", len(out["steps"]), "steps")


@torch.no_grad()
def gen_fifo(num_chunks=2, dtype=torch.float32, tag="fifo_tiny"):
    """Run the reference driver cogvideo_fifo_mp_v2 end-to-end on CPU (1 forked worker) with the tiny DiT."""
    fifo = load_ref_module("longvgen/fifo_sampling/cogvideo_sampling_mp_fifo.py", "ref_fifo")
    import threading

    class _ThreadProcess(threading.Thread):
        """mp.Process stand-in: a forked torch worker deadlocks on CPU (OpenMP after fork), and the driver
        blocks on the output queue while the worker runs, so a thread gives the same serial order — worker
        and driver then share ONE global RNG stream, which the oracle test replays with one generator."""
        def __init__(self, target, args):
            super().__init__(target=target, args=args, daemon=True)

        def close(self):
            pass

    fifo.mp = types.SimpleNamespace(Queue=queue.Queue, Process=_ThreadProcess)
    H, W, nf, T = 4, 6, 13, 52
    m, sd = tiny_model(400)
    m = m.to(dtype)
    sched = make_sched()
    pipe = types.SimpleNamespace(
        device=torch.device("cpu"), transformer=m, scheduler=sched, guidance_scale=6.0,
        _prepare_vip_rotary_positional_embeddings=lambda grid_t, grid_h, grid_w, device:
            get_3d_rotary_pos_embed_v2(64, grid_t, grid_h, grid_w))
    g = torch.Generator().manual_seed(401)
    fifo_latents = torch.randn(1, T, 16, H, W, generator=g).to(dtype)
    # base stage pushes to the FRONT each step (pipeline_cogvideox_mp_fifo.py:1190-1194): entry 51 (noisiest) has no x0 yet
    fifo_old = [torch.randn(1, 1, 16, H, W, generator=g).to(dtype) for _ in range(T - 1)] + [None]
    prompt = torch.randn(2, 8, TINY["text_embed_dim"], generator=g).to(dtype)
    n_groups = 4 * (num_chunks + 1)
    image_embeddings = torch.randn(1, n_groups, 128, 2, 3, generator=g).to(dtype).repeat(2, 1, 1, 1, 1)
    f32 = np.float32
    grid_t = np.linspace(0, num_chunks * nf, num_chunks * nf, endpoint=False, dtype=f32)
    grid_h = np.linspace(0, H // 2, H // 2, endpoint=False, dtype=f32)
    grid_w = np.linspace(0, W // 2, W // 2, endpoint=False, dtype=f32)
    cond_t = np.concatenate([np.linspace(1000 + i * nf, 1000 + (i + 1) * nf, 4, endpoint=False, dtype=f32)
                             for i in range(num_chunks + 1)])
    cond_h = np.linspace(0, H // 2, 2, endpoint=False, dtype=f32)
    cond_w = np.linspace(0, W // 2, 3, endpoint=False, dtype=f32)
    rope = get_3d_rotary_pos_embed(64, ((0, 0, 0), (nf, H // 2, W // 2)), (nf, H // 2, W // 2))
    base = types.SimpleNamespace(
        sampling_params=dict(use_adaptive_padding=True, num_partitions=4), fifo_latents=fifo_latents.clone(),
        fifo_old_pred_original_sample=list(fifo_old), nf_per_chunk=nf, vip_nf_per_chunk=4, num_frames=num_chunks * nf,
        image_embeddings=image_embeddings, timesteps=sched.timesteps, num_inference_steps=T,
        do_classifier_free_guidance=True, use_separate_guidance=False, use_dynamic_cfg=False, prompt_embeds=prompt,
        image_rotary_emb=rope, vip_image_rotary_grid=[grid_t.copy(), grid_h, grid_w],
        vip_condition_rotary_grid=[cond_t.copy(), cond_h, cond_w], attention_kwargs=None, guidance_scale=6.0,
        guidance_scale_img=None, extra_step_kwargs={}, cache_idx=[], condition_frames=None,
        video_ipadapter_start_frame_idx=1000, output_type="latent", return_dict=False,
        orig_latents=fifo_latents[:, :nf].clone())
    torch.manual_seed(4242)          # worker thread and driver both draw from this global stream, serially
    orig, video, cache = fifo.cogvideo_fifo_mp_v2([pipe], base)
    torch.save(dict(weight_seed=400, input_seed=401, rng_seed=4242, num_chunks=num_chunks, dtype=str(dtype), H=H, W=W,
                    fifo_latents=fifo_latents, fifo_old=fifo_old, prompt=prompt, image_embeddings=image_embeddings,
                    grid_t=grid_t, grid_h=grid_h, grid_w=grid_w, cond_t=cond_t, cond_h=cond_h, cond_w=cond_w,
                    video=video.clone(), sd_checksum=sd_checksum(sd)), os.path.join(GOLD, tag + ".pt"))
    print(tag + ".pt", tuple(video.shape))


@torch.no_grad()
def gen_full_block():
    """Full-width CogVideoX-5B block with VIP (BASELINE config 1), B=1: sampled outputs + statistics."""
    D, heads = 3072, 48
    cfg = dict(num_attention_heads=heads, attention_head_dim=64, num_layers=1, time_embed_dim=512)
    sd = O.make_state_dict(cfg, n_vip_dim=3072, seed=500)
    bsd = {k[len("transformer_blocks.0."):]: v for k, v in sd.items() if k.startswith("transformer_blocks.0.")}
    blk = CogVideoXBlock(dim=D, num_attention_heads=heads, attention_head_dim=64, time_embed_dim=512,
                         attention_bias=True)
    blk.set_vip_layers(length=480, func_type="1", scale=[0.6])
    blk.load_state_dict(bsd, strict=True)
    blk.eval()
    g = torch.Generator().manual_seed(501)
    hid = torch.randn(1, 17550, D, generator=g)
    enc = torch.randn(1, 706, D, generator=g)
    temb = torch.randn(1, 13, 512, generator=g)
    f32 = np.float32
    rope = get_3d_rotary_pos_embed_v2(64, np.arange(13, dtype=f32), np.arange(30, dtype=f32), np.arange(45, dtype=f32))
    crope = get_3d_rotary_pos_embed_v2(64, np.linspace(1000, 1016.25, 5, dtype=f32),
                                       np.linspace(0, 30, 8, endpoint=False, dtype=f32),
                                       np.linspace(0, 45, 12, endpoint=False, dtype=f32))
    idx_h = torch.randint(0, 17550 * D, (4096,), generator=g)
    idx_e = torch.randint(0, 706 * D, (2048,), generator=g)
    res = dict(weight_seed=500, input_seed=501, idx_h=idx_h, idx_e=idx_e, sd_checksum=sd_checksum(bsd))
    for dt in (torch.bfloat16, torch.float32):
        b = blk.to(dt)
        oh, oe = b(hid.to(dt), enc.to(dt), temb.to(dt), image_rotary_emb=rope, vip_image_rotary_emb=rope,
                   vip_condition_rotary_emb=crope)
        res[str(dt)] = dict(h_samples=oh.flatten()[idx_h].clone(), e_samples=oe.flatten()[idx_e].clone(),
                            h_mean=oh.float().mean().item(), h_std=oh.float().std().item(),
                            h_absmax=oh.float().abs().max().item(), e_mean=oe.float().mean().item(),
                            e_std=oe.float().std().item(), e_absmax=oe.float().abs().max().item())
        print("full block", dt, res[str(dt)]["h_std"], res[str(dt)]["e_std"])
    torch.save(res, os.path.join(GOLD, "block_full.pt"))


VAE_TINY = dict(block_out_channels=(64, 128, 128, 128), layers_per_block=1, latent_channels=16, sample_height=64, sample_width=96)


@torch.no_grad()
def gen_vae():
    """Vendored VAE twin (longvgen/models/autoencoder_kl_cogvideox.py) on a tiny config: plain + tiled encode/decode.
    CogVideoXDownsample3D / Upsample3D / DiagonalGaussianDistribution come from the shim (restated, unpinned)."""
    from oracle import vae_ref as V
    rv = load_ref_module("longvgen/models/autoencoder_kl_cogvideox.py", "ref_vae")
    cfg = VAE_TINY
    vae = rv.AutoencoderKLCogVideoX(in_channels=3, out_channels=3, block_out_channels=cfg["block_out_channels"], latent_channels=16,
                                    layers_per_block=1, sample_height=64, sample_width=96, temporal_compression_ratio=4)
    sd = V.make_state_dict(cfg, seed=600)
    vae.load_state_dict(sd, strict=True)
    vae.eval()
    g = torch.Generator().manual_seed(601)
    x = torch.rand(1, 3, 17, 64, 96, generator=g) * 2 - 1
    z5 = torch.randn(1, 16, 5, 8, 12, generator=g)
    z13 = torch.randn(1, 16, 13, 8, 12, generator=g)
    idx = torch.randint(0, 3 * 17 * 64 * 96, (4096,), generator=g)

    def samp(t):
        f = t.flatten()
        return dict(shape=tuple(t.shape), samples=f[idx % f.numel()].clone(), mean=t.mean().item(), std=t.std().item(), absmax=t.abs().max().item())
    out = dict(cfg=cfg, weight_seed=600, input_seed=601, idx=idx, sd_checksum=sd_checksum(sd))
    out["encode_plain"] = vae._encode(x).clone()
    out["decode_plain"] = samp(vae._decode(z5).sample)
    vae.enable_tiling()
    out["encode_tiled"] = vae._encode(x).clone()
    out["decode_tiled"] = samp(vae.tiled_decode(z13).sample)
    # building blocks for kernel-level parity: causal conv across two calls (cache), spatial norm, resnet
    conv = rv.CogVideoXCausalConv3d(64, 128, 3)
    xs = torch.randn(1, 64, 5, 6, 10, generator=g)
    y1 = conv(xs[:, :, :3]); y2 = conv(xs[:, :, 3:])
    out["causal_conv"] = dict(w=conv.conv.weight.detach().clone(), b=conv.conv.bias.detach().clone(), x=xs, y=torch.cat([y1, y2], 2))
    torch.save(out, os.path.join(GOLD, "vae_tiny.pt"))
    print("vae_tiny.pt", out["encode_tiled"].shape, out["decode_tiled"]["shape"])


@torch.no_grad()
def gen_vae_t26():
    """tiled_decode over TWO 13-frame latent chunks (autoencoder_kl_cogvideox.py:1313-1336): the reference restarts its frame batching every 13
    frames and carries the conv cache across the chunks of a tile.  Same tiny VAE / weights as gen_vae; separate file so that vae_tiny.pt stays as it is."""
    from oracle import vae_ref as V
    rv = load_ref_module("longvgen/models/autoencoder_kl_cogvideox.py", "ref_vae_t26")
    cfg = VAE_TINY
    vae = rv.AutoencoderKLCogVideoX(in_channels=3, out_channels=3, block_out_channels=cfg["block_out_channels"], latent_channels=16,
                                    layers_per_block=1, sample_height=64, sample_width=96, temporal_compression_ratio=4)
    vae.load_state_dict(V.make_state_dict(cfg, seed=600), strict=True)
    vae.eval()
    vae.enable_tiling()
    g = torch.Generator().manual_seed(611)
    z26 = torch.randn(1, 16, 26, 8, 12, generator=g)
    calls = []
    dec = vae.decoder
    orig = dec.forward

    def rec(z, *a, **k):
        calls.append(z.shape[2])
        return orig(z, *a, **k)
    dec.forward = rec
    y = vae.tiled_decode(z26).sample
    idx = torch.randint(0, y.numel(), (8192,), generator=g)
    out = dict(cfg=cfg, weight_seed=600, input_seed=611, shape=tuple(y.shape), idx=idx, samples=y.flatten()[idx].clone(), frames_per_decoder_call=calls,
               mean=y.mean().item(), std=y.std().item())
    torch.save(out, os.path.join(GOLD, "vae_tiled_decode_t26.pt"))
    print("vae_tiled_decode_t26.pt", out["shape"], calls[:14])


RESAMPLER_TINY = dict(dim=128, depth=2, dim_head=64, heads=2, num_height_queries=2, num_width_queries=3, num_temporal_queries=4,
                      embedding_dim=128, output_dim=128, ff_mult=4, max_height_seq_len=4, max_width_seq_len=6, max_temporal_seq_len=13)


@torch.no_grad()
def gen_resampler():
    """Reference Resampler (longvgen/video_ipadapter/resampler.py) on a tiny config, fp32 and bf16, with both RoPE tables."""
    from oracle import resampler_ref as RR
    rs = load_ref_module("longvgen/video_ipadapter/resampler.py", "ref_resampler")
    cfg = RESAMPLER_TINY
    m = rs.Resampler(**cfg)
    sd = RR.make_state_dict(cfg, seed=700)
    m.load_state_dict(sd, strict=True)
    m.eval()
    g = torch.Generator().manual_seed(701)
    x = torch.randn(1, 13, 24, 128, generator=g)
    f32 = np.float32
    img = get_3d_rotary_pos_embed_v2(64, np.arange(13, dtype=f32), np.arange(4, dtype=f32), np.arange(6, dtype=f32))
    smp = get_3d_rotary_pos_embed_v2(64, np.linspace(1000, 1013, 4, endpoint=False, dtype=f32), np.linspace(0, 4, 2, endpoint=False, dtype=f32),
                                     np.linspace(0, 6, 3, endpoint=False, dtype=f32))
    out = dict(cfg=cfg, weight_seed=700, input_seed=701, sd_checksum=sd_checksum(sd))
    out["fp32"] = m(x, image_rotary_emb=img, sampling_rotary_emb=smp).clone()
    out["bf16"] = m.to(torch.bfloat16)(x.bfloat16(), image_rotary_emb=img, sampling_rotary_emb=smp).clone()
    # batch 2 (the gate-of-ones residual epilogue must be batch invariant) and the optional PCA low-rank filter (resampler.py:201-207,
    # 230-237) exactly as gen.yaml sets it up: a pca.PCA fitted by the reference's own class, pickled whole, loaded by set_pca(path)
    import tempfile
    refpca = load_ref_module("pca.py", "pca")
    sys.modules["pca"] = refpca
    torch.serialization.add_safe_globals([refpca.PCA])
    x2 = torch.randn(2, 13, 24, 128, generator=g)
    pca = refpca.PCA(n_components=None).fit(torch.randn(800, 128, generator=g) * torch.linspace(2.0, 0.1, 128))
    out["pca_mean"], out["pca_components"] = pca.mean_.clone(), pca.components_.clone()
    for tag, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        m2 = rs.Resampler(**cfg)
        m2.load_state_dict(sd, strict=True)
        m2 = m2.eval().to(dt)
        out[tag + "_b2"] = m2(x2.to(dt), image_rotary_emb=img, sampling_rotary_emb=smp).clone()
        with tempfile.TemporaryDirectory() as td:
            torch.save(pca, os.path.join(td, "pca.pt"))
            m2.set_pca(os.path.join(td, "pca.pt"), device="cpu")
        out[tag + "_b2_pca"] = m2(x2.to(dt), image_rotary_emb=img, sampling_rotary_emb=smp).clone()
    torch.save(out, os.path.join(GOLD, "resampler_tiny.pt"))
    print("resampler_tiny.pt", tuple(out["fp32"].shape), tuple(out["fp32_b2_pca"].shape))



@torch.no_grad()
def gen_t2to():
    """Run the reference LongVGenCogVideoXPipeline.__call__ (pipeline_cogvideox_t2to.py:566-912) on CPU with a tiny plain
    patch-1 DiT: 6 DPM steps, dynamic CFG, CPU generator (initial latents + every SDE draw), real PCA fitted with the
    reference's pca.PCA on seeded data.  Stored: inputs, the initial latents, every gaussian in draw order, the sampled
    latents before the tail, the final condensed tokens, and the 16 PCA rows / mean the tail actually uses."""
    import contextlib
    import tempfile
    ref = load_ref_module("longvgen/pipeline/pipeline_cogvideox_t2to.py", "ref_t2to")
    refpca = load_ref_module("pca.py", "pca")
    sys.modules["pca"] = refpca                       # the pipeline torch.load()s a pickled pca.PCA
    torch.serialization.add_safe_globals([refpca.PCA])   # torch >= 2.6 defaults weights_only=True; the reference predates it
    cfg = dict(TINY, patch_size=1)
    H, W, nfc, chunks, steps = 2, 3, 4, 3, 6

    class Pipe(ref.LongVGenCogVideoXPipeline):
        def __init__(self, transformer, scheduler):      # skip DiffusionPipeline.register_modules / T5 / VAE
            self.transformer, self.scheduler = transformer, scheduler
        _execution_device = property(lambda self: torch.device("cpu"))

        @contextlib.contextmanager
        def progress_bar(self, total=None):
            yield types.SimpleNamespace(update=lambda *a: None)

        def maybe_free_model_hooks(self):
            pass

    results = {}
    for dt in (torch.float32, torch.bfloat16):
        m = CogVideoXTransformer3DModel(num_attention_heads=2, attention_head_dim=64, num_layers=2, in_channels=16, out_channels=16,
                                        text_embed_dim=TINY["text_embed_dim"], time_embed_dim=TINY["time_embed_dim"], sample_width=W,
                                        sample_height=H, sample_frames=49, use_rotary_positional_embeddings=True, max_text_seq_length=8,
                                        patch_size=1, use_output_projection=True)
        sd = O.make_state_dict(cfg, seed=800)
        m.load_state_dict(sd, strict=True)
        m = m.eval().to(dt)
        sched = CogVideoXDPMScheduler(prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=1.0,
                                      timestep_spacing="trailing")
        g = torch.Generator().manual_seed(801)
        prompt = torch.randn(1, 8, TINY["text_embed_dim"], generator=g).to(dt)
        negative = torch.randn(1, 8, TINY["text_embed_dim"], generator=g).to(dt)
        mean = torch.randn(1, 32, generator=g) * 0.5
        std = torch.rand(1, 32, generator=g) + 0.5
        pca = refpca.PCA(n_components=None).fit(torch.randn(3200, 3072, generator=g) * torch.linspace(2.0, 0.1, 3072))
        draws = []
        real_randn = ref.randn_tensor

        def spy(shape, generator=None, device=None, dtype=None, layout=None):
            t = real_randn(shape, generator=generator, device=device, dtype=dtype)
            draws.append(t.clone())
            return t
        ref.randn_tensor = spy
        sched_mod = sys.modules[CogVideoXDPMScheduler.__module__]
        sched_randn = sched_mod.randn_tensor
        sched_mod.randn_tensor = spy
        taps = {}
        pipe = Pipe(m, sched)
        with tempfile.TemporaryDirectory() as td:
            torch.save(mean, os.path.join(td, "mean.pt")); torch.save(std, os.path.join(td, "std.pt")); torch.save(pca, os.path.join(td, "pca.pt"))
            # the tail overwrites `latents`; keep the sampled ones through the (reference-supported) step-end callback
            def cb(pp, i, t, kw):
                taps["latents"] = kw["latents"].clone()
                return {}
            out = pipe(prompt_embeds=prompt, negative_prompt_embeds=negative, height=H, width=W, num_frames_per_chunk=nfc,
                       num_chunks=chunks, num_inference_steps=steps, use_dynamic_cfg=True, guidance_scale=6.0,
                       generator=torch.Generator().manual_seed(802), longvgen_mean=os.path.join(td, "mean.pt"),
                       longvgen_std=os.path.join(td, "std.pt"), longvgen_pca=os.path.join(td, "pca.pt"),
                       callback_on_step_end=cb, callback_on_step_end_tensor_inputs=["latents"]).frames
        ref.randn_tensor = real_randn
        sched_mod.randn_tensor = sched_randn
        results[str(dt)] = dict(prompt=prompt, negative=negative, init_latents=draws[0], step_draws=draws[1:], sampled=taps["latents"],
                                frames=out.clone(), timesteps=sched.timesteps.clone())
        if dt == torch.float32:
            common = dict(mean=mean, std=std, pca_mean=pca.mean_.clone(), pca_components16=pca.components_[:16].clone(),
                          sd_checksum=sd_checksum(sd))
    torch.save(dict(cfg=cfg, weight_seed=800, input_seed=801, gen_seed=802, H=H, W=W, nfc=nfc, chunks=chunks, steps=steps,
                    guidance_scale=6.0, **common, cases=results), os.path.join(GOLD, "t2to_tiny.pt"))
    print("t2to_tiny.pt", tuple(results[str(torch.float32)]["frames"].shape), len(results[str(torch.float32)]["step_draws"]), "draws")


@torch.no_grad()
def gen_base_stage(variant=False):
    """variant=True -> base_stage_dyn_sep.pt: the same run with use_dynamic_cfg + use_separate_guidance (guidance_scale_img 4.0, 16 steps) — the
    branch in which the reference re-assigns its LOCAL `guidance_scale_img = 1 + guidance_scale_img * ramp` on every step (:1257), so the image
    weight compounds from step to step and the compounded value is what the stage exports to the FIFO driver (:1336).  The Resampler stub returns
    a fixed seeded tensor there (the "tokens of an all-zero video" of the third CFG branch), zeros in the default fixture.
    SURVEY §8c G11: the reference's OWN `MPFIFOVideoIPAdapterCogVideoXPipeline.__call__` (pipeline_cogvideox_mp_fifo.py:837-1344, the
    52-step base stage that seeds the FIFO queue, :1186-1307) run on CPU with the tiny To2V DiT.  Only the constructor is bypassed
    (DiffusionPipeline.register_modules / T5 / VAE weights): `vae.encode` and the Resampler are touched solely by the zero-video "uncond"
    branch of vae_encode_image (:618-646), whose result the reference discards unless use_separate_guidance — they are stubs returning
    zeros of the right shape.  Every gaussian (initial latents + per-step SDE draws) is recorded in draw order.  Stored: inputs, draws,
    fifo_latents, fifo_old_pred_original_sample, orig_latents, the position grids and the RoPE table the pipeline built."""
    import contextlib
    ref = load_ref_module("longvgen/pipeline/pipeline_cogvideox_mp_fifo.py", "ref_pipe_mp_fifo")
    H, W, nf, T, chunks = 4, 6, 13, (16 if variant else 52), 2
    unc_tok = torch.randn(1, 4, 128, 2, 3, generator=torch.Generator().manual_seed(903)) if variant else torch.zeros(1, 4, 128, 2, 3)
    rq = types.SimpleNamespace(num_temporal_queries=4, num_height_queries=2, num_width_queries=3, max_temporal_seq_len=13, max_height_seq_len=2,
                               max_width_seq_len=3)

    class _Resampler:
        config = rq

        def __call__(self, x, image_rotary_emb=None, sampling_rotary_emb=None):
            return unc_tok.to(x.dtype).expand(x.shape[0], -1, -1, -1, -1).clone()

    class _Vae:
        config = types.SimpleNamespace(scaling_factor=1.15258426)

        def __init__(self, dt):
            self.p = torch.nn.Parameter(torch.zeros(1, dtype=dt))

        def parameters(self):
            return iter([self.p])

        def encode(self, video):
            z = torch.zeros(video.shape[0], 16, (video.shape[2] - 1) // 4 + 1, H, W, dtype=video.dtype)
            return types.SimpleNamespace(latent_dist=types.SimpleNamespace(sample=lambda: z))

    class Pipe(ref.MPFIFOVideoIPAdapterCogVideoXPipeline):
        def __init__(self, transformer, scheduler, dt):
            self.transformer, self.scheduler = transformer, scheduler
            self.resampler, self.image_encoder, self.vae = _Resampler(), None, _Vae(dt)
            self.vae_scale_factor_spatial, self.vae_scale_factor_temporal, self.vae_scaling_factor_image = 8, 4, 1.15258426
        _execution_device = property(lambda self: torch.device("cpu"))

        @contextlib.contextmanager
        def progress_bar(self, total=None):
            yield types.SimpleNamespace(update=lambda *a: None, set_description=lambda *a: None)

        def maybe_free_model_hooks(self):
            pass

    cases = {}
    for dt in (torch.float32, torch.bfloat16):
        m, sd = tiny_model(900)
        m = m.to(dt)
        sched = make_sched()
        g = torch.Generator().manual_seed(901)
        prompt = torch.randn(1, 8, TINY["text_embed_dim"], generator=g).to(dt)
        negative = torch.randn(1, 8, TINY["text_embed_dim"], generator=g).to(dt)
        emb = torch.randn(1, 4 * chunks, 128, 2, 3, generator=g).to(dt)               # what the T2To stage / Resampler hands over (:611-616)
        draws = []
        sched_mod = sys.modules[CogVideoXDPMScheduler.__module__]
        real_pipe_randn, real_sched_randn = ref.randn_tensor, sched_mod.randn_tensor

        def spy(shape, generator=None, device=None, dtype=None, layout=None):
            t = real_sched_randn(shape, generator=generator, device=device, dtype=dtype)
            draws.append(t.clone())
            return t
        ref.randn_tensor = sched_mod.randn_tensor = spy
        try:
            out = Pipe(m, sched, dt)(prompt_embeds=prompt, negative_prompt_embeds=negative, image_embeddings=emb, height=8 * H, width=8 * W,
                                     num_frames_per_chunk=49, max_num_chunks=chunks, max_num_chunks_wo_fifo=1, num_inference_steps=T,
                                     guidance_scale=6.0, use_dynamic_cfg=bool(variant), generator=torch.Generator().manual_seed(902), vip_scale=[0.6],
                                     **(dict(use_separate_guidance=True, guidance_scale_img=4.0) if variant else {}),
                                     sampling_mode="fifo", sampling_params=dict(use_adaptive_padding=True, num_partitions=4), output_type="latent",
                                     video_ipadapter_start_frame_idx=1000)
        finally:
            ref.randn_tensor, sched_mod.randn_tensor = real_pipe_randn, real_sched_randn
        assert out.fifo_latents.shape == (1, T, 16, H, W) and out.nf_per_chunk == nf and out.num_frames == chunks * nf
        cases[str(dt)] = dict(prompt=prompt, negative=negative, emb_in=emb, init_latents=draws[0], step_draws=draws[1:],
                              fifo_latents=out.fifo_latents.clone(), fifo_old=[None if o is None else o.clone() for o in out.fifo_old_pred_original_sample],
                              orig_latents=out.orig_latents.clone(), image_embeddings=out.image_embeddings.clone(), prompt_embeds=out.prompt_embeds.clone())
        if variant:
            cases[str(dt)].update(guidance_scale_out=float(out.guidance_scale), guidance_scale_img_out=float(out.guidance_scale_img))
        if dt == torch.float32:
            common = dict(timesteps=out.timesteps.clone(), image_rotary_emb=tuple(t.clone() for t in out.image_rotary_emb),
                          vip_image_rotary_grid=[np.asarray(a).copy() for a in out.vip_image_rotary_grid],
                          vip_condition_rotary_grid=[np.asarray(a).copy() for a in out.vip_condition_rotary_grid],
                          vip_nf_per_chunk=out.vip_nf_per_chunk, sd_checksum=sd_checksum(sd))
    name = "base_stage_dyn_sep.pt" if variant else "base_stage_tiny.pt"
    torch.save(dict(weight_seed=900, input_seed=901, gen_seed=902, H=H, W=W, chunks=chunks, steps=T, guidance_scale=6.0, vip_scale=[0.6],
                    **(dict(guidance_scale_img=4.0, use_dynamic_cfg=True, use_separate_guidance=True, unc_tok=unc_tok) if variant else {}),
                    **common, cases=cases), os.path.join(GOLD, name))
    c = cases[str(torch.float32)]
    print(name, tuple(c["fifo_latents"].shape), len(c["step_draws"]), "step draws", tuple(c["image_embeddings"].shape))


@torch.no_grad()
def gen_vae_geometry():
    """the tile / temporal-batch geometry of the reference VAE object AT 480 x 720 x 49 (13 x 60 x 90 latent), read from the
    reference's own loops instead of typed in: the vendored AutoencoderKLCogVideoX is built at sample 480 x 720 with 4 (narrow) blocks, tiling
    enabled (autoencoder_kl_cogvideox.py:1028-1062), and its `encoder` / `decoder` are replaced by recorders that log the exact slice
    tiled_encode (:1225-1259) / tiled_decode (:1303-1336) hands them and return zeros of the right output shape; blend_v / blend_h are
    wrapped to log the extents.  Stored: attributes, per-call slices (frame range, tile shape, in order), blend extents, output shapes."""
    rv = load_ref_module("longvgen/models/autoencoder_kl_cogvideox.py", "ref_vae_geom")
    vae = rv.AutoencoderKLCogVideoX(in_channels=3, out_channels=3, block_out_channels=(32, 32, 32, 32), latent_channels=16, layers_per_block=1,
                                    sample_height=480, sample_width=720, temporal_compression_ratio=4)
    vae.eval()
    vae.enable_tiling()
    attrs = {k: getattr(vae, k) for k in ("tile_sample_min_height", "tile_sample_min_width", "tile_latent_min_height", "tile_latent_min_width",
                                          "tile_overlap_factor_height", "tile_overlap_factor_width", "num_latent_frames_batch_size",
                                          "num_sample_frames_batch_size")}
    log = {"encode": [], "decode": [], "blend_v": [], "blend_h": []}

    class Enc(torch.nn.Module):
        def forward(self, x):
            b, c, t, h, w = x.shape
            log["encode"].append((t, h, w))
            return torch.zeros(b, 32, (t - 1) // 4 + 1 if t % 2 else t // 4, h // 8, w // 8)

    class Dec(torch.nn.Module):
        def forward(self, z):
            b, c, t, h, w = z.shape
            log["decode"].append((t, h, w))
            return torch.zeros(b, 3, 4 * (t - 1) + 1 if t % 2 else 4 * t, 8 * h, 8 * w)
    vae.encoder, vae.decoder = Enc(), Dec()
    bv, bh = vae.blend_v, vae.blend_h
    vae.blend_v = lambda a, b, e: (log["blend_v"].append((e, tuple(a.shape[-2:]), tuple(b.shape[-2:]))), bv(a, b, e))[1]
    vae.blend_h = lambda a, b, e: (log["blend_h"].append((e, tuple(a.shape[-2:]), tuple(b.shape[-2:]))), bh(a, b, e))[1]
    # frame ranges: mark every frame with its index so the recorder's slices can be mapped back
    x = torch.zeros(1, 3, 49, 480, 720)
    z = torch.zeros(1, 16, 13, 60, 90)
    enc = vae.tiled_encode(x)
    dec = vae.tiled_decode(z).sample
    out = dict(attrs=attrs, encode_calls=list(log["encode"]), decode_calls=list(log["decode"]), blend_v=list(log["blend_v"]), blend_h=list(log["blend_h"]),
               encode_out=tuple(enc.shape), decode_out=tuple(dec.shape))
    # the temporal batch boundaries themselves: re-run with a recorder that sees WHICH frames arrive (frame index written into the data)
    fr = {"encode": [], "decode": []}
    x = torch.arange(49, dtype=torch.float32).view(1, 1, 49, 1, 1).expand(1, 3, 49, 480, 720)
    z = torch.arange(13, dtype=torch.float32).view(1, 1, 13, 1, 1).expand(1, 16, 13, 60, 90)

    class Enc2(Enc):
        def forward(self, x):
            fr["encode"].append((int(x[0, 0, 0, 0, 0]), int(x[0, 0, -1, 0, 0]) + 1))
            return super().forward(x)

    class Dec2(Dec):
        def forward(self, z):
            fr["decode"].append((int(z[0, 0, 0, 0, 0]), int(z[0, 0, -1, 0, 0]) + 1))
            return super().forward(z)
    log["encode"], log["decode"] = [], []
    vae.encoder, vae.decoder = Enc2(), Dec2()
    vae.tiled_encode(x); vae.tiled_decode(z)
    out["encode_frames"], out["decode_frames"] = fr["encode"], fr["decode"]
    torch.save(out, os.path.join(GOLD, "vae_geometry_480x720.pt"))
    print("vae_geometry_480x720.pt", attrs, len(out["encode_calls"]), len(out["decode_calls"]), out["encode_out"], out["decode_out"])


@torch.no_grad()
def gen_fifo_worker():
    """The reference worker body `fifo_onestep_per_gpu` (cogvideo_sampling_mp_fifo.py:408-579) run IN PROCESS (plain queue.Queue: one work item,
    then None) for the guidance / prediction branches the shipped configs do not select but the yaml can (VERDICT r2 missing #4):
    `use_separate_guidance` (3-way batch + guidance_scale_img, :493-497, 528-530), `use_dynamic_cfg` (:519-527), both together, and an
    epsilon-prediction scheduler — on a head window (prev_t = -1 frames) and a tail window (t = 999, next_t -> None), fp32 and bf16."""
    fifo = load_ref_module("longvgen/fifo_sampling/cogvideo_sampling_mp_fifo.py", "ref_fifo_worker")
    H, W, nf, T = 4, 6, 13, 52
    f32 = np.float32
    g = torch.Generator().manual_seed(811)
    ts = make_sched().timesteps.tolist()
    lvl = [ts[-1]] * 6 + ts[::-1]                       # per queue position, ascending noise
    cases = []

    class Pipe:                                          # the attributes the worker touches; guidance_scale is a property on the real pipeline
        def __init__(self, m, sched, gs):
            self.transformer, self.scheduler, self._guidance_scale = m, sched, gs

        @property
        def guidance_scale(self):
            return self._guidance_scale

        def _prepare_vip_rotary_positional_embeddings(self, grid_t, grid_h, grid_w, device):
            return get_3d_rotary_pos_embed_v2(64, grid_t, grid_h, grid_w)

    variants = [("separate", True, False, "v_prediction"), ("dynamic", False, True, "v_prediction"), ("separate_dynamic", True, True, "v_prediction"),
                ("epsilon_static", False, False, "epsilon"), ("no_cfg", False, False, "v_prediction"), ("no_cfg_dynamic", False, True, "v_prediction")]
    for name, sep, dyn, ptype in variants:
        cfg_on = not name.startswith("no_cfg")            # do_classifier_free_guidance=False (:497-498): batch of one, model output = prediction
        for dt in (torch.float32, torch.bfloat16):
            # alphas_cumprod[999] = 0 under zero terminal SNR and epsilon prediction divides by sqrt(alpha) (inf in the reference too): the epsilon
            # tail window stops one position short of t = 999 (its last frame then has t_back = 999: r = inf, d = x0)
            for start in ((0, 44) if ptype == "epsilon" else (0, 45)):
                m, sd = tiny_model(820)
                m = m.to(dt)
                sched = CogVideoXDPMScheduler(prediction_type=ptype, rescale_betas_zero_snr=True, snr_shift_scale=1.0, timestep_spacing="trailing")
                sched.set_timesteps(T)
                nb = (3 if sep else 2) if cfg_on else 1
                pipe = Pipe(m, sched, 6.0)
                t = torch.tensor(lvl[start:start + nf])
                prev_t = torch.tensor([(-1 if q <= 6 else lvl[q - 1]) for q in range(start, start + nf)])
                next_t = torch.tensor([(lvl[q + 1] if q + 1 < len(lvl) else -1) for q in range(start, start + nf)])
                lat = torch.randn(1, nf, 16, H, W, generator=g).to(dt)
                old = [torch.randn(1, 1, 16, H, W, generator=g).to(dt) if int(next_t[j]) > 0 else None for j in range(nf)]
                prompt = torch.randn(nb, 8, TINY["text_embed_dim"], generator=g).to(dt)
                emb = torch.randn(nb, 5, 128, 2, 3, generator=g).to(dt)
                grid_t = np.arange(nf, dtype=f32) + f32(start)
                grid_h, grid_w = np.arange(H // 2, dtype=f32), np.arange(W // 2, dtype=f32)
                cond_t = np.linspace(1000, 1016.25, 5, dtype=f32)
                cond_h, cond_w = np.linspace(0, H // 2, 2, endpoint=False, dtype=f32), np.linspace(0, W // 2, 3, endpoint=False, dtype=f32)
                rope = get_3d_rotary_pos_embed(64, ((0, 0, 0), (nf, H // 2, W // 2)), (nf, H // 2, W // 2))
                qin, qout = queue.Queue(), queue.Queue()
                qin.put((0, 0, 0, start, start + 6, start + nf, start + nf, t.clone(), prev_t.clone(), next_t.clone(), lat.clone(), list(old),
                         grid_t, grid_h, grid_w, cond_t, cond_h, cond_w, emb.clone(), []))
                qin.put(None)
                seed = 9000 + len(cases)
                torch.manual_seed(seed)
                fifo.fifo_onestep_per_gpu(0, qin, qout, pipe, prompt, rope, T, cfg_on, sep, 6.0, 4.0, dyn, None)
                (_, _, _, _, _, out_lat, out_x0, _) = qout.get()
                cases.append(dict(name=name, separate=sep, dynamic=dyn, cfg=cfg_on, prediction_type=ptype, dtype=str(dt), start=start, t=t, prev_t=prev_t, next_t=next_t,
                                  latents=lat, old=old, prompt=prompt, image_embeddings=emb, grid_t=grid_t, grid_h=grid_h, grid_w=grid_w, cond_t=cond_t,
                                  cond_h=cond_h, cond_w=cond_w, rng_seed=seed, guidance_scale=6.0, guidance_scale_img=4.0, out_latents=out_lat.clone(),
                                  out_x0=[x.clone() for x in out_x0]))
    torch.save(dict(weight_seed=820, H=H, W=W, cases=cases), os.path.join(GOLD, "fifo_worker_variants.pt"))
    print("fifo_worker_variants.pt", len(cases), "cases", {c["name"]: str(c["out_latents"].dtype) for c in cases})


@torch.no_grad()
def gen_train():
    """Training-loss arithmetic of the reference (VERDICT r2 missing #1): the scheduler's `add_noise` and `get_velocity`
    (scheduling_dpm_cogvideox.py:470-495, 521-538) on the REFERENCE class, and the weighted loss evaluated exactly as the training script does
    (train_cogvideo_to2v.py:1990-2010: flatten per-frame timesteps, get_velocity, weights = 1/(1 - alphas_cumprod[t]), per-item mean, batch
    mean) on fixed tensors — fp32 and bf16, timesteps [B] (the ordinary branch, :1815-1823) and per-frame [B, F] (the FIFO-style branch,
    :1773-1795, incl. how it reshapes around add_noise)."""
    from einops import rearrange
    s = make_sched()
    g = torch.Generator().manual_seed(77)
    B, F, C, H, W = 2, 3, 4, 2, 3
    cases = []
    for dt in (torch.float32, torch.bfloat16):
        for per_frame in (False, True):
            model_input = torch.randn(B, F, C, H, W, generator=g).to(dt)
            noise = torch.randn(B, F, C, H, W, generator=g).to(dt)
            model_output = torch.randn(B, F, C, H, W, generator=g).to(dt)
            if per_frame:
                timesteps = torch.randint(0, 1000, (B * F,), generator=g).long()
                mi, nz = rearrange(model_input, "b f c h w -> (b f) c h w"), rearrange(noise, "b f c h w -> (b f) c h w")
                noisy = rearrange(s.add_noise(mi, nz, timesteps), "(b f) c h w -> b f c h w", b=B)
                timesteps = rearrange(timesteps, "(b f) -> b f", b=B)
            else:
                timesteps = torch.randint(0, 1000, (B,), generator=g).long()
                timesteps[0] = 999                                   # zero terminal SNR: alphas_cumprod = 0, weight exactly 1
                noisy = s.add_noise(model_input, noise, timesteps)
            # --- train_cogvideo_to2v.py:1990-2010 ---
            ts, mo, nmi, mit = timesteps, model_output, noisy, model_input
            if ts.ndim > 1:
                ts = rearrange(ts, "b f -> (b f)")
                mo, nmi, mit = (rearrange(t_, "b f c h w -> (b f) c h w") for t_ in (mo, nmi, mit))
            model_pred = s.get_velocity(mo, nmi, ts)
            alphas_cumprod = s.alphas_cumprod[ts]
            weights = 1 / (1 - alphas_cumprod)
            while len(weights.shape) < len(model_pred.shape):
                weights = weights.unsqueeze(-1)
            target = mit
            per_item = torch.mean((weights * (model_pred - target) ** 2).reshape(B, -1), dim=1)
            loss = per_item.mean()
            cases.append(dict(dtype=str(dt), per_frame=per_frame, model_input=model_input, noise=noise, model_output=model_output,
                              timesteps=timesteps, noisy=noisy.clone(), velocity=model_pred.clone(), per_item=per_item.clone(), loss=loss.clone()))
    torch.save(dict(alphas_cumprod=s.alphas_cumprod.clone(), cases=cases), os.path.join(GOLD, "train_loss.pt"))
    print("train_loss.pt", len(cases), "cases", [str(c["loss"].dtype) for c in cases])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full-block", action="store_true")
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    jobs = dict(dit=gen_dit_tiny, vip=gen_vip_processor, sched=gen_scheduler, fifo=gen_fifo, vae=gen_vae, resampler=gen_resampler, t2to=gen_t2to, base=gen_base_stage, base_dyn_sep=lambda: gen_base_stage(variant=True), vae_geom=gen_vae_geometry, vae_t26=gen_vae_t26, train=gen_train, fifo_worker=gen_fifo_worker)
    if a.only:
        jobs = {a.only: jobs.get(a.only, gen_full_block)}
    for k, fn in jobs.items():
        fn()
    if a.full_block and not a.only:
        gen_full_block()
