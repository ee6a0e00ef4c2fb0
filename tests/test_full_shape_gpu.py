"""BASELINE configs 2 and 5 at their FULL shapes under `-m gpu` (VERDICT r2 missing #5 / next #2c, #6): the 42-layer x 3072-wide To2V forward
over 18 256 tokens at CFG batch 2, and one training micro-step at the yaml's shapes.  No oracle can run these sizes in seconds, so the checks are
the size-independent properties: finite, run-to-run bitwise, every block a pure function of its input (re-running block i alone on the recorded
stream reproduces the recorded output bitwise), both softmax paths agree, both activation schedules give bitwise the same gradients.  Parity of
the arithmetic itself is established at tiny sizes / one full-width block against the reference's own outputs (test_dit_gpu.py, test_train_gpu.py)."""
import os
import sys

import numpy as np
import pytest
import torch
from conftest import measured

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel(a, b):
    a, b = a.float(), b.float()
    return measured(((a - b).norm() / (b.norm() + 1e-12)).item())


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


@pytest.mark.timeout(900)
def test_full_shape_to2v_forward_properties():
    """CogVideoX-5B To2V forward (cogvideox_transformer_3d.py:636-770) at the headline shape: B = 2, 13 x 60 x 90 latents = 17 550 video + 226 text
    + 480 condensed tokens, 42 layers, D = 3072, per-frame timesteps — the launch sequence bench.py times."""
    from tokensgen_amd import rope as R
    torch.cuda.empty_cache()
    model = _bench().build_model(DEV, 42)
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(2, 13, 16, 60, 90, generator=g, device=DEV).to(BF)
    prompt = (torch.randn(2, 226, 4096, generator=g, device=DEV) * 0.1).to(BF)
    emb = torch.nn.functional.layer_norm(torch.randn(1, 5, 8, 12, 3072, generator=g, device=DEV), (3072,))
    emb = emb.permute(0, 1, 4, 2, 3).to(BF).repeat(2, 1, 1, 1, 1).contiguous()
    ts = torch.tensor([[999 - 19 * k for k in range(13)]] * 2, device=DEV)
    f32 = np.float32
    rope = R.rope_3d_crop(64, (0, 0, 0), (13, 30, 45), (13, 30, 45))
    vr = R.rope_3d(64, np.arange(13, dtype=f32) + f32(26), np.arange(30, dtype=f32), np.arange(45, dtype=f32), device=DEV)
    cr = R.rope_3d(64, np.linspace(1026, 1042.25, 5, dtype=f32), np.linspace(0, 30, 8, endpoint=False, dtype=f32),
                   np.linspace(0, 45, 12, endpoint=False, dtype=f32), device=DEV)
    kw = dict(hidden_states=x, encoder_hidden_states=prompt, timestep=ts, image_rotary_emb=rope, vip_image_rotary_emb=vr,
              vip_condition_rotary_emb=cr, vip_encoder_hidden_states=emb, return_dict=False)
    # record the residual stream around every block of the first pass
    stream_in, stream_out, args = [], [], {}
    orig = model._run_block

    def spy(i, ws, *a):
        stream_in.append(ws.X.clone())
        orig(i, ws, *a)
        stream_out.append(ws.X.clone())
        args[i] = (ws, a)
    model._run_block = spy
    y1 = model(**kw)[0]
    model._run_block = orig
    assert y1.shape == x.shape and bool(torch.isfinite(y1).all())
    assert all(bool(torch.isfinite(s).all()) for s in stream_out)
    assert model.attn_path == "constant_shift"
    ws = next(iter(model._ws.values()))
    assert ws.retry.count() == 0                               # the constant-shift softmax stood in every workgroup of every layer
    # every block alone on its recorded input reproduces its recorded output bitwise (out of order: 41 first — no hidden state between layers)
    for i in [41, 0, 20] + list(range(1, 41, 3)):
        ws_i, a = args[i]
        ws_i.X.copy_(stream_in[i])
        orig(i, ws_i, *a)
        assert torch.equal(ws_i.X, stream_out[i]), i
    del stream_in, stream_out
    y2 = model(**kw)[0]
    assert torch.equal(y1, y2)                                 # run-to-run bitwise
    # the running-max softmax is the same function up to rounding, 42 layers deep
    model.attn_path = "running_max"
    y3 = model(**kw)[0]
    model.attn_path = "constant_shift"
    assert bool(torch.isfinite(y3).all()) and not torch.equal(y3, y1)
    assert _rel(y3, y1) < 3e-2                                  # measured 1.5e-2: two valid softmax kernels, 42 random-weight layers deep
    del model
    torch.cuda.empty_cache()


@pytest.mark.timeout(1500)
def test_full_shape_training_micro_step_properties():
    """BASELINE config 5 at the yaml's shapes (train_cogvideo_to2v.py:1721-2021; cogvideo_5b_vaevip_4x8x12_to2v.yaml): per_gpu_batch_size 2, 13 latent
    frames of 60 x 90, 226 text tokens, Resampler over two chunks of 17 550 tokens -> 480 vip tokens, 42 layers + Resampler = 1.97 B trainable
    parameters.  Finite loss / gradient norm; run-to-run bitwise gradients; the keep-activations schedule and the reference's per-block recompute
    give bitwise the same gradients."""
    from tokensgen_amd import optim, train
    from tokensgen_amd import rope as R
    from tokensgen_amd.scheduler import CogVideoXDPMScheduler
    torch.cuda.empty_cache()
    bench = _bench()
    model = bench.build_model(DEV, 42)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    tr = train.To2VTrainer(sd, 48, 42, patch_size=2, vip_scale=1.0)
    rsd, depth, heads = bench.build_resampler_sd(DEV)
    rt = train.ResamplerTrainer(rsd, depth=depth, heads=heads)
    params = {k: sd[k] for k in tr.trainable}
    params.update({"resampler." + k: v for k, v in rsd.items()})
    arena = optim.ParamArena(params, optim.arena_order(list(params), 42), DEV)
    tr.use_arena(arena); rt.use_arena(arena)
    del params
    n_clip = arena.prefix_elems(lambda n: not n.startswith("resampler."))
    opt = optim.AdamW(arena, lr=2e-4, betas=(0.9, 0.95), weight_decay=1e-4, max_grad_norm=1.0, clip_elems=n_clip)
    sched = CogVideoXDPMScheduler(prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=1.0, timestep_spacing="trailing")
    step = train.To2VTrainStep(tr, arena, opt, sched.alphas_cumprod.to(torch.float32), accumulation_steps=1000, resampler=rt)
    g = torch.Generator(device=DEV).manual_seed(7)
    B, nf, C, H, W = 2, 13, 16, 60, 90
    x0, noise = (torch.randn(B, nf, C, H, W, generator=g, device=DEV).to(BF) for _ in range(2))
    text = (torch.randn(B, 226, 4096, generator=g, device=DEV) * 0.1).to(BF)
    emb = (torch.randn(B, 2 * nf, 1350, 3072, generator=g, device=DEV) * 0.5).to(BF)
    f32 = np.float32
    rope = R.rope_3d_crop(64, (0, 0, 0), (nf, 30, 45), (nf, 30, 45))
    crope = R.rope_3d(64, np.linspace(1000, 1016.25, 5, dtype=f32), np.linspace(0, 30, 8, endpoint=False, dtype=f32), np.linspace(0, 45, 12, endpoint=False, dtype=f32))
    img = R.rope_3d(64, np.arange(13, dtype=f32), np.arange(30, dtype=f32), np.arange(45, dtype=f32))
    smp = R.rope_3d(64, np.linspace(1000, 1013, 4, endpoint=False, dtype=f32), np.linspace(0, 30, 8, endpoint=False, dtype=f32), np.linspace(0, 45, 12, endpoint=False, dtype=f32))
    ts = torch.tensor([137, 803])

    def run(budget):
        tr.activation_budget_bytes = budget
        arena.grad.zero_()
        loss, did = step.micro_step(x0, noise, ts, text, None, rope, rope, crope, image_embeddings=emb, emb_start_idx=[1, 2], resampler_ropes=(img, smp))
        assert not did
        return float(loss), arena.grad.clone()
    l1, g1 = run(48 * 2 ** 30)                   # ~9 blocks keep their activations, the rest recompute
    kept = tr.blocks_kept
    l2, g2 = run(48 * 2 ** 30)
    assert np.isfinite(l1) and l1 > 0 and bool(torch.isfinite(g1).all())
    gn = float(g1.double().norm())
    assert np.isfinite(gn) and gn > 0
    assert l1 == l2 and torch.equal(g1, g2)                     # run-to-run bitwise
    del g2
    l3, g3 = run(0)                              # the reference's schedule: every block recomputed in the backward
    assert 0 < kept < 42 and tr.blocks_kept == 0
    assert l3 == l1 and torch.equal(g3, g1)      # the two schedules are the same computation
    nz = float((g1 != 0).float().mean())
    assert nz > 0.9                              # every trainable tensor received a gradient
    del model, tr, rt, arena, g1, g3
    torch.cuda.empty_cache()
