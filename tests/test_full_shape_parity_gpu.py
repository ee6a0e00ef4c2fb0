"""GPU: VALUE parity at the full BASELINE shapes (VERDICT r5 missing #2 / next #2) — the launches bench.py times, compared with fp32 numbers computed ON
THE GPU by the oracle's own torch code (the oracle stays the checker; nothing of it is measured or shipped):

 (i)   one To2V block forward at CFG batch 2 over 226 text + 17 550 video + 480 condensed tokens (the 96 (batch, head) pairs of the bench's attention launch:
       3 360 + 96 rider workgroups, key-split tail, constant shift) against `oracle.dit_ref.block_forward` in fp32 — cogvideox_transformer_3d.py:221-332;
 (ii)  `tg_attention_bwd_multi` at (2, 48, 17 776, 17 776) with the nk = 480 vip-key problem riding in the same launch (70 chained key blocks x 96 heads, the ordered
       dQ exchange) against fp32 autograd on one (batch, head) pair per XCD — autograd of attention_processor.py:2066-2125;
 (iii) one FULL-SIZE VAE tile over the whole clip — 13 latent frames of 30 x 45 = 49 frames of 240 x 360, six temporal batches, the 128-channel full-resolution
       stages at 8 frames x 240 x 360 — decode and encode against `oracle.vae_ref` in fp32 — autoencoder_kl_cogvideox.py:1085-1163.

The fp32 checkers need tens of GB (one 17 776^2 score matrix is 1.26 GB): they run head by head / chunk by chunk."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from oracle import dit_ref as O
from oracle import vae_ref as V

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def _sdpa_fp32_by_heads(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None):
    """F.scaled_dot_product_attention restated as softmax(q k^T / sqrt(d)) v in fp32, four heads at a time (a [4, 17776, 17776] fp32 block is 5 GB; all 96 at once are not
    available even in 288 GB once the softmax keeps its own copy)."""
    assert attn_mask is None and not is_causal and dropout_p == 0.0
    sc = (1.0 / math.sqrt(q.shape[-1])) if scale is None else scale
    out = torch.empty(q.shape[:-1] + (v.shape[-1],), dtype=q.dtype, device=q.device)
    for b in range(q.shape[0]):
        for h0 in range(0, q.shape[1], 4):
            s = torch.matmul(q[b, h0:h0 + 4], k[b, h0:h0 + 4].transpose(-1, -2)) * sc
            out[b, h0:h0 + 4] = torch.matmul(torch.softmax(s, dim=-1), v[b, h0:h0 + 4])
            del s
    return out


@pytest.mark.timeout(1500)
def test_bench_shape_block_forward_vs_fp32_oracle(parity, monkeypatch):
    """(i)  B = 2 with DIFFERENT items (conditional / unconditional halves of a CFG batch), per-frame timestep embeddings, the bench's rotary tables."""
    import block_runner
    from tokensgen_amd import rope as R
    from tokensgen_amd.transformer import CogVideoXTransformer3DModel
    torch.cuda.empty_cache()
    D, heads, Np = 3072, 48, 480
    cfg = dict(num_attention_heads=heads, attention_head_dim=64, num_layers=1, time_embed_dim=512)
    pre = "transformer_blocks.0"
    sd = O.make_state_dict(cfg, n_vip_dim=3072, seed=700)
    m = CogVideoXTransformer3DModel(num_attention_heads=heads, attention_head_dim=64, num_layers=1, time_embed_dim=512, text_embed_dim=4096,
                                    use_rotary_positional_embeddings=True, device=DEV)
    m.set_vip_layers(None, length=Np, func_type="1", scale=[0.6],
                     resampler_params=dict(output_dim=3072, num_height_queries=8, num_width_queries=12, num_temporal_queries=4))
    m.load_state_dict({k: v.to(BF) for k, v in sd.items()}, strict=False)
    g = torch.Generator().manual_seed(701)
    hid = torch.randn(2, 17550, D, generator=g).to(DEV, BF)
    enc = torch.randn(2, 226 + Np, D, generator=g).to(DEV, BF)
    temb = torch.randn(2, 13, 512, generator=g).to(DEV, BF)
    f32 = np.float32
    rope = R.rope_3d_crop(64, (0, 0, 0), (13, 30, 45), (13, 30, 45))
    vrope = R.rope_3d(64, np.arange(13, dtype=f32) + f32(26), np.arange(30, dtype=f32), np.arange(45, dtype=f32))
    crope = R.rope_3d(64, np.linspace(1026, 1042.25, 5, dtype=f32), np.linspace(0, 30, 8, endpoint=False, dtype=f32), np.linspace(0, 45, 12, endpoint=False, dtype=f32))
    oh, oe = block_runner.block_forward(m, 0, hid, enc, temb, rope, vrope, crope)
    assert m.attn_path == "constant_shift"
    retried = next(iter(m._ws.values())).retry.count()
    assert bool(torch.isfinite(oh).all()) and bool(torch.isfinite(oe).all())
    # fp32 oracle on the SAME bf16-rounded weights and inputs, on the GPU
    sd32 = {k: v.to(BF).float().to(DEV) for k, v in sd.items() if k.startswith(pre + ".")}
    dev = lambda r: tuple(torch.as_tensor(t).to(DEV, torch.float32) for t in r)
    monkeypatch.setattr(torch.nn.functional, "scaled_dot_product_attention", _sdpa_fp32_by_heads)
    with torch.no_grad():
        rh, re = O.block_forward(sd32, pre, hid.float(), enc.float(), temb.float(), heads, Np, [0.6], dev(rope), dev(vrope), dev(crope))
    monkeypatch.undo()
    parity(_rel(oh, rh), 1e-2, "block forward at B = 2 x 18 256 tokens, video rows, HIP bf16 vs fp32 oracle on the GPU (SURVEY 8c per-block bound)")
    parity(_rel(oe, re), 1e-2, "the same, text | vip rows")
    for b in range(2):                                   # each CFG half on its own (a batch-stride slip would hide in the joint norm)
        parity(_rel(oh[b], rh[b]), 1e-2, f"the same, video rows of batch item {b}")
    # sampled rows across the launch: first / last query tile, the key-split tail region, the vip rider's rows
    rows = torch.tensor([0, 1, 255, 256, 4095, 8191, 12287, 16383, 17295, 17549], device=DEV)
    parity(_rel(oh[:, rows], rh[:, rows]), 1.2e-2, "the same, ten sampled video rows")
    parity(_rel(oe[:, 226:], re[:, 226:]), 1e-2, "the same, the 480 vip rows (SDPA #3, the rider workgroups)")
    parity(float(retried), 1.0, "workgroups that left the constant-shift path (informative: 0)")
    del m, sd32, rh, re
    torch.cuda.empty_cache()


@pytest.mark.timeout(1500)
def test_bench_shape_attention_backward_vs_fp32_autograd(parity):
    """(ii)  The training step's backward call of one layer exactly as train.py issues it: the 17 776^2 problem on K rows prescaled by scale x log2 e in the real
    norm kernel (scale = ln 2), the forward's own log-sum-exp, and the vip-key problem (17 776 queries x 480 keys, scale 1/8, dK / dV accumulated) in the SAME call."""
    from tokensgen_amd import kernels as K
    torch.cuda.empty_cache()
    B, H, N1, NP = 2, 48, 17776, 480
    D = H * 64
    kscale = 0.125 * 1.4426950408889634
    gen = torch.Generator(device=DEV).manual_seed(93)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=gen, device=DEV, dtype=torch.float32) * sc).to(BF)
    qkv = rnd(B, N1, 3 * D, sc=1.5)
    w = lambda: (1.0 + 0.1 * torch.randn(64, generator=gen, device=DEV)).to(BF)
    bb = lambda: (0.1 * torch.randn(64, generator=gen, device=DEV)).to(BF)
    km, kws = torch.zeros(B, H, dtype=torch.float32, device=DEV), K.kmax_workspace(N1, H, B, DEV)
    K.qk_layernorm_rope_pair(qkv[:, :, :D], qkv[:, :, D:2 * D], H, w(), bb(), w(), bb(), 1e-6, k_scale=kscale, kmax=km, kmax_ws=kws)
    q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
    vt = K.transpose_v(v, H, 0, N1, torch.zeros(B, H, 64, (N1 + 63) // 64 * 64, dtype=BF, device=DEV))
    retry = K.AttnRetry(N1, 0, H, B, DEV)
    o1 = torch.zeros(B, N1, D, dtype=BF, device=DEV)
    _, lse1 = K.attention_lse(q, k, vt, N1, o1, H, 0.125, k_prescaled=True, kmax=km, retry=retry)
    # the vip-key problem: its own queries over the same rows, 480 keys
    qx, kv, vv = rnd(B, N1, D, sc=1.2), rnd(B, NP, D, sc=1.2), rnd(B, NP, D)
    vt2 = K.transpose_v(vv, H, 0, NP, torch.zeros(B, H, 64, (NP + 63) // 64 * 64, dtype=BF, device=DEV))
    o2 = torch.zeros(B, N1, D, dtype=BF, device=DEV)
    _, lse2 = K.attention_lse(qx, kv, vt2, NP, o2, H, 0.125)
    g1, g2 = rnd(B, N1, D), rnd(B, N1, D)
    dq2 = torch.empty(B, N1, D, dtype=torch.float32, device=DEV)
    dk2, dv2 = (torch.zeros(B, NP, D, dtype=torch.float32, device=DEV) for _ in range(2))
    st = K.BwdDeviceState.get(torch.device(DEV, torch.cuda.current_device()))

    def call():
        dk2.zero_(); dv2.zero_()
        return K.attention_bwd_multi([
            dict(q=q, k=k, v=v, o=o1, dout=g1, scale=math.log(2.0), lse=lse1),
            dict(q=qx, k=kv, v=vv, o=o2, dout=g2, scale=0.125, dq=dq2, dk=dk2, dv=dv2, accumulate=2, lse=lse2)], H)
    (dq, dk, dv), _ = call()
    K.attention_bwd_check(DEV)                       # no exchange poll gave up; every head's key blocks sat on one XCD
    assert st.one_kernel, "the device probe should offer the one-kernel form on an MI355X: this test is about ITS ordered dQ exchange"
    keep = [t.clone() for t in (dq, dk, dv, dq2, dk2, dv2)]
    (dqb, dkb, dvb), _ = call()
    assert all(torch.equal(a, b_) for a, b_ in zip(keep, (dqb, dkb, dvb, dq2, dk2, dv2))), "the backward is not run-to-run bitwise at the full shape"
    del keep
    # one (batch, head) pair per XCD (pair index % 8) + the last pair, fp32 autograd one head at a time
    pairs = [0, 9, 18, 27, 36, 45, 54, 63, 95]
    assert sorted({p % 8 for p in pairs}) == list(range(8))
    worst = {n: 0.0 for n in ("dq", "dk", "dv", "dq_vipkey", "dk_vipkey", "dv_vipkey")}
    for p in pairs:
        b, h = divmod(p, H)
        sl = slice(h * 64, h * 64 + 64)
        qf, kf, vf = (t[b, :, sl].float().clone().requires_grad_(True) for t in (q, k, v))
        (((torch.softmax(qf @ kf.t() * math.log(2.0), -1) @ vf)) * g1[b, :, sl].float()).sum().backward()
        worst["dq"] = max(worst["dq"], _rel(dq[b, :, sl], qf.grad)); worst["dk"] = max(worst["dk"], _rel(dk[b, :, sl], kf.grad))
        worst["dv"] = max(worst["dv"], _rel(dv[b, :, sl], vf.grad))
        del qf, kf, vf
        qf, kf, vf = (t[b, :, sl].float().clone().requires_grad_(True) for t in (qx, kv, vv))
        (((torch.softmax(qf @ kf.t() * 0.125, -1) @ vf)) * g2[b, :, sl].float()).sum().backward()
        worst["dq_vipkey"] = max(worst["dq_vipkey"], _rel(dq2[b, :, sl], qf.grad)); worst["dk_vipkey"] = max(worst["dk_vipkey"], _rel(dk2[b, :, sl], kf.grad))
        worst["dv_vipkey"] = max(worst["dv_vipkey"], _rel(dv2[b, :, sl], vf.grad))
        del qf, kf, vf
    for n, tol in (("dq", 5.5e-3), ("dk", 5.5e-3), ("dv", 4.5e-3), ("dq_vipkey", 5.5e-3), ("dk_vipkey", 5.5e-3), ("dv_vipkey", 4.5e-3)):
        parity(worst[n], tol, f"{n}: worst of 9 (batch, head) pairs (one per XCD) at 2 x 48 x 17776 x {'480' if 'vip' in n else '17776'}, HIP vs fp32 autograd on the GPU")
    # every OTHER head is at least finite and of the sampled heads' magnitude (a head skipped by the launch would be zeros or garbage)
    per_head = dq.view(B, N1, H, 64).float().pow(2).sum(dim=(1, 3)).sqrt()
    assert bool(torch.isfinite(per_head).all()) and float(per_head.min()) > 0.2 * float(per_head.median())
    torch.cuda.empty_cache()


FULL = dict(block_out_channels=(128, 256, 256, 512), layers_per_block=3, latent_channels=16, sample_height=480, sample_width=720)


@pytest.mark.timeout(2400)
def test_full_size_vae_tile_vs_fp32_oracle(parity):
    """(iii)  One tile of the 480 x 720 geometry over the WHOLE clip — every temporal batch, the carried causal cache across all six, the full-resolution 128-channel
    stages (halo kernel, narrow conv_out, 8-channel conv_in) at their real launch sizes — against the oracle in fp32 on the GPU (same bf16-rounded weights / inputs)."""
    from tokensgen_amd.vae import AutoencoderKLCogVideoX
    torch.cuda.empty_cache()
    sd = V.make_state_dict(FULL, seed=5, dtype=BF)
    vae = AutoencoderKLCogVideoX(device=DEV)
    vae.load_state_dict(sd)
    vae.enable_slicing()
    g = torch.Generator().manual_seed(8)
    z = (torch.randn(1, 16, 13, 30, 45, generator=g) / 1.15258426).to(BF).to(DEV)
    x = (torch.rand(1, 3, 49, 240, 360, generator=g) * 2 - 1).to(BF).to(DEV)
    d = vae.decode(z).sample
    h = vae.encode(x).latent_dist.parameters
    sd32 = {k: v.float().to(DEV) for k, v in sd.items()}
    with torch.no_grad():
        ref_d = V.decode(sd32, FULL, z.float(), tiling=False)
        ref_h = V.encode(sd32, FULL, x.float(), tiling=False)
    assert d.shape == ref_d.shape == (1, 3, 49, 240, 360) and h.shape == ref_h.shape == (1, 32, 13, 30, 45)
    parity(_rel(d, ref_d), 2.8e-2, "decode of one full-size tile over the whole clip (13 latent frames, 6 temporal batches) vs fp32 oracle on the GPU")
    parity(_rel(h, ref_h), 2.8e-2, "encode of one full-size tile over the whole clip (49 frames of 240 x 360) vs fp32 oracle on the GPU")
    for t0, t1 in ((0, 1), (1, 9), (41, 49)):           # first frame (replicated context), the first full batch, the last batch (cache carried five times)
        parity(_rel(d[:, :, t0:t1], ref_d[:, :, t0:t1]), 3.2e-2, f"decode, frames {t0}..{t1 - 1}")
    torch.cuda.empty_cache()
