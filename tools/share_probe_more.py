#!/usr/bin/env python3
"""GPU-sharing repeatability of the two paths tools/race_hunt_big.py does not cover (round 6): `vae` — the real-width VAE on a small tile (decode + encode, every conv / norm kernel family), `train` —
one real-width To2V block forward + backward on a short stream (attention backward, training elementwise kernels).  Each repeated N times against its first run.   python tools/share_probe_more.py MODE N TAG"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode, N, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]
DEV, BF = torch.device("cuda", 0), torch.bfloat16
g = torch.Generator().manual_seed(8)
if mode == "vae":
    from tokensgen_amd.vae import AutoencoderKLCogVideoX
    vae = AutoencoderKLCogVideoX(device=DEV).init_random(seed=1)
    z = (torch.randn(1, 16, 3, 30, 45, generator=g) / 1.15258426).to(BF).to(DEV)
    x = (torch.rand(1, 3, 9, 240, 360, generator=g) * 2 - 1).to(BF).to(DEV)

    def op():
        return [vae.decode(z).sample, vae.encode(x).latent_dist.parameters]
else:
    from oracle import dit_ref as O  # (weights only: diagnostic tool)
    from tokensgen_amd import train
    B, H, Nt, Fr, hw, Np = 1, 48, 16, 2, 192, 64
    Nv, D = Fr * hw, H * 64
    f32 = np.float32
    cfg = dict(num_attention_heads=H, attention_head_dim=64, num_layers=1, patch_size=2, time_embed_dim=512, text_embed_dim=64, in_channels=16, out_channels=16)
    pre = "transformer_blocks.0"
    sd = {k: v.to(BF).to(DEV).contiguous() for k, v in O.make_state_dict(cfg, n_vip_dim=128, seed=111, std=0.02).items() if k.startswith(pre + ".")}
    rnd = lambda *s: torch.randn(*s, generator=g).to(BF).to(DEV)
    hidden, enc, temb, Gh, Ge = rnd(B, Nv, D), rnd(B, Nt + Np, D), rnd(B, Fr, 512), rnd(B, Nv, D), rnd(B, Nt + Np, D)
    rope = O.rope_3d(64, np.arange(2, dtype=f32), np.arange(12, dtype=f32), np.arange(16, dtype=f32))
    vrope = O.rope_3d(64, np.arange(2, dtype=f32) + f32(3), np.arange(12, dtype=f32), np.arange(16, dtype=f32))
    crope = O.rope_3d(64, np.linspace(1000, 1016.25, 4, dtype=f32), np.arange(4, dtype=f32), np.arange(4, dtype=f32))
    blk = train.To2VBlockTrainer(sd, pre, H, Nt, Np, Fr, 1.0)

    def op():
        gh, ge = blk.forward(hidden, enc, temb, rope, vrope, crope)
        grads, dh, de = blk.backward(Gh, Ge)
        return [gh, ge, dh, de] + [grads[k] for k in sorted(grads)]
op()
ref = [t.clone() for t in op()]
torch.cuda.synchronize()
bad = 0
for r in range(N):
    out = op()
    d = [i for i, (a, b) in enumerate(zip(out, ref)) if not torch.equal(a, b)]
    if d:
        bad += 1
        if bad <= 5:
            print(f"[{tag}] {mode} repeat {r}: outputs {d} differ ({[int((out[i] != ref[i]).sum()) for i in d[:6]]} elements)", flush=True)
print(f"[{tag}] SHARE_PROBE_MORE {mode}: {bad} of {N} repeats differed")
