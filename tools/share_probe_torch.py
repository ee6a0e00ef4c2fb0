#!/usr/bin/env python3
"""CONTROL for tools/race_hunt2.py: the same tiny To2V forward computed by PyTorch's OWN kernels (the oracle's torch code moved to the GPU: rocBLAS / hipBLASLt GEMMs, SDPA,
elementwise) — no kernel of this repository — repeated N times while other processes share the GPU.  If this also differs from itself now and then, the cause is the platform
(time-slicing several processes on one device), not a kernel of tokensgen_amd.     python tools/share_probe_torch.py N TAG"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dit_ref as O  # noqa: E402
N, tag = int(sys.argv[1]), sys.argv[2]
DEV, BF = torch.device("cuda", 0), torch.bfloat16
dt = torch.float32 if "--fp32" in sys.argv else BF
cfg = dict(num_attention_heads=2, attention_head_dim=64, num_layers=1, patch_size=2, time_embed_dim=128, text_embed_dim=64, in_channels=16, out_channels=16)
sd = {k: v.to(dt).to(DEV) for k, v in O.make_state_dict(cfg, 128, seed=31).items()}
g = torch.Generator().manual_seed(5)
H, W, nf = 4, 6, 13
f32 = np.float32
pre = "transformer_blocks.0"
hs = [torch.randn(2, nf * 2 * 3, 128, generator=g).to(dt).to(DEV) for _ in range(4)]
enc = torch.randn(2, 8 + 30, 128, generator=g).to(dt).to(DEV)
temb = torch.randn(2, nf, 128, generator=g).to(dt).to(DEV)
dev = lambda r: tuple(torch.as_tensor(t).to(DEV) for t in r)
rope = dev(O.rope_3d_crop(64, (0, 0, 0), (nf, H // 2, W // 2), (nf, H // 2, W // 2)))
vr = dev(O.rope_3d(64, np.arange(nf, dtype=f32) + f32(5), np.arange(H // 2, dtype=f32), np.arange(W // 2, dtype=f32)))
cr = dev(O.rope_3d(64, np.linspace(1000, 1016.25, 5, dtype=f32), np.linspace(0, H // 2, 2, endpoint=False, dtype=f32), np.linspace(0, W // 2, 3, endpoint=False, dtype=f32)))
with torch.no_grad():
    def fwd(h):
        a, b = O.block_forward(sd, pre, h, enc, temb, 2, 30, [0.6], rope, vr, cr)
        return torch.cat([a.reshape(-1), b.reshape(-1)])
    ref = [fwd(x).clone() for x in hs]
    torch.cuda.synchronize()
    bad = 0
    for r in range(N):
        y = fwd(hs[r % 4])
        if not torch.equal(y, ref[r % 4]):
            bad += 1
            d = (y != ref[r % 4]).nonzero()
            print(f"[{tag}] torch block forward {r}: {d.shape[0]} elements differ, max |diff| {float((y.float() - ref[r % 4].float()).abs().max()):.4g}", flush=True)
print(f"[{tag}] SHARE_PROBE_TORCH {bad} of {N} block forwards differed ({dt})")
