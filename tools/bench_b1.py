#!/usr/bin/env python3
"""Batch-1 against batch-2 forward of the two DiTs (round 6, for tools/project_scaling.py): under CFG parallelism (cfg_parallel.py) a rank of the base stage / the T2To
stage runs ONE CFG half per step.  To2V window shape (226 + 17 550 + 480 tokens, scalar timestep as in the base stage) and T2To shape (patch 1, 226 + 4 x chunks x 96
tokens).  Prints {"to2v": {"b2_ms", "b1_ms", "b1_over_b2"}, "t2to": {...}}."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tokensgen_amd import rope as R  # noqa: E402
from tokensgen_amd.transformer import CogVideoXTransformer3DModel  # noqa: E402

DEV = torch.device("cuda", 0)
BF = torch.bfloat16
f32 = np.float32


def timed(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


out = {}
m = bench.build_model(DEV, 42)
g = torch.Generator(device=DEV).manual_seed(3)
rope = R.rope_3d_crop(64, (0, 0, 0), (13, 30, 45), (13, 30, 45))
vr = R.rope_3d(64, np.arange(13, dtype=f32), np.arange(30, dtype=f32), np.arange(45, dtype=f32), device=DEV)
cr = R.rope_3d(64, np.linspace(1000, 1016.25, 5, dtype=f32), np.linspace(0, 30, 8, endpoint=False, dtype=f32), np.linspace(0, 45, 12, endpoint=False, dtype=f32), device=DEV)
rec = {}
for B in (2, 1):
    x = torch.randn(B, 13, 16, 60, 90, generator=g, device=DEV).to(BF)
    pe = (torch.randn(B, 226, 4096, generator=g, device=DEV) * 0.1).to(BF)
    emb = torch.nn.functional.layer_norm(torch.randn(B, 5, 8, 12, 3072, generator=g, device=DEV), (3072,)).permute(0, 1, 4, 2, 3).to(BF).contiguous()
    ts = torch.full((B,), 500, dtype=torch.int64, device=DEV)
    rec[f"b{B}_ms"] = timed(lambda: m(hidden_states=x, encoder_hidden_states=pe, timestep=ts, image_rotary_emb=rope, vip_image_rotary_emb=vr,
                                      vip_condition_rotary_emb=cr, vip_encoder_hidden_states=emb, return_dict=False)[0])
rec["b1_over_b2"] = rec["b1_ms"] / rec["b2_ms"]
out["to2v"] = rec
del m
torch.cuda.empty_cache()
chunks = int(os.environ.get("TG_B1_CHUNKS", "24"))
m2 = CogVideoXTransformer3DModel(num_attention_heads=48, attention_head_dim=64, num_layers=42, time_embed_dim=512, text_embed_dim=4096, patch_size=1,
                                 use_rotary_positional_embeddings=True, device=DEV)
for name, t in m2._fused.items():
    t.copy_(torch.randn(t.shape, generator=g, device=DEV, dtype=torch.float32) * 0.02)
    if name.endswith(("ln", "qknorm")):
        t[0::2] += 1.0
nf = 4 * chunks
rope2 = R.rope_3d(64, np.arange(nf, dtype=f32), np.arange(8, dtype=f32), np.arange(12, dtype=f32), dim_t=52, dim_h=6, dim_w=6)
rec = {"tokens": 226 + nf * 96}
for B in (2, 1):
    x = torch.randn(B, nf, 16, 8, 12, generator=g, device=DEV).to(BF)
    pe = (torch.randn(B, 226, 4096, generator=g, device=DEV) * 0.1).to(BF)
    ts = torch.full((B,), 500, dtype=torch.int64, device=DEV)
    rec[f"b{B}_ms"] = timed(lambda: m2(hidden_states=x, encoder_hidden_states=pe, timestep=ts, image_rotary_emb=rope2, return_dict=False)[0])
rec["b1_over_b2"] = rec["b1_ms"] / rec["b2_ms"]
out["t2to"] = rec
print(json.dumps(out, indent=1))
