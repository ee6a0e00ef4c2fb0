#!/usr/bin/env python3
"""UPPER BOUND of what one launch per layer over ALL temporal batches of a tile would buy (round 6): the same full-size decode / encode with the temporal batch size
set to the whole clip (13 latent / 49 sample frames) — the launches of a merged design, minus its per-batch norm launches.  The RESULT differs from the reference's
(GroupNorm statistics over the whole clip instead of per batch): timing only."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tokensgen_amd.vae import AutoencoderKLCogVideoX  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(0)
z = (torch.randn(1, 16, 13, 60, 90, generator=g, device=DEV) / 1.15258426).to(torch.bfloat16)
x = (torch.rand(1, 3, 49, 480, 720, generator=g, device=DEV) * 2 - 1).to(torch.bfloat16)
out = {}
for tag, lb, sb in (("product (2 latent / 8 sample frames per batch)", 2, 8), ("whole clip per launch (13 / 49)", 13, 49), ("product again", 2, 8)):
    vae = AutoencoderKLCogVideoX(device=DEV).init_random(seed=1)
    vae.enable_tiling(); vae.enable_slicing()
    vae.num_latent_frames_batch_size, vae.num_sample_frames_batch_size = lb, sb
    rec = {}
    for name, fn in (("decode", lambda: vae.decode(z).sample), ("encode", lambda: vae.encode(x).latent_dist.parameters)):
        y = fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); y = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        rec[name] = {"seconds": round(best, 4), "finite": bool(torch.isfinite(y).all()), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
    out[tag] = rec
    del vae
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
