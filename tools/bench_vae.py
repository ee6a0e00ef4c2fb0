#!/usr/bin/env python3
"""BASELINE config 4: 3-D causal VAE encode/decode only, 720x480x49 frames, 1 GPU (tiling + slicing on, like the pipeline)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tokensgen_amd import kernels as K  # noqa: E402
from tokensgen_amd.vae import AutoencoderKLCogVideoX  # noqa: E402

DEV = "cuda"
vae = AutoencoderKLCogVideoX(device=DEV).init_random(seed=1)
vae.enable_tiling(); vae.enable_slicing()
g = torch.Generator(device=DEV).manual_seed(0)
z = (torch.randn(1, 16, 13, 60, 90, generator=g, device=DEV) / 1.15258426).to(torch.bfloat16)
x = (torch.rand(1, 3, 49, 480, 720, generator=g, device=DEV) * 2 - 1).to(torch.bfloat16)
PLAIN = "--plain" in sys.argv       # no per-launch events (for runs under rocprofv3)
what = [w for w in sys.argv[1:] if not w.startswith("--")] or ["decode", "encode"]
for name in what:
    fn = (lambda: vae.decode(z).sample) if name == "decode" else (lambda: vae.encode(x).latent_dist.mode())
    out = fn(); torch.cuda.synchronize()
    dt = 1e9
    for _ in range(4):        # (the 1st repeat captures the tile graphs) wall time WITHOUT the per-launch profiling events (~8300 launches: the events alone cost ~80 ms)
        t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); dt = min(dt, time.perf_counter() - t0)
    if PLAIN:
        print(json.dumps({"op": f"vae_{name}", "seconds": dt, "out_shape": list(out.shape)}))
        continue
    K.PROFILE.clear(); K.PROFILE_ON[0] = True
    t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); dt_prof = time.perf_counter() - t0
    K.PROFILE_ON[0] = False
    prof = K.profile_summary()
    flop = {"decode": 3.1e14, "encode": 1.5e14}[name]      # untiled algorithmic count; executed = x1.40
    top = sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:int(os.environ.get("TG_BENCH_VAE_TOP", "8"))]
    print(json.dumps({"op": f"vae_{name}", "out_shape": list(out.shape), "seconds": dt, "seconds_with_profiling_events": dt_prof, "clips_per_s": 1 / dt,
                      "algorithmic_TFLOPs": flop / dt / 1e12, "executed_TFLOPs": 1.4 * flop / dt / 1e12, "finite": bool(torch.isfinite(out).all()),
                      "kernel_total_ms": {k: round(v["total_ms"], 1) for k, v in top}, "launches": sum(v["n"] for v in prof.values())}))
