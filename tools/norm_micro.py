#!/usr/bin/env python3
"""The VAE's streaming norm passes at the decoder's large shapes, each in a loop: achieved bytes/s (one read + one write of the tensor).
usage: norm_micro.py [repeats]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tokensgen_amd import kernels as K  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for name, (T, H, W, C, Tz, Hz, Wz) in {"sn128": (8, 240, 360, 128, 2, 30, 45), "sn256": (8, 120, 180, 256, 2, 30, 45), "sn256s": (4, 60, 90, 256, 2, 30, 45),
                                        "sn512": (2, 30, 45, 512, 2, 30, 45), "gn128": (8, 240, 360, 128, 0, 0, 0), "gn256": (8, 120, 180, 256, 0, 0, 0)}.items():
    g = torch.Generator(device=DEV).manual_seed(0)
    f = torch.randn(T, H, W, C, generator=g, device=DEV).to(BF)
    st = K.groupnorm_stats(f.view(-1, C), 1e-6)
    gamma, beta = torch.ones(C, dtype=BF, device=DEV), torch.zeros(C, dtype=BF, device=DEV)
    if Tz:
        yz = torch.randn(Tz * Hz * Wz, C, generator=g, device=DEV).to(BF)
        bz = torch.randn(Tz * Hz * Wz, C, generator=g, device=DEV).to(BF)
        nosilu = os.environ.get("TG_NORM_MICRO_NOSILU") == "1"          # timing probe: how much of the pass is the SiLU's exp / rcp
        run = lambda: K.spatialnorm_silu(f, st, gamma, beta, yz, bz, (Tz, Hz, Wz), silu=not nosilu)
    else:
        run = lambda: K.groupnorm_silu(f, st, gamma, beta)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        y = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"case": name, "us": round(dt * 1e6, 1), "TB/s": round(2 * f.numel() * 2 / dt / 1e12, 2)}))
