#!/usr/bin/env python3
"""The VAE decoder's large 3x3x3 convolution shapes, each in a loop: timing / PMC target for the conv kernels.  The dispatch switches of
tg_conv3d_cl (TG_CONV_HALO, TG_CONV_W4, ...) select the kernel; same-box A/B = one process per setting.
usage: conv_micro.py [repeats] [case ...]      cases: c128 (128->128 @ 8x240x360), c256 (256->256 @ 8x120x180), c256_128 (256->128 @ 8x240x360),
                                                c512_256 (512->256 @ 4x60x90), c256s (256->256 @ 4x60x90), c512 / c512t3 (512->512 @ 2 / 3 x 30x45), cout3 (128->3 @ 8x240x360: the decoder's conv_out)"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tokensgen_amd import kernels as K  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
CASES = {"c128": (128, 128, 8, 240, 360), "c256": (256, 256, 8, 120, 180), "c256_128": (256, 128, 8, 240, 360), "c512_256": (512, 256, 4, 60, 90),
         "c256s": (256, 256, 4, 60, 90), "c512": (512, 512, 2, 30, 45), "c512t3": (512, 512, 3, 30, 45), "cout3": (128, 3, 8, 240, 360)}
args = sys.argv[1:]
n = int(args[0]) if args and args[0].isdigit() else 20
names = [a for a in args if a in CASES] or list(CASES)
env = {k: v for k, v in os.environ.items() if k.startswith("TG_CONV")}
for name in names:
    Ci, Co, T, H, W = CASES[name]
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(T, H, W, Ci, generator=g, device=DEV).to(BF)
    cache = torch.randn(2, H, W, Ci, generator=g, device=DEV).to(BF)
    Cp = Co if Co % 128 == 0 else (Co + 15) // 16 * 16
    w = torch.zeros(Cp, 27, Ci, dtype=BF, device=DEV)
    w[:Co] = (torch.randn(Co, 27, Ci, generator=g, device=DEV) * 0.02).to(BF)
    b = torch.zeros(Co, dtype=BF, device=DEV)
    if os.environ.get("TG_CONV_MICRO_ZERO") == "1":          # power probe: the same launch on all-zero operands (no toggling in the matrix pipe)
        x.zero_(); cache.zero_(); w.zero_()
    gn = 1e-6 if Co % 128 == 0 else None
    for _ in range(3):
        y = K.conv3d_cl(x, w, b, Co, 3, 3, 3, cache=cache, gn_stats_eps=gn)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        y = K.conv3d_cl(x, w, b, Co, 3, 3, 3, cache=cache, gn_stats_eps=gn)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    fl = 2.0 * T * H * W * 27 * Ci * Co
    print(json.dumps({"case": name, "env": env, "ms": round(dt * 1e3, 4), "TFLOPs": round(fl / dt / 1e12, 1), "checksum": float(y.float().abs().mean())}))
