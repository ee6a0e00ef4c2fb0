#!/usr/bin/env python3
"""One full-resolution 128 -> 128 3x3x3 convolution (the VAE's largest Cout = 128 shape) in a loop: timing / PMC target for the conv kernels.
TG_CONV_HALO selects the kernel (0: 4-wave 512 x 128 GEMM-shaped kernel, 3: halo-tiled)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tokensgen_amd import kernels as K  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
T, H, W, C = 8, 240, 360, 128
g = torch.Generator(device=DEV).manual_seed(0)
x = torch.randn(T, H, W, C, generator=g, device=DEV).to(BF)
cache = torch.randn(2, H, W, C, generator=g, device=DEV).to(BF)
w = (torch.randn(128, 27, C, generator=g, device=DEV) * 0.02).to(BF)
b = torch.zeros(128, dtype=BF, device=DEV)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3):
    y = K.conv3d_cl(x, w, b, 128, 3, 3, 3, cache=cache, gn_stats_eps=1e-6)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    y = K.conv3d_cl(x, w, b, 128, 3, 3, 3, cache=cache, gn_stats_eps=1e-6)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
fl = 2.0 * T * H * W * 27 * C * 128
print(f"TG_CONV_HALO={os.environ.get('TG_CONV_HALO', '1')}: {dt * 1e3:.3f} ms, {fl / dt / 1e12:.1f} TFLOP/s")
