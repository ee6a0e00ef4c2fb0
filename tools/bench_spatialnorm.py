#!/usr/bin/env python3
"""Streaming rate of tg_spatialnorm_silu at the VAE decoder's largest shapes (bytes moved = f read + y written)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokensgen_amd import kernels as K
DEV, BF = "cuda", torch.bfloat16
def timeit(fn, n=10):
    for _ in range(3): fn()
    ev = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record(); fn(); e.record(); ev.append((s, e))
    torch.cuda.synchronize(); ms = sorted(s.elapsed_time(e) for s, e in ev); return ms[len(ms) // 2]
for (T, H, W, C, Tz, Hz, Wz) in [(8, 192, 288, 128, 2, 24, 36), (8, 96, 144, 256, 2, 24, 36), (4, 48, 72, 512, 2, 24, 36)]:
    f = torch.randn(T, H, W, C, device=DEV).to(BF)
    stats = torch.stack([torch.zeros(32, device=DEV), torch.ones(32, device=DEV)], 1).contiguous()
    g, b = torch.ones(C, device=DEV, dtype=BF), torch.zeros(C, device=DEV, dtype=BF)
    yb = torch.randn(Tz * Hz * Wz, 2 * C, device=DEV).to(BF)
    ms = timeit(lambda: K.spatialnorm_silu(f, stats, g, b, yb[:, :C], yb[:, C:], (Tz, Hz, Wz)))
    print(json.dumps({"shape": [T, H, W, C], "ms": ms, "GBps": 2 * f.numel() * 2 / ms / 1e6}))
    ms = timeit(lambda: K.groupnorm_silu(f, stats, g, b))
    print(json.dumps({"groupnorm_silu": [T, H, W, C], "ms": ms, "GBps": 2 * f.numel() * 2 / ms / 1e6}))
