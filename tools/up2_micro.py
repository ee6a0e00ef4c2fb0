#!/usr/bin/env python3
"""Stand-alone time of the decoder's upsampler convolutions: the 9-tap kernel on the upsampled grid (tg_conv3d_cl up = 2) against the four 2x2 phase launches
(tg_conv3d_up2_subpixel), at the tile shapes of the 480 x 720 decode.   python tools/up2_micro.py"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokensgen_amd import kernels as K  # noqa: E402
from tokensgen_amd.vae import pack_up2_phases  # noqa: E402
DEV, BF = "cuda", torch.bfloat16


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


out = []
for name, T, H, W, C, tx2 in (("up_blocks.2 (spatial only)", 8, 120, 180, 256, False), ("up_blocks.1 (time x2)", 4, 60, 90, 256, True), ("up_blocks.0 (time x2)", 2, 30, 45, 512, True)):
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(T, H, W, C, generator=g, device=DEV).to(BF)
    w = (torch.randn(C, C, 3, 3, generator=g, device=DEV) * 0.03).to(BF)
    b = torch.randn(C, generator=g, device=DEV).to(BF)
    wp = torch.zeros(C, 9, C, dtype=BF, device=DEV); wp[:] = w.reshape(C, C, 9).permute(0, 2, 1)
    ph = pack_up2_phases(w)
    To = T if not tx2 else 2 * T
    tm = torch.tensor([t for t in range(T) for _ in (0, 1)], dtype=torch.int32, device=DEV) if tx2 else None
    flop9 = 2.0 * To * 4 * H * W * C * C * 9
    rec = {"layer": name, "in": [T, H, W, C], "us_9tap": timed(lambda: K.conv3d_cl(x, wp, b, C, 1, 3, 3, up=2, t_map=tm, out_dims=(To, 2 * H, 2 * W), gn_stats_eps=1e-6))}
    rec["PF_9tap"] = flop9 / rec["us_9tap"] / 1e9
    if K.conv3d_up2_subpixel_ok(T, H, W, C, C):
        rec["us_phases"] = timed(lambda: K.conv3d_up2_subpixel(x, ph, b, C, gn_stats_eps=1e-6, time_x2=tx2))
        rec["PF_phases_executed"] = 2.0 * T * H * W * C * C * 4 * 4 / rec["us_phases"] / 1e9
        rec["PF_phases_as_9tap_work"] = flop9 / rec["us_phases"] / 1e9
    out.append(rec)
print(json.dumps(out, indent=1))
