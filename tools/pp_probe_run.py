#!/usr/bin/env python3
"""Reads the stamps a `ppprobe` library variant (tools/patches/pp_probe.py) leaves in dV: average s_memtime ticks per tile and segment of the ping-pong one-kernel
attention backward at the training step's main shape.  GPU box only; TG_LIB_PATH must point at the variant."""
import json, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tokensgen_amd import kernels as K  # noqa: E402
assert "TG_LIB_PATH" in os.environ
B, H, D, N1 = 2, 48, 3072, 17776
rnd = lambda *sh, scale=1.0: (torch.randn(*sh, device="cuda") * scale).to(torch.bfloat16)
qkv = rnd(B, N1, 3 * D, scale=0.6)
o, do = rnd(B, N1, D, scale=0.3), rnd(B, N1, D, scale=0.3)
dq, dk, dv = (torch.empty(B, N1, D, dtype=torch.float32, device="cuda") for _ in range(3))
unit = os.environ.get("TG_BENCH_BWD_UNIT") == "1"
kk = (qkv[:, :, D:2 * D].float() * (0.125 * 1.4426950408889634)).to(torch.bfloat16) if unit else qkv[:, :, D:2 * D]
for _ in range(2):
    K.attention_bwd(qkv[:, :, :D], kk, qkv[:, :, 2 * D:], o, do, H, math.log(2.0) if unit else 0.125, dq=dq, dk=dk, dv=dv)
torch.cuda.synchronize()
names = ["X", "barrier after X", "Y top: math + store + request (after the lgkm wait)", "Y softmax", "Y tail (check, prefetch)", "barrier after Y", "Y top: sDQ read issue", "Y top: vmcnt(0) wait", "Y top: stage write + fetch", "Y top: lgkmcnt wait for the sDQ reads"]
raw = dv.view(-1)[:2 * 12 * 12].view(torch.int64).cpu().tolist()
for wgi, wgname in enumerate(("wg40", "wg320", "wg2400")):
    for w, wn in enumerate(("wave0", "wave1", "wave4", "wave5")):
        c = raw[12 * (4 * wgi + w): 12 * (4 * wgi + w) + 12]
        nt = max(c[10], 1)
        print(json.dumps({"wg": wgname, "wave": wn, "iterations": c[10], "ticks_per_tile": {n: round(c[i] / nt, 1) for i, n in enumerate(names)}, "sum": round(sum(c[:10]) / nt, 1)}))
