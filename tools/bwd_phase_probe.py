#!/usr/bin/env python3
"""Reads the cycle counters a `bwdprobe` library variant (tools/patches/bwd_probe.py) leaves in dV / dQ: average s_memtime ticks per tile and
phase of the attention-backward kernels at the training step's main shape.  GPU box only; run with TG_LIB_PATH pointing at the variant."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tokensgen_amd import kernels as K  # noqa: E402

assert "TG_LIB_PATH" in os.environ, "point TG_LIB_PATH at csrc/variants/bwdprobe.so"
B, H, D, N1 = 2, 48, 3072, 17776
rnd = lambda *sh, scale=1.0: (torch.randn(*sh, device="cuda") * scale).to(torch.bfloat16)
qkv = rnd(B, N1, 3 * D, scale=0.6)
o, do = rnd(B, N1, D, scale=0.3), rnd(B, N1, D, scale=0.3)
dq, dk, dv = (torch.empty(B, N1, D, dtype=torch.float32, device="cuda") for _ in range(3))
for _ in range(2):
    K.attention_bwd(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], o, do, H, 0.125, dq=dq, dk=dk, dv=dv)
torch.cuda.synchronize()
names = {"dkdv": ["X: 18 mfma + all LDS reads", "barrier after X", "Y: softmax", "barrier after Y", "Y: stage write (+ landed)", "Y: fetch issue"],
         "dq": ["A + transposed B reads", "S/dP mfma (x2)", "softmax (x2)", "dQ mfma (x2)", "stash", "barrier"]}
for name, t in (("dkdv", dv), ("dq", dq)):
    raw = t.view(-1)[:32].view(torch.int64).cpu().tolist()
    for w in range(2):
        c = raw[8 * w: 8 * w + 8]
        nt = max(c[6], 1)
        print(json.dumps({"kernel": name, "wave_group": w, "tiles": c[6], "ticks_per_tile": {n: round(c[i] / nt, 1) for i, n in enumerate(names[name])},
                          "sum": round(sum(c[:6]) / nt, 1)}))
