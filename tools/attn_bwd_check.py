#!/usr/bin/env python3
"""Error of tg_attention_bwd against autograd on a few shapes + time at the full shape (A/B of library variants via TG_LIB_PATH)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokensgen_amd import kernels as K
BF = torch.bfloat16
def rel(a, b): return ((a.float().cpu() - b.float()).norm() / b.float().norm()).item()
for (B, H, nq, nk) in [(1, 2, 50, 70), (2, 3, 513, 1500), (1, 4, 200, 33), (1, 8, 50, 70), (2, 4, 513, 1500), (1, 8, 200, 33), (2, 8, 1000, 777)]:
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(B, n, H * 64, generator=g).mul(1.5).to(BF) for n in (nq, nk, nk))
    do = torch.randn(B, nq, H * 64, generator=g).to(BF)
    qf, kf, vf = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    sp = lambda t: t.view(B, t.shape[1], H, 64).transpose(1, 2)
    o = (torch.softmax(sp(qf) @ sp(kf).transpose(-1, -2) / 8.0, -1) @ sp(vf)).transpose(1, 2).reshape(B, nq, H * 64)
    (o * do.float()).sum().backward()
    dq, dk, dv = K.attention_bwd(q.cuda(), k.cuda(), v.cuda(), o.detach().to(BF).cuda(), do.cuda(), H, 0.125)
    print((B, H, nq, nk), "dq %.4f dk %.4f dv %.4f" % (rel(dq, qf.grad), rel(dk, kf.grad), rel(dv, vf.grad)))
