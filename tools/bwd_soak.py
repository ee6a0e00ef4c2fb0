#!/usr/bin/env python3
"""Soak of the one-kernel attention backward at the training step's real shapes: the main call (17776^2, 96 (batch, head) pairs) with the vip-key call riding, N times; every
run must be BITWISE equal to the first (the ordered dQ exchange is deterministic by construction: any race shows up here) and the status words must stay clean.
   python tools/bwd_soak.py [runs]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokensgen_amd import kernels as K  # noqa: E402
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, H, D, N1, NP = 2, 48, 3072, 17776, 480
rnd = lambda *sh, scale=1.0: (torch.randn(*sh, device="cuda") * scale).to(torch.bfloat16)
q, k, v, o, g = rnd(B, N1, D, scale=0.6), rnd(B, N1, D, scale=0.5), rnd(B, N1, D, scale=0.6), rnd(B, N1, D, scale=0.3), rnd(B, N1, D, scale=0.3)
q2, k2, v2, o2 = rnd(B, N1, D, scale=0.6), rnd(B, NP, D, scale=0.6), rnd(B, NP, D, scale=0.6), rnd(B, N1, D, scale=0.3)
f32 = torch.float32
ref = None
bad = 0
for r in range(runs):
    dq, dk, dv = (torch.empty(B, N1, D, dtype=f32, device="cuda") for _ in range(3))
    dq2 = torch.empty(B, N1, D, dtype=f32, device="cuda"); dk2, dv2 = (torch.zeros(B, NP, D, dtype=f32, device="cuda") for _ in range(2))
    K.attention_bwd_multi([dict(q=q, k=k, v=v, o=o, dout=g, scale=math.log(2.0), dq=dq, dk=dk, dv=dv),
                           dict(q=q2, k=k2, v=v2, o=o2, dout=g, scale=0.125, dq=dq2, dk=dk2, dv=dv2, accumulate=2)], H)
    torch.cuda.synchronize()
    polls, xcd = K.attention_bwd_status("cuda")
    cur = [t.clone() for t in (dq, dk, dv, dq2, dk2, dv2)]
    fin = all(bool(torch.isfinite(t).all()) for t in cur)
    if ref is None:
        ref = cur
        same = True
    else:
        same = all(torch.equal(a, b) for a, b in zip(ref, cur))
    ok = same and fin and not polls and not xcd
    bad += not ok
    print(f"run {r}: bitwise_equal_to_run_0={same} finite={fin} status={polls},{xcd} {'OK' if ok else 'FAIL'}", flush=True)
print("SOAK", "FAILED" if bad else "OK", bad)
sys.exit(1 if bad else 0)
