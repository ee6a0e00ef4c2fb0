#!/usr/bin/env python3
"""First contact of a new attention-backward kernel with the GPU: small shapes first, the exchange's poll limit forced LOW (a protocol bug then ends in
milliseconds with status words set instead of a long spin), every result against fp32 autograd.  Usage: python tools/bwd_first_contact.py [poll_limit]"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokensgen_amd import kernels as K  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
limit = int(sys.argv[1]) if len(sys.argv) > 1 else 20000


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


def sdpa(q, k, v, H, scale):
    B, nq, nk = q.shape[0], q.shape[1], k.shape[1]
    sp = lambda t: t.view(B, t.shape[1], H, 64).transpose(1, 2)
    p = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * scale, dim=-1)
    return (p @ sp(v)).transpose(1, 2).reshape(B, nq, H * 64)


st = K.BwdDeviceState.get(torch.device(DEV))
print("probe", st.one_kernel, st.probe, flush=True)
st.status[1] = limit
bad = 0
for (B, H, nq, nk, scale) in [(1, 8, 129, 5, 0.125), (1, 8, 1100, 700, 0.125), (2, 4, 1030, 257, 0.125), (2, 4, 2500, 300, math.log(2.0)), (1, 8, 4100, 1000, 0.125),
                              (1, 8, 4100, 1000, math.log(2.0)), (2, 48, 3000, 1500, math.log(2.0))]:
    g = torch.Generator().manual_seed(nq + nk)
    D = H * 64
    amp = 1.5 if scale == 0.125 else 0.6
    fused = (torch.randn(B, max(nq, nk), 3 * D, generator=g) * amp).to(BF)
    q, k, v = fused[:, :nq, :D], fused[:, :nk, D:2 * D], fused[:, :nk, 2 * D:]
    go = torch.randn(B, nq, D, generator=g).to(BF)
    big = B * H * nq * nk > 3e8
    if not big:
        qf, kf, vf = (t.float().clone().requires_grad_(True) for t in (q, k, v))
        o = sdpa(qf, kf, vf, H, scale)
        (o * go.float()).sum().backward()
        ref = (qf.grad, kf.grad, vf.grad)
    else:       # reference on the GPU in fp32, head by head
        fd32 = fused.to(DEV).float()
        qf, kf, vf = (fd32[:, :n, c * D:(c + 1) * D].clone().requires_grad_(True) for n, c in ((nq, 0), (nk, 1), (nk, 2)))
        o = sdpa(qf, kf, vf, H, scale)
        (o * go.to(DEV).float()).sum().backward()
        ref = (qf.grad, kf.grad, vf.grad)
    fd = fused.to(DEV)
    qd, kd, vd = fd[:, :nq, :D], fd[:, :nk, D:2 * D], fd[:, :nk, 2 * D:]
    od, gd = o.detach().to(BF).to(DEV), go.to(DEV)
    t0 = time.time()
    dq, dk, dv = K.attention_bwd(qd, kd, vd, od, gd, H, scale)
    torch.cuda.synchronize()
    dt = time.time() - t0
    polls, xcd = K.attention_bwd_status(DEV)
    dq2, dk2, dv2 = K.attention_bwd(qd, kd, vd, od, gd, H, scale)
    same = torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)
    K.attention_bwd(qd, kd, vd, od, gd, H, scale, dq=dq2, dk=dk2, dv=dv2, accumulate=True)
    acc_ok = torch.allclose(dq2, 2 * dq, rtol=1e-4, atol=1e-5) and torch.allclose(dk2, 2 * dk) and torch.allclose(dv2, 2 * dv)
    polls2, xcd2 = K.attention_bwd_status(DEV)
    e = (rel(dq, ref[0]), rel(dk, ref[1]), rel(dv, ref[2]))
    ok = max(e) < 6e-3 and same and acc_ok and not (polls or xcd or polls2 or xcd2)
    bad += not ok
    print(f"B={B} H={H} nq={nq} nk={nk} scale={scale:.3f}: rel dq/dk/dv = {e[0]:.2e} {e[1]:.2e} {e[2]:.2e} deterministic={same} accumulate={acc_ok} "
          f"status={polls},{xcd},{polls2},{xcd2} first_call_s={dt:.3f} finite={bool(torch.isfinite(dq).all())} {'OK' if ok else 'FAIL'}", flush=True)
print("FAILED" if bad else "ALL OK", bad)
sys.exit(1 if bad else 0)
