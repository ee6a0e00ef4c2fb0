#!/usr/bin/env python3
"""Localise an attention-backward error: per-tensor determinism and per-(batch, head, key block) error against fp32 autograd on the GPU."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokensgen_amd import kernels as K  # noqa: E402
DEV, BF = "cuda", torch.bfloat16
B, H, nq, nk = (int(x) for x in sys.argv[1:5])
scale = math.log(2.0) if len(sys.argv) > 5 and sys.argv[5] == "unit" else 0.125
g = torch.Generator().manual_seed(nq + nk)
D = H * 64
fused = (torch.randn(B, max(nq, nk), 3 * D, generator=g) * (0.6 if scale != 0.125 else 1.5)).to(BF).to(DEV)
go = torch.randn(B, nq, D, generator=g).to(BF).to(DEV)
q, k, v = fused[:, :nq, :D], fused[:, :nk, D:2 * D], fused[:, :nk, 2 * D:]
sp = lambda t: t.view(B, t.shape[1], H, 64).transpose(1, 2)
qf, kf, vf = (t.float().clone().requires_grad_(True) for t in (q, k, v))
p = torch.softmax(sp(qf) @ sp(kf).transpose(-1, -2) * scale, dim=-1)
o = (p @ sp(vf)).transpose(1, 2).reshape(B, nq, D)
(o * go.float()).sum().backward()
od = o.detach().to(BF)
K.BwdDeviceState.get(torch.device(DEV)).status[1] = 20000
r1 = K.attention_bwd(q, k, v, od, go, H, scale)
r2 = K.attention_bwd(q, k, v, od, go, H, scale)
torch.cuda.synchronize()
print("status", K.attention_bwd_status(DEV))
for name, a, b, ref in zip(("dq", "dk", "dv"), r1, r2, (qf.grad, kf.grad, vf.grad)):
    diff = (a != b)
    print(name, "run-to-run differing elements:", int(diff.sum()), "rel err", ((a - ref).norm() / ref.norm()).item(), "nan", int(torch.isnan(a).sum()))
    n = a.shape[1]
    e = (a - ref).view(B, n, H, 64)
    rr = ref.view(B, n, H, 64)
    per_bh = (e.pow(2).sum(dim=(1, 3)) / rr.pow(2).sum(dim=(1, 3))).sqrt()          # [B, H]
    worst = torch.topk(per_bh.flatten(), min(6, B * H))
    print("   worst (b,h):", [(int(i) // H, int(i) % H, round(float(x), 4)) for x, i in zip(worst.values, worst.indices)])
    nb = (n + 255) // 256
    rows = torch.arange(n, device=DEV) // 256
    per_blk = torch.zeros(nb, device=DEV).index_add_(0, rows, e.pow(2).sum(dim=(0, 2, 3))) / torch.zeros(nb, device=DEV).index_add_(0, rows, rr.pow(2).sum(dim=(0, 2, 3)))
    print("   per 256-row block rel:", [round(float(x) ** 0.5, 4) for x in per_blk])
    if int(diff.sum()):
        idx = diff.nonzero()
        print("   differing: batches", idx[:, 0].unique().tolist(), "heads", (idx[:, 2] // 64).unique().tolist()[:20], "row range", int(idx[:, 1].min()), int(idx[:, 1].max()),
              "rows mod 32 ->", (idx[:, 1] % 32).unique().tolist()[:32], "cols mod 64", (idx[:, 2] % 64).unique().tolist()[:64])
