"""Reads the counters a `fprobe` library variant (tools/patches/fused_probe.py) leaves in dV: three workgroups, waves 0 and 5, at the training shape."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from tokensgen_amd import kernels as K
B, H, D, N1 = 2, 48, 3072, 17776
rnd = lambda *sh, scale=1.0: (torch.randn(*sh, device="cuda") * scale).to(torch.bfloat16)
qkv = rnd(B, N1, 3 * D, scale=0.6)
o, do = rnd(B, N1, D, scale=0.3), rnd(B, N1, D, scale=0.3)
dq, dk, dv = (torch.empty(B, N1, D, dtype=torch.float32, device="cuda") for _ in range(3))
for _ in range(2):
    K.attention_bwd(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], o, do, 48, 0.125, dq=dq, dk=dk, dv=dv)
torch.cuda.synchronize()
raw = dv.view(-1)[:128].view(torch.int64).cpu().tolist()
names = ["exchange at the top", "check (sample wait + polls)", "vmcnt + barrier", "rest of the iteration", "polls"]
for i in range(6):
    c = raw[8 * i: 8 * i + 8]
    nt = max(c[5], 1)
    print("workgroup", [40, 320, 2400][i // 2], "wave", [0, 5][i % 2], {n: round(c[k] / nt, 1) for k, n in enumerate(names)}, "tiles", c[5])
