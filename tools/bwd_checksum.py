#!/usr/bin/env python3
"""SHA-256 of the six gradient tensors of the training step's backward call at its real shape (17776^2 main problem + the vip-key rider, seeded operands):
run under two libraries (TG_LIB_PATH) to show that a kernel change is BITWISE neutral.   python tools/bwd_checksum.py"""
import hashlib, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokensgen_amd import kernels as K  # noqa: E402
B, H, D, N1, NP = 2, 48, 3072, 17776, 480
g = torch.Generator(device="cuda").manual_seed(1234)
rnd = lambda *sh, scale=1.0: (torch.randn(*sh, device="cuda", generator=g) * scale).to(torch.bfloat16)
q, k, v, o, do = rnd(B, N1, D, scale=0.6), rnd(B, N1, D, scale=0.5), rnd(B, N1, D, scale=0.6), rnd(B, N1, D, scale=0.3), rnd(B, N1, D, scale=0.3)
q2, k2, v2, o2 = rnd(B, N1, D, scale=0.6), rnd(B, NP, D, scale=0.6), rnd(B, NP, D, scale=0.6), rnd(B, N1, D, scale=0.3)
f32 = torch.float32
dq, dk, dv = (torch.empty(B, N1, D, dtype=f32, device="cuda") for _ in range(3))
dq2 = torch.empty(B, N1, D, dtype=f32, device="cuda"); dk2, dv2 = (torch.zeros(B, NP, D, dtype=f32, device="cuda") for _ in range(2))
K.attention_bwd_multi([dict(q=q, k=k, v=v, o=o, dout=do, scale=math.log(2.0), dq=dq, dk=dk, dv=dv),
                       dict(q=q2, k=k2, v=v2, o=o2, dout=do, scale=0.125, dq=dq2, dk=dk2, dv=dv2, accumulate=2)], H)
torch.cuda.synchronize()
K.attention_bwd_check("cuda")
h = hashlib.sha256()
for t in (dq, dk, dv, dq2, dk2, dv2):
    h.update(t.cpu().numpy().tobytes())
print(os.path.basename(os.environ.get("TG_LIB_PATH", "product")), h.hexdigest(), "one_kernel" if K.BwdDeviceState.get(torch.device("cuda", 0)).one_kernel else "two_kernel")
