#!/usr/bin/env python3
"""Times the To2V processor's vip-query backward call (480 queries x 18256 keys, 48 heads, batch 2: the two-launch form) — same-box A/B through TG_LIB_PATH."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokensgen_amd import kernels as K  # noqa: E402
B, H, D, NQ, NK = 2, 48, 3072, 480, 18256
rnd = lambda *sh, scale=1.0: (torch.randn(*sh, device="cuda") * scale).to(torch.bfloat16)
q, k, v, o, do = rnd(B, NQ, D, scale=0.6), rnd(B, NK, D, scale=0.6), rnd(B, NK, D, scale=0.6), rnd(B, NQ, D, scale=0.3), rnd(B, NQ, D, scale=0.3)
dq = torch.empty(B, NQ, D, dtype=torch.float32, device="cuda"); dk, dv = (torch.empty(B, NK, D, dtype=torch.float32, device="cuda") for _ in range(2))
lse = torch.zeros(B, H, NQ, dtype=torch.float32, device="cuda") + 5.0
fn = lambda: K.attention_bwd(q, k, v, o, do, H, 0.125, dq=dq, dk=dk, dv=dv, lse=lse)
for _ in range(3): fn()
torch.cuda.synchronize()
evs = []
for _ in range(20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record(); evs.append((s, e))
torch.cuda.synchronize()
ms = sorted(s.elapsed_time(e) for s, e in evs)
print(json.dumps({"call": "vip-query backward (stats + dK/dV + dQ [+ join])", "lib": os.path.basename(os.environ.get("TG_LIB_PATH", "product")), "ms_median": ms[10], "ms_min": ms[0]}))
