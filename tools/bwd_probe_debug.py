#!/usr/bin/env python3
"""Print what tg_attention_bwd_probe left in its buffer, idle and behind other work on the stream (debugging aid for the one-kernel backward's device probe)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tokensgen_amd import kernels as K, lib as L
lib = L.load()
nb = lib.tg_attention_bwd_probe_bytes()


def probe(tag):
    buf = torch.empty(nb, dtype=torch.uint8, device="cuda")
    L.check(lib.tg_attention_bwd_probe(buf.data_ptr(), nb, K._stream()), "probe")
    host = buf.cpu()
    ints = host[:4 * 2052].view(torch.int32)
    data = host[4 * 2052:].view(torch.float32)
    print(tag, "verdict", lib.tg_attention_bwd_probe_verdict(host.data_ptr(), nb), "bad", int(ints[0]), "cnt min/max", int(ints[4:].view(-1, 32)[:, 0].min()),
          int(ints[4:].view(-1, 32)[:, 0].max()), "data uniq", torch.unique(data).tolist()[:8], flush=True)


for t in range(2):
    probe(f"idle{t}")
a = torch.randn(4099, 4099, device="cuda")
for t in range(3):
    for _ in range(20):
        b = a @ a
        c = torch.relu(b[:1237, :777]).sum()
    probe(f"behind_matmuls{t}")
x = torch.randn(13, 1001, 517, device="cuda", dtype=torch.bfloat16)
for t in range(3):
    for _ in range(50):
        y = torch.nn.functional.gelu(x) * 1.5 + x
    probe(f"behind_elementwise{t}")
big = torch.empty(60 * 2**30, dtype=torch.uint8, device="cuda")
big.zero_()
probe("after_60GB_alloc")
