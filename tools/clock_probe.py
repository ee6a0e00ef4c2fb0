#!/usr/bin/env python3
"""Sample the shader clock / power (rocm-smi) while one GEMM shape runs in a loop: which kernels are clock(power)-limited.
Usage: python tools/clock_probe.py [tg|vendor] [seconds]"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokensgen_amd import kernels as K  # noqa: E402
from tokensgen_amd import lib as L  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "tg"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
M, N, Kk = 36512, 12288, 3072
a = torch.randn(1, M, Kk, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, Kk, device="cuda") * 0.02).to(torch.bfloat16)
b = torch.randn(N, device="cuda").to(torch.bfloat16)
out = torch.empty(1, M, N, dtype=torch.bfloat16, device="cuda")
fn = (lambda: K.gemm(a, w, b, out, L.EPI_BIAS)) if which == "tg" else (lambda: torch.nn.functional.linear(a[0], w, b))
if which == "attn":          # the DiT's main attention launch (2 x 48 heads, 17776 queries x 17776 keys + 480 vip keys)
    B, H, N1, NP = 2, 48, 17776, 480
    D = H * 64
    qkv = (torch.randn(B, N1, 3 * D, device="cuda") * 0.4).to(torch.bfloat16)
    qkvv = (torch.randn(B, N1 + NP, 3 * D, device="cuda") * 0.4).to(torch.bfloat16)
    pad = lambda n: (n + 63) // 64 * 64
    vt1 = torch.empty(B, H, 64, pad(N1), dtype=torch.bfloat16, device="cuda")
    vt2 = torch.empty(B, H, 64, pad(NP), dtype=torch.bfloat16, device="cuda")
    K.transpose_v(qkv[:, :, 2 * D:], H, 0, N1, vt1)
    K.transpose_v(qkvv[:, :, 2 * D:], H, N1, NP, vt2)
    ao = torch.empty(B, N1, D, dtype=torch.bfloat16, device="cuda")
    fn = lambda: K.attention(qkv[:, :, :D], qkv[:, :, D:2 * D], vt1, N1, ao, H, 0.125, qkvv[:, :N1, :D], qkvv[:, N1:, D:2 * D], vt2, NP, 0.6,
                             k_prescaled=True)
    M, N, Kk = 1, 1, (B * (4.0 * N1 * N1 * D + 4.0 * N1 * NP * D)) / 2.0
if which == "attn_bwd":      # the training step's main attention backward call (dK/dV + dQ + statistics)
    B, H, N1 = 2, 48, 17776
    D = H * 64
    qkv = (torch.randn(B, N1, 3 * D, device="cuda") * 0.6).to(torch.bfloat16)
    o, do = ((torch.randn(B, N1, D, device="cuda") * 0.3).to(torch.bfloat16) for _ in range(2))
    dq, dk, dv = (torch.empty(B, N1, D, dtype=torch.float32, device="cuda") for _ in range(3))
    fn = lambda: K.attention_bwd(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], o, do, H, 0.125, dq=dq, dk=dk, dv=dv)
    M, N, Kk = 1, 1, 7 * B * N1 * N1 * D / 2.0
samples, stop = [], False


def sampler():
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append(r.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            samples.append(str(e))
        time.sleep(0.3)


for _ in range(5):
    fn()
torch.cuda.synchronize()
t = threading.Thread(target=sampler); t.start()
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(20):
        fn()
    torch.cuda.synchronize(); n += 20
dt = time.time() - t0
stop = True; t.join()
print(which, "ms/gemm %.3f" % (dt / n * 1e3), "TFLOP/s %.0f" % (2.0 * M * N * Kk * n / dt / 1e12))
for s in samples[:2] + samples[-4:]:
    print(s[:300])
