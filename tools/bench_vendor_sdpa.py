#!/usr/bin/env python3
"""Reference point only (never on the product path): torch's scaled_dot_product_attention (the ROCm flash kernel torch ships) on
the main attention shape, next to tg_attention_fwd's single-segment call on the same random bf16 data."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokensgen_amd import kernels as K  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
B, H, N, D = 2, 48, 17776, 3072


def timeit(fn, n=8):
    for _ in range(2):
        fn()
    evs = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); evs.append((s, e))
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in evs)
    return ms[len(ms) // 2]


def main():
    qkv = (torch.randn(B, N, 3 * D, device=DEV) * 0.4).to(BF)
    q, k, v = (qkv[:, :, i * D:(i + 1) * D].view(B, N, H, 64).transpose(1, 2) for i in range(3))
    fl = 4.0 * B * N * N * D
    res = {}
    for name, backend in (("flash", torch.nn.attention.SDPBackend.FLASH_ATTENTION), ("efficient", torch.nn.attention.SDPBackend.EFFICIENT_ATTENTION)):
        try:
            with torch.nn.attention.sdpa_kernel(backend):
                ms = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
            res[name] = {"ms": ms, "tflops": fl / ms / 1e9}
        except Exception as e:  # noqa: BLE001
            res[name] = {"error": str(e)[:120]}
    vt = torch.empty(B, H, 64, (N + 63) // 64 * 64, dtype=BF, device=DEV)
    K.transpose_v(qkv[:, :, 2 * D:], H, 0, N, vt)
    out = torch.empty(B, N, D, dtype=BF, device=DEV)
    ms = timeit(lambda: K.attention(qkv[:, :, :D], qkv[:, :, D:2 * D], vt, N, out, H, 0.125))
    res["tg_attention_fwd(1 segment, scale in kernel)"] = {"ms": ms, "tflops": fl / ms / 1e9}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
