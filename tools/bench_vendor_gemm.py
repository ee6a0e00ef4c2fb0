#!/usr/bin/env python3
"""Reference point only (never on the product path): the vendor GEMM (torch.matmul -> hipBLASLt/rocBLAS) on the DiT's GEMM
shapes next to tg_gemm_bf16, same random bf16 data, same box."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokensgen_amd import kernels as K  # noqa: E402
from tokensgen_amd import lib as L  # noqa: E402

DEV, BF = "cuda", torch.bfloat16


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    evs = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); evs.append((s, e))
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in evs)
    return ms[len(ms) // 2]


def main():
    for (M, N, Kk, name) in [(35552, 9216, 3072, "qkv"), (36512, 3072, 3072, "out"), (36512, 12288, 3072, "ff1"), (36512, 3072, 12288, "ff2"),
                             (18884, 9216, 3072, "t2to_qkv")]:
        a = (torch.randn(M, Kk, device=DEV) * 1.0).to(BF)
        w = (torch.randn(N, Kk, device=DEV) * 0.02).to(BF)
        b = torch.randn(N, device=DEV).to(BF)
        out = torch.empty(M, N, dtype=BF, device=DEV)
        fl = 2.0 * M * N * Kk
        ms_v = timeit(lambda: torch.nn.functional.linear(a, w, b))
        ms_m = timeit(lambda: K.gemm(a[None], w, b, out[None], L.EPI_BIAS))
        print(json.dumps({"gemm": name, "M": M, "N": N, "K": Kk, "vendor_ms": ms_v, "vendor_tflops": fl / ms_v / 1e9, "tg_ms": ms_m,
                          "tg_tflops": fl / ms_m / 1e9}))


if __name__ == "__main__":
    main()
