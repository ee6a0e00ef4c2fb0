#!/usr/bin/env python3
"""FF1 / FF2-dgrad of the training step at the block's shape: GEMM + tg_act pass against the GEMM's activation epilogues.  GPU box only."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_kernels as bk
from tokensgen_amd import kernels as K, lib as L, train
B, D, N = bk.B, bk.D, bk.N
x, w1, b1 = bk.rnd(B, N, D), bk.rnd(4 * D, D, scale=0.02), bk.rnd(4 * D)
pre, h = torch.empty(B, N, 4 * D, dtype=bk.BF, device="cuda"), torch.empty(B, N, 4 * D, dtype=bk.BF, device="cuda")
dy, w2 = bk.rnd(B * N, D), bk.rnd(D, 4 * D, scale=0.02)
cache = {}
def fwd2(): K.gemm(x, w1, b1, pre, L.EPI_BIAS); train._act(pre, gelu=True)
def fwd1(): K.gemm(x, w1, b1, pre, L.EPI_BIAS_KEEP_GELU, residual=h)
pre2 = pre.view(B * N, -1)
def bwd2(): train._act(pre2, train.linear_backward_dx(dy, w2, frozen=(cache, "w")))
def bwd1(): train.linear_backward_dx(dy, w2, frozen=(cache, "w"), gelu_pre=pre2)
fwd2()
for r in range(3):
    print(json.dumps({"FF1 + gelu pass ms": round(bk.timeit(fwd2, iters=9), 4), "FF1 keep-GELU epilogue ms": round(bk.timeit(fwd1, iters=9), 4),
                      "FF2 dgrad + gelu' pass ms": round(bk.timeit(bwd2, iters=9), 4), "FF2 dgrad gelu' epilogue ms": round(bk.timeit(bwd1, iters=9), 4)}))
