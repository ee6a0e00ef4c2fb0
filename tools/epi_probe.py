import json, os, sys, torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_kernels as bk
from tokensgen_amd import kernels as K, lib as L
B, D, N = bk.B, bk.D, bk.N
for rnd_ in range(3):
    for (M, Nn, Kk, name) in [(bk.N1, 3 * D, D, "qkv-shape"), (N, 4 * D, D, "ff1-shape")]:
        a, w, bias = bk.rnd(B, M, Kk), bk.rnd(Nn, Kk, scale=0.02), bk.rnd(Nn)
        out = torch.empty(B, M, Nn, dtype=bk.BF, device="cuda")
        r = {}
        for epi, en in ((L.EPI_BIAS, "bias"), (L.EPI_BIAS_GELU, "gelu"), (L.EPI_BIAS_SILU, "silu")):
            r[en] = round(bk.timeit(lambda: K.gemm(a, w, bias, out, epi), iters=9), 4)
        print(json.dumps({"shape": name, "ms": r}))
