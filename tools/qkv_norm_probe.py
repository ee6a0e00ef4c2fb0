#!/usr/bin/env python3
"""QKV projection + QK-norm / RoPE at the DiT's shapes: two launches (tg_gemm_bf16_qkv, tg_qk_layernorm_rope_pair_kmax per problem) against the fused epilogue
(tg_gemm_bf16_qkv_norm).  GPU box only.
The fused entry is NOT in the tree (measured slower, profiles/NOTES.md §F): apply tools/patches/gemm_qkv_norm_epilogue.diff and rebuild first."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_kernels as bk
from tokensgen_amd import kernels as K, lib as L
B, H, D, Nt, N1, N = bk.B, bk.H, bk.D, bk.NT, bk.N1, bk.N
x = bk.rnd(B, N, D)
w1, w2, b1, b2 = bk.rnd(3 * D, D, scale=0.02), bk.rnd(3 * D, D, scale=0.02), bk.rnd(3 * D), bk.rnd(3 * D)
ln = [bk.rnd(64) for _ in range(8)]
def table(n):
    ang = torch.randn(n, 32, device="cuda") * 3
    return ang.cos().repeat_interleave(2, dim=1).contiguous(), ang.sin().repeat_interleave(2, dim=1).contiguous()
rope, vrope, crope = table(N1 - Nt), table(N1 - Nt), table(N - N1)
pad = lambda n: (n + 63) // 64 * 64
c1, c2 = torch.empty(B, N1, 3 * D, dtype=bk.BF, device="cuda"), torch.empty(B, N, 3 * D, dtype=bk.BF, device="cuda")
vt1, vt2 = torch.empty(B, H, 64, pad(N1), dtype=bk.BF, device="cuda"), torch.empty(B, H, 64, pad(N), dtype=bk.BF, device="cuda")
km1, km2 = torch.zeros(B, H, device="cuda"), torch.zeros(B, H, device="cuda")
kws = torch.empty(L.load().tg_qk_kmax_ws_floats(N, H, B), device="cuda")
ks = 0.125 * 1.4426950408889634
def two_launch():
    K.gemm_qkv(x[:, :N1], w1, b1, c1, vt1, x, w2, b2, c2, vt2)
    K.qk_layernorm_rope_pair(c1[:, :, :D], c1[:, :, D:2 * D], H, *ln[:4], 1e-6, (Nt, rope), k_scale=ks, kmax=km1, kmax_ws=kws)
    K.qk_layernorm_rope_pair(c2[:, :, :D], c2[:, :, D:2 * D], H, *ln[4:], 1e-6, (Nt, vrope), (N1, crope), k_scale=ks, kmax=km2, kmax_ws=kws)
n1 = K.qk_norm_params(H, *ln[:4], 1e-6, (Nt, rope), None, k_scale=ks, kmax=km1, kmax_ws=kws)
n2 = K.qk_norm_params(H, *ln[4:], 1e-6, (Nt, vrope), (N1, crope), k_scale=ks, kmax=km2, kmax_ws=kws[kws.numel() // 2:])
def fused():
    K.gemm_qkv(x[:, :N1], w1, b1, c1, vt1, x, w2, b2, c2, vt2, norm1=n1, norm2=n2)
def gemm_only():
    K.gemm_qkv(x[:, :N1], w1, b1, c1, vt1, x, w2, b2, c2, vt2)
for r in range(3):
    print(json.dumps({"gemm_qkv alone ms": round(bk.timeit(gemm_only, iters=9), 4), "gemm_qkv + 2 x qk_layernorm_rope_pair_kmax ms": round(bk.timeit(two_launch, iters=9), 4),
                      "fused epilogue ms": round(bk.timeit(fused, iters=9), 4)}))
