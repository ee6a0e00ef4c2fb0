#!/usr/bin/env python3
"""Kernel micro-benchmarks at the CogVideoX-5B To2V shapes (GPU box only): TFLOP/s per hot kernel.
Usage: python tools/bench_kernels.py [attn] [attn_bwd] [gemm] [norm] [train_elem]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tokensgen_amd import kernels as K  # noqa: E402
from tokensgen_amd import lib as L  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16
B, H, D, NT, NV, NP = 2, 48, 3072, 226, 17550, 480
N1, N = NT + NV, NT + NV + NP


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in evs)
    return ms[len(ms) // 2]


def rnd(*shape, scale=1.0):
    if os.environ.get("TG_BENCH_ZERO") == "1":       # power probe: the same launches on all-zero operands (timing only)
        return torch.zeros(*shape, device=DEV, dtype=BF)
    return (torch.randn(*shape, device=DEV, dtype=torch.float32) * scale).to(BF)


def bench_attn():
    PRE = os.environ.get("TG_BENCH_PRESCALED", "1") == "1"      # the model folds scale*log2(e) into K
    qkv = rnd(B, N1, 3 * D, scale=0.4 if PRE else 1.0)
    qkvv = rnd(B, N, 3 * D, scale=0.4 if PRE else 1.0)
    pad = lambda n: (n + 63) // 64 * 64
    vt1 = torch.empty(B, H, 64, pad(N1), dtype=BF, device=DEV)
    vt2 = torch.empty(B, H, 64, pad(NP), dtype=BF, device=DEV)
    vt3 = torch.empty(B, H, 64, pad(N), dtype=BF, device=DEV)
    K.transpose_v(qkv[:, :, 2 * D:], H, 0, N1, vt1)
    K.transpose_v(qkvv[:, :, 2 * D:], H, N1, NP, vt2)
    K.transpose_v(qkvv[:, :, 2 * D:], H, 0, N, vt3)
    ao = torch.empty(B, N, D, dtype=BF, device=DEV)
    f_main = lambda: K.attention(qkv[:, :, :D], qkv[:, :, D:2 * D], vt1, N1, ao[:, :N1], H, 0.125,
                                 qkvv[:, :N1, :D], qkvv[:, N1:, D:2 * D], vt2, NP, 0.6, k_prescaled=PRE)
    f_vip = lambda: K.attention(qkvv[:, N1:, :D], qkvv[:, :, D:2 * D], vt3, N, ao[:, N1:], H, 0.125)
    ms = timeit(f_main)
    fl = B * (4.0 * N1 * N1 * D + 4.0 * N1 * NP * D)
    print(json.dumps({"kernel": "attention_main_2seg", "ms": ms, "tflops": fl / ms / 1e9}))
    ms = timeit(f_vip)
    print(json.dumps({"kernel": "attention_vip", "ms": ms, "tflops": B * 4.0 * NP * N * D / ms / 1e9}))


def bench_attn_dit():
    """The To2V block's attention launch as the model issues it (tg_attention_fwd_multi: SDPA#1+#2 with SDPA#3 riding), q / k from the real
    K-norm kernel (LayerNorm gains ~ TG_BENCH_GAIN, default 1), on both softmax paths: constant shift (+ its retry launch) and running max."""
    gain = float(os.environ.get("TG_BENCH_GAIN", "1"))
    kscale = 0.125 * 1.4426950408889634
    qkv, qkvv = rnd(B, N1, 3 * D), rnd(B, N, 3 * D)
    w = lambda: ((1.0 + 0.05 * torch.randn(64, device=DEV)) * gain).to(BF)
    bb = lambda: (0.05 * torch.randn(64, device=DEV)).to(BF)
    km1, km2 = (torch.zeros(B, H, dtype=torch.float32, device=DEV) for _ in range(2))
    kws = K.kmax_workspace(N, H, B, DEV)
    wq, bq, wk, bk = w(), bb(), w(), bb()
    K.qk_layernorm_rope_pair(qkv[:, :, :D], qkv[:, :, D:2 * D], H, wq, bq, wk, bk, 1e-6, k_scale=kscale, kmax=km1, kmax_ws=kws)
    K.qk_layernorm_rope_pair(qkvv[:, :, :D], qkvv[:, :, D:2 * D], H, wq, bq, wk, bk, 1e-6, k_scale=kscale, kmax=km2, kmax_ws=kws)
    pad = lambda n: (n + 63) // 64 * 64
    vt1 = torch.zeros(B, H, 64, pad(N1), dtype=BF, device=DEV); K.transpose_v(qkv[:, :, 2 * D:], H, 0, N1, vt1)
    vt3 = torch.zeros(B, H, 64, pad(N), dtype=BF, device=DEV); K.transpose_v(qkvv[:, :, 2 * D:], H, 0, N, vt3)
    vt2 = vt3[:, :, :, N1:]
    ao = torch.empty(B, N, D, dtype=BF, device=DEV)
    retry = K.AttnRetry(N1, NP, H, B, DEV)

    def run(fast):
        K.attention_multi(dict(q1=qkv[:, :, :D], k1=qkv[:, :, D:2 * D], vt1=vt1, nk1=N1, out=ao[:, :N1], q2=qkvv[:, :N1, :D], k2=qkvv[:, N1:, D:2 * D],
                               vt2=vt2, nk2=NP, seg2_scale=0.6, kmax1=km1 if fast else None, kmax2=km2 if fast else None),
                          dict(q1=qkvv[:, N1:, :D], k1=qkvv[:, :, D:2 * D], vt1=vt3, nk1=N, out=ao[:, N1:], kmax1=km2 if fast else None), H, 0.125,
                          k_prescaled=True, retry=retry if fast else None, split=retry.split if os.environ.get("TG_BENCH_SPLIT", "1") == "1" else None)
    fl = B * (4.0 * N1 * N1 * D + 4.0 * N1 * NP * D + 4.0 * NP * N * D)
    res = {}
    for name, fast in (("constant_shift", True), ("running_max", False)):
        ms = timeit(lambda: run(fast), iters=7)
        res[name] = {"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}
    qk_ms = timeit(lambda: K.qk_layernorm_rope_pair(qkv[:, :, :D], qkv[:, :, D:2 * D], H, wq, bq, wk, bk, 1e-6, k_scale=1.0, kmax=km1, kmax_ws=kws))
    qk0_ms = timeit(lambda: K.qk_layernorm_rope_pair(qkv[:, :, :D], qkv[:, :, D:2 * D], H, wq, bq, wk, bk, 1e-6, k_scale=1.0))
    print(json.dumps({"kernel": "attention_dit_launch", "gain": gain, **res, "retried": retry.count(), "qk_norm_pair_kmax_ms": round(qk_ms, 4),
                      "qk_norm_pair_ms": round(qk0_ms, 4)}))


def bench_attn_bwd():
    """The training step's main attention call (SDPA #1 of the To2V processor): nq = nk = 17776, 48 heads, batch 2."""
    qkv = rnd(B, N1, 3 * D, scale=0.6)
    o, do = rnd(B, N1, D, scale=0.3), rnd(B, N1, D, scale=0.3)
    f32 = torch.float32
    dq, dk, dv = (torch.empty(B, N1, D, dtype=f32, device=DEV) for _ in range(3))
    # TG_BENCH_BWD_UNIT=1: the training step's form — K prescaled by scale * log2 e, scale = ln 2 (the kernel's exp2 needs no multiply)
    unit = os.environ.get("TG_BENCH_BWD_UNIT") == "1"
    import math
    kk = (qkv[:, :, D:2 * D].float() * (0.125 * 1.4426950408889634)).to(BF) if unit else qkv[:, :, D:2 * D]
    fn = lambda: K.attention_bwd(qkv[:, :, :D], kk, qkv[:, :, 2 * D:], o, do, H, math.log(2.0) if unit else 0.125, dq=dq, dk=dk, dv=dv)
    ms = timeit(fn, iters=3, warm=1)
    K.attention_bwd_check()
    fl = 5 * 2.0 * B * N1 * N1 * D
    print(json.dumps({"kernel": "attention_bwd_main", "unit_scale": unit, "pp": os.environ.get("TG_ATTN_BWD_PP", "1"), "ms": ms, "tflops_algorithmic(5 GEMMs)": fl / ms / 1e9, "tflops_executed(7 GEMMs)": fl * 1.4 / ms / 1e9}))


def bench_gemm():
    for (M, Nn, Kk, epi, name) in [(N1, 3 * D, D, L.EPI_BIAS, "qkv"), (N, D, D, L.EPI_BIAS, "out(bias)"),
                                   (N, 4 * D, D, L.EPI_BIAS_GELU, "ff1"), (N, D, 4 * D, L.EPI_BIAS, "ff2(bias)")]:
        a, w, bias = rnd(B, M, Kk), rnd(Nn, Kk, scale=0.02), rnd(Nn)
        out = torch.empty(B, M, Nn, dtype=BF, device=DEV)
        ms = timeit(lambda: K.gemm(a, w, bias, out, epi))
        print(json.dumps({"kernel": f"gemm_{name}_M{M}_N{Nn}_K{Kk}", "ms": ms, "tflops": 2.0 * B * M * Nn * Kk / ms / 1e9}))
    # the block's two gated-residual GEMMs as the DiT runs them (in place on the hidden states, 2 token groups)
    mod = rnd(B, 2, 6 * D, scale=0.5)
    tok_group = (torch.arange(N, device=DEV) >= N - N1).to(torch.uint8)
    tab = K.GroupTable(mod, tok_group, [0, 1], [0, 0], [D, D], [2 * D, 2 * D])
    for (Kk, name) in [(D, "out(gate)"), (4 * D, "ff2(gate)")]:
        a, w, bias = rnd(B, N, Kk), rnd(D, Kk, scale=0.02), rnd(D)
        x = rnd(B, N, D)
        ms = timeit(lambda: K.gemm(a, w, bias, x, L.EPI_BIAS_GATE_RES, residual=x, gate=tab))
        print(json.dumps({"kernel": f"gemm_{name}_M{N}_N{D}_K{Kk}", "ms": ms, "tflops": 2.0 * B * N * D * Kk / ms / 1e9}))


def bench_norm():
    x, y = rnd(B, N, D), torch.empty(B, N, D, dtype=BF, device=DEV)
    w, b = rnd(D), rnd(D)
    ms = timeit(lambda: K.adaln_modulate(x, y, w, b, 1e-5, None))
    print(json.dumps({"kernel": "adaln(plain)", "ms": ms, "GBps": 2 * x.numel() * 2 / ms / 1e6}))
    qkv = rnd(B, N, 3 * D)
    ms = timeit(lambda: K.qk_layernorm_rope(qkv[:, :, :D], H, w[:64], b[:64], 1e-6))
    print(json.dumps({"kernel": "qk_layernorm_rope", "ms": ms, "GBps": 2 * B * N * D * 2 / ms / 1e6}))
    vt = torch.empty(B, H, 64, (N + 63) // 64 * 64, dtype=BF, device=DEV)
    ms = timeit(lambda: K.transpose_v(qkv[:, :, 2 * D:], H, 0, N, vt))
    print(json.dumps({"kernel": "transpose_v", "ms": ms, "GBps": 2 * B * N * D * 2 / ms / 1e6}))


def bench_train_elem():
    """The HBM-bound passes of the training backward at the block's shapes (frozen-norm AdaLN backward with the residual add; GELU forward / backward)."""
    from tokensgen_amd import train
    x, dy, add, dx = rnd(B, N1, D), rnd(B, N1, D), rnd(B, N1, D), torch.empty(B, N1, D, dtype=BF, device=DEV)
    w, b = rnd(D), rnd(D)
    tok = torch.zeros(N1, dtype=torch.uint8, device=DEV)
    table = K.GroupTable(rnd(B, 1, 6 * D, scale=0.3), tok, [0], [0], [D], [2 * D])
    ms = timeit(lambda: train._adaln_bwd(x, dy, dx, w, b, 1e-5, table, products=False, add=add))
    print(json.dumps({"kernel": "adaln_bwd(frozen norm, modulated, + residual)", "ms": ms, "GBps": 4 * x.numel() * 2 / ms / 1e6}))
    pre, dh = rnd(B * N, 4 * D), rnd(B * N, 4 * D)
    ms = timeit(lambda: train._act(pre, gelu=True))
    print(json.dumps({"kernel": "gelu forward pass", "ms": ms, "GBps": 2 * pre.numel() * 2 / ms / 1e6}))
    ms = timeit(lambda: train._act(pre, dh))
    print(json.dumps({"kernel": "gelu backward pass", "ms": ms, "GBps": 3 * pre.numel() * 2 / ms / 1e6}))


if __name__ == "__main__":
    what = sys.argv[1:] or ["attn", "gemm", "norm"]
    if "attn" in what:
        bench_attn()
    if "attn_dit" in what:
        bench_attn_dit()
    if "attn_bwd" in what:
        bench_attn_bwd()
    if "gemm" in what:
        bench_gemm()
    if "norm" in what:
        bench_norm()
    if "train_elem" in what:
        bench_train_elem()
