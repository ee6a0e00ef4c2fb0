#!/usr/bin/env python3
"""Does ONE small kernel give the same bits every time while other processes share the GPU?  (round 6: the tiny To2V forward differed from itself in ~1 of 10 000 runs under sharing,
always in a few elements of the Q / K rows the QK-LayerNorm + RoPE kernel had just written.)   python tools/share_probe.py MODE N TAG
   qk       tg_qk_layernorm_rope_pair in place on a [2, 86, 384] fused buffer (the tiny model's shape), copy-in before every launch
   qkmax    the same through tg_qk_layernorm_rope_pair_kmax
   gemmgelu / gemmkeep   the 4-wave GEMM with its GELU epilogue (TG_EPI_BIAS_GELU / TG_EPI_BIAS_KEEP_GELU)
   torch    a chain of PyTorch's own kernels on the same buffer (layer_norm, mul, add, sin): the control — no code of this repository runs
   copy     only the copy-in (torch copy_) + comparison: the control of the control"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode, N, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]
DEV, BF = torch.device("cuda", 0), torch.bfloat16
g = torch.Generator().manual_seed(3)
B, T, H = 2, 86, 2
D = H * 64
src = torch.randn(B, T, 3 * D, generator=g).to(BF).to(DEV)
buf = torch.empty_like(src)
w = [(1 + 0.1 * torch.randn(64, generator=g)).to(BF).to(DEV) for _ in range(2)]
b_ = [(0.1 * torch.randn(64, generator=g)).to(BF).to(DEV) for _ in range(2)]
if mode in ("qk", "qkmax"):
    from tokensgen_amd import kernels as K
    from tokensgen_amd import rope as R
    f32 = np.float32
    rope = R.rope_3d(64, np.arange(13, dtype=f32), np.arange(2, dtype=f32), np.arange(3, dtype=f32), device=DEV)
    km, kws = torch.zeros(B, H, dtype=torch.float32, device=DEV), K.kmax_workspace(T, H, B, DEV)

    def op():
        buf.copy_(src)
        K.qk_layernorm_rope_pair(buf[:, :, :D], buf[:, :, D:2 * D], H, w[0], b_[0], w[1], b_[1], 1e-6, (8, rope), k_scale=0.18,
                                 kmax=km if mode == "qkmax" else None, kmax_ws=kws if mode == "qkmax" else None)
        return buf
elif mode in ("gemm", "gemmqk", "gemm_gap_qk", "gemmqk_out"):
    from tokensgen_amd import kernels as K
    from tokensgen_amd import lib as L
    from tokensgen_amd import rope as R
    f32 = np.float32
    rope = R.rope_3d(64, np.arange(13, dtype=f32), np.arange(2, dtype=f32), np.arange(3, dtype=f32), device=DEV)
    xn = torch.randn(B, T, D, generator=g).to(BF).to(DEV)
    wq = (torch.randn(3 * D, D, generator=g) * 0.1).to(BF).to(DEV)
    bq = torch.randn(3 * D, generator=g).to(BF).to(DEV)
    gap = torch.zeros(4096, device=DEV)
    out2 = torch.empty(B, T, 2 * D, dtype=BF, device=DEV)

    def op():
        K.gemm(xn, wq, bq, buf, L.EPI_BIAS)
        if mode == "gemm":
            return buf
        if mode == "gemm_gap_qk":
            gap.add_(1.0)                                    # an unrelated kernel between the projection and the norm
        if mode == "gemmqk_out":
            K.qk_layernorm_rope_pair(buf[:, :, :D], buf[:, :, D:2 * D], H, w[0], b_[0], w[1], b_[1], 1e-6, (8, rope), k_scale=0.18, out=(out2[:, :, :D], out2[:, :, D:]))
            return out2
        K.qk_layernorm_rope_pair(buf[:, :, :D], buf[:, :, D:2 * D], H, w[0], b_[0], w[1], b_[1], 1e-6, (8, rope), k_scale=0.18)
        return buf
elif mode in ("gemmgelu", "gemmkeep"):
    # the 4-wave GEMM's GELU epilogues (the library's remaining packed-fp32 instructions with an op_sel modifier: v_pk_mul_f32 / v_pk_fma_f32 by a broadcast constant) on a
    # shape where the epilogue is most of the kernel: [1024 x 256] x [512 x 256]^T
    from tokensgen_amd import kernels as K
    from tokensgen_amd import lib as L
    Mg, Ng, Kg = 1024, 512, 256
    assert K.gemm_act_supported(Mg, Ng, Kg)
    xg = torch.randn(Mg, Kg, generator=g).to(BF).to(DEV)
    wg = (torch.randn(Ng, Kg, generator=g) * 0.1).to(BF).to(DEV)
    bg = torch.randn(Ng, generator=g).to(BF).to(DEV)
    og, og2 = torch.empty(Mg, Ng, dtype=BF, device=DEV), torch.empty(Mg, Ng, dtype=BF, device=DEV)

    def op():
        if mode == "gemmgelu":
            K.gemm(xg, wg, bg, og, L.EPI_BIAS_GELU)
            return og
        K.gemm(xg, wg, bg, og, L.EPI_BIAS_KEEP_GELU, residual=og2)
        return og2
elif mode == "torch":
    wt = torch.randn(3 * D, generator=g).to(BF).to(DEV)

    def op():
        buf.copy_(src)
        y = torch.nn.functional.layer_norm(buf.float().view(B, T, 6, 64), (64,)).view(B, T, 3 * D)
        buf.copy_((y * wt.float() + torch.sin(y)).to(BF))
        return buf
else:
    def op():
        buf.copy_(src)
        return buf
ref = op().clone()
torch.cuda.synchronize()
bad = 0
for r in range(N):
    y = op()
    if not torch.equal(y, ref):
        bad += 1
        idx = (y != ref).nonzero()
        extra = ""
        if mode in ("gemmqk", "gemm_gap_qk"):
            ycopy = y.clone()
            pre = torch.empty_like(buf)
            K.gemm(xn, wq, bq, pre, L.EPI_BIAS)                 # the projection's own output (what the norm kernel read and overwrote in place)
            sel = tuple(idx.t())
            extra = f"; of them equal to the PRE-norm projection value at that place: {int((ycopy[sel] == pre[sel]).sum())}"
            # candidates for what the wrong values ARE: LayerNorm output without the rotation (x scale), the rotation with this lane's OTHER pair member, ...
            for (bi, ti, ci) in idx[:6].tolist():
                sec, hd, d = ci // D, (ci % D) // 64, ci % 64
                row = pre[bi, ti, sec * D + hd * 64: sec * D + hd * 64 + 64].float()
                ln = torch.nn.functional.layer_norm(row, (64,), w[sec].float(), b_[sec].float(), 1e-6).to(BF).float()
                sc = 1.0 if sec == 0 else 0.18
                cosv, sinv = (rope[0][ti - 8, d].item(), rope[1][ti - 8, d].item()) if ti >= 8 else (1.0, 0.0)
                partner = ln[d ^ 1].item()
                cands = {"ln*scale (no rotation)": ln[d].item() * sc, "a*cos (no sin term)": ln[d].item() * cosv * sc, "-b*sin only": -partner * sinv * sc,
                         "a*cos + b*sin (sign flipped)": (ln[d].item() * cosv + partner * sinv) * sc, "expected a*cos - b*sin": (ln[d].item() * cosv - partner * sinv) * sc,
                         "zero tables -> 0": 0.0}
                extra += f"\n      ({bi},{ti},{ci}): got {ycopy[bi, ti, ci].item():.6g} want {ref[bi, ti, ci].item():.6g} pre {pre[bi, ti, ci].item():.6g} | " + ", ".join(f"{k} {v:.5g}" for k, v in cands.items())
        print(f"[{tag}] {mode} launch {r}: {idx.shape[0]} elements differ, first {idx[:10].tolist()}{extra}", flush=True)
print(f"[{tag}] SHARE_PROBE {mode}: {bad} of {N} launches differed")
