"""GPU: the training step's first operator (BASELINE config 5) — flash-attention BACKWARD on MFMA (tg_attention_bwd) —
against torch.autograd of a plain fp32 restatement of the same op on the same bf16-rounded inputs.  Covers the single attention call
(ragged tiles, several heads / batch items, strided fused-QKV views) and the To2V processor's three-call composition
(attention_processor.py:2066-2135: O = cat(sdpa(q,k,v) + s * sdpa(qx,kv,vv), sdpa(qv, cat(kx,kv), cat(vx,vv)))) where K / V tensors are
shared between calls and their gradients accumulate."""
import math

import pytest
import torch
from conftest import measured

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return measured(((a - b).norm() / (b.norm() + 1e-12)).item())


def _sdpa(q, k, v, heads, scale):
    """[B, n, heads*64] fp32 tensors -> softmax(scale q k^T) v per head, merged back."""
    B, nq, _ = q.shape
    sp = lambda t: t.view(B, t.shape[1], heads, 64).transpose(1, 2)
    p = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * scale, dim=-1)
    return (p @ sp(v)).transpose(1, 2).reshape(B, nq, heads * 64)


def _rand(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


@pytest.mark.parametrize("B,H,nq,nk", [(1, 2, 50, 70), (2, 3, 513, 1500), (1, 1, 64, 64), (1, 4, 200, 33)])
def test_attention_bwd_vs_autograd(B, H, nq, nk):
    from tokensgen_amd import kernels as K
    scale = 1.0 / math.sqrt(64)
    # q|k|v as column slices of one fused buffer (what the QKV GEMM writes): exercises the row / batch strides
    nmax = max(nq, nk)
    fused = _rand(B, nmax, 3 * H * 64, seed=1, scale=1.5)
    q, k, v = fused[:, :nq, :H * 64], fused[:, :nk, H * 64:2 * H * 64], fused[:, :nk, 2 * H * 64:]
    g = _rand(B, nq, H * 64, seed=2)
    qf, kf, vf = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    o = _sdpa(qf, kf, vf, H, scale)
    (o * g.float()).sum().backward()
    fd = fused.to(DEV)
    qd, kd, vd = fd[:, :nq, :H * 64], fd[:, :nk, H * 64:2 * H * 64], fd[:, :nk, 2 * H * 64:]
    dq, dk, dv = K.attention_bwd(qd, kd, vd, o.detach().to(BF).to(DEV), g.to(DEV), H, scale)
    assert dq.shape == (B, nq, H * 64) and dk.shape == dv.shape == (B, nk, H * 64) and dq.dtype == torch.float32
    assert _rel(dv, vf.grad) < 8e-3
    assert _rel(dq, qf.grad) < 1.5e-2
    assert _rel(dk, kf.grad) < 1.5e-2
    # deterministic (no atomics) and accumulate adds
    dq2, dk2, dv2 = K.attention_bwd(qd, kd, vd, o.detach().to(BF).to(DEV), g.to(DEV), H, scale)
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)
    K.attention_bwd(qd, kd, vd, o.detach().to(BF).to(DEV), g.to(DEV), H, scale, dq=dq2, dk=dk2, dv=dv2, accumulate=True)
    assert torch.allclose(dq2, 2 * dq) and torch.allclose(dk2, 2 * dk) and torch.allclose(dv2, 2 * dv)


def test_to2v_processor_attention_gradients():
    """The three SDPA calls of VideoIPAdapterCogVideoXAttnProcessor2_0 (func_type "1") composed from tg_attention_bwd calls: gradients of
    every q / k / v tensor of the base and the vip projections (the vip ones are the TRAINABLE path, cogvideox_transformer_3d.py:207-218)
    against autograd of the fp32 restatement."""
    from tokensgen_amd import kernels as K
    B, H, N1, Np, s = 2, 2, 150, 40, 0.6
    scale = 1.0 / math.sqrt(64)
    names = ("q", "k", "v", "qx", "kx", "vx", "qv", "kv", "vv")
    shapes = dict(q=N1, k=N1, v=N1, qx=N1, kx=N1, vx=N1, qv=Np, kv=Np, vv=Np)
    t = {n: _rand(B, shapes[n], H * 64, seed=10 + i, scale=1.2) for i, n in enumerate(names)}
    g = _rand(B, N1 + Np, H * 64, seed=30)
    f = {n: x.float().clone().requires_grad_(True) for n, x in t.items()}
    o1 = _sdpa(f["q"], f["k"], f["v"], H, scale)
    o2 = _sdpa(f["qx"], f["kv"], f["vv"], H, scale)
    o3 = _sdpa(f["qv"], torch.cat([f["kx"], f["kv"]], 1), torch.cat([f["vx"], f["vv"]], 1), H, scale)
    out = torch.cat([o1 + s * o2, o3], dim=1)
    (out * g.float()).sum().backward()
    d = {n: x.to(DEV) for n, x in t.items()}
    gd = g.to(DEV)
    bf = lambda x: x.detach().to(BF).to(DEV)
    f32 = torch.float32
    # SDPA #1
    dq, dk, dv = K.attention_bwd(d["q"], d["k"], d["v"], bf(o1), gd[:, :N1], H, scale)
    # SDPA #2: dO2 = s * dO (bf16, like the forward's `scale * O2` tensor)
    g2 = (gd[:, :N1].float() * s).to(BF)
    dqx, dkv, dvv = K.attention_bwd(d["qx"], d["kv"], d["vv"], bf(o2), g2, H, scale)
    # SDPA #3: keys / values = cat(x-part, vip-part): gradients land in one [N1 + Np] buffer, whose tail ADDS to SDPA #2's kv / vv gradients
    kcat, vcat = torch.cat([d["kx"], d["kv"]], 1), torch.cat([d["vx"], d["vv"]], 1)
    dkcat = torch.zeros(B, N1 + Np, H * 64, dtype=f32, device=DEV)
    dvcat = torch.zeros_like(dkcat)
    dkcat[:, N1:], dvcat[:, N1:] = dkv, dvv
    dqv, _, _ = K.attention_bwd(d["qv"], kcat, vcat, bf(o3), gd[:, N1:], H, scale, dk=dkcat, dv=dvcat, accumulate=True,
                                dq=torch.zeros(B, Np, H * 64, dtype=f32, device=DEV))
    got = dict(q=dq, k=dk, v=dv, qx=dqx, kx=dkcat[:, :N1], vx=dvcat[:, :N1], qv=dqv, kv=dkcat[:, N1:], vv=dvcat[:, N1:])
    for n in names:
        assert _rel(got[n], f[n].grad) < 2e-2, n


def test_vpred_loss_and_gradient_vs_autograd():
    """tg_vpred_loss_grad against autograd of the oracle's restatement of train_cogvideo_to2v.py:1990-2010 (bf16 tensors, per-frame timesteps)."""
    from oracle import scheduler_ref as S
    from oracle import train_ref as T
    from tokensgen_amd import train
    _, ac = S.alphas_cumprod()
    ac = torch.as_tensor(ac, dtype=torch.float32)
    B, F, C, H, W = 2, 13, 16, 6, 10
    g = torch.Generator().manual_seed(5)
    out, noisy, x0 = (torch.randn(B, F, C, H, W, generator=g).to(BF) for _ in range(3))
    ts = torch.randint(20, 980, (B, F), generator=g)
    o = out.clone().requires_grad_(True)
    loss, per_item = T.vpred_loss(ac, o, noisy, x0, ts)
    loss.backward()
    l2, p2, grad = train.vpred_loss_and_grad(out.to(DEV), noisy.to(DEV), x0.to(DEV), ts, ac)
    assert grad.shape == out.shape and grad.dtype == BF
    assert abs(l2.item() - loss.item()) < 2e-3 * abs(loss.item())
    assert torch.allclose(p2.cpu(), per_item.detach().float(), rtol=2e-3)
    assert _rel(grad, o.grad) < 1e-2
    # scalar timestep per batch item
    ts1 = torch.tensor([100, 900])
    o = out.clone().requires_grad_(True)
    loss, _ = T.vpred_loss(ac, o, noisy, x0, ts1)
    loss.backward()
    l3, _, grad3 = train.vpred_loss_and_grad(out.to(DEV), noisy.to(DEV), x0.to(DEV), ts1, ac)
    assert abs(l3.item() - loss.item()) < 2e-3 * abs(loss.item()) and _rel(grad3, o.grad) < 1e-2
