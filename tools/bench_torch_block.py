#!/usr/bin/env python3
"""Reference point only (never on the product path, imports nothing from oracle/): ONE To2V CogVideoX block forward written with the plain PyTorch ops
the reference runs on a GPU — F.layer_norm, F.linear (hipBLASLt), F.scaled_dot_product_attention (the ROCm flash kernel torch ships), F.gelu — in the op
order of SURVEY App. B (cogvideox_transformer_3d.py:221-332 + attention_processor.py:1982-2155), bf16, B = 2, the headline shapes (226 text + 17550 video
+ 480 condensed tokens, D = 3072, 48 heads), random weights.  Printed next to the same block through tokensgen_amd on the same box: what PyTorch-ROCm
eager would give the reference on this GPU, per block and extrapolated to the 42-layer CFG step."""
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DEV, BF = "cuda", torch.bfloat16
B, H, D, NT, NV, NP, FR = 2, 48, 3072, 226, 17550, 480, 13
N1, N = NT + NV, NT + NV + NP


def rnd(*s, sc=1.0):
    return (torch.randn(*s, device=DEV) * sc).to(BF)


def rope(x, cos, sin):
    """rotate pairs (2i, 2i+1) of the last axis (apply_rotary_emb, use_real_unbind_dim=-1)"""
    xr, xi = x.float().reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(-2)
    return (x.float() * cos + rot * sin).to(x.dtype)


class TorchBlock:
    def __init__(self):
        lin = lambda o, i: (rnd(o, i, sc=0.02), rnd(o, sc=0.02))
        self.mod1, self.mod2 = lin(6 * D, 512), lin(6 * D, 512)
        self.vmod1, self.vmod2 = lin(3 * D, 512), lin(3 * D, 512)
        self.ln = [(torch.ones(D, device=DEV, dtype=BF), torch.zeros(D, device=DEV, dtype=BF)) for _ in range(4)]       # norm1, vip_norm1, norm2, vip_norm2
        self.qkv = [lin(D, D) for _ in range(3)]
        self.vqkv = [lin(D, D) for _ in range(3)]
        self.hn = [(torch.ones(64, device=DEV, dtype=BF), torch.zeros(64, device=DEV, dtype=BF)) for _ in range(4)]      # norm_q, norm_k, vip_norm_q, vip_norm_k
        self.out = lin(D, D)
        self.ff1, self.ff2 = lin(4 * D, D), lin(D, 4 * D)
        ang = torch.rand(NV, 32, device=DEV) * 6.28
        self.cs = (torch.cos(ang).repeat_interleave(2, -1), torch.sin(ang).repeat_interleave(2, -1))
        ang2 = torch.rand(NP, 32, device=DEV) * 6.28
        self.ccs = (torch.cos(ang2).repeat_interleave(2, -1), torch.sin(ang2).repeat_interleave(2, -1))

    def heads(self, t):
        return t.view(B, t.shape[1], H, 64).transpose(1, 2)

    def adaln(self, h, text, vip, temb, mod, vmod, ln, vln):
        m = F.linear(F.silu(temb), *mod)                                   # [B, F, 6D]
        sh, sc, g, esh, esc, eg = m.chunk(6, dim=-1)
        per = NV // FR
        exp = lambda t: t.repeat_interleave(per, dim=1)                  # per-frame rows -> per-token rows
        nh = F.layer_norm(h, (D,), *ln) * (1 + exp(sc)) + exp(sh)
        nt = F.layer_norm(text, (D,), *ln) * (1 + esc[:, :1]) + esh[:, :1]
        vm = F.linear(F.silu(temb[:, :1]), *vmod)
        vsh, vsc, vg = vm.chunk(3, dim=-1)
        nv = F.layer_norm(vip, (D,), *vln) * (1 + vsc) + vsh
        return nh, nt, nv, exp(g), eg[:, :1], vg

    def forward(self, h, e, temb):
        text, vip = e[:, :NT], e[:, NT:]
        nh, nt, nv, g, eg, vg = self.adaln(h, text, vip, temb, self.mod1, self.vmod1, self.ln[0], self.ln[1])
        x = torch.cat([nt, nh], dim=1)
        q, k, v = (self.heads(F.linear(x, *w)) for w in self.qkv)
        qx, kx, vx = (self.heads(F.linear(x, *w)) for w in self.vqkv)
        qv, kv, vv = (self.heads(F.linear(nv, *w)) for w in self.vqkv)
        hn = lambda t, i: F.layer_norm(t, (64,), *self.hn[i], eps=1e-6)
        q, k, qx, qv, kx, kv = hn(q, 0), hn(k, 1), hn(qx, 2), hn(qv, 2), hn(kx, 3), hn(kv, 3)

        def rot_tail(t):                                                  # text rows are never rotated
            return torch.cat([t[:, :, :NT], rope(t[:, :, NT:], *self.cs)], dim=2)
        q, k, qx, kx = rot_tail(q), rot_tail(k), rot_tail(qx), rot_tail(kx)
        qv, kv = rope(qv, *self.ccs), rope(kv, *self.ccs)
        o1 = F.scaled_dot_product_attention(q, k, v)
        o2 = F.scaled_dot_product_attention(qx, kv, vv)
        o3 = F.scaled_dot_product_attention(qv, torch.cat([kx, kv], dim=2), torch.cat([vx, vv], dim=2))
        o = torch.cat([o1 + 0.6 * o2, o3], dim=2).transpose(1, 2).reshape(B, N, D)
        a = F.linear(o, *self.out)
        h = h + g * a[:, NT:N1]
        text = text + eg * a[:, :NT]
        vip = vip + vg * a[:, N1:]
        nh, nt, nv, g, eg, vg = self.adaln(h, text, vip, temb, self.mod2, self.vmod2, self.ln[2], self.ln[3])
        ff = lambda t: F.linear(F.gelu(F.linear(t, *self.ff1), approximate="tanh"), *self.ff2)
        y = ff(torch.cat([nt, nh], dim=1))
        h = h + g * y[:, NT:]
        text = text + eg * y[:, :NT]
        vip = vip + vg * ff(nv)
        return h, torch.cat([text, vip], dim=1)


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


@torch.no_grad()
def main():
    blk = TorchBlock()
    h, e, temb = rnd(B, NV, D), rnd(B, NT + NP, D), rnd(B, FR, 512)
    ms_torch = timeit(lambda: blk.forward(h, e, temb))
    res = {"torch_eager_block_ms": ms_torch, "torch_eager_steps_per_s_extrapolated_x42": 1e3 / (42 * ms_torch),
           "flops_per_block_B2": 2 * 9.237e12, "torch_eager_TFLOPs": 2 * 9.237 / ms_torch * 1e3}
    # the same block through the product (one layer of the real model class, same shapes), same box
    sys.argv = [sys.argv[0]]
    import bench
    from tokensgen_amd import rope as R
    model = bench.build_model(torch.device(DEV), 1)
    f32 = np.float32
    lat = rnd(B, FR, 16, 60, 90)
    prompt = rnd(B, NT, 4096, sc=0.1)
    emb = rnd(B, 5, 3072, 8, 12)
    tt = torch.full((B, FR), 500, dtype=torch.int64, device=DEV)
    ropes = R.rope_3d_crop(64, (0, 0, 0), (FR, 30, 45), (FR, 30, 45))
    vr = R.rope_3d(64, np.arange(FR, dtype=f32), np.arange(30, dtype=f32), np.arange(45, dtype=f32))
    cr = R.rope_3d(64, np.linspace(1000, 1016.25, 5, dtype=f32), np.linspace(0, 30, 8, endpoint=False, dtype=f32), np.linspace(0, 45, 12, endpoint=False, dtype=f32))
    fn = lambda: model(hidden_states=lat, encoder_hidden_states=prompt, timestep=tt, image_rotary_emb=ropes, vip_image_rotary_emb=vr,
                       vip_condition_rotary_emb=cr, vip_encoder_hidden_states=emb, return_dict=False)
    ms1 = timeit(fn)
    model2 = bench.build_model(torch.device(DEV), 3)
    fn2 = lambda: model2(hidden_states=lat, encoder_hidden_states=prompt, timestep=tt, image_rotary_emb=ropes, vip_image_rotary_emb=vr,
                         vip_condition_rotary_emb=cr, vip_encoder_hidden_states=emb, return_dict=False)
    ms3 = timeit(fn2)
    per_block = (ms3 - ms1) / 2.0                     # front / back ends cancel
    res.update({"tokensgen_block_ms": per_block, "tokensgen_TFLOPs": 2 * 9.237 / per_block * 1e3, "speedup_per_block": ms_torch / per_block})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
