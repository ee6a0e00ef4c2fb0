"""TEST HELPER (lives under tests/, not in the product package).  Run ONE CogVideoXBlock (with the To2V branch) at an arbitrary width through the HIP path — the unit
BASELINE config 1 measures (`CogVideoXBlock.forward`, cogvideox_transformer_3d.py:221-332).

The block is executed by a 1-layer CogVideoXTransformer3DModel whose residual stream / temb are injected
directly, so exactly the same kernels and launch sequence as the full model are exercised."""
import numpy as np
import torch

from tokensgen_amd import kernels as K
from tokensgen_amd import lib as L
from tokensgen_amd import rope as R
from tokensgen_amd.transformer import BF16, CogVideoXTransformer3DModel


def block_forward(model, layer, hidden, enc, temb, rope, vrope=None, crope=None):
    """hidden [B,Nv,D], enc [B,Nt+Np,D] (text|vip), temb [B,F,te] -> (hidden', enc').  Mirrors block.forward."""
    c, D, H = model.config, model.inner_dim, model.config.num_attention_heads
    B, Nv, _ = hidden.shape
    Np = model.vip_length if model.use_vip else 0
    Nt = enc.shape[1] - Np
    Fm = temb.shape[1]
    ws = model._workspace(B, Nt, Nv, Np, Fm)
    N1 = Nt + Nv
    ws.X[:, :Nt] = enc[:, :Nt]
    ws.X[:, Nt:N1] = hidden
    if Np:
        ws.X[:, N1:] = enc[:, Nt:]
    # silu(temb) -> modulation table (the full model gets silu(emb) from the timestep GEMM epilogue)
    ws.semb.copy_(torch.nn.functional.silu(temb.to(BF16)).reshape(B * Fm, -1))
    K.gemm(ws.semb.view(B, Fm, -1), model._fused["mod.w"], model._fused["mod.b"], ws.mod, L.EPI_BIAS)
    dev = lambda t: t.to(model.device, torch.float32).contiguous()
    model._run_block(layer, ws, B, Nt, Nv, Np, Fm, tuple(dev(t) for t in rope),
                     None if vrope is None else tuple(dev(t) for t in vrope),
                     None if crope is None else tuple(dev(t) for t in crope))
    out_h = ws.X[:, Nt:N1].clone()
    out_e = torch.cat([ws.X[:, :Nt], ws.X[:, N1:]], dim=1)
    return out_h, out_e


def run_full_width_block(state_dict, input_seed, device):
    """Build the seeded full-width block + inputs of BASELINE config 1 and run it on the GPU.

    `state_dict` (reference key names) must be supplied by the caller (tests/bench generate it with the
    oracle's seeded initialiser; the product itself never imports the oracle)."""
    D, heads = 3072, 48
    m = CogVideoXTransformer3DModel(num_attention_heads=heads, attention_head_dim=64, num_layers=1, time_embed_dim=512,
                                    text_embed_dim=4096, use_rotary_positional_embeddings=True, device=device)
    m.set_vip_layers(None, length=480, func_type="1", scale=[0.6],
                     resampler_params=dict(output_dim=3072, num_height_queries=8, num_width_queries=12, num_temporal_queries=4))
    m.load_state_dict({k: v.to(BF16) for k, v in state_dict.items()}, strict=False)
    g = torch.Generator().manual_seed(input_seed)
    hid = torch.randn(1, 17550, D, generator=g).to(device, BF16)
    enc = torch.randn(1, 706, D, generator=g).to(device, BF16)
    temb = torch.randn(1, 13, 512, generator=g).to(device, BF16)
    f32 = np.float32
    rope = R.rope_3d(64, np.arange(13, dtype=f32), np.arange(30, dtype=f32), np.arange(45, dtype=f32))
    crope = R.rope_3d(64, np.linspace(1000, 1016.25, 5, dtype=f32), np.linspace(0, 30, 8, endpoint=False, dtype=f32),
                      np.linspace(0, 45, 12, endpoint=False, dtype=f32))
    return block_forward(m, 0, hid, enc, temb, rope, rope, crope)
