"""Oracle: CogVideoX-5B DiT forward with the To2V condensed-token ("vip") branch.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Functional restatement over a plain
state-dict whose keys are the reference's parameter names.  Runs in whatever
dtype the state-dict / inputs carry (fp32 or bf16) using stock torch CPU ops, so per-op rounding
follows the reference's PyTorch path.

Reference: longvgen/models/cogvideox_transformer_3d.py, attention_processor.py,
normalization.py, embeddings.py (line numbers per function below).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# RoPE tables  (embeddings.py:774-828 get_1d_rotary_pos_embed, :641-707 get_3d_rotary_pos_embed_v2)
# ----------------------------------------------------------------------------------------------
def rope_1d(dim, pos, theta=10000.0):
    """cos/sin [S, dim], each frequency repeated twice (interleaved pairs). embeddings.py:806-819"""
    pos = torch.as_tensor(np.asarray(pos))
    inv = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
    ang = torch.outer(pos, inv)
    return (ang.cos().repeat_interleave(2, dim=1).float(),
            ang.sin().repeat_interleave(2, dim=1).float())


def rope_3d(head_dim, grid_t, grid_h, grid_w, dim_t=None, dim_h=None, dim_w=None):
    """3-D RoPE table for tokens ordered (t, h, w). embeddings.py:641-707 (and :571-639 via linspace grids).

    Returns (cos, sin), each [T*H*W, head_dim] fp32; channel split t|h|w = d/4 | 3d/8 | 3d/8."""
    dim_t = head_dim // 4 if dim_t is None else dim_t
    dim_h = head_dim // 8 * 3 if dim_h is None else dim_h
    dim_w = head_dim // 8 * 3 if dim_w is None else dim_w
    T, H, W = len(grid_t), len(grid_h), len(grid_w)
    out = []
    for which in (0, 1):
        ft = rope_1d(dim_t, grid_t)[which][:, None, None, :].expand(T, H, W, dim_t)
        fh = rope_1d(dim_h, grid_h)[which][None, :, None, :].expand(T, H, W, dim_h)
        fw = rope_1d(dim_w, grid_w)[which][None, None, :, :].expand(T, H, W, dim_w)
        out.append(torch.cat([ft, fh, fw], dim=-1).reshape(T * H * W, -1))
    return out[0], out[1]


def rope_3d_crop(head_dim, start, stop, grid_size):
    """embeddings.py:571-639: grids are linspace(start, stop, n, endpoint=False) in fp32."""
    gt, gh, gw = (np.linspace(start[i], stop[i], grid_size[i], endpoint=False, dtype=np.float32) for i in range(3))
    return rope_3d(head_dim, gt, gh, gw)


def apply_rope(x, cos_sin):
    """x [B,H,S,D]; pairs (x0,x1)->(x0 c - x1 s, x1 c + x0 s); fp32 math then cast. embeddings.py:866-885"""
    cos, sin = cos_sin
    cos, sin = cos[None, None], sin[None, None]
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos + rot.float() * sin).to(x.dtype)


# ----------------------------------------------------------------------------------------------
# small pieces
# ----------------------------------------------------------------------------------------------
def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ln(sd, name, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd.get(name + ".weight"), sd.get(name + ".bias"), eps)


def timestep_sinusoid(timesteps, dim, flip_sin_to_cos=True, freq_shift=0.0):
    """embeddings.py:28-79 — [N] -> [N, dim] fp32; (cos | sin) when flipped."""
    half = dim // 2
    expo = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / (half - freq_shift)
    ang = timesteps[:, None].float() * torch.exp(expo)[None, :]
    emb = torch.cat([ang.sin(), ang.cos()], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def time_embedding(sd, timestep, batch, inner_dim, dtype):
    """cogvideox_transformer_3d.py:669-680 + embeddings.py:920-965. timestep [B] or [B,F] -> [B,F',512]."""
    t = timestep.reshape(-1)
    e = timestep_sinusoid(t, inner_dim).to(dtype)
    e = _lin(sd, "time_embedding.linear_1", e)
    e = F.silu(e)
    e = _lin(sd, "time_embedding.linear_2", e)
    return e.reshape(batch, -1, e.shape[-1])


def patch_embed(sd, text, latents, vip, patch):
    """embeddings.py:502-568 (rotary model: no abs-pos-emb). Returns [B, Nt+Nv+Np, D] = text|video|vip."""
    text = _lin(sd, "patch_embed.text_proj", text)
    parts = [text]
    B, Fr, C, H, W = latents.shape
    x = F.conv2d(latents.reshape(-1, C, H, W), sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=patch)
    x = x.view(B, Fr, *x.shape[1:]).flatten(3).transpose(2, 3).flatten(1, 2)
    parts.append(x)
    if vip is not None:
        b, f, c, h, w = vip.shape
        v = vip.permute(0, 1, 3, 4, 2).reshape(b, f * h * w, c)
        parts.append(_lin(sd, "patch_embed.vip_proj", v))
    return torch.cat(parts, dim=1).contiguous()


def layer_norm_zero(sd, pre, hidden, enc, temb, eps):
    """normalization.py:441-460. temb [B,F,C]; video tokens use per-frame modulation, enc uses frame 0."""
    B, Fr, _ = temb.shape
    hw = hidden.shape[1] // Fr
    mod = _lin(sd, pre + ".linear", F.silu(temb.reshape(B * Fr, -1)))
    shift, scale, gate, esh, esc, eg = (c.reshape(B, Fr, -1) for c in mod.chunk(6, dim=1))
    shift, scale, gate = (c.repeat_interleave(hw, dim=1) for c in (shift, scale, gate))
    nh = _ln(sd, pre + ".norm", hidden, eps) * (1 + scale) + shift
    ne = _ln(sd, pre + ".norm", enc, eps) * (1 + esc)[:, [0], :] + esh[:, [0], :]
    return nh, ne, gate, eg[:, [0], :]


def vip_layer_norm_zero(sd, pre, vip, temb, eps):
    """normalization.py:477-488: 3 chunks, frame-0 modulation only."""
    B, Fr, _ = temb.shape
    mod = _lin(sd, pre + ".linear", F.silu(temb.reshape(B * Fr, -1)))
    sh, sc, g = (c.reshape(B, Fr, -1) for c in mod.chunk(3, dim=1))
    return _ln(sd, pre + ".norm", vip, eps) * (1 + sc)[:, [0], :] + sh[:, [0], :], g[:, [0], :]


def _heads(x, heads):
    B, S, D = x.shape
    return x.view(B, S, heads, D // heads).transpose(1, 2)


def plain_attention(sd, pre, hidden, enc, heads, rope):
    """attention_processor.py:1895-1953 (CogVideoXAttnProcessor2_0): joint attention, RoPE on video rows."""
    Nt = enc.shape[1]
    x = torch.cat([enc, hidden], dim=1)
    q, k, v = (_heads(_lin(sd, f"{pre}.to_{n}", x), heads) for n in "qkv")
    q = _ln(sd, pre + ".norm_q", q, 1e-6)
    k = _ln(sd, pre + ".norm_k", k, 1e-6)
    if rope is not None:
        q[:, :, Nt:] = apply_rope(q[:, :, Nt:], rope)
        k[:, :, Nt:] = apply_rope(k[:, :, Nt:], rope)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(x.shape[0], -1, x.shape[-1])
    o = _lin(sd, pre + ".to_out.0", o)
    return o[:, Nt:], o[:, :Nt]


def vip_attention(sd, pre, hidden, enc, heads, n_vip, scale, rope, vip_rope, cond_rope, taps=None):
    """attention_processor.py:1982-2155 (VideoIPAdapterCogVideoXAttnProcessor2_0, func_type "1").

    enc = text|vip.  Three SDPAs: main (base weights), text+video -> vip K/V (vip weights),
    vip -> all K/V (vip weights).  out rows text+video = O1 + scale*O2, vip rows = O3."""
    text, vip = enc[:, :-n_vip], enc[:, -n_vip:]
    Nt = text.shape[1]
    x = torch.cat([text, hidden], dim=1)
    P = pre + ".processor"
    q, k, v = (_heads(_lin(sd, f"{pre}.to_{n}", x), heads) for n in "qkv")
    qx, kx, vx = (_heads(_lin(sd, f"{P}.vip_to_{n}", x), heads) for n in "qkv")
    qv, kv, vv = (_heads(_lin(sd, f"{P}.vip_to_{n}", vip), heads) for n in "qkv")
    q = _ln(sd, pre + ".norm_q", q, 1e-6)
    k = _ln(sd, pre + ".norm_k", k, 1e-6)
    qx, qv = _ln(sd, P + ".vip_norm_q", qx, 1e-6), _ln(sd, P + ".vip_norm_q", qv, 1e-6)
    kx, kv = _ln(sd, P + ".vip_norm_k", kx, 1e-6), _ln(sd, P + ".vip_norm_k", kv, 1e-6)
    if rope is not None:
        q[:, :, Nt:] = apply_rope(q[:, :, Nt:], rope)
        qx[:, :, Nt:] = apply_rope(qx[:, :, Nt:], vip_rope)
        qv = apply_rope(qv, cond_rope)
        k[:, :, Nt:] = apply_rope(k[:, :, Nt:], rope)
        kx[:, :, Nt:] = apply_rope(kx[:, :, Nt:], vip_rope)
        kv = apply_rope(kv, cond_rope)
    o1 = F.scaled_dot_product_attention(q, k, v)
    o2 = F.scaled_dot_product_attention(qx, kv, vv)
    o3 = F.scaled_dot_product_attention(qv, torch.cat([kx, kv], dim=2), torch.cat([vx, vv], dim=2))
    s = torch.tensor(float(scale[0] if isinstance(scale, (list, tuple)) else scale), dtype=o2.dtype)
    if taps is not None:
        taps.update(q=q, k=k, v=v, qx=qx, kx=kx, vx=vx, qv=qv, kv=kv, vv=vv, o1=o1, o2=o2, o3=o3)
    o = torch.cat([o1 + s * o2, o3], dim=2)
    o = o.transpose(1, 2).reshape(x.shape[0], -1, x.shape[-1])
    o = _lin(sd, pre + ".to_out.0", o)
    N1 = x.shape[1]
    return o[:, Nt:N1], torch.cat([o[:, :Nt], o[:, N1:]], dim=1)


def feed_forward(sd, pre, x):
    """diffusers FeedForward(activation_fn="gelu-approximate") — call site cogvideox_transformer_3d.py:136-143."""
    return _lin(sd, pre + ".net.2", F.gelu(_lin(sd, pre + ".net.0.proj", x), approximate="tanh"))


def block_forward(sd, pre, hidden, enc, temb, heads, n_vip, scale, rope, vip_rope, cond_rope, eps=1e-5):
    """cogvideox_transformer_3d.py:221-332 (use_vip, func_type "1"; n_vip=0 -> plain block)."""
    use_vip = n_vip > 0
    if use_vip:
        text, vip = enc[:, :-n_vip], enc[:, -n_vip:]
    else:
        text = enc
    Nt = text.shape[1]
    nh, nt, g, eg = layer_norm_zero(sd, pre + ".norm1", hidden, text, temb, eps)
    if use_vip:
        nv, vg = vip_layer_norm_zero(sd, pre + ".vip_norm1", vip, temb, eps)
        a_h, a_e = vip_attention(sd, pre + ".attn1", nh, torch.cat([nt, nv], dim=1), heads, n_vip, scale,
                                 rope, vip_rope, cond_rope)
        a_t, a_v = a_e[:, :Nt], a_e[:, Nt:]
    else:
        a_h, a_t = plain_attention(sd, pre + ".attn1", nh, nt, heads, rope)
    hidden = hidden + g * a_h
    text = text + eg * a_t
    if use_vip:
        vip = vip + vg * a_v
    nh, nt, g, eg = layer_norm_zero(sd, pre + ".norm2", hidden, text, temb, eps)
    if use_vip:
        nv, vg = vip_layer_norm_zero(sd, pre + ".vip_norm2", vip, temb, eps)
    y = feed_forward(sd, pre + ".ff", torch.cat([nt, nh], dim=1))
    hidden = hidden + g * y[:, Nt:]
    text = text + eg * y[:, :Nt]
    if use_vip:
        vip = vip + vg * feed_forward(sd, pre + ".ff", nv)
        return hidden, torch.cat([text, vip], dim=1)
    return hidden, text


def final_layer(sd, hidden, enc, temb, n_frames, hw_shape, patch, eps=1e-5):
    """cogvideox_transformer_3d.py:736-759 + normalization.py:70-92 (AdaLayerNorm chunk_dim=1: shift, scale)."""
    B = hidden.shape[0]
    x = _ln(sd, "norm_final", torch.cat([enc, hidden], dim=1), eps)[:, enc.shape[1]:]
    Fr = temb.shape[1]
    mod = _lin(sd, "norm_out.linear", F.silu(temb.reshape(B * Fr, -1)))
    shift, scale = (c.reshape(B, Fr, -1).repeat_interleave(x.shape[1] // Fr, dim=1) for c in mod.chunk(2, dim=1))
    x = _ln(sd, "norm_out.norm", x, eps) * (1 + scale) + shift
    x = _lin(sd, "proj_out", x)
    h, w = hw_shape
    out = x.reshape(B, n_frames, h // patch, w // patch, -1, patch, patch)
    return out.permute(0, 1, 4, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)


def dit_forward(sd, cfg, hidden_states, encoder_hidden_states, timestep, vip_encoder_hidden_states=None,
                image_rotary_emb=None, vip_image_rotary_emb=None, vip_condition_rotary_emb=None,
                vip_scale=(1.0,), taps=None):
    """cogvideox_transformer_3d.py:636-770. cfg: dict(num_attention_heads, attention_head_dim, num_layers, patch_size).

    hidden_states [B,F,C,H,W]; timestep [B] or [B,F]; returns [B,F,C_out,H,W]."""
    heads = cfg["num_attention_heads"]
    inner = heads * cfg["attention_head_dim"]
    patch = cfg.get("patch_size", 2)
    B, Fr, C, H, W = hidden_states.shape
    temb = time_embedding(sd, timestep, B, inner, hidden_states.dtype)
    x = patch_embed(sd, encoder_hidden_states, hidden_states, vip_encoder_hidden_states, patch)
    Nt = encoder_hidden_states.shape[1]
    n_vip = 0
    if vip_encoder_hidden_states is not None:
        n_vip = vip_encoder_hidden_states.shape[1] * vip_encoder_hidden_states.shape[3] * vip_encoder_hidden_states.shape[4]
        enc = torch.cat([x[:, :Nt], x[:, -n_vip:]], dim=1)
        hid = x[:, Nt:-n_vip]
    else:
        enc, hid = x[:, :Nt], x[:, Nt:]
    for i in range(cfg["num_layers"]):
        hid, enc = block_forward(sd, f"transformer_blocks.{i}", hid, enc, temb, heads, n_vip, vip_scale,
                                 image_rotary_emb, vip_image_rotary_emb, vip_condition_rotary_emb)
        if taps is not None:
            taps[f"block{i}.hidden"] = hid
            taps[f"block{i}.enc"] = enc
    return final_layer(sd, hid, enc, temb, Fr, (H, W), patch)


# ----------------------------------------------------------------------------------------------
# synthetic weights with the reference's state-dict names (real checkpoints are not available offline)
# ----------------------------------------------------------------------------------------------
def make_state_dict(cfg, n_vip_dim=None, seed=0, dtype=torch.float32, std=0.02):
    """Random weights keyed like the reference model + set_vip_layers (SURVEY §8b loading contract)."""
    g = torch.Generator().manual_seed(seed)
    heads, hd, L = cfg["num_attention_heads"], cfg["attention_head_dim"], cfg["num_layers"]
    D = heads * hd
    te = cfg.get("time_embed_dim", 512)
    txt = cfg.get("text_embed_dim", 4096)
    cin = cfg.get("in_channels", 16)
    cout = cfg.get("out_channels", 16)
    p = cfg.get("patch_size", 2)
    sd = {}

    def lin(name, o, i, s=std, bias=True):
        sd[name + ".weight"] = (torch.randn(o, i, generator=g) * s).to(dtype)
        if bias:
            sd[name + ".bias"] = (torch.randn(o, generator=g) * s).to(dtype)

    def ln(name, n):
        sd[name + ".weight"] = (1 + 0.1 * torch.randn(n, generator=g)).to(dtype)
        sd[name + ".bias"] = (0.1 * torch.randn(n, generator=g)).to(dtype)

    sd["patch_embed.proj.weight"] = (torch.randn(D, cin, p, p, generator=g) * 0.1).to(dtype)
    sd["patch_embed.proj.bias"] = (torch.randn(D, generator=g) * std).to(dtype)
    lin("patch_embed.text_proj", D, txt)
    if n_vip_dim:
        lin("patch_embed.vip_proj", D, n_vip_dim)
    lin("time_embedding.linear_1", te, D)
    lin("time_embedding.linear_2", te, te, s=0.05)
    for i in range(L):
        b = f"transformer_blocks.{i}"
        for n in ("norm1", "norm2"):
            lin(f"{b}.{n}.linear", 6 * D, te, s=0.05)
            ln(f"{b}.{n}.norm", D)
        for n in "qkv":
            lin(f"{b}.attn1.to_{n}", D, D)
        ln(f"{b}.attn1.norm_q", hd)
        ln(f"{b}.attn1.norm_k", hd)
        lin(f"{b}.attn1.to_out.0", D, D)
        lin(f"{b}.ff.net.0.proj", 4 * D, D)
        lin(f"{b}.ff.net.2", D, 4 * D)
        if n_vip_dim:
            for n in ("vip_norm1", "vip_norm2"):
                lin(f"{b}.{n}.linear", 3 * D, te, s=0.05)
                ln(f"{b}.{n}.norm", D)
            for n in "qkv":
                lin(f"{b}.attn1.processor.vip_to_{n}", D, D)
            ln(f"{b}.attn1.processor.vip_norm_q", hd)
            ln(f"{b}.attn1.processor.vip_norm_k", hd)
    ln("norm_final", D)
    lin("norm_out.linear", 2 * D, te, s=0.05)
    ln("norm_out.norm", D)
    lin("proj_out", p * p * cout, D)
    return sd
