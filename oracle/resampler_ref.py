"""Oracle: Resampler (condensed-token encoder, Perceiver style) — TEST INFRASTRUCTURE ONLY.

Restates longvgen/video_ipadapter/resampler.py:66-244 functionally over a state dict with the reference's key names
(`latents`, `proj_in`, `layers.{i}.0.{norm1,norm2,to_q,to_kv,to_out,norm_q,norm_k}`, `layers.{i}.1.net.{0.proj,2}`,
`proj_out`, `norm_out`), including the optional PCA low-rank filter (:230-237 with pca.py:56-66; gen.yaml passes a real `longvgen_pca`
path to `set_pca`, infer_cogvideo_mp_fifo.py:118).  Pinned bit-exact against the reference class (tests/golden/resampler_tiny.pt: fp32 and
bf16, batch 1 and 2, with and without the filter).  diffusers' FeedForward is restated (unpinned, see dit_ref.py)."""
import math

import torch
import torch.nn.functional as F

from .dit_ref import apply_rope


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _ln(sd, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def _heads(x, heads):
    b, n, w = x.shape
    return x.view(b, n, heads, -1).transpose(1, 2)


def perceiver_attention(sd, pre, x, latents, heads, image_rope, sampling_rope):
    """PerceiverAttention.forward, :86-129: queries = latents, keys/values = cat(x, latents)."""
    x = _ln(sd, pre + ".norm1", x)
    latents = _ln(sd, pre + ".norm2", latents)
    b, l, _ = latents.shape
    q = _heads(_lin(sd, pre + ".to_q", latents), heads)
    k, v = _lin(sd, pre + ".to_kv", torch.cat((x, latents), dim=-2)).chunk(2, dim=-1)
    k, v = _heads(k, heads), _heads(v, heads)
    q = _ln(sd, pre + ".norm_q", q, 1e-6)
    k = _ln(sd, pre + ".norm_k", k, 1e-6)
    if image_rope is not None:
        k[:, :, :-l] = apply_rope(k[:, :, :-l], image_rope)
    if sampling_rope is not None:
        q = apply_rope(q, sampling_rope)
        k[:, :, -l:] = apply_rope(k[:, :, -l:], sampling_rope)
    out = F.scaled_dot_product_attention(q, k, v, scale=1 / math.sqrt(q.shape[-1]))
    out = out.permute(0, 2, 1, 3).reshape(b, l, -1)
    return _lin(sd, pre + ".to_out", out)


def resampler_forward(sd, cfg, x, image_rope=None, sampling_rope=None, pca=None):
    """Resampler.forward, :209-244.  x [b, f, n, embedding_dim] -> [b, Tq, output_dim, Hq, Wq].
    pca: optional (components_ [d, D], mean_ [1, D]) of a fitted pca.PCA -> the low-rank filter of :230-237."""
    b = x.shape[0]
    x = _lin(sd, "proj_in", x.flatten(0, 1)).reshape(b, -1, sd["proj_in.weight"].shape[0])
    lat = sd["latents"].expand(b, -1, -1)
    for i in range(cfg["depth"]):
        lat = perceiver_attention(sd, f"layers.{i}.0", x, lat, cfg["heads"], image_rope, sampling_rope) + lat
        h = F.gelu(_lin(sd, f"layers.{i}.1.net.0.proj", lat), approximate="tanh")
        lat = _lin(sd, f"layers.{i}.1.net.2", h) + lat
    lat = _ln(sd, "norm_out", _lin(sd, "proj_out", lat))
    if pca is not None:
        comp, mean = pca
        dt = lat.dtype
        y = torch.matmul(lat.flatten(0, 1).to(comp.dtype) - mean, comp.t())          # pca.transform, pca.py:56-58
        y[:, 16:] = 0.0
        lat = (torch.matmul(y, comp) + mean).reshape(lat.shape).to(dt)              # pca.inverse_transform, :64-66
    tq, hq, wq = cfg["num_temporal_queries"], cfg["num_height_queries"], cfg["num_width_queries"]
    return lat.reshape(b, tq, hq, wq, -1).permute(0, 1, 4, 2, 3)


def make_state_dict(cfg, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    dim, inner = cfg["dim"], cfg["dim_head"] * cfg["heads"]
    nq = cfg["num_temporal_queries"] * cfg["num_height_queries"] * cfg["num_width_queries"]
    sd = {"latents": (torch.randn(1, nq, dim, generator=g) / dim ** 0.5).to(dtype)}

    def lin(name, o, i, bias=True, s=0.03):
        sd[name + ".weight"] = (torch.randn(o, i, generator=g) * s).to(dtype)
        if bias:
            sd[name + ".bias"] = (torch.randn(o, generator=g) * s).to(dtype)

    def ln(name, n):
        sd[name + ".weight"] = (1 + 0.1 * torch.randn(n, generator=g)).to(dtype)
        sd[name + ".bias"] = (0.1 * torch.randn(n, generator=g)).to(dtype)

    lin("proj_in", dim, cfg["embedding_dim"])
    lin("proj_out", cfg["output_dim"], dim)
    ln("norm_out", cfg["output_dim"])
    for i in range(cfg["depth"]):
        p = f"layers.{i}.0"
        ln(p + ".norm1", dim); ln(p + ".norm2", dim)
        lin(p + ".to_q", inner, dim, bias=False); lin(p + ".to_kv", 2 * inner, dim, bias=False); lin(p + ".to_out", dim, inner, bias=False)
        ln(p + ".norm_q", cfg["dim_head"]); ln(p + ".norm_k", cfg["dim_head"])
        lin(f"layers.{i}.1.net.0.proj", dim * cfg.get("ff_mult", 4), dim)
        lin(f"layers.{i}.1.net.2", dim, dim * cfg.get("ff_mult", 4))
    return sd
