"""Oracle: CogVideoX 3-D causal VAE (encode / decode, frame batching, 9-tile tiling + blend) — TEST INFRASTRUCTURE ONLY.

Functional restatement over a plain state dict with the diffusers key names.  Follows the vendored twin
longvgen/models/autoencoder_kl_cogvideox.py for everything that is in the reference tree (causal conv + cache
:67-145, SpatialNorm3D :148-188, ResnetBlock3D :191-309, Down/Mid/Up blocks :312-608, Encoder3D :611-742,
Decoder3D :745-883, frame batching :1085-1108,1138-1163, tiling/blend :1190-1359).  Three classes live only in
diffusers 0.31.0.dev0 (not in /root/reference): CogVideoXDownsample3D, CogVideoXUpsample3D,
DiagonalGaussianDistribution — restated here from the published upstream semantics: **parity unpinned** for
those three; everything else is pinned bit-exact against the vendored twin (tests/golden/vae_tiny.pt).
The runtime VAE is the upstream diffusers class; it decodes one 13-latent-frame chunk per call
(pipeline_cogvideox_mp_fifo.py:676-684), for which upstream's and the twin's tiled_decode loops coincide.
"""
import torch
import torch.nn.functional as F

GROUPS = 32


class ConvCache(dict):
    """conv name -> last (kt-1) input frames (the reference's per-module `conv_cache`)."""


def causal_conv3d(sd, name, x, cache):
    """CogVideoXCausalConv3d.forward, :120-145.  x [B,C,T,H,W]."""
    w, b = sd[name + ".conv.weight"], sd.get(name + ".conv.bias")
    kt, kh, kw = w.shape[2:]
    if kt > 1:
        prev = cache.get(name)
        ctx = [prev] if prev is not None else [x[:, :, :1]] * (kt - 1)
        x = torch.cat(ctx + [x], dim=2)
        cache[name] = x[:, :, -(kt - 1):].clone()
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2), mode="constant", value=0)
    return F.conv3d(x, w, b)


def group_norm(sd, name, x, eps=1e-6):
    return F.group_norm(x, GROUPS, sd[name + ".weight"], sd[name + ".bias"], eps)


def spatial_norm(sd, name, f, zq, cache):
    """CogVideoXSpatialNorm3D.forward, :171-188: zq nearest-resized to f (first frame separately when T odd > 1)."""
    if f.shape[2] > 1 and f.shape[2] % 2 == 1:
        z_first = F.interpolate(zq[:, :, :1], size=f[:, :, :1].shape[-3:])
        z_rest = F.interpolate(zq[:, :, 1:], size=f[:, :, 1:].shape[-3:])
        zq = torch.cat([z_first, z_rest], dim=2)
    else:
        zq = F.interpolate(zq, size=f.shape[-3:])
    nf = group_norm(sd, name + ".norm_layer", f)
    return nf * causal_conv3d(sd, name + ".conv_y", zq, cache) + causal_conv3d(sd, name + ".conv_b", zq, cache)


def resnet(sd, name, x, zq, cache):
    """CogVideoXResnetBlock3D.forward, :277-309 (temb is None on this path)."""
    h = spatial_norm(sd, name + ".norm1", x, zq, cache) if zq is not None else group_norm(sd, name + ".norm1", x)
    h = causal_conv3d(sd, name + ".conv1", F.silu(h), cache)
    h = spatial_norm(sd, name + ".norm2", h, zq, cache) if zq is not None else group_norm(sd, name + ".norm2", h)
    h = causal_conv3d(sd, name + ".conv2", F.silu(h), cache)
    if (name + ".conv_shortcut.weight") in sd:
        x = F.conv3d(x, sd[name + ".conv_shortcut.weight"], sd[name + ".conv_shortcut.bias"])
    return h + x


def downsample3d(sd, name, x, compress_time):
    """diffusers CogVideoXDownsample3D (restated, unpinned)."""
    if compress_time:
        b, c, f, h, w = x.shape
        x = x.permute(0, 3, 4, 1, 2).reshape(b * h * w, c, f)
        if f % 2 == 1:
            first, rest = x[..., 0], x[..., 1:]
            if rest.shape[-1] > 0:
                rest = F.avg_pool1d(rest, kernel_size=2, stride=2)
            x = torch.cat([first[..., None], rest], dim=-1)
        else:
            x = F.avg_pool1d(x, kernel_size=2, stride=2)
        x = x.reshape(b, h, w, c, x.shape[-1]).permute(0, 3, 4, 1, 2)
    x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
    b, c, f, h, w = x.shape
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w), sd[name + ".conv.weight"], sd[name + ".conv.bias"], stride=2)
    return y.reshape(b, f, *y.shape[1:]).permute(0, 2, 1, 3, 4)


def upsample3d(sd, name, x, compress_time):
    """diffusers CogVideoXUpsample3D (restated, unpinned)."""
    if compress_time:
        if x.shape[2] > 1 and x.shape[2] % 2 == 1:
            first = F.interpolate(x[:, :, 0], scale_factor=2.0)
            rest = F.interpolate(x[:, :, 1:], scale_factor=2.0)
            x = torch.cat([first[:, :, None], rest], dim=2)
        elif x.shape[2] > 1:
            x = F.interpolate(x, scale_factor=2.0)
        else:
            x = F.interpolate(x.squeeze(2), scale_factor=2.0)[:, :, None]
    else:
        b, c, t, h, w = x.shape
        x = F.interpolate(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w), scale_factor=2.0)
        x = x.reshape(b, t, c, *x.shape[2:]).permute(0, 2, 1, 3, 4)
    b, c, t, h, w = x.shape
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w), sd[name + ".conv.weight"], sd[name + ".conv.bias"], padding=1)
    return y.reshape(b, t, *y.shape[1:]).permute(0, 2, 1, 3, 4)


def encoder(sd, cfg, x, cache):
    """CogVideoXEncoder3D.forward, :708-742."""
    nb = len(cfg["block_out_channels"])
    tl = cfg.get("temporal_compress_level", 2)
    h = causal_conv3d(sd, "encoder.conv_in", x, cache)
    for i in range(nb):
        for j in range(cfg["layers_per_block"]):
            h = resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, None, cache)
        if i != nb - 1:
            h = downsample3d(sd, f"encoder.down_blocks.{i}.downsamplers.0", h, i < tl)
    for j in range(2):
        h = resnet(sd, f"encoder.mid_block.resnets.{j}", h, None, cache)
    h = F.silu(group_norm(sd, "encoder.norm_out", h))
    return causal_conv3d(sd, "encoder.conv_out", h, cache)


def decoder(sd, cfg, z, cache):
    """CogVideoXDecoder3D.forward, :849-883 (zq = the latent tile itself)."""
    nb = len(cfg["block_out_channels"])
    tl = cfg.get("temporal_compress_level", 2)
    h = causal_conv3d(sd, "decoder.conv_in", z, cache)
    for j in range(2):
        h = resnet(sd, f"decoder.mid_block.resnets.{j}", h, z, cache)
    for i in range(nb):
        for j in range(cfg["layers_per_block"] + 1):
            h = resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, z, cache)
        if i != nb - 1:
            h = upsample3d(sd, f"decoder.up_blocks.{i}.upsamplers.0", h, i < tl)
    h = F.silu(spatial_norm(sd, "decoder.norm_out", h, z, cache))
    return causal_conv3d(sd, "decoder.conv_out", h, cache)


def frame_batches(num_frames, batch, chunk=None):
    """:1092-1097 / :1146-1151: the remainder is folded into the first batch.  chunk = 13 (tiled_decode, :1313-1325): the rule restarts every 13
    latent frames (cache carried across the chunks of a tile); used for whole multiples of the chunk beyond one (the reference drops frames past the last
    whole chunk and fails below one)."""
    if chunk and num_frames > chunk and num_frames % chunk == 0:
        rem = chunk % batch
        return [(c0 + batch * k + (0 if k == 0 else rem), c0 + batch * (k + 1) + rem) for c0 in range(0, num_frames, chunk) for k in range(chunk // batch)]
    n = max(num_frames // batch, 1) if num_frames > 1 else 1
    rem = num_frames % batch
    return [(batch * k + (0 if k == 0 else rem), batch * (k + 1) + rem) for k in range(n)]


def _encode_plain(sd, cfg, x):
    cache = ConvCache()
    out = [encoder(sd, cfg, x[:, :, a:b], cache) for a, b in frame_batches(x.shape[2], 8)]
    return torch.cat(out, dim=2)


def _decode_plain(sd, cfg, z):
    cache = ConvCache()
    out = [decoder(sd, cfg, z[:, :, a:b], cache) for a, b in frame_batches(z.shape[2], 2)]
    return torch.cat(out, dim=2)


def blend_v(a, b, extent):
    extent = min(a.shape[3], b.shape[3], extent)
    for y in range(extent):
        b[:, :, :, y, :] = a[:, :, :, -extent + y, :] * (1 - y / extent) + b[:, :, :, y, :] * (y / extent)
    return b


def blend_h(a, b, extent):
    extent = min(a.shape[4], b.shape[4], extent)
    for x in range(extent):
        b[:, :, :, :, x] = a[:, :, :, :, -extent + x] * (1 - x / extent) + b[:, :, :, :, x] * (x / extent)
    return b


def tile_geometry(cfg, for_decode):
    """enable_tiling defaults, :1006-1025: sample tile = sample/2, latent tile = /8, overlap factors 1/6, 1/5."""
    sh, sw = cfg["sample_height"] // 2, cfg["sample_width"] // 2
    scale = 2 ** (len(cfg["block_out_channels"]) - 1)
    lh, lw = int(sh / scale), int(sw / scale)
    oh, ow = 1 / 6, 1 / 5
    if for_decode:
        return dict(tile=(lh, lw), stride=(int(lh * (1 - oh)), int(lw * (1 - ow))), blend=(int(sh * oh), int(sw * ow)),
                    limit=(sh - int(sh * oh), sw - int(sw * ow)))
    return dict(tile=(sh, sw), stride=(int(sh * (1 - oh)), int(sw * (1 - ow))), blend=(int(lh * oh), int(lw * ow)),
                limit=(lh - int(lh * oh), lw - int(lw * ow)))


def _tiled(sd, cfg, x, for_decode):
    """tiled_encode / tiled_decode, :1206-1359: per tile, frame batches with a carried conv cache; blend; crop; concat."""
    g = tile_geometry(cfg, for_decode)
    (th, tw), (st_h, st_w), (bh, bw), (lim_h, lim_w) = g["tile"], g["stride"], g["blend"], g["limit"]
    fn, fb = (decoder, 2) if for_decode else (encoder, 8)
    rows = []
    for i in range(0, x.shape[3], st_h):
        row = []
        for j in range(0, x.shape[4], st_w):
            cache = ConvCache()
            parts = [fn(sd, cfg, x[:, :, a:b, i:i + th, j:j + tw], cache) for a, b in frame_batches(x.shape[2], fb, 13 if for_decode else None)]
            row.append(torch.cat(parts, dim=2))
        rows.append(row)
    out_rows = []
    for i, row in enumerate(rows):
        out = []
        for j, tile in enumerate(row):
            if i > 0:
                tile = blend_v(rows[i - 1][j], tile, bh)
            if j > 0:
                tile = blend_h(row[j - 1], tile, bw)
            out.append(tile[:, :, :, :lim_h, :lim_w])
        out_rows.append(torch.cat(out, dim=4))
    return torch.cat(out_rows, dim=3)


def encode(sd, cfg, x, tiling=True):
    """AutoencoderKLCogVideoX._encode (:1085-1108): returns the moments tensor h [B, 2*latent, T', H/8, W/8]."""
    g = tile_geometry(cfg, False)
    if tiling and (x.shape[4] > g["tile"][1] or x.shape[3] > g["tile"][0]):
        return _tiled(sd, cfg, x, False)
    return _encode_plain(sd, cfg, x)


def decode(sd, cfg, z, tiling=True):
    """AutoencoderKLCogVideoX._decode (:1138-1163)."""
    g = tile_geometry(cfg, True)
    if tiling and (z.shape[4] > g["tile"][1] or z.shape[3] > g["tile"][0]):
        return _tiled(sd, cfg, z, True)
    return _decode_plain(sd, cfg, z)


def gaussian_sample(h, noise=None):
    """DiagonalGaussianDistribution (restated, unpinned): mean, logvar = chunk(2, 1); logvar clamp(-30, 20)."""
    mean, logvar = torch.chunk(h, 2, dim=1)
    if noise is None:
        return mean
    return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise


def make_state_dict(cfg, seed=0, dtype=torch.float32):
    """Seeded random VAE weights with the diffusers key names (real checkpoints are not available offline)."""
    g = torch.Generator().manual_seed(seed)
    boc, lpb, lat = cfg["block_out_channels"], cfg["layers_per_block"], cfg["latent_channels"]
    sd = {}

    def conv(name, co, ci, k, causal=True, dims=3):
        shape = (co, ci) + (k,) * dims
        fan = ci * k ** dims
        key = name + (".conv" if causal else "")
        sd[key + ".weight"] = (torch.randn(shape, generator=g) / fan ** 0.5).to(dtype)
        sd[key + ".bias"] = (torch.randn(co, generator=g) * 0.05).to(dtype)

    def norm(name, c):
        sd[name + ".weight"] = (1 + 0.1 * torch.randn(c, generator=g)).to(dtype)
        sd[name + ".bias"] = (0.1 * torch.randn(c, generator=g)).to(dtype)

    def res(name, ci, co, zq):
        for n, c in (("norm1", ci), ("norm2", co)):
            if zq:
                norm(f"{name}.{n}.norm_layer", c)
                conv(f"{name}.{n}.conv_y", c, zq, 1)
                conv(f"{name}.{n}.conv_b", c, zq, 1)
                sd[f"{name}.{n}.conv_y.conv.bias"] += 1.0
            else:
                norm(f"{name}.{n}", c)
        conv(name + ".conv1", co, ci, 3)
        conv(name + ".conv2", co, co, 3)
        if ci != co:
            conv(name + ".conv_shortcut", co, ci, 1, causal=False)

    conv("encoder.conv_in", boc[0], cfg.get("in_channels", 3), 3)
    c = boc[0]
    for i, co in enumerate(boc):
        for j in range(lpb):
            res(f"encoder.down_blocks.{i}.resnets.{j}", c if j == 0 else co, co, 0)
        c = co
        if i != len(boc) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co, 3, causal=False, dims=2)
    for j in range(2):
        res(f"encoder.mid_block.resnets.{j}", c, c, 0)
    norm("encoder.norm_out", c)
    conv("encoder.conv_out", 2 * lat, c, 3)
    rb = list(reversed(boc))
    conv("decoder.conv_in", rb[0], lat, 3)
    for j in range(2):
        res(f"decoder.mid_block.resnets.{j}", rb[0], rb[0], lat)
    c = rb[0]
    for i, co in enumerate(rb):
        for j in range(lpb + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}", c if j == 0 else co, co, lat)
        c = co
        if i != len(rb) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3, causal=False, dims=2)
    norm("decoder.norm_out.norm_layer", c)
    conv("decoder.norm_out.conv_y", c, lat, 1)
    conv("decoder.norm_out.conv_b", c, lat, 1)
    sd["decoder.norm_out.conv_y.conv.bias"] += 1.0
    conv("decoder.conv_out", cfg.get("out_channels", 3), c, 3)
    return sd
