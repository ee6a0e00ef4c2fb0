"""AutoencoderKLCogVideoX — host mirror of the 3-D causal VAE the pipeline calls (`vae.encode(x).latent_dist.sample()`,
`vae.decode(z).sample`, `enable_tiling/slicing`, `config.scaling_factor`; pipeline_cogvideox_mp_fifo.py:585,682,
infer_cogvideo_mp_fifo.py:131-132).  Reference twin: longvgen/models/autoencoder_kl_cogvideox.py.

Same state-dict keys as diffusers (`encoder.conv_in.conv.weight`, `decoder.up_blocks.0.resnets.0.norm1.conv_y.conv.weight`,
`...upsamplers.0.conv.weight`, ...).  The arithmetic is HIP only (vae.hip): channels-last bf16 activations, implicit-GEMM
MFMA convolutions with the causal cache / zero padding / nearest upsampling folded into the loader, fused
GroupNorm|SpatialNorm+SiLU.  The reference's result-defining structure is kept exactly: 9 spatial tiles with linear
seam blending, temporal batches of 2 latent (8 sample) frames with the remainder folded into the first batch, GroupNorm
statistics per (tile, temporal batch), causal cache carried across the batches of a tile and cleared per tile.
"""
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import kernels as K
from . import lib as L

BF16 = torch.bfloat16


def _pad_to(n, m):
    return (n + m - 1) // m * m


class DiagonalGaussianDistribution:
    """diffusers.models.autoencoders.vae.DiagonalGaussianDistribution semantics (mean | logvar, clamp(-30, 20))."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=torch.float32).to(self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


def pack_up2_phases(w):
    """Conv2d 3x3 weight [co, ci, 3, 3] behind a nearest x2 upsampling -> the four 2x2 phase kernels [4 = py*2+px, co, 4 = a*2+b, ci] (bf16) of tg_conv3d_up2_subpixel:
    of the three upsampled rows a tap row reads, two are the same low-resolution row — their weights are added (fp32) and rounded to bf16 ONCE.
    py = 0: a = 0 <- dy 0, a = 1 <- dy 1 + dy 2;  py = 1: a = 0 <- dy 0 + dy 1, a = 1 <- dy 2; columns likewise."""
    sets = (((0,), (1, 2)), ((0, 1), (2,)))
    wf = w.float()
    co, ci = w.shape[:2]
    out = torch.zeros(4, co, 4, ci, dtype=torch.float32, device=w.device)
    for py in range(2):
        for px in range(2):
            for a in range(2):
                for b in range(2):
                    for dy in sets[py][a]:
                        for dx in sets[px][b]:
                            out[py * 2 + px, :, a * 2 + b] += wf[:, :, dy, dx]
    return out.to(BF16).contiguous()


class AutoencoderKLCogVideoX:
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 256, 512), latent_channels=16, layers_per_block=3,
                 act_fn="silu", norm_eps=1e-6, norm_num_groups=32, temporal_compression_ratio=4, sample_height=480, sample_width=720,
                 scaling_factor=1.15258426, use_quant_conv=False, use_post_quant_conv=False, device="cuda", **unused):
        if act_fn != "silu" or norm_num_groups != 32 or use_quant_conv or use_post_quant_conv or latent_channels != 16:
            raise NotImplementedError("configuration differs from CogVideoX-5b's VAE in a way the fused kernels do not cover")
        if any(c % 64 for c in block_out_channels):
            raise NotImplementedError("block_out_channels must be multiples of 64")
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, block_out_channels=tuple(block_out_channels),
                                      latent_channels=latent_channels, layers_per_block=layers_per_block, norm_eps=norm_eps,
                                      temporal_compression_ratio=temporal_compression_ratio, sample_height=sample_height,
                                      sample_width=sample_width, scaling_factor=scaling_factor)
        self.device = torch.device(device)
        self.dtype = BF16
        self.use_slicing = self.use_tiling = False
        self.num_latent_frames_batch_size, self.num_sample_frames_batch_size = 2, 8
        self.tile_sample_min_height, self.tile_sample_min_width = sample_height // 2, sample_width // 2
        self.tile_overlap_factor_height, self.tile_overlap_factor_width = 1 / 6, 1 / 5
        self._retile()
        self._sd, self._packed = {}, {}
        self.tile_streams = int(os.environ.get("TG_VAE_STREAMS", "3"))     # concurrent spatial tiles (see _run_tiles)
        self._streams = []
        self._tmaps = {}
        # HIP graphs: a tile program (all temporal batches of one spatial tile: ~950 launches) is captured the second time its shape is seen
        # and replayed afterwards — the host then issues 9 graph launches per clip instead of ~8 500 kernel launches, which is what lets the
        # tile streams actually overlap (the decode was bound by Python's launch rate, not by the GPU, once the tiles ran concurrently)
        self.use_graphs = os.environ.get("TG_VAE_GRAPHS", "1") != "0"
        self.subpixel_upsample = os.environ.get("TG_VAE_SUBPIXEL", "1") != "0"    # spatial-only upsamplers as four 2x2 phase convolutions (_upsample); flip BEFORE the first decode (graphs)
        self._graphs, self._seen = {}, {}
        self._tl = int(np.log2(temporal_compression_ratio))

    # ---- reference API -----------------------------------------------------------------------------------
    def _retile(self):
        s = 2 ** (len(self.config.block_out_channels) - 1)
        self.tile_latent_min_height = int(self.tile_sample_min_height / s)
        self.tile_latent_min_width = int(self.tile_sample_min_width / s)

    def enable_tiling(self, tile_sample_min_height=None, tile_sample_min_width=None, tile_overlap_factor_height=None,
                      tile_overlap_factor_width=None):
        self.use_tiling = True
        self.tile_sample_min_height = tile_sample_min_height or self.tile_sample_min_height
        self.tile_sample_min_width = tile_sample_min_width or self.tile_sample_min_width
        self.tile_overlap_factor_height = tile_overlap_factor_height or self.tile_overlap_factor_height
        self.tile_overlap_factor_width = tile_overlap_factor_width or self.tile_overlap_factor_width
        self._retile()

    def disable_tiling(self):
        self.use_tiling = False

    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    def state_dict(self):
        return dict(self._sd)

    def load_state_dict(self, sd, strict=True):
        """Keeps the bf16 originals under the diffusers names and builds the packed GEMM operands."""
        want = self.param_shapes()
        missing = [k for k in want if k not in sd]
        unexpected = [k for k in sd if k not in want]
        bad = [k for k in want if k in sd and tuple(sd[k].shape) != tuple(want[k])]
        if bad or (strict and (missing or unexpected)):
            raise RuntimeError(f"AutoencoderKLCogVideoX.load_state_dict: missing {missing[:4]}, unexpected {unexpected[:4]}, shape mismatch {bad[:4]}")
        self._sd = {k: v.detach().to(self.device, BF16).contiguous() for k, v in sd.items()}
        self._packed = {}
        # captured tile programs hold raw pointers into the old _sd / _packed tensors: a reload must drop them
        self._graphs, self._seen = {}, {}
        K.zero_page(self.device)          # created on the caller's stream before any tile stream can race for it
        for k, v in self._sd.items():
            if k.endswith(".weight") and v.dim() >= 4:
                co, ci = v.shape[:2]
                taps = v.shape[2:]
                if all(t == 1 for t in taps) and ci == 16 and (k.endswith("conv_y.conv.weight") or k.endswith("conv_b.conv.weight")):
                    continue                                                               # SpatialNorm convs: fused below
                if k == "encoder.conv_in.conv.weight" and ci == 3 and co == 128 and tuple(taps) == (3, 3, 3):
                    # the encoder's input convolution: reduction index k = tap * 3 + channel (81 -> 96), input tiles carry 8 channels (csrc/vae.hip conv3d_in_kernel)
                    w = torch.zeros(co, 96, dtype=BF16, device=self.device)
                    w[:, :81] = v.reshape(co, ci, 27).permute(0, 2, 1).reshape(co, 81)
                    self._packed[k + ".k96"] = w.contiguous()          # beside the general packing: tiles too small for the patch kernel take that
                if ".upsamplers." in k and tuple(taps) == (3, 3) and co % 256 == 0 and ci % 64 == 0:
                    self._packed[k + ".up2"] = pack_up2_phases(v)       # beside the general packing (time-upsampling layers and small tiles keep that)
                cin_p = _pad_to(ci, 64)
                cout_p = _pad_to(co, 16) if co <= 32 else _pad_to(co, 128)     # conv_out (3 / 32 channels): the narrow 128 x 16 tile
                w = torch.zeros(cout_p, int(np.prod(taps)), cin_p, dtype=BF16, device=self.device)
                w[:co, :, :ci] = v.reshape(co, ci, -1).permute(0, 2, 1)
                self._packed[k] = w.contiguous()
        # SpatialNorm: conv_y | conv_b of one norm layer -> ONE [2*Cpad, 64] GEMM weight over the (zero-padded) latent channels
        for k in list(self._sd):
            if k.endswith(".conv_y.conv.weight"):
                base = k[: -len(".conv_y.conv.weight")]
                wy, wb = self._sd[k], self._sd[base + ".conv_b.conv.weight"]
                C = wy.shape[0]
                Cp = _pad_to(C, 128)
                w = torch.zeros(2 * Cp, 64, dtype=BF16, device=self.device)
                w[:C, :16], w[Cp:Cp + C, :16] = wy.reshape(C, 16), wb.reshape(C, 16)
                b = torch.zeros(2 * Cp, dtype=BF16, device=self.device)
                b[:C], b[Cp:Cp + C] = self._sd[base + ".conv_y.conv.bias"], self._sd[base + ".conv_b.conv.bias"]
                self._packed[base + ".yb.weight"], self._packed[base + ".yb.bias"] = w, b
        # ... and ALL SpatialNorm layers of the decoder -> one [sum 2*Cpad, 64] weight: every one of them reads the same latent tile, so one
        # GEMM per (tile, temporal batch) replaces 37 small ones (43 us each on the tile's critical path, 1 400 launches per clip)
        names = [k[: -len(".yb.weight")] for k in self._packed if k.endswith(".yb.weight")]
        self._yb_off, off = {}, 0
        for nme in names:
            self._yb_off[nme] = off
            off += self._packed[nme + ".yb.weight"].shape[0]
        if names:
            self._packed["yb_all.weight"] = torch.cat([self._packed[nme + ".yb.weight"] for nme in names]).contiguous()
            self._packed["yb_all.bias"] = torch.cat([self._packed[nme + ".yb.bias"] for nme in names]).contiguous()
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)


    def _after_weight_update(self):
        """runtime.broadcast_weights wrote into the `_sd` tensors: rebuild the packed GEMM operands (and drop captured graphs) from them."""
        self.load_state_dict(self._sd)

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=BF16, device="cuda", broadcast=False, **unused):
        """diffusers' ModelMixin.from_pretrained for `<CogVideoX-5b>/vae`: config.json + diffusion_pytorch_model.safetensors.
        broadcast=True: rank 0 reads the file, the others receive the weights over RCCL (runtime.broadcast_weights)."""
        import json
        from safetensors.torch import load_file
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, "config.json")) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        if "down_block_types" in cfg:          # diffusers lists block types; CogVideoX has exactly one kind of each
            cfg.pop("down_block_types"), cfg.pop("up_block_types", None)
        vae = cls(**cfg, device=device)
        rank = 0
        if broadcast:
            import torch.distributed as dist
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        if rank == 0:
            vae.load_state_dict(load_file(os.path.join(d, "diffusion_pytorch_model.safetensors")))
        else:
            vae.load_state_dict({k: torch.zeros(sh, dtype=BF16) for k, sh in vae.param_shapes().items()})      # storages to receive into
        if broadcast:
            from .runtime import broadcast_weights
            broadcast_weights(vae, src=0)
        return vae

    def save_pretrained(self, path):
        import json
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(dict({k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(self.config).items()}, _class_name="AutoencoderKLCogVideoX"), f, indent=1)
        save_file({k: v.detach().cpu().contiguous() for k, v in self._sd.items()}, os.path.join(path, "diffusion_pytorch_model.safetensors"))

    def to(self, *args, **kwargs):
        """`.to(device)` of the reference call sites (`pipe.to(device)`); bf16 on the GPU whatever dtype is asked for."""
        dev = kwargs.get("device")
        for a in args:
            if isinstance(a, (str, torch.device)):
                dev = a
        if dev is not None and torch.device(dev) != self.device:
            self.device = torch.device(dev)
            if self._sd:
                self.load_state_dict(self._sd)
            self._streams, self._tmaps, self._graphs, self._seen = [], {}, {}, {}
        return self

    def param_shapes(self):
        """name -> shape of every parameter of this configuration, under the diffusers key names (what load_state_dict expects)."""
        c = self.config
        boc, lpb, lat = list(c.block_out_channels), c.layers_per_block, c.latent_channels
        out = {}

        def conv(name, co, ci, taps):
            out[name + ".weight"], out[name + ".bias"] = (co, ci) + taps, (co,)

        def res(name, ci, co, zq):
            for n, ch in (("norm1", ci), ("norm2", co)):
                if zq:
                    out[f"{name}.{n}.norm_layer.weight"] = out[f"{name}.{n}.norm_layer.bias"] = (ch,)
                    conv(f"{name}.{n}.conv_y.conv", ch, lat, (1, 1, 1))
                    conv(f"{name}.{n}.conv_b.conv", ch, lat, (1, 1, 1))
                else:
                    out[f"{name}.{n}.weight"] = out[f"{name}.{n}.bias"] = (ch,)
            conv(name + ".conv1.conv", co, ci, (3, 3, 3))
            conv(name + ".conv2.conv", co, co, (3, 3, 3))
            if ci != co:
                conv(name + ".conv_shortcut", co, ci, (1, 1, 1))

        conv("encoder.conv_in.conv", boc[0], c.in_channels, (3, 3, 3))
        ch = boc[0]
        for i, co in enumerate(boc):
            for j in range(lpb):
                res(f"encoder.down_blocks.{i}.resnets.{j}", ch if j == 0 else co, co, False)
            ch = co
            if i != len(boc) - 1:
                conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co, (3, 3))
        for j in range(2):
            res(f"encoder.mid_block.resnets.{j}", ch, ch, False)
        out["encoder.norm_out.weight"] = out["encoder.norm_out.bias"] = (ch,)
        conv("encoder.conv_out.conv", 2 * lat, ch, (3, 3, 3))
        rb = boc[::-1]
        conv("decoder.conv_in.conv", rb[0], lat, (3, 3, 3))
        for j in range(2):
            res(f"decoder.mid_block.resnets.{j}", rb[0], rb[0], True)
        ch = rb[0]
        for i, co in enumerate(rb):
            for j in range(lpb + 1):
                res(f"decoder.up_blocks.{i}.resnets.{j}", ch if j == 0 else co, co, True)
            ch = co
            if i != len(rb) - 1:
                conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, (3, 3))
        out["decoder.norm_out.norm_layer.weight"] = out["decoder.norm_out.norm_layer.bias"] = (ch,)
        conv("decoder.norm_out.conv_y.conv", ch, lat, (1, 1, 1))
        conv("decoder.norm_out.conv_b.conv", ch, lat, (1, 1, 1))
        conv("decoder.conv_out.conv", c.out_channels, ch, (3, 3, 3))
        return out

    def init_random(self, seed=0):
        """Seeded random weights at this configuration's shapes (benchmarks: real checkpoints are not available offline):
        fan-in scaled convolutions, norm scales near 1, SpatialNorm conv_y biased to 1."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        sd = {}
        for k, shp in self.param_shapes().items():
            r = torch.randn(shp, generator=g, device=self.device, dtype=torch.float32)
            if len(shp) > 1:
                r = r / float(np.prod(shp[1:])) ** 0.5
            elif k.endswith(".weight"):
                r = 1 + 0.1 * r
            else:
                r = 0.05 * r + (1.0 if k.endswith("conv_y.conv.bias") else 0.0)
            sd[k] = r.to(BF16)
        self.load_state_dict(sd)
        return self

    # ---- building blocks -----------------------------------------------------------------------------------
    def _conv(self, name, x, cache, residual=None, packed=None):
        """CogVideoXCausalConv3d (autoencoder_kl_cogvideox.py:120-145): the carried cache is read in place by the kernel."""
        w = self._sd[name + ".conv.weight"]
        co, kt = w.shape[0], w.shape[2]
        prev = cache.get(name) if kt > 1 else None
        y = K.conv3d_cl(x, self._packed[packed or name + ".conv.weight"], self._sd[name + ".conv.bias"], co, kt, w.shape[3], w.shape[4], cache=prev,
                        residual=residual, gn_stats_eps=self.config.norm_eps)    # the next norm's statistics come with y
        if kt > 1:
            need = kt - 1
            if x.shape[0] >= need:
                cache[name] = x[-need:]            # a view: x (a fresh norm output / staging tensor) is never written again — no copy launch per conv
            else:
                head = prev if prev is not None else x[:1].expand(need, -1, -1, -1)
                cache[name] = torch.cat([head, x], dim=0)[-need:].contiguous()
        return y

    def _norm_act(self, name, x, zq, silu=True):
        stats = getattr(x, "gn_sums", None)                  # left by the convolution that produced x (its epilogue's per-tile sums; the norm pass finalises them)
        if stats is None:
            stats = K.groupnorm_stats(x.view(-1, x.shape[-1]), self.config.norm_eps)
        if zq is None:
            return K.groupnorm_silu(x, stats, self._sd[name + ".weight"], self._sd[name + ".bias"], silu)
        z64, zdims, yb_all = zq                            # latent tile [Vz, 64] (channels-last, padded), its dims, conv_y|conv_b of every norm layer
        off, Cp = self._yb_off[name], self._packed[name + ".yb.weight"].shape[0] // 2
        return K.spatialnorm_silu(x, stats, self._sd[name + ".norm_layer.weight"], self._sd[name + ".norm_layer.bias"],
                                  yb_all[:, off:off + Cp], yb_all[:, off + Cp:off + 2 * Cp], zdims, silu)

    def _spatialnorm_tables(self, z64):
        """conv_y(z) | conv_b(z) per LATENT voxel for all SpatialNorm layers at once (a 1x1x1 conv commutes with the nearest resize of zq,
        autoencoder_kl_cogvideox.py:171-188): [Vz, sum 2*Cpad] bf16."""
        w, b = self._packed["yb_all.weight"], self._packed["yb_all.bias"]
        yb = torch.empty(z64.shape[0], w.shape[0], dtype=BF16, device=z64.device)
        K.gemm(z64, w, b, yb, L.EPI_BIAS)
        return yb

    def _resnet(self, name, x, zq, cache):
        """CogVideoXResnetBlock3D.forward (:277-309): the `+ inputs` is the second conv's epilogue."""
        h = self._conv(name + ".conv1", self._norm_act(name + ".norm1", x, zq), cache)
        h = self._norm_act(name + ".norm2", h, zq)
        skip = x
        if (name + ".conv_shortcut.weight") in self._sd:
            co = self._sd[name + ".conv_shortcut.weight"].shape[0]           # 1x1x1 CogVideoXSafeConv3d (:262-265)
            skip = K.conv3d_cl(x, self._packed[name + ".conv_shortcut.weight"], self._sd[name + ".conv_shortcut.bias"], co, 1, 1, 1, pad=0)
        return self._conv(name + ".conv2", h, cache, residual=skip)

    def _upsample(self, name, x, compress_time):
        """diffusers CogVideoXUpsample3D: nearest x2 (first frame 2-D only when T odd > 1) + Conv2d 3x3 — folded into one conv."""
        T, H, W, C = x.shape
        w = self._sd[name + ".conv.weight"]
        ph = self._packed.get(name + ".conv.weight.up2") if self.subpixel_upsample else None
        if ph is not None and K.conv3d_up2_subpixel_ok(T, H, W, C, w.shape[0]):
            # four 2x2 phase convolutions on the low-resolution input, 4 taps instead of 9 (pre-summed weights: the deviation stated in include/tokensgen_hip.h); the
            # time-doubling layers convolve every frame once and store it twice (exact).  The last upsampler (256 -> 256 at 8 x 240 x 360 per tile) was 5 % of the decode's kernel time
            return K.conv3d_up2_subpixel(x, ph, self._sd[name + ".conv.bias"], w.shape[0], gn_stats_eps=self.config.norm_eps, time_x2=bool(compress_time and T > 1))
        tmap = None
        To = T
        if compress_time and T > 1:
            idx = [0] + [t for t in range(1, T) for _ in (0, 1)] if T % 2 == 1 else [t for t in range(T) for _ in (0, 1)]
            To = len(idx)
            tmap = self._tmaps.get(T)
            if tmap is None:                      # cached: a host->device copy cannot sit inside a captured tile program
                tmap = self._tmaps[T] = torch.tensor(idx, dtype=torch.int32, device=x.device)
                # created on whichever tile stream needs it first; the other tile streams use it with no dependency on this stream, so the
                # upload must have LANDED before the host enqueues anything else (once per distinct T; never inside a capture: the eager
                # first pass of a shape has already created it)
                torch.cuda.current_stream(x.device).synchronize()
        return K.conv3d_cl(x, self._packed[name + ".conv.weight"], self._sd[name + ".conv.bias"], w.shape[0], 1, 3, 3, stride=1, pad=1, up=2,
                           t_map=tmap, out_dims=(To, 2 * H, 2 * W), gn_stats_eps=self.config.norm_eps)

    def _downsample(self, name, x, compress_time):
        """diffusers CogVideoXDownsample3D: temporal avg-pool (first frame kept when T odd), pad (0,1,0,1), Conv2d 3x3 stride 2."""
        if compress_time:
            x = K.avgpool_time(x)
        T, H, W, C = x.shape
        w = self._sd[name + ".conv.weight"]
        return K.conv3d_cl(x, self._packed[name + ".conv.weight"], self._sd[name + ".conv.bias"], w.shape[0], 1, 3, 3, stride=2, pad=0,
                           out_dims=(T, (H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1), gn_stats_eps=self.config.norm_eps)

    def _encoder(self, x_cl, cache):
        """CogVideoXEncoder3D.forward (:708-742) on a channels-last tile [T,H,W,8 or 64 (3 used)] -> [T',H/8,W/8,32]."""
        c = self.config
        nb = len(c.block_out_channels)
        h = self._conv("encoder.conv_in", x_cl, cache, packed="encoder.conv_in.conv.weight.k96" if x_cl.shape[-1] == 8 else None)
        for i in range(nb):
            for j in range(c.layers_per_block):
                h = self._resnet(f"encoder.down_blocks.{i}.resnets.{j}", h, None, cache)
            if i != nb - 1:
                h = self._downsample(f"encoder.down_blocks.{i}.downsamplers.0", h, i < self._tl)
        for j in range(2):
            h = self._resnet(f"encoder.mid_block.resnets.{j}", h, None, cache)
        h = self._norm_act("encoder.norm_out", h, None)
        return self._conv("encoder.conv_out", h, cache)

    def _decoder(self, z_cl64, zq, cache):
        """CogVideoXDecoder3D.forward (:849-883); zq = the latent tile (channels-last, 16 ch)."""
        c = self.config
        nb = len(c.block_out_channels)
        h = self._conv("decoder.conv_in", z_cl64, cache)
        for j in range(2):
            h = self._resnet(f"decoder.mid_block.resnets.{j}", h, zq, cache)
        for i in range(nb):
            for j in range(c.layers_per_block + 1):
                h = self._resnet(f"decoder.up_blocks.{i}.resnets.{j}", h, zq, cache)
            if i != nb - 1:
                h = self._upsample(f"decoder.up_blocks.{i}.upsamplers.0", h, i < self._tl)
        h = self._norm_act("decoder.norm_out", h, zq)
        return self._conv("decoder.conv_out", h, cache)

    @staticmethod
    def _frame_batches(n, batch, chunk=None):
        """:1092-1097, 1146-1151 — remainder folded into the first batch.  chunk (tiled_decode only, :1313-1325): the reference restarts that rule
        every 13 latent frames — (0,3),(3,5),..,(11,13),(13,16),(16,18),.. — with the conv cache carried across the chunks of a tile.  Applied when
        the frame count is a whole number of chunks > 1 (the reference drops frames past the last whole chunk and fails below one chunk; shorter
        inputs keep the plain rule, which is identical for exactly one chunk)."""
        if chunk and n > chunk and n % chunk == 0:
            rem = chunk % batch
            return [(c0 + batch * k + (0 if k == 0 else rem), c0 + batch * (k + 1) + rem) for c0 in range(0, n, chunk) for k in range(chunk // batch)]
        nb = max(n // batch, 1) if n > 1 else 1
        rem = n % batch
        return [(batch * k + (0 if k == 0 else rem), batch * (k + 1) + rem) for k in range(nb)]

    def _run_tile(self, src, i, j, th, tw, decode):
        """All temporal batches of one spatial tile (cache carried, cleared per tile) -> NCDHW bf16 tile output."""
        C, Tt, Ht, Wt = src.shape
        Hc, Wc = min(th, Ht - i), min(tw, Wt - j)
        cache = {}
        outs = []
        for a, b in self._frame_batches(Tt, self.num_latent_frames_batch_size if decode else self.num_sample_frames_batch_size,
                                        13 if (decode and getattr(self, "_tiled_pass", False)) else None):
            if decode:
                z64 = K.ncdhw_to_cl(src, a, b - a, i, Hc, j, Wc, 64)
                z2 = z64.view(-1, 64)
                y = self._decoder(z64, (z2, (b - a, Hc, Wc), self._spatialnorm_tables(z2)), cache)
            else:
                # 8-channel input + the k = tap * 3 + channel kernel when every 16 x 32 patch owns a row (128 voxels) of the GroupNorm partial buffer
                in8 = os.environ.get("TG_VAE_IN8", "1") != "0" and "encoder.conv_in.conv.weight.k96" in self._packed and -(-Hc // 16) * -(-Wc // 32) * 128 <= Hc * Wc     # (per frame: the same for every batch of the tile)
                y = self._encoder(K.ncdhw_to_cl(src, a, b - a, i, Hc, j, Wc, 8 if in8 else 64), cache)
            outs.append(y)
        To = sum(o.shape[0] for o in outs)
        Ho, Wo, Co = outs[0].shape[1:]
        tile = torch.empty(Co, To, Ho, Wo, dtype=BF16, device=self.device)
        t0 = 0
        for o in outs:
            K.cl_to_ncdhw(o, tile, t0, 0, 0)
            t0 += o.shape[0]
        return tile

    def _run_tile_maybe_graph(self, src, i, j, th, tw, decode, slot):
        """_run_tile, replayed from a captured HIP graph once the same (direction, dtype, tile shape, slot) has been seen before.
        `slot` separates tiles that may be in flight at the same time (a graph owns its buffers).  The graph reads a private staging copy
        of the tile's input window and returns its private output tile: the caller consumes it (blend, crop, cat) before the next replay
        of that slot, which the stream order of _run_tiles / _process guarantees."""
        if not self.use_graphs or K.PROFILE_ON[0]:
            return self._run_tile(src, i, j, th, tw, decode)
        C, Tt, Ht, Wt = src.shape
        Hc, Wc = min(th, Ht - i), min(tw, Wt - j)
        # everything a captured program bakes in besides the weights (load_state_dict drops the cache): shapes, the temporal batching, eps
        key = (decode, src.dtype, C, Tt, Hc, Wc, slot, bool(getattr(self, "_tiled_pass", False)), bool(self.subpixel_upsample), self.num_latent_frames_batch_size,
               self.num_sample_frames_batch_size, float(self.config.norm_eps))
        g = self._graphs.get(key)
        if g is None:
            if not self._seen.get(key):            # first sight: run eagerly (this is also the warm-up a capture needs)
                self._seen[key] = True
                return self._run_tile(src, i, j, th, tw, decode)
            cur = torch.cuda.current_stream(self.device)
            inp = torch.empty(C, Tt, Hc, Wc, dtype=src.dtype, device=self.device)
            graph = torch.cuda.CUDAGraph()
            cap = torch.cuda.Stream(device=self.device)
            cap.wait_stream(cur)
            with torch.cuda.graph(graph, stream=cap):
                out = self._run_tile(inp, 0, 0, Hc, Wc, decode)
            cur.wait_stream(cap)
            g = self._graphs[key] = (graph, inp, out)
        graph, inp, out = g
        inp.copy_(src[:, :, i:i + Hc, j:j + Wc])
        graph.replay()
        return out

    def _run_tiles(self, src, origins, th, tw, decode):
        """The spatial tiles are independent until the blend (cache cleared per tile, :1240/:1321): they are dealt round-robin to
        `tile_streams` HIP streams so that the small launches of one tile (the 512-channel layers have 88 workgroups on 256 CUs; the
        statistics finalise launches and cache copies are a few microseconds) run beside another tile's large convolutions instead of
        leaving most of the chip idle.  Each tile's arithmetic is unchanged, so the result does not depend on the stream count."""
        n = max(1, int(self.tile_streams))
        if n == 1:
            flat = [ij for row in origins for ij in row]            # slot = tile index: every tile keeps its own graph buffers until the blend
            outs = [self._run_tile_maybe_graph(src, i, j, th, tw, decode, k) for k, (i, j) in enumerate(flat)]
            it = iter(outs)
            return [[next(it) for _ in row] for row in origins]
        main = torch.cuda.current_stream(self.device)
        if len(self._streams) < n:
            self._streams += [torch.cuda.Stream(device=self.device) for _ in range(n - len(self._streams))]
        # stream assignment: round-robin in the reference's row-major tile order (measured best at 480 x 720 on 3 streams: the two streams
        # that carry the full-size tiles and the one that carries the edge tiles drift out of phase, so small-kernel phases of one tile meet
        # large convolutions of another; longest-processing-time-first balancing measured 3 % slower)
        flat = [(i, j) for row in origins for (i, j) in row]
        assign = [k % n for k in range(len(flat))]
        order = list(range(len(flat)))
        res = [None] * len(flat)
        for k in order:
            i, j = flat[k]
            st = self._streams[assign[k]]
            st.wait_stream(main)
            with torch.cuda.stream(st):
                t = self._run_tile_maybe_graph(src, i, j, th, tw, decode, k)
            t.record_stream(main)
            res[k] = t
        it = iter(res)
        out = [[next(it) for _ in row] for row in origins]
        for st in self._streams[:n]:
            main.wait_stream(st)
        return out

    def _process(self, src, decode):
        """_encode/_decode incl. tiled_encode/tiled_decode (:1085-1108, 1138-1163, 1206-1359) for one batch item [C,T,H,W]."""
        if decode:
            th, tw = self.tile_latent_min_height, self.tile_latent_min_width
            tsh, tsw = self.tile_sample_min_height, self.tile_sample_min_width
        else:
            th, tw = self.tile_sample_min_height, self.tile_sample_min_width
            tsh, tsw = self.tile_latent_min_height, self.tile_latent_min_width
        H, W = src.shape[2:]
        self._tiled_pass = bool(self.use_tiling and (W > tw or H > th))
        if not self._tiled_pass:
            return self._run_tile_maybe_graph(src, 0, 0, H, W, decode, 0).clone()      # a graph's output buffer is reused by its next replay
        st_h, st_w = int(th * (1 - self.tile_overlap_factor_height)), int(tw * (1 - self.tile_overlap_factor_width))
        bh, bw = int(tsh * self.tile_overlap_factor_height), int(tsw * self.tile_overlap_factor_width)
        lim_h, lim_w = tsh - bh, tsw - bw
        rows = self._run_tiles(src, [[(i, j) for j in range(0, W, st_w)] for i in range(0, H, st_h)], th, tw, decode)
        out_rows = []
        for i, row in enumerate(rows):
            out = []
            for j, tile in enumerate(row):
                if i > 0:
                    K.tile_blend(rows[i - 1][j], tile, 3, bh)      # in place, same order as the reference (blend_v then blend_h)
                if j > 0:
                    K.tile_blend(row[j - 1], tile, 4, bw)
                out.append(tile[:, :, :lim_h, :lim_w])
            out_rows.append(torch.cat(out, dim=3))
        return torch.cat(out_rows, dim=2)

    @torch.no_grad()
    def encode(self, x, return_dict=True):
        """x [B,3,T,H,W] -> latent_dist over [B,16,T',H/8,W/8] (moments computed on the GPU)."""
        if not x.is_cuda:
            raise RuntimeError("AutoencoderKLCogVideoX.encode: expected a GPU tensor (tokensgen_amd has no CPU fallback)")
        h = torch.stack([self._process(xi.contiguous(), False) for xi in x])
        post = DiagonalGaussianDistribution(h)
        return SimpleNamespace(latent_dist=post) if return_dict else (post,)

    @torch.no_grad()
    def decode(self, z, return_dict=True):
        """z [B,16,T,h,w] -> sample [B,3,4(T-1)+1,8h,8w] bf16."""
        if not z.is_cuda:
            raise RuntimeError("AutoencoderKLCogVideoX.decode: expected a GPU tensor (tokensgen_amd has no CPU fallback)")
        d = torch.stack([self._process(zi.contiguous(), True) for zi in z])
        return SimpleNamespace(sample=d) if return_dict else (d,)
