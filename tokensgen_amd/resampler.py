"""Resampler — host mirror of longvgen/video_ipadapter/resampler.py:132-244 (the condensed-token encoder):
384 learned queries x `depth` Perceiver layers over the 17 550 patch tokens of one 49-frame clip.  Same constructor keywords
and state-dict keys as the reference (`latents`, `proj_in`, `layers.{i}.0.*`, `layers.{i}.1.net.*`, `proj_out`, `norm_out`).

Runs entirely on the DiT kernels: LayerNorm = tg_adaln_modulate (no modulation), projections = tg_gemm_bf16 (the `+ latents`
residuals are the GEMM's gated-residual epilogue with a gate of ones), per-head QK-LayerNorm + the two RoPE tables in one
tg_qk_layernorm_rope launch per operand, attention = tg_attention_fwd (16 heads x 64).  K/V of image tokens and of the
queries live in one [b, Nx + Nq, 2*inner] buffer (the reference concatenates them, :100-101).
The optional PCA low-rank filter (:201-207, 230-237; gen.yaml passes a real `longvgen_pca` path to `set_pca`, infer_cogvideo_mp_fifo.py:118,167)
runs as one fp32 kernel per forward (tg_pca_lowrank_filter)."""
import json
import math
import os
from types import SimpleNamespace

import torch

from . import kernels as K
from . import lib as L

BF16 = torch.bfloat16


class Resampler:
    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_height_queries=6, num_width_queries=6, num_temporal_queries=13,
                 embedding_dim=1280, output_dim=1024, ff_mult=4, max_height_seq_len=16, max_width_seq_len=16, max_temporal_seq_len=49,
                 dropout=0.0, activation_fn="gelu-approximate", ff_inner_dim=None, final_dropout=True, ff_bias=True, device="cuda", **kwargs):
        if dim_head != 64 or activation_fn != "gelu-approximate" or not ff_bias or ff_inner_dim not in (None, dim * ff_mult):
            raise NotImplementedError("Resampler configuration outside what the kernels cover (head_dim 64, gelu-approximate FF with bias)")
        self.config = SimpleNamespace(dim=dim, depth=depth, dim_head=dim_head, heads=heads, num_height_queries=num_height_queries,
                                      num_width_queries=num_width_queries, num_temporal_queries=num_temporal_queries, embedding_dim=embedding_dim,
                                      output_dim=output_dim, ff_mult=ff_mult, max_height_seq_len=max_height_seq_len,
                                      max_width_seq_len=max_width_seq_len, max_temporal_seq_len=max_temporal_seq_len)
        self.num_height_queries, self.num_width_queries, self.num_temporal_queries = num_height_queries, num_width_queries, num_temporal_queries
        self.device = torch.device(device)
        self.dtype = BF16
        self.pca = None
        self._pca_dev = None
        self._sd = {}
        self._ones = None

    def set_pca(self, pca_path=None, device=None):
        """resampler.py:201-207 — `self.pca = torch.load(pca_path).to(device)`: the file is a pickled `pca.PCA` module (pca.py:6-66; the class
        is importable as `pca.PCA` through tokensgen_amd.compat / tokensgen_amd.pca).  Only its buffers are used: forward() projects every
        token onto the first 16 components and back (:230-237)."""
        if pca_path is None:
            self.pca = None
            self._pca_dev = None
            return
        from . import compat
        compat.ensure_pca_module()
        obj = torch.load(pca_path, map_location="cpu", weights_only=False) if isinstance(pca_path, (str, os.PathLike)) else pca_path
        if not (hasattr(obj, "components_") and hasattr(obj, "mean_")):
            raise ValueError(f"{pca_path}: expected a fitted pca.PCA (buffers mean_, components_)")
        if obj.components_.shape[1] != self.config.output_dim:
            raise ValueError(f"PCA width {obj.components_.shape[1]} != Resampler output_dim {self.config.output_dim}")
        self.pca = obj
        keep = min(16, obj.components_.shape[0])                       # `latents[:, 16:] = 0.0`
        dev = self.device if device is None else torch.device(device)
        self._pca_dev = (obj.components_[:keep].to(dev, torch.float32).contiguous(), obj.mean_.reshape(-1).to(dev, torch.float32).contiguous())
        print(f"Successfully set pca: {pca_path}")

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=BF16, device="cuda", broadcast=False, **unused):
        """diffusers' ModelMixin.from_pretrained as the entry script uses it (infer_cogvideo_mp_fifo.py:113-117, 162-166):
        <path>/<subfolder>/config.json + diffusion_pytorch_model.safetensors (or the sharded index).
        broadcast=True (torch.distributed initialised): rank 0 reads the weights, the other ranks receive them (runtime.broadcast_weights)."""
        from safetensors.torch import load_file
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, "config.json")) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        m = cls(**cfg, device=device)
        rank = 0
        if broadcast:
            import torch.distributed as dist
            broadcast = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
            rank = dist.get_rank() if broadcast else 0
        if rank == 0:
            idx = os.path.join(d, "diffusion_pytorch_model.safetensors.index.json")
            if os.path.exists(idx):
                with open(idx) as f:
                    files = sorted(set(json.load(f)["weight_map"].values()))
            else:
                files = ["diffusion_pytorch_model.safetensors"]
            sd = {}
            for fn in files:
                sd.update(load_file(os.path.join(d, fn)))
            m.load_state_dict(sd)
        if broadcast:
            from .runtime import broadcast_weights
            meta = [[(k, tuple(v.shape)) for k, v in sorted(m._sd.items())] if rank == 0 else None]
            dist.broadcast_object_list(meta, src=0)              # the Resampler has no static shape table: names and shapes travel first
            if rank != 0:
                m.load_state_dict({k: torch.zeros(sh, dtype=BF16) for k, sh in meta[0]})
            broadcast_weights(m, src=0)
        return m

    def save_pretrained(self, path):
        """config.json + diffusion_pytorch_model.safetensors in diffusers' layout (what from_pretrained reads)."""
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(dict(vars(self.config), _class_name="Resampler"), f, indent=1)
        save_file({k: v.detach().cpu().contiguous() for k, v in self._sd.items()}, os.path.join(path, "diffusion_pytorch_model.safetensors"))

    def to(self, *args, **kwargs):
        """`.to(device)` / `.to(dtype)` of the reference call sites; the arithmetic is bf16 on the GPU whatever dtype is asked for."""
        dev = kwargs.get("device")
        for a in args:
            if isinstance(a, (str, torch.device)):
                dev = a
        if dev is not None and torch.device(dev) != self.device:
            self.device = torch.device(dev)
            self._sd = {k: v.to(self.device) for k, v in self._sd.items()}
            self._ones = None
            if self._pca_dev is not None:
                self._pca_dev = tuple(t.to(self.device) for t in self._pca_dev)
        return self

    def expected_keys(self):
        c = self.config
        keys = ["latents", "proj_in.weight", "proj_in.bias", "proj_out.weight", "proj_out.bias", "norm_out.weight", "norm_out.bias"]
        for i in range(c.depth):
            keys += [f"layers.{i}.0.{n}.{wb}" for n in ("norm1", "norm2", "norm_q", "norm_k") for wb in ("weight", "bias")]
            keys += [f"layers.{i}.0.{n}.weight" for n in ("to_q", "to_kv", "to_out")]
            keys += [f"layers.{i}.1.net.0.proj.weight", f"layers.{i}.1.net.0.proj.bias", f"layers.{i}.1.net.2.weight", f"layers.{i}.1.net.2.bias"]
        return keys

    def state_dict(self):
        return dict(self._sd)

    def load_state_dict(self, sd, strict=True):
        want = self.expected_keys()
        missing, unexpected = [k for k in want if k not in sd], [k for k in sd if k not in want]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Resampler.load_state_dict: missing {missing[:4]}, unexpected {unexpected[:4]}")
        self._sd = {k: v.detach().to(self.device, BF16).contiguous() for k, v in sd.items()}
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    def _ones_gate(self, tokens, width, batch=1):
        """Gate table of ones: turns the gated-residual GEMM epilogue into `y = residual + linear`."""
        if self._ones is None or self._ones[0].shape[-1] < width or self._ones[1].shape[0] < tokens:
            self._ones = (torch.ones(1, 1, width, dtype=BF16, device=self.device), torch.zeros(max(tokens, 1024), dtype=torch.uint8, device=self.device))
        # expand: batch stride 0 — the GATE_RES epilogue reads mod + b * mod_batch_stride, every batch item must see the same ones
        return K.GroupTable(self._ones[0].expand(max(batch, 1), 1, -1), self._ones[1], [0], [0], [0], [0])

    @torch.no_grad()
    def forward(self, x, image_rotary_emb=None, sampling_rotary_emb=None):
        """x [b, f, n, embedding_dim] bf16 on the GPU -> [b, Tq, output_dim, Hq, Wq]  (resampler.py:209-244)."""
        c, sd = self.config, self._sd
        if not x.is_cuda:
            raise RuntimeError("Resampler.forward: expected a GPU tensor (tokensgen_amd has no CPU fallback)")
        b = x.shape[0]
        H, inner, dim = c.heads, c.heads * c.dim_head, c.dim
        e = lambda *s: torch.empty(*s, dtype=BF16, device=self.device)
        xin = x.to(BF16).reshape(b, -1, c.embedding_dim).contiguous()
        Nx = xin.shape[1]
        Nq = sd["latents"].shape[1]
        xp = e(b, Nx, dim)
        K.gemm(xin, sd["proj_in.weight"], sd["proj_in.bias"], xp, L.EPI_BIAS)
        lat = sd["latents"].expand(b, -1, -1).clone()          # clone: the layers update it in place
        cat = e(b, Nx + Nq, dim)                      # norm1(x) | norm2(latents): the kv input of every layer
        kv = e(b, Nx + Nq, 2 * inner)
        q = e(b, Nq, inner)
        vt = e(b, H, 64, (Nx + Nq + 63) // 64 * 64)
        ao = e(b, Nq, inner)
        ffh = e(b, Nq, dim * c.ff_mult)
        dev = lambda t: t.to(self.device, torch.float32).contiguous()
        img = None if image_rotary_emb is None else tuple(dev(t) for t in image_rotary_emb)
        smp = None if sampling_rotary_emb is None else tuple(dev(t) for t in sampling_rotary_emb)
        ones = self._ones_gate(Nq, dim, b)
        for i in range(c.depth):
            p = f"layers.{i}.0"
            K.adaln_modulate(xp, cat[:, :Nx], sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5, None)
            K.adaln_modulate(lat, cat[:, Nx:], sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5, None)
            K.gemm(cat[:, Nx:], sd[p + ".to_q.weight"], None, q, L.EPI_BIAS)
            K.gemm(cat, sd[p + ".to_kv.weight"], None, kv, L.EPI_BIAS)
            K.qk_layernorm_rope(q, H, sd[p + ".norm_q.weight"], sd[p + ".norm_q.bias"], 1e-6, None if smp is None else (0, smp))
            K.qk_layernorm_rope(kv[:, :, :inner], H, sd[p + ".norm_k.weight"], sd[p + ".norm_k.bias"], 1e-6,
                                None if img is None else (0, img), None if smp is None else (Nx, smp))
            K.transpose_v(kv[:, :, inner:], H, 0, Nx + Nq, vt)
            K.attention(q, kv[:, :, :inner], vt, Nx + Nq, ao, H, 1.0 / math.sqrt(c.dim_head))
            K.gemm(ao, sd[p + ".to_out.weight"], None, lat, L.EPI_BIAS_GATE_RES, residual=lat, gate=ones)          # + latents
            f = f"layers.{i}.1"
            K.gemm(lat, sd[f + ".net.0.proj.weight"], sd[f + ".net.0.proj.bias"], ffh, L.EPI_BIAS_GELU)
            K.gemm(ffh, sd[f + ".net.2.weight"], sd[f + ".net.2.bias"], lat, L.EPI_BIAS_GATE_RES, residual=lat, gate=ones)   # + latents
        po = e(b, Nq, c.output_dim)
        K.gemm(lat, sd["proj_out.weight"], sd["proj_out.bias"], po, L.EPI_BIAS)
        out = e(b, Nq, c.output_dim)
        K.adaln_modulate(po, out, sd["norm_out.weight"], sd["norm_out.bias"], 1e-5, None)
        if self._pca_dev is not None:                      # :230-237, fp32 like the reference (`.to(self.pca.components_.dtype)`)
            comp, mean = self._pca_dev
            K.pca_lowrank_filter(out.view(-1, c.output_dim), comp, mean, out.view(-1, c.output_dim))
        return out.reshape(b, self.num_temporal_queries, self.num_height_queries, self.num_width_queries, -1).permute(0, 1, 4, 2, 3)

    __call__ = forward
