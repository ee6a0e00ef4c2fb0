"""CogVideoXTransformer3DModel — host mirror of longvgen/models/cogvideox_transformer_3d.py for the To2V
("vip", func_type "1") and plain (T2To / base) processors, driving the HIP kernels of libtokensgen_hip.so.

Same constructor keywords, `forward` signature, `set_vip_layers`, `config` and state-dict key names as the
reference, so `infer_cogvideo_mp_fifo.py` can build it and load the same checkpoints.

MI355X-first layout decisions:
  * one residual stream buffer X [B, Nt+Nv+Np, D] ordered text | video | vip — the reference's
    cat/split/slice traffic (cogvideox_transformer_3d.py:232-235, 267-288, 315-324) disappears: the three
    token kinds are row ranges of the same buffer;
  * parameters that feed one GEMM share ONE storage: to_q|to_k|to_v -> [3D, D]; vip_to_q|k|v -> [3D, D];
    every AdaLN modulation linear of every layer (+ norm_out.linear) -> one [L*18D + 2D, 512] matrix that is
    multiplied ONCE per forward (temb is layer independent).  The nn.Parameters registered under the
    reference's names are views into those storages, so state_dict()/load_state_dict() keep the reference
    key names with zero copies;
  * gated residuals, GELU, SiLU, bias are GEMM epilogues; LayerNorm+modulate, QK-LayerNorm+RoPE are single
    passes; the three SDPAs of the To2V processor are two launches of one flash kernel.
"""
import json
import math
import os
from types import SimpleNamespace

import torch
from torch import nn

from . import kernels as K
from . import lib as L

BF16 = torch.bfloat16


class Transformer2DModelOutput:
    def __init__(self, sample):
        self.sample = sample


class VideoIPAdapterCogVideoXAttnProcessor2_0(nn.Module):
    """Holder of the To2V branch weights under the reference's names (attention_processor.py:1955-1980).
    The class name is load-bearing: the pipeline sets `.scale` on modules named exactly like this
    (pipeline_cogvideox_mp_fifo.py:981-983).  Inside the model the arithmetic is fused into CogVideoXTransformer3DModel.forward;
    called on its own (`__call__`, the reference's processor signature :1982-1991) it runs the same kernels as a standalone op."""

    def __init__(self, scale=1.0, num_tokens=None):
        super().__init__()
        self.scale = scale
        self.num_tokens = num_tokens
        self.vip_to_q, self.vip_to_k, self.vip_to_v = _Lin(), _Lin(), _Lin()
        self.vip_norm_q, self.vip_norm_k = _Lin(), _Lin()

    def __call__(self, attn, hidden_states, encoder_hidden_states, attention_mask=None, image_rotary_emb=None,
                 vip_image_rotary_emb=None, vip_condition_rotary_emb=None):
        return _processor_call(attn, hidden_states, encoder_hidden_states, attention_mask, image_rotary_emb, vip_image_rotary_emb,
                               vip_condition_rotary_emb)


def _processor_call(attn, hidden_states, encoder_hidden_states, attention_mask, image_rotary_emb, vip_image_rotary_emb=None,
                    vip_condition_rotary_emb=None):
    if attention_mask is not None:
        raise NotImplementedError("attention masks are not used on the TokensGen hot path")
    model = attn._owner() if getattr(attn, "_owner", None) is not None else None
    if model is None:
        raise RuntimeError("this Attention module is not attached to a CogVideoXTransformer3DModel (its weights live in the model's fused storages)")
    return model.attention_op(attn._layer, hidden_states, encoder_hidden_states, image_rotary_emb, vip_image_rotary_emb, vip_condition_rotary_emb)


class CogVideoXAttnProcessor2_0:
    """Plain joint-attention processor (attention_processor.py:1885-1953): same call signature, runs on the HIP kernels."""

    def __call__(self, attn, hidden_states, encoder_hidden_states, attention_mask=None, image_rotary_emb=None, **unused):
        return _processor_call(attn, hidden_states, encoder_hidden_states, attention_mask, image_rotary_emb)


class _Lin(nn.Module):
    """Parameter holder (`weight`, `bias`) — parameters are attached later as views of fused storages."""


class _Attention(nn.Module):
    def __init__(self):
        super().__init__()
        self.to_q, self.to_k, self.to_v = _Lin(), _Lin(), _Lin()
        self.norm_q, self.norm_k = _Lin(), _Lin()
        self.to_out = nn.ModuleList([_Lin()])
        object.__setattr__(self, "processor", CogVideoXAttnProcessor2_0())

    def set_processor(self, processor):
        """attention_processor.py:423-441: processors that own weights are registered sub-modules."""
        self.__dict__.pop("processor", None)
        self._modules.pop("processor", None)
        if isinstance(processor, nn.Module):
            self._modules["processor"] = processor
        else:
            object.__setattr__(self, "processor", processor)

    def get_processor(self):
        return self.processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        """attention_processor.py:457-501: hand over to the processor with the keyword arguments its signature accepts."""
        import inspect
        ok = set(inspect.signature(self.processor.__call__).parameters)
        kw = {k: v for k, v in cross_attention_kwargs.items() if k in ok}
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask, **kw)


class _Norm(nn.Module):
    def __init__(self):
        super().__init__()
        self.linear, self.norm = _Lin(), _Lin()


class _FF(nn.Module):
    """diffusers FeedForward key layout: net.0.proj / net.2"""

    def __init__(self):
        super().__init__()
        g = nn.Module()
        g.proj = _Lin()
        self.net = nn.ModuleList([g, nn.Identity(), _Lin()])


class CogVideoXBlock(nn.Module):
    def __init__(self):
        super().__init__()
        self.norm1, self.norm2 = _Norm(), _Norm()
        self.attn1 = _Attention()
        self.ff = _FF()
        self.use_vip = False


class _PatchEmbed(nn.Module):
    def __init__(self):
        super().__init__()
        self.proj, self.text_proj = _Lin(), _Lin()


def _pad_to(n, m):
    return (n + m - 1) // m * m


class CogVideoXTransformer3DModel(nn.Module):
    """See module docstring.  Construct directly on the GPU: CogVideoXTransformer3DModel(**config, device="cuda")."""

    def __init__(self, num_attention_heads=30, attention_head_dim=64, in_channels=16, out_channels=16,
                 flip_sin_to_cos=True, freq_shift=0, time_embed_dim=512, text_embed_dim=4096, num_layers=30,
                 dropout=0.0, attention_bias=True, sample_width=90, sample_height=60, sample_frames=49, patch_size=2,
                 temporal_compression_ratio=4, max_text_seq_length=226, activation_fn="gelu-approximate",
                 timestep_activation_fn="silu", norm_elementwise_affine=True, norm_eps=1e-5,
                 spatial_interpolation_scale=1.875, temporal_interpolation_scale=1.0,
                 use_rotary_positional_embeddings=False, use_learned_positional_embeddings=False,
                 use_output_projection=True, device="cuda", dtype=BF16):
        super().__init__()
        cfg = {k: v for k, v in locals().items() if k not in ("self", "__class__", "device", "dtype")}
        self.config = SimpleNamespace(**cfg)
        if attention_head_dim != 64:
            raise NotImplementedError("the gfx950 attention kernel is specialised for head_dim 64 (CogVideoX)")
        if not use_rotary_positional_embeddings:
            raise NotImplementedError("only the rotary (CogVideoX-5B) model is on the hot path; the 2B sin-cos variant is out of scope")
        if activation_fn != "gelu-approximate" or timestep_activation_fn != "silu" or not attention_bias \
                or not norm_elementwise_affine or patch_size not in (1, 2) or not flip_sin_to_cos or freq_shift != 0 \
                or not use_output_projection or dtype != BF16:
            raise NotImplementedError("configuration differs from CogVideoX-5b in a way the fused kernels do not cover")
        self.use_vip = False
        self.vip_length = 0
        self.vip_func_type = None
        self._device = torch.device(device)
        self._fused = {}          # name -> storage tensor
        self._views = []          # (holder module, attr, fused name, row slice, shape)
        self._ws = {}
        # "constant_shift" (default) | "running_max": see _attn_fast; TG_ATTN_FIXEDM=0 in the environment forces the latter inside the library
        self.attn_path = "constant_shift"
        D = num_attention_heads * attention_head_dim
        self.inner_dim = D
        self.patch_embed = _PatchEmbed()
        self.time_embedding = nn.Module()
        self.time_embedding.linear_1, self.time_embedding.linear_2 = _Lin(), _Lin()
        self.transformer_blocks = nn.ModuleList([CogVideoXBlock() for _ in range(num_layers)])
        import weakref
        for li, blk in enumerate(self.transformer_blocks):      # operator-level seam: blk.attn1(...) finds its fused weights
            object.__setattr__(blk.attn1, "_owner", weakref.ref(self))
            blk.attn1._layer = li
        self.norm_final = _Lin()
        self.norm_out = _Norm()
        self.proj_out = _Lin()
        self._build_storage()

    # ------------------------------------------------------------------------------------------ storage
    def _alloc(self, name, *shape):
        t = torch.zeros(*shape, dtype=BF16, device=self._device)
        self._fused[name] = t
        return t

    def _bind(self, holder, attr, fused, rows, shape=None):
        """Register holder.<attr> as an nn.Parameter that is a view of self._fused[fused][rows]."""
        self._views.append((holder, attr, fused, rows, shape))
        v = self._fused[fused][rows]
        if shape is not None:
            v = v.view(*shape)
        holder._parameters[attr] = nn.Parameter(v, requires_grad=False)

    def _mod_cols(self):
        """Column (= output feature) layout of the fused modulation matrix."""
        D, Lyr = self.inner_dim, self.config.num_layers
        per = 18 * D if self.use_vip else 12 * D
        return per, Lyr * per, Lyr * per + 2 * D

    def _build_storage(self):
        c, D = self.config, self.inner_dim
        te, Lyr = c.time_embed_dim, c.num_layers
        ps = c.patch_size                   # 2: To2V / base model; 1: the T2To token model (train_cogvideo_t2to.py:1277)
        kin = c.in_channels * ps * ps
        self._alloc("patch.w", D, _pad_to(kin, 64)); self._alloc("patch.b", D)       # K padded to the GEMM granule (zeros)
        self._bind(self.patch_embed.proj, "weight", "patch.w", (slice(None), slice(0, kin)), (D, c.in_channels, ps, ps))
        self._bind(self.patch_embed.proj, "bias", "patch.b", slice(None))
        self._alloc("text.w", D, c.text_embed_dim); self._alloc("text.b", D)
        self._bind(self.patch_embed.text_proj, "weight", "text.w", slice(None))
        self._bind(self.patch_embed.text_proj, "bias", "text.b", slice(None))
        self._alloc("t1.w", te, D); self._alloc("t1.b", te); self._alloc("t2.w", te, te); self._alloc("t2.b", te)
        self._bind(self.time_embedding.linear_1, "weight", "t1.w", slice(None)); self._bind(self.time_embedding.linear_1, "bias", "t1.b", slice(None))
        self._bind(self.time_embedding.linear_2, "weight", "t2.w", slice(None)); self._bind(self.time_embedding.linear_2, "bias", "t2.b", slice(None))
        for i, blk in enumerate(self.transformer_blocks):
            p = f"l{i}."
            self._alloc(p + "qkv.w", 3 * D, D); self._alloc(p + "qkv.b", 3 * D)
            for j, n in enumerate(("to_q", "to_k", "to_v")):
                self._bind(getattr(blk.attn1, n), "weight", p + "qkv.w", slice(j * D, (j + 1) * D))
                self._bind(getattr(blk.attn1, n), "bias", p + "qkv.b", slice(j * D, (j + 1) * D))
            self._alloc(p + "qknorm", 4, 64)   # norm_q.w, norm_q.b, norm_k.w, norm_k.b
            self._bind(blk.attn1.norm_q, "weight", p + "qknorm", 0); self._bind(blk.attn1.norm_q, "bias", p + "qknorm", 1)
            self._bind(blk.attn1.norm_k, "weight", p + "qknorm", 2); self._bind(blk.attn1.norm_k, "bias", p + "qknorm", 3)
            self._alloc(p + "out.w", D, D); self._alloc(p + "out.b", D)
            self._bind(blk.attn1.to_out[0], "weight", p + "out.w", slice(None)); self._bind(blk.attn1.to_out[0], "bias", p + "out.b", slice(None))
            self._alloc(p + "ff1.w", 4 * D, D); self._alloc(p + "ff1.b", 4 * D)
            self._alloc(p + "ff2.w", D, 4 * D); self._alloc(p + "ff2.b", D)
            self._bind(blk.ff.net[0].proj, "weight", p + "ff1.w", slice(None)); self._bind(blk.ff.net[0].proj, "bias", p + "ff1.b", slice(None))
            self._bind(blk.ff.net[2], "weight", p + "ff2.w", slice(None)); self._bind(blk.ff.net[2], "bias", p + "ff2.b", slice(None))
            self._alloc(p + "ln", 4, D)        # norm1.norm.w/b, norm2.norm.w/b
            self._bind(blk.norm1.norm, "weight", p + "ln", 0); self._bind(blk.norm1.norm, "bias", p + "ln", 1)
            self._bind(blk.norm2.norm, "weight", p + "ln", 2); self._bind(blk.norm2.norm, "bias", p + "ln", 3)
        self._alloc("final.ln", 4, D)          # norm_final.w/b, norm_out.norm.w/b
        self._bind(self.norm_final, "weight", "final.ln", 0); self._bind(self.norm_final, "bias", "final.ln", 1)
        self._bind(self.norm_out.norm, "weight", "final.ln", 2); self._bind(self.norm_out.norm, "bias", "final.ln", 3)
        npo = ps * ps * c.out_channels
        self._alloc("proj_out.w", _pad_to(npo, 128), D); self._alloc("proj_out.b", _pad_to(npo, 128))   # N padded to the GEMM tile
        self._bind(self.proj_out, "weight", "proj_out.w", slice(0, npo)); self._bind(self.proj_out, "bias", "proj_out.b", slice(0, npo))
        self._build_mod_storage()

    def _build_mod_storage(self):
        """(Re)build the fused modulation matrix; called again by set_vip_layers (layout gains the vip rows)."""
        c, D, te = self.config, self.inner_dim, self.config.time_embed_dim
        old = {}
        for (holder, attr, fused, rows, shape) in self._views:
            if fused in ("mod.w", "mod.b"):
                old[(id(holder), attr)] = holder._parameters[attr].detach().clone()
        self._views = [v for v in self._views if v[2] not in ("mod.w", "mod.b")]
        per, out_base, total = self._mod_cols()
        self._alloc("mod.w", _pad_to(total, 128), te); self._alloc("mod.b", _pad_to(total, 128))
        for i, blk in enumerate(self.transformer_blocks):
            base = i * per
            order = [(blk.norm1.linear, 6 * D)]
            if self.use_vip:
                order.append((blk.vip_norm1.linear, 3 * D))
            order.append((blk.norm2.linear, 6 * D))
            if self.use_vip:
                order.append((blk.vip_norm2.linear, 3 * D))
            for holder, n in order:
                self._bind(holder, "weight", "mod.w", slice(base, base + n)); self._bind(holder, "bias", "mod.b", slice(base, base + n))
                base += n
        self._bind(self.norm_out.linear, "weight", "mod.w", slice(out_base, out_base + 2 * D))
        self._bind(self.norm_out.linear, "bias", "mod.b", slice(out_base, out_base + 2 * D))
        for (holder, attr, fused, rows, shape) in self._views:
            if fused in ("mod.w", "mod.b") and (id(holder), attr) in old:
                holder._parameters[attr].data.copy_(old[(id(holder), attr)])

    def _apply(self, fn, recurse=True):
        """Keep the shared storages shared across .to()/.cuda(): move storages, then re-create the views."""
        for name in list(self._fused):
            self._fused[name] = fn(self._fused[name])
        for (holder, attr, fused, rows, shape) in self._views:
            v = self._fused[fused][rows]
            if shape is not None:
                v = v.view(*shape)
            holder._parameters[attr] = nn.Parameter(v, requires_grad=False)
        any_t = next(iter(self._fused.values()))
        if any_t.dtype != BF16:
            raise NotImplementedError("tokensgen_amd kernels are bf16-only")
        self._device = any_t.device
        self._ws = {}
        return self

    @property
    def device(self):
        return self._device

    @property
    def dtype(self):
        return BF16

    # ------------------------------------------------------------------------------------------ vip layers
    def set_vip_layers(self, vip_ckpt_dir=None, **kwargs):
        """cogvideox_transformer_3d.py:591-622 (+ block :145-218, patch_embed embeddings.py:423-431)."""
        func_type = kwargs["func_type"]
        if func_type != "1":
            raise NotImplementedError(f"vip func_type {func_type!r}: only \"1\" (the shipped To2V configs) is on the hot path")
        self.use_vip = True
        self.vip_length = kwargs["length"]
        self.vip_func_type = func_type
        D = self.inner_dim
        rp = kwargs["resampler_params"]
        pe = self.patch_embed
        pe.vip_proj = _Lin()
        pe.vip_num_height_queries, pe.vip_num_width_queries = rp["num_height_queries"], rp["num_width_queries"]
        pe.vip_num_temporal_queries = rp["num_temporal_queries"]
        self._alloc("vipproj.w", D, rp["output_dim"]); self._alloc("vipproj.b", D)
        self._bind(pe.vip_proj, "weight", "vipproj.w", slice(None)); self._bind(pe.vip_proj, "bias", "vipproj.b", slice(None))
        for i, blk in enumerate(self.transformer_blocks):
            p = f"l{i}."
            blk.use_vip = True
            blk.vip_length = kwargs["length"]
            blk.vip_norm1, blk.vip_norm2 = _Norm(), _Norm()
            proc = VideoIPAdapterCogVideoXAttnProcessor2_0(scale=kwargs["scale"], num_tokens=kwargs["length"])
            blk.attn1.set_processor(proc)
            self._alloc(p + "vqkv.w", 3 * D, D); self._alloc(p + "vqkv.b", 3 * D)
            for j, n in enumerate(("vip_to_q", "vip_to_k", "vip_to_v")):
                self._bind(getattr(proc, n), "weight", p + "vqkv.w", slice(j * D, (j + 1) * D))
                self._bind(getattr(proc, n), "bias", p + "vqkv.b", slice(j * D, (j + 1) * D))
            self._alloc(p + "vqknorm", 4, 64)
            self._bind(proc.vip_norm_q, "weight", p + "vqknorm", 0); self._bind(proc.vip_norm_q, "bias", p + "vqknorm", 1)
            self._bind(proc.vip_norm_k, "weight", p + "vqknorm", 2); self._bind(proc.vip_norm_k, "bias", p + "vqknorm", 3)
            self._alloc(p + "vln", 4, D)
            self._bind(blk.vip_norm1.norm, "weight", p + "vln", 0); self._bind(blk.vip_norm1.norm, "bias", p + "vln", 1)
            self._bind(blk.vip_norm2.norm, "weight", p + "vln", 2); self._bind(blk.vip_norm2.norm, "bias", p + "vln", 3)
            # the reference initialises the vip projections / norms as copies of the base ones (:207-218)
            self._fused[p + "vqkv.w"].copy_(self._fused[p + "qkv.w"]); self._fused[p + "vqkv.b"].copy_(self._fused[p + "qkv.b"])
            self._fused[p + "vqknorm"].copy_(self._fused[p + "qknorm"])
            self._fused[p + "vln"][0].fill_(1.0); self._fused[p + "vln"][2].fill_(1.0)
        self._build_mod_storage()
        self._ws = {}
        if vip_ckpt_dir is not None:
            path = os.path.join(vip_ckpt_dir, "vip.pt")
            if not os.path.exists(path):
                raise IOError(f"no vip weights found in {vip_ckpt_dir}")
            sd = torch.load(path, weights_only=True)
            own = self.state_dict().keys()
            for k in sd:
                assert k in own, k
            self.load_state_dict(sd, strict=False)

    def save_vip_layers(self, vip_ckpt_dir=None):
        """cogvideox_transformer_3d.py:624-634"""
        assert self.use_vip
        sd = {n: p.detach().to("cpu").to(torch.float32) for n, p in self.named_parameters() if "vip_" in n}
        os.makedirs(vip_ckpt_dir, exist_ok=True)
        torch.save(sd, os.path.join(vip_ckpt_dir, "vip.pt"))

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=BF16, device="cuda", broadcast=False, **kw):
        """from_pretrained-lite: <dir>/config.json + diffusion_pytorch_model*.safetensors.
        broadcast=True (with torch.distributed initialised): only rank 0 reads the safetensors files, the other ranks build the empty model from
        config.json and receive the weights with `runtime.broadcast_weights` (RCCL over xGMI) — 14.3 GB from one disk read instead of N.  Call
        `runtime.broadcast_weights(model)` again after `set_vip_layers(vip_ckpt_dir)` if only rank 0 loaded vip.pt."""
        from safetensors.torch import load_file
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, "config.json")) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        m = cls(**cfg, device=device, dtype=torch_dtype)
        rank = 0
        if broadcast:
            import torch.distributed as dist
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        if rank == 0:
            idx = os.path.join(d, "diffusion_pytorch_model.safetensors.index.json")
            files = sorted(set(json.load(open(idx))["weight_map"].values())) if os.path.exists(idx) else ["diffusion_pytorch_model.safetensors"]
            for fn in files:
                m.load_state_dict(load_file(os.path.join(d, fn)), strict=False)
        if broadcast:
            from .runtime import broadcast_weights
            broadcast_weights(m, src=0)
        return m

    def _after_weight_update(self):
        """Weights changed underneath the parameter views (runtime.broadcast_weights, an in-place optimizer step): forget what was derived."""
        self.attn_path = "constant_shift"

    # ------------------------------------------------------------------------------------------ attention path
    def _attn_fast(self, ws):
        """The constant-shift softmax path of the 512-row attention kernel (tg_attn_segment.k_norm2_max): its range bound comes from the
        DATA of every launch (key norms from the K-norm kernel, query norms inside the attention kernel), so it holds for any checkpoint,
        and rows it cannot cover are verified + recomputed on the device.  The host only watches the retry counter without ever waiting
        for the device: weights for which more than 2 % of the workgroups keep being recomputed are better served by the running
        maximum, and the model then stays on it until the weights change (`attn_path`)."""
        if self.attn_path == "running_max" or ws.retry is None:
            return False
        r = ws.retry                                       # launch count and poll mark live with the counter they describe (one per workspace shape)
        got = r.poll(r.launches)
        if got is not None:
            n, mark = got
            launches, retries = mark - r.mark[0], n - r.mark[1]
            r.mark = (mark, n)
            if launches > 0 and retries > 0.02 * launches * (r.ints - 1):
                self.attn_path = "running_max"
                return False
        r.launches += 1
        return True

    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self.attn_path = "constant_shift"
        return r

    # ------------------------------------------------------------------------------------------ workspace
    def _workspace(self, B, Nt, Nv, Np, Fm):
        key = (B, Nt, Nv, Np, Fm)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        D, H = self.inner_dim, self.config.num_attention_heads
        N1, N = Nt + Nv, Nt + Nv + Np
        dev = self._device
        e = lambda *s: torch.empty(*s, dtype=BF16, device=dev)
        ws = SimpleNamespace()
        ws.X, ws.Xn, ws.AO = e(B, N, D), e(B, N, D), e(B, N, D)
        ws.QKV = e(B, N1, 3 * D)
        ws.FF = e(B, N, 4 * D)
        ws.Vt1 = e(B, H, 64, _pad_to(N1, 64))
        # constant-shift attention: key-norm bounds per (batch, head) of the two K projections, their reduction scratch, the retry flags
        ws.kmax1 = torch.zeros(B, H, dtype=torch.float32, device=dev)
        ws.kmax2 = torch.zeros(B, H, dtype=torch.float32, device=dev)
        ws.kmax_ws = K.kmax_workspace(N, H, B, dev)
        ws.retry = K.AttnRetry(N1, Np, H, B, dev)
        if Np:
            ws.QKVv = e(B, N, 3 * D)
            ws.Vt2 = e(B, H, 64, _pad_to(Np, 64))
            ws.Vt3 = e(B, H, 64, _pad_to(N, 64))
        per, out_base, total = self._mod_cols()
        ws.mod = e(B, Fm, _pad_to(total, 128))
        ws.sin = e(B * Fm, D)
        ws.t1 = e(B * Fm, self.config.time_embed_dim)
        ws.semb = e(B * Fm, self.config.time_embed_dim)
        ws.patches = torch.zeros(B, Nv, self._fused["patch.w"].shape[1], dtype=BF16, device=dev)   # pad columns stay zero
        ws.po = e(B, Nv, self._fused["proj_out.w"].shape[0])
        hw = Nv // Fm if Fm > 1 else Nv
        tg = torch.empty(N, dtype=torch.uint8)
        tg[:Nt] = Fm
        tg[Nt:N1] = (torch.arange(Nv) // hw).to(torch.uint8) if Fm > 1 else 0
        tg[N1:] = Fm + 1
        ws.tok_group = tg.to(dev)
        self._ws = {key: ws}          # keep one shape resident
        return ws

    def _tables(self, ws, layer, Fm, which):
        """Group table for norm{which} of `layer`: groups 0..Fm-1 video frames, Fm text, Fm+1 vip."""
        D = self.inner_dim
        per, _, _ = self._mod_cols()
        base = layer * per
        if self.use_vip:
            main = base + (0 if which == 1 else 9 * D)
            vip = main + 6 * D
        else:
            main = base + (0 if which == 1 else 6 * D)
            vip = main
        rows = list(range(Fm)) + [0, 0]
        sh = [main] * Fm + [main + 3 * D, vip]
        sc = [main + D] * Fm + [main + 4 * D, vip + D]
        ga = [main + 2 * D] * Fm + [main + 5 * D, vip + 2 * D]
        return K.GroupTable(ws.mod, ws.tok_group, rows, sh, sc, ga)

    def _run_block(self, i, ws, B, Nt, Nv, Np, Fm, rope, vrope, crope):
        """One CogVideoXBlock (cogvideox_transformer_3d.py:221-332 + attention_processor.py:1982-2155) on the
        residual stream ws.X = text | video | vip, in place."""
        c, D, H = self.config, self.inner_dim, self.config.num_attention_heads
        F = self._fused
        X = ws.X
        N1, N = Nt + Nv, Nt + Nv + Np
        use_vip = Np > 0
        sm_scale = 1.0 / math.sqrt(64)
        blk = self.transformer_blocks[i]
        p = f"l{i}."
        ln = F[p + "ln"]
        t1 = self._tables(ws, i, Fm, 1)
        K.adaln_modulate(X[:, :N1], ws.Xn[:, :N1], ln[0], ln[1], c.norm_eps, t1)
        if use_vip:
            vln = F[p + "vln"]
            K.adaln_modulate(X[:, N1:], ws.Xn[:, N1:], vln[0], vln[1], c.norm_eps, t1.offset(N1))
        self._attn_core(i, ws, B, Nt, Nv, Np, rope, vrope, crope)
        # out projection with the gated residual as epilogue (cogvideox_transformer_3d.py:290-293)
        K.gemm(ws.AO, F[p + "out.w"], F[p + "out.b"], X, L.EPI_BIAS_GATE_RES, residual=X, gate=t1)
        t2 = self._tables(ws, i, Fm, 2)
        K.adaln_modulate(X[:, :N1], ws.Xn[:, :N1], ln[2], ln[3], c.norm_eps, t2)
        if use_vip:
            K.adaln_modulate(X[:, N1:], ws.Xn[:, N1:], vln[2], vln[3], c.norm_eps, t2.offset(N1))
        # feed-forward over all tokens (same ff weights for the vip rows, :315-324)
        K.gemm(ws.Xn, F[p + "ff1.w"], F[p + "ff1.b"], ws.FF, L.EPI_BIAS_GELU)
        K.gemm(ws.FF, F[p + "ff2.w"], F[p + "ff2.b"], X, L.EPI_BIAS_GATE_RES, residual=X, gate=t2)


    def _attn_core(self, i, ws, B, Nt, Nv, Np, rope, vrope, crope):
        """Attention of block i between the input norm and `to_out` (attention_processor.py:1982-2135 / 1895-1945): reads ws.Xn
        (text | video | vip rows), leaves the pre-projection attention output in ws.AO.  ws needs Xn, QKV, Vt1, AO and, with vip
        tokens, QKVv, Vt2, Vt3."""
        D, H = self.inner_dim, self.config.num_attention_heads
        F = self._fused
        N1, N = Nt + Nv, Nt + Nv + Np
        use_vip = Np > 0
        sm_scale = 1.0 / math.sqrt(64)
        blk = self.transformer_blocks[i]
        p = f"l{i}."
        # QKV projections (+ vip-weight projections over ALL tokens: x rows and vip rows share vip_to_*)
        paired = use_vip and N1 >= 1024 and (3 * D) % 256 == 0       # both projections in one launch of the 256^2 kernel
        # ... with V written transposed by the GEMM epilogue (no transpose_v pass; the V columns of QKV / QKVv are then never written)
        pad64 = lambda n: (n + 63) // 64 * 64
        # the vip keys' V^T can be read as the tail columns of the all-keys image only if pad64(Np) keys from column N1 stay inside a row
        vt2_view_ok = N1 % 8 == 0 and N1 + pad64(Np) <= pad64(N)
        fused_vt = K.gemm_qkv_supported(N1, 3 * D, D, 2 * D) and (not use_vip or vt2_view_ok)
        qn = F[p + "qknorm"]
        kscale = sm_scale * 1.4426950408889634       # softmax scale * log2(e) folded into K before its single bf16 rounding
        if fused_vt and use_vip:
            K.gemm_qkv(ws.Xn[:, :N1], F[p + "qkv.w"], F[p + "qkv.b"], ws.QKV, ws.Vt1, ws.Xn, F[p + "vqkv.w"], F[p + "vqkv.b"], ws.QKVv, ws.Vt3)
        elif fused_vt:
            K.gemm_qkv(ws.Xn[:, :N1], F[p + "qkv.w"], F[p + "qkv.b"], ws.QKV, ws.Vt1)
        elif paired:
            K.gemm_pair(ws.Xn[:, :N1], F[p + "qkv.w"], F[p + "qkv.b"], ws.QKV, ws.Xn, F[p + "vqkv.w"], F[p + "vqkv.b"], ws.QKVv, L.EPI_BIAS)
        else:
            K.gemm(ws.Xn[:, :N1], F[p + "qkv.w"], F[p + "qkv.b"], ws.QKV, L.EPI_BIAS)
        fast = self._attn_fast(ws) if getattr(ws, "retry", None) is not None else False
        K.qk_layernorm_rope_pair(ws.QKV[:, :, :D], ws.QKV[:, :, D:2 * D], H, qn[0], qn[1], qn[2], qn[3], 1e-6, (Nt, rope), k_scale=kscale,
                                 kmax=ws.kmax1 if fast else None, kmax_ws=ws.kmax_ws if fast else None)
        if not fused_vt:
            K.transpose_v(ws.QKV[:, :, 2 * D:], H, 0, N1, ws.Vt1)
        if use_vip:
            if not paired and not fused_vt:
                K.gemm(ws.Xn, F[p + "vqkv.w"], F[p + "vqkv.b"], ws.QKVv, L.EPI_BIAS)
            vqn = F[p + "vqknorm"]
            K.qk_layernorm_rope_pair(ws.QKVv[:, :, :D], ws.QKVv[:, :, D:2 * D], H, vqn[0], vqn[1], vqn[2], vqn[3], 1e-6, (Nt, vrope),
                                     (N1, crope), k_scale=kscale, kmax=ws.kmax2 if fast else None, kmax_ws=ws.kmax_ws if fast else None)
            if not fused_vt:
                K.transpose_v(ws.QKVv[:, :, 2 * D:], H, 0, N, ws.Vt3)
            if vt2_view_ok and N1 + pad64(Np) <= ws.Vt3.shape[3]:
                # V^T of the vip keys = the tail columns of the all-keys image (16-B aligned start; its zero padding out to a multiple of
                # 64 keys is the image's own).  The kernel reads pad64(Np) keys from that offset: only when they stay inside the row
                # (not e.g. N1 % 64 = 8, Np % 64 = 8, where N1 + pad64(Np) > pad64(N) runs into the next d-row / past the allocation)
                vt2 = ws.Vt3[:, :, :, N1:]
            else:
                vt2 = K.transpose_v(ws.QKVv[:, :, 2 * D:], H, N1, Np, ws.Vt2)
            # attention_processor.py:2126-2131: `scale` becomes a tensor of the activations' dtype (0.6 -> bf16 0.6015625); a list as long as
            # the batch weighs every batch item with its own entry, any other list uses entry 0
            s = blk.attn1.processor.scale
            sl = [float(v) for v in s] if isinstance(s, (list, tuple)) else [float(s)]
            sl = [float(torch.tensor(v, dtype=torch.float32).to(BF16)) for v in sl]
            s_batch = sl if (len(sl) == B and len(set(sl)) > 1) else None
            s = sl[0]
            # text+video rows: softmax(q k^T) v  +  s * softmax(qx kv^T) vv   (attention_processor.py:2066-2069,2117-2134)
            # vip rows: qv against cat(kx, kv) / cat(vx, vv)                  (:2120-2125) — rides in the main launch's last round
            # kmax2 is the maximum over ALL rows of the vip-weight K projection: an upper bound for the vip keys of segment 2 as well
            km1, km2 = (ws.kmax1, ws.kmax2) if fast else (None, None)
            K.attention_multi(dict(q1=ws.QKV[:, :, :D], k1=ws.QKV[:, :, D:2 * D], vt1=ws.Vt1, nk1=N1, out=ws.AO[:, :N1],
                                   q2=ws.QKVv[:, :N1, :D], k2=ws.QKVv[:, N1:, D:2 * D], vt2=vt2, nk2=Np, seg2_scale=s, kmax1=km1, kmax2=km2, seg2_scale_batch=s_batch),
                              dict(q1=ws.QKVv[:, N1:, :D], k1=ws.QKVv[:, :, D:2 * D], vt1=ws.Vt3, nk1=N, out=ws.AO[:, N1:], kmax1=km2),
                              H, sm_scale, k_prescaled=True, retry=ws.retry if fast else None, split=getattr(ws.retry, "split", None))
        elif fast:
            K.attention_multi(dict(q1=ws.QKV[:, :, :D], k1=ws.QKV[:, :, D:2 * D], vt1=ws.Vt1, nk1=N1, out=ws.AO[:, :N1], kmax1=ws.kmax1), None,
                              H, sm_scale, k_prescaled=True, retry=ws.retry)
        else:
            K.attention(ws.QKV[:, :, :D], ws.QKV[:, :, D:2 * D], ws.Vt1, N1, ws.AO[:, :N1], H, sm_scale, k_prescaled=True)

    @torch.no_grad()
    def attention_op(self, i, hidden_states, encoder_hidden_states, image_rotary_emb=None, vip_image_rotary_emb=None,
                     vip_condition_rotary_emb=None):
        """The reference's operator-level seam for block i: what `blk.attn1(hidden_states, encoder_hidden_states=..., rotary tables)`
        returns there (Attention.forward -> processor.__call__, attention_processor.py:457-501, 1982-2155 / 1895-1953):
        (video rows [B,Nv,D], cat(text, vip) rows [B,Nt+Np,D]) AFTER `to_out`, from already normalised + modulated inputs."""
        dev = self._device
        B, Nv, D = hidden_states.shape
        Np = self.vip_length if self.transformer_blocks[i].use_vip else 0
        Nt = encoder_hidden_states.shape[1] - Np
        N1, N = Nt + Nv, Nt + Nv + Np
        H = self.config.num_attention_heads
        if image_rotary_emb is None or (Np and (vip_image_rotary_emb is None or vip_condition_rotary_emb is None)):
            raise ValueError("rotary tables are required (CogVideoX-5B attention)")
        e = lambda *sh: torch.empty(*sh, dtype=BF16, device=dev)
        ws = SimpleNamespace(Xn=e(B, N, D), QKV=e(B, N1, 3 * D), Vt1=e(B, H, 64, _pad_to(N1, 64)), AO=e(B, N, D))
        ws.kmax1 = torch.zeros(B, H, dtype=torch.float32, device=dev)
        ws.kmax2 = torch.zeros(B, H, dtype=torch.float32, device=dev)
        ws.kmax_ws, ws.retry = K.kmax_workspace(N, H, B, dev), K.AttnRetry(N1, Np, H, B, dev)
        ws.Xn[:, :Nt] = encoder_hidden_states[:, :Nt].to(dev, BF16)
        ws.Xn[:, Nt:N1] = hidden_states.to(dev, BF16)
        f32 = lambda t: tuple(x.to(dev, torch.float32).contiguous() for x in t)
        vr = cr = None
        if Np:
            ws.Xn[:, N1:] = encoder_hidden_states[:, Nt:].to(dev, BF16)
            ws.QKVv, ws.Vt2, ws.Vt3 = e(B, N, 3 * D), e(B, H, 64, _pad_to(Np, 64)), e(B, H, 64, _pad_to(N, 64))
            vr, cr = f32(vip_image_rotary_emb), f32(vip_condition_rotary_emb)
        self._attn_core(i, ws, B, Nt, Nv, Np, f32(image_rotary_emb), vr, cr)
        out = e(B, N, D)
        F = self._fused
        K.gemm(ws.AO, F[f"l{i}.out.w"], F[f"l{i}.out.b"], out, L.EPI_BIAS)
        return out[:, Nt:N1], torch.cat([out[:, :Nt], out[:, N1:]], dim=1)

    @torch.no_grad()
    def patch_embed_proj(self, latents):
        """`transformer.patch_embed.proj` as the pipeline calls it directly (pipeline_cogvideox_mp_fifo.py:596): Conv2d(k=2, s=2) on
        latents [b, f, C, h, w] -> patch tokens [b, f, (h/2)(w/2), D], as one patch gather + GEMM (embeddings.py:516-523)."""
        b, f, C, h, w = latents.shape
        F = self._fused
        ps = self.config.patch_size
        ntok = b * f * (h // ps) * (w // ps)
        cols = torch.zeros(ntok, F["patch.w"].shape[1], dtype=BF16, device=self._device)
        K.patchify(latents.to(self._device, BF16).reshape(b * f, C, h, w).contiguous(), cols, ps)
        out = torch.empty(ntok, self.inner_dim, dtype=BF16, device=self._device)
        K.gemm(cols, F["patch.w"], F["patch.b"], out, L.EPI_BIAS)
        return out.view(b, f, (h // ps) * (w // ps), self.inner_dim)

    # ------------------------------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, hidden_states, encoder_hidden_states, timestep, vip_encoder_hidden_states=None,
                timestep_cond=None, image_rotary_emb=None, vip_image_rotary_emb=None, vip_condition_rotary_emb=None,
                vip_grid_t=None, attention_kwargs=None, return_dict=True):
        """cogvideox_transformer_3d.py:636-770.  hidden_states [B,F,C,H,W] bf16 on the GPU."""
        if timestep_cond is not None:
            raise NotImplementedError("timestep_cond is unused by CogVideoX")
        if attention_kwargs and attention_kwargs.get("attention_masks") is not None:
            raise NotImplementedError("attention masks are not used on the TokensGen hot path")
        if image_rotary_emb is None:
            raise ValueError("image_rotary_emb is required (CogVideoX-5B uses rotary embeddings)")
        c, D, H = self.config, self.inner_dim, self.config.num_attention_heads
        F = self._fused
        B, Fr, C, Hh, Ww = hidden_states.shape
        ps = c.patch_size
        hw = (Hh // ps) * (Ww // ps)
        Nt, Nv = encoder_hidden_states.shape[1], Fr * hw
        use_vip = self.use_vip
        if use_vip:
            if vip_encoder_hidden_states is None or vip_image_rotary_emb is None or vip_condition_rotary_emb is None:
                raise ValueError("vip layers are set: vip_encoder_hidden_states and both vip rotary tables are required")
            vb, vf, vc, vh, vw = vip_encoder_hidden_states.shape
            Np = vf * vh * vw
            if Np != self.vip_length:
                raise ValueError(f"vip token count {Np} != configured length {self.vip_length}")
        else:
            Np = 0
        N1, N = Nt + Nv, Nt + Nv + Np
        timestep = torch.as_tensor(timestep, device=self._device)
        if timestep.ndim == 0:
            timestep = timestep.expand(B)
        Fm = timestep.shape[1] if timestep.ndim == 2 else 1
        if Fm not in (1, Fr):
            raise ValueError("timestep must be [B] or [B, num_frames]")
        ws = self._workspace(B, Nt, Nv, Np, Fm)
        dev = lambda t: t.to(self._device, torch.float32).contiguous()
        rope = tuple(dev(t) for t in image_rotary_emb)

        # 1. timestep embedding -> silu(emb) -> every modulation vector of every layer in one GEMM
        K.timestep_sinusoid(timestep.reshape(-1).to(torch.int64).contiguous(), D, ws.sin)
        K.gemm(ws.sin, F["t1.w"], F["t1.b"], ws.t1, L.EPI_BIAS_SILU)
        K.gemm(ws.t1, F["t2.w"], F["t2.b"], ws.semb, L.EPI_BIAS_SILU)     # block/out norms all consume silu(emb)
        K.gemm(ws.semb.view(B, Fm, -1), F["mod.w"], F["mod.b"], ws.mod, L.EPI_BIAS)

        # 2. patch / text / vip embeddings straight into the residual stream X = text | video | vip
        X = ws.X
        K.patchify(hidden_states.to(BF16).reshape(B * Fr, C, Hh, Ww).contiguous(), ws.patches.view(B * Nv, -1), ps)
        K.gemm(ws.patches, F["patch.w"], F["patch.b"], X[:, Nt:N1], L.EPI_BIAS)
        K.gemm(encoder_hidden_states.to(BF16).contiguous(), F["text.w"], F["text.b"], X[:, :Nt], L.EPI_BIAS)
        if use_vip:
            vtok = vip_encoder_hidden_states.to(BF16).permute(0, 1, 3, 4, 2).reshape(B, Np, vc).contiguous()
            K.gemm(vtok, F["vipproj.w"], F["vipproj.b"], X[:, N1:], L.EPI_BIAS)
            vrope = tuple(dev(t) for t in vip_image_rotary_emb)
            crope = tuple(dev(t) for t in vip_condition_rotary_emb)
        # 3. blocks
        for i in range(len(self.transformer_blocks)):
            self._run_block(i, ws, B, Nt, Nv, Np, Fm, rope, vrope if use_vip else None, crope if use_vip else None)

        # 4. final norm (per-token, so only the video rows matter), AdaLayerNorm(shift, scale), proj_out, unpatchify
        fl = F["final.ln"]
        vid, vidn, vid2 = X[:, Nt:N1], ws.Xn[:, Nt:N1], ws.AO[:, Nt:N1]
        K.adaln_modulate(vid, vidn, fl[0], fl[1], c.norm_eps, None)
        _, out_base, _ = self._mod_cols()
        tout = K.GroupTable(ws.mod, ws.tok_group, list(range(Fm)) + [0, 0], [out_base] * (Fm + 2), [out_base + D] * (Fm + 2),
                            [out_base] * (Fm + 2)).offset(Nt)
        K.adaln_modulate(vidn, vid2, fl[2], fl[3], c.norm_eps, tout)
        K.gemm(vid2, F["proj_out.w"], F["proj_out.b"], ws.po, L.EPI_BIAS)
        out = torch.empty(B, Fr, c.out_channels, Hh, Ww, dtype=BF16, device=self._device)
        K.unpatchify(ws.po.view(B * Nv, -1), out.view(B * Fr, c.out_channels, Hh, Ww), ps)
        if not return_dict:
            return (out,)
        return Transformer2DModelOutput(sample=out)
