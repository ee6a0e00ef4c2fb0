"""Optimizer side of the To2V training step (train_cogvideo_to2v.py:1083-1098 optimizer, :1157-1164 DDP, :1726 accumulate,
:2012-2021 clip / step / zero_grad).

MI355X-first layout: every trainable parameter lives in ONE flat bf16 arena (the state-dict entries the kernels read are views of it), with one
flat fp32 arena each for the accumulated gradient and the two AdamW moments.  The optimizer step is then three streaming launches over the arena
(sum of squares -> clip coefficient on the device -> AdamW + zero_grad), and the data-parallel gradient exchange is a handful of large RCCL
all-reduces over slices of the same buffer (xGMI rings are per-link bound: few big buckets, once per `gradient_accumulation_steps` micro-steps —
the reference's `no_sync` for the other eight).  288 GB of HBM is why the moments are fp32 instead of bitsandbytes' 8-bit blocks."""
import torch

from . import kernels as K
from . import lib as L

BF16 = torch.bfloat16
_ALIGN = 64          # elements: keeps every view 128-byte aligned (the GEMM wants 16 B, the streaming kernels like full lines)


def arena_order(names, num_layers):
    """Arena order = the order gradients become final in the backward (last block first, embeddings last, Resampler after the transformer), so
    that a bucket can be handed to RCCL as soon as the backward has passed its end.  Inside a block vip_to_{q,k,v} weights (and biases) are
    adjacent: the fused [3D, D] projection weight is then a view of the arena, not a copy."""
    def block_key(n):
        for j, pat in enumerate(("vip_to_q.weight", "vip_to_k.weight", "vip_to_v.weight", "vip_to_q.bias", "vip_to_k.bias", "vip_to_v.bias")):
            if n.endswith(pat):
                return (0, j, n)
        return (1, 0, n)
    out = []
    for i in reversed(range(num_layers)):
        pre = f"transformer_blocks.{i}."
        out += sorted((n for n in names if n.startswith(pre)), key=block_key)
    rest = [n for n in names if not n.startswith("transformer_blocks.")]
    out += sorted(n for n in rest if not n.startswith("resampler."))
    out += sorted(n for n in rest if n.startswith("resampler."))
    assert sorted(out) == sorted(names)
    return out


class ParamArena:
    """Flat storage for the trainable parameters.  `params`: {name: tensor}; `order`: names in arena order.  After construction `views[name]` is a
    bf16 view of the arena holding the parameter (install these in the state dict the kernels use)."""

    def __init__(self, params, order, device):
        self.names = list(order)
        self.offsets, self.shapes = {}, {}
        off = 0
        for n in self.names:
            self.offsets[n], self.shapes[n] = off, tuple(params[n].shape)
            off += (params[n].numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        self.param = torch.zeros(off, dtype=BF16, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=device)
        self.views = {}
        for n in self.names:
            v = self.param[self.offsets[n]: self.offsets[n] + params[n].numel()].view(self.shapes[n])
            v.copy_(params[n])
            self.views[n] = v

    def end_of(self, name):
        n = self.offsets[name]
        k = 1
        for s in self.shapes[name]:
            k *= s
        return n + (k + _ALIGN - 1) // _ALIGN * _ALIGN

    def prefix_elems(self, pred):
        """Number of leading arena elements whose names satisfy pred (they must form a prefix of the order)."""
        end, seen_other = 0, False
        for n in self.names:
            if pred(n):
                assert not seen_other, "names selected by pred must be a prefix of the arena order"
                end = self.end_of(n)
            else:
                seen_other = True
        return end

    def grad_view(self, name):
        o = self.offsets[name]
        k = self.views[name].numel()
        return self.grad[o:o + k].view(self.shapes[name])

    @torch.no_grad()
    def accumulate(self, grads, scale=1.0):
        """grad arena += scale * grads[name] for every entry (bf16 or fp32 tensors shaped like the parameter) — one launch per L.TG_ACCUM_MAX entries."""
        if not grads:
            return
        lib = L.load()
        items = (L.AccumItem * len(grads))()
        keep = []                                          # the contiguous copies must outlive the launch call
        for i, (n, g) in enumerate(grads.items()):
            if tuple(g.shape) != self.shapes[n]:
                raise ValueError(f"gradient of {n}: shape {tuple(g.shape)} != parameter shape {self.shapes[n]}")
            g = g.contiguous()
            if g.dtype not in (BF16, torch.float32):
                raise TypeError(f"gradient of {n}: dtype {g.dtype}")
            keep.append(g)
            items[i].grad, items[i].acc, items[i].n, items[i].grad_is_bf16 = g.data_ptr(), self.grad.data_ptr() + 4 * self.offsets[n], g.numel(), 1 if g.dtype == BF16 else 0
        L.check(lib.tg_grad_accumulate_multi(items, len(grads), float(scale), K._stream()), "tg_grad_accumulate_multi")

    def state_dict(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}

    def layout(self):
        """(name, offset, shape) of every parameter in the flat arenas — stored with a checkpoint so that a resume can check it is loading
        moments of the same geometry."""
        return [(n, int(self.offsets[n]), tuple(self.shapes[n])) for n in self.names]


class AdamW:
    """torch.optim.AdamW semantics (decoupled weight decay) on a ParamArena, with the reference's gradient clipping folded in:
    `clip_elems` leading arena elements (the transformer's parameters, train_cogvideo_to2v.py:2014-2015) are clipped to `max_grad_norm` by their
    global L2 norm; the rest (the Resampler) is stepped unclipped, as in the reference."""

    def __init__(self, arena, lr=2e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-4, max_grad_norm=1.0, clip_elems=None):
        self.arena, self.lr, self.betas, self.eps, self.wd, self.max_norm = arena, lr, betas, eps, weight_decay, max_grad_norm
        self.clip_elems = arena.numel if clip_elems is None else int(clip_elems)
        self.t = 0
        dev = arena.param.device
        self._ws = torch.empty(L.load().tg_grad_norm_ws_floats(), dtype=torch.float32, device=dev)
        self.coef = torch.ones(2, dtype=torch.float32, device=dev)        # [total norm, clip coefficient] of the last step (device side)

    @torch.no_grad()
    def step(self, lr=None, zero_grad=True):
        a, lib = self.arena, L.load()
        self.t += 1
        lr = self.lr if lr is None else lr
        st = K._stream()
        clip_ptr = None
        nc = self.clip_elems
        if self.max_norm is not None and self.max_norm > 0 and nc > 0:
            L.check(lib.tg_grad_clip_coef(a.grad.data_ptr(), nc, float(self.max_norm), self._ws.data_ptr(), self.coef.data_ptr(), st), "tg_grad_clip_coef")
            clip_ptr = self.coef.data_ptr() + 4
        for lo, hi, cp in ((0, nc, clip_ptr), (nc, a.numel, None)):
            if hi > lo:
                L.check(lib.tg_adamw_step(a.param.data_ptr() + 2 * lo, a.grad.data_ptr() + 4 * lo, a.exp_avg.data_ptr() + 4 * lo, a.exp_avg_sq.data_ptr() + 4 * lo,
                                          hi - lo, self.t, float(lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd), cp,
                                          1 if zero_grad else 0, st), "tg_adamw_step")


    # ---- checkpoint / resume (the reference: accelerator.save_state / load_state behind --resume_from_checkpoint, train_cogvideo_to2v.py:1690-1716,
    # 2030-2047).  Layout of the dict: {"t": optimizer steps taken (bias correction), "exp_avg" / "exp_avg_sq": the flat fp32 moment arenas,
    # "grad": the flat fp32 gradient arena (non-zero only in the middle of an accumulation window), "layout": ParamArena.layout(),
    # "hyper": lr / betas / eps / weight_decay / max_grad_norm / clip_elems}.  Parameters are saved separately under the reference's names
    # (CogVideoXTransformer3DModel.save_vip_layers -> vip.pt, Resampler.save_pretrained).
    def state_dict(self):
        a = self.arena
        return {"t": int(self.t), "exp_avg": a.exp_avg.detach().cpu().clone(), "exp_avg_sq": a.exp_avg_sq.detach().cpu().clone(),
                "grad": a.grad.detach().cpu().clone(), "layout": a.layout(),
                "hyper": {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "max_grad_norm": self.max_norm,
                          "clip_elems": self.clip_elems}}

    @torch.no_grad()
    def load_state_dict(self, sd):
        a = self.arena
        if [(n, int(o), tuple(sh)) for n, o, sh in sd["layout"]] != a.layout():
            raise ValueError("AdamW.load_state_dict: the checkpoint's arena layout (names / offsets / shapes) differs from this arena's")
        for name in ("exp_avg", "exp_avg_sq", "grad"):
            t = sd[name]
            if t.numel() != a.numel or t.dtype != torch.float32:
                raise ValueError(f"AdamW.load_state_dict: {name} has {t.numel()} {t.dtype} elements, arena has {a.numel} fp32")
            getattr(a, name).copy_(t.to(a.param.device))
        self.t = int(sd["t"])
        h = sd.get("hyper", {})
        self.lr, self.betas, self.eps = h.get("lr", self.lr), tuple(h.get("betas", self.betas)), h.get("eps", self.eps)
        self.wd, self.max_norm, self.clip_elems = h.get("weight_decay", self.wd), h.get("max_grad_norm", self.max_norm), int(h.get("clip_elems", self.clip_elems))


def constant_with_warmup(step, base_lr, warmup_steps):
    """diffusers get_scheduler("constant") ignores warm-up; "constant_with_warmup" ramps linearly (optimization.py).  The yaml uses "constant"."""
    return base_lr if warmup_steps <= 0 else base_lr * min(1.0, step / float(warmup_steps))


class GradSync:
    """Data-parallel gradient exchange (accelerate DDP, train_cogvideo_to2v.py:1157-1164): SUM all-reduce of the flat gradient in a few large
    buckets; averaging is folded into the accumulation scale (1 / (accumulation_steps * world_size)).  `ready(end)` may be called during the
    backward of the LAST micro-step of an accumulation window: every bucket that lies entirely below `end` (arena order = backward order) is
    handed to the collective asynchronously, so the exchange overlaps the remaining blocks' backward; `finish()` launches what is left and waits.
    Works on any flat tensor (gloo on CPU in the tests, RCCL on the GPU)."""

    def __init__(self, flat, group=None, bucket_elems=64 * 1024 * 1024):
        import torch.distributed as dist
        self.dist, self.flat, self.group = dist, flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = flat.numel()
        self.bounds = [(lo, min(n, lo + bucket_elems)) for lo in range(0, n, bucket_elems)]
        self._next, self._work = 0, []

    def ready(self, end):
        if not self.dist.is_initialized():
            return                                   # single process without a process group: nothing to exchange
        while self._next < len(self.bounds) and self.bounds[self._next][1] <= end:
            lo, hi = self.bounds[self._next]
            self._work.append(self.dist.all_reduce(self.flat[lo:hi], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
            self._next += 1

    def finish(self):
        self.ready(self.flat.numel())
        for w in self._work:
            w.wait()
        self._next, self._work = 0, []
