"""Throw-away import shim used ONLY in the build container to import the reference
(`/root/reference/longvgen`, Python) on CPU so that golden fixtures can be generated
(tools/make_golden.py).  It never travels into the product or the tests.

The reference needs `diffusers==0.31.0.dev0` and `xformers`, neither of which is installed
here.  `install()` registers permissive placeholder modules for every `diffusers.*` /
`xformers.*` import and fills in the handful of names that carry behaviour the reference
relies on.  Pieces of *arithmetic* that live in diffusers (not under /root/reference) are
restated here from the published upstream semantics and are therefore "parity unpinned":
    FeedForward / GELU(tanh), get_activation, randn_tensor,
    CogVideoXDownsample3D, CogVideoXUpsample3D, DiagonalGaussianDistribution.
"""
import functools
import importlib.abc
import importlib.machinery
import inspect
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = "/root/reference"


# ----------------------------------------------------------------------------------------------
# behaviour-carrying stand-ins
# ----------------------------------------------------------------------------------------------
class _Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def register_to_config(init):
    @functools.wraps(init)
    def wrapper(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        self._internal_cfg = _Cfg(cfg)
        init(self, *args, **kwargs)
    return wrapper


class ConfigMixin:
    @property
    def config(self):
        return self._internal_cfg

    def register_to_config(self, **kw):
        if not hasattr(self, "_internal_cfg"):
            self._internal_cfg = _Cfg()
        self._internal_cfg.update(kw)


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


class SchedulerMixin:
    pass


class PeftAdapterMixin:
    pass


class FromOriginalModelMixin:
    pass


class BaseOutput(dict):
    def __post_init__(self):
        for k, v in self.__dict__.items():
            self[k] = v


class Transformer2DModelOutput:
    def __init__(self, sample):
        self.sample = sample


class AutoencoderKLOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class DecoderOutput:
    def __init__(self, sample, commit_loss=None):
        self.sample = sample


class _Logger:
    def __getattr__(self, name):
        return lambda *a, **k: None


class _logging:
    @staticmethod
    def get_logger(name=None):
        return _Logger()


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    return torch.randn(shape, generator=generator, device=device, dtype=dtype)


def get_activation(name):
    name = name.lower()
    return {"silu": nn.SiLU, "swish": nn.SiLU, "mish": nn.Mish, "gelu": nn.GELU, "relu": nn.ReLU}[name]()


class FP32SiLU(nn.Module):
    def forward(self, x):
        return F.silu(x.float()).to(x.dtype)


class GELU(nn.Module):
    """diffusers.models.activations.GELU (restated): Linear then F.gelu(approximate=...)."""

    def __init__(self, dim_in, dim_out, approximate="none", bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate

    def forward(self, x):
        return F.gelu(self.proj(x), approximate=self.approximate)


class FeedForward(nn.Module):
    """diffusers.models.attention.FeedForward (restated): net = [act(proj), Dropout, Linear, (Dropout)]."""

    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu",
                 final_dropout=False, inner_dim=None, bias=True):
        super().__init__()
        inner_dim = int(dim * mult) if inner_dim is None else inner_dim
        dim_out = dim if dim_out is None else dim_out
        if activation_fn == "gelu":
            act = GELU(dim, inner_dim, bias=bias)
        elif activation_fn == "gelu-approximate":
            act = GELU(dim, inner_dim, approximate="tanh", bias=bias)
        else:
            raise NotImplementedError(activation_fn)
        self.net = nn.ModuleList([act, nn.Dropout(dropout), nn.Linear(inner_dim, dim_out, bias=bias)])
        if final_dropout:
            self.net.append(nn.Dropout(dropout))

    def forward(self, x, *a, **k):
        for m in self.net:
            x = m(x)
        return x


class CogVideoXDownsample3D(nn.Module):
    """diffusers.models.downsampling.CogVideoXDownsample3D (restated from upstream 0.31 semantics)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=2, padding=0, compress_time=False):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding)
        self.compress_time = compress_time

    def forward(self, x):
        if self.compress_time:
            b, c, f, h, w = x.shape
            x = x.permute(0, 3, 4, 1, 2).reshape(b * h * w, c, f)
            if x.shape[-1] % 2 == 1:
                x_first, x_rest = x[..., 0], x[..., 1:]
                if x_rest.shape[-1] > 0:
                    x_rest = F.avg_pool1d(x_rest, kernel_size=2, stride=2)
                x = torch.cat([x_first[..., None], x_rest], dim=-1)
            else:
                x = F.avg_pool1d(x, kernel_size=2, stride=2)
            x = x.reshape(b, h, w, c, x.shape[-1]).permute(0, 3, 4, 1, 2)
        x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
        b, c, f, h, w = x.shape
        x = x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
        x = self.conv(x)
        x = x.reshape(b, f, x.shape[1], x.shape[2], x.shape[3]).permute(0, 2, 1, 3, 4)
        return x


class CogVideoXUpsample3D(nn.Module):
    """diffusers.models.upsampling.CogVideoXUpsample3D (restated from upstream 0.31 semantics)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, compress_time=False):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding)
        self.compress_time = compress_time

    def forward(self, inputs):
        if self.compress_time:
            if inputs.shape[2] > 1 and inputs.shape[2] % 2 == 1:
                x_first, x_rest = inputs[:, :, 0], inputs[:, :, 1:]
                x_first = F.interpolate(x_first, scale_factor=2.0)
                x_rest = F.interpolate(x_rest, scale_factor=2.0)
                inputs = torch.cat([x_first[:, :, None, :, :], x_rest], dim=2)
            elif inputs.shape[2] > 1:
                inputs = F.interpolate(inputs, scale_factor=2.0)
            else:
                inputs = inputs.squeeze(2)
                inputs = F.interpolate(inputs, scale_factor=2.0)
                inputs = inputs[:, :, None, :, :]
        else:
            b, c, t, h, w = inputs.shape
            inputs = inputs.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
            inputs = F.interpolate(inputs, scale_factor=2.0)
            inputs = inputs.reshape(b, t, c, *inputs.shape[2:]).permute(0, 2, 1, 3, 4)
        b, c, t, h, w = inputs.shape
        inputs = inputs.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
        inputs = self.conv(inputs)
        inputs = inputs.reshape(b, t, *inputs.shape[1:]).permute(0, 2, 1, 3, 4)
        return inputs


class DiagonalGaussianDistribution:
    """diffusers.models.autoencoders.vae.DiagonalGaussianDistribution (restated)."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator=None):
        noise = randn_tensor(self.mean.shape, generator=generator, device=self.parameters.device,
                             dtype=self.parameters.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


def _identity_decorator(fn=None, *a, **k):
    return fn


def is_torch_version(op, ver):
    return True


_EXPLICIT = {
    "ConfigMixin": ConfigMixin, "register_to_config": register_to_config, "ModelMixin": ModelMixin,
    "SchedulerMixin": SchedulerMixin, "PeftAdapterMixin": PeftAdapterMixin,
    "FromOriginalModelMixin": FromOriginalModelMixin, "BaseOutput": BaseOutput,
    "Transformer2DModelOutput": Transformer2DModelOutput, "AutoencoderKLOutput": AutoencoderKLOutput,
    "DecoderOutput": DecoderOutput, "logging": _logging, "randn_tensor": randn_tensor,
    "get_activation": get_activation, "FP32SiLU": FP32SiLU, "GELU": GELU, "FeedForward": FeedForward,
    "CogVideoXDownsample3D": CogVideoXDownsample3D, "CogVideoXUpsample3D": CogVideoXUpsample3D,
    "DiagonalGaussianDistribution": DiagonalGaussianDistribution,
    "maybe_allow_in_graph": _identity_decorator, "apply_forward_hook": _identity_decorator,
    "is_torch_version": is_torch_version, "USE_PEFT_BACKEND": False,
    "scale_lora_layers": lambda *a, **k: None, "unscale_lora_layers": lambda *a, **k: None,
    "deprecate": lambda *a, **k: None, "is_torch_npu_available": lambda: False,
    "is_xformers_available": lambda: False, "check_min_version": lambda *a, **k: None,
    "__version__": "0.0.28",
    "replace_example_docstring": lambda doc: (lambda fn: fn),          # decorator factory on the pipelines' __call__
}


class _PlaceholderMeta(type):
    def __iter__(cls):
        return iter(())


class _Placeholder(metaclass=_PlaceholderMeta):
    """Anything else imported from diffusers/xformers: an inert class usable as a base or enum."""
    def __init__(self, *a, **k):
        pass


class _ShimModule(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__") and name not in _EXPLICIT:
            raise AttributeError(name)
        if name in _EXPLICIT:
            return _EXPLICIT[name]
        return _PlaceholderMeta(name, (_Placeholder,), {})


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("diffusers", "xformers")

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _ShimModule(spec.name)

    def exec_module(self, module):
        pass


_installed = False


def install():
    """Make `import longvgen...` work against /root/reference on CPU."""
    global _installed
    if _installed:
        return
    sys.meta_path.insert(0, _Finder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True
