"""3-D rotary position tables for the DiT (host side, fp32), uploaded once per window.

Same arithmetic as the reference's get_1d_rotary_pos_embed / get_3d_rotary_pos_embed(_v2)
(longvgen/models/embeddings.py:774-828, 571-707): per axis angle = pos * theta^(-2i/dim), cos/sin with every
frequency repeated twice (interleaved pairs), channels split t|h|w = d/4 | 3d/8 | 3d/8, tokens ordered (t,h,w).
The reference rebuilds these on the CPU for every window (cogvideo_sampling_mp_fifo.py:478-489); the tables are
a few MB; with a GPU device they are generated on the device by tg_rope_table_3d from the three position vectors.
"""
import numpy as np
import torch


def _axis(dim, pos, theta=10000.0):
    pos = torch.from_numpy(np.ascontiguousarray(np.asarray(pos)))
    inv = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
    ang = torch.outer(pos, inv)
    return ang.cos().repeat_interleave(2, dim=1).float(), ang.sin().repeat_interleave(2, dim=1).float()


_INV = {}


def _inv_freq(dim, device, theta=10000.0):
    key = (dim, str(device), theta)
    if key not in _INV:
        _INV[key] = (1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))).to(device)
    return _INV[key]


def rope_3d_device(head_dim, grid_t, grid_h, grid_w, dim_t, dim_h, dim_w, device):
    """The same tables built by tg_rope_table_3d on the GPU: only the three position vectors (a few dozen floats) cross PCIe, instead
    of two [T*H*W, 64] fp32 tables per window (the reference rebuilds + uploads them, fifo:478-489)."""
    from . import kernels as K
    from . import lib as L
    pos = [torch.from_numpy(np.ascontiguousarray(np.asarray(g, dtype=np.float32))).to(device) for g in (grid_t, grid_h, grid_w)]
    inv = [_inv_freq(d, device) for d in (dim_t, dim_h, dim_w)]
    n = len(grid_t) * len(grid_h) * len(grid_w)
    cos = torch.empty(n, dim_t + dim_h + dim_w, dtype=torch.float32, device=device)
    sin = torch.empty_like(cos)
    L.check(L.load().tg_rope_table_3d(pos[0].data_ptr(), len(grid_t), pos[1].data_ptr(), len(grid_h), pos[2].data_ptr(), len(grid_w),
                                      inv[0].data_ptr(), dim_t, inv[1].data_ptr(), dim_h, inv[2].data_ptr(), dim_w, cos.data_ptr(),
                                      sin.data_ptr(), K._stream()), "tg_rope_table_3d")
    return cos, sin


def rope_3d(head_dim, grid_t, grid_h, grid_w, dim_t=None, dim_h=None, dim_w=None, device=None):
    """(cos, sin), each [T*H*W, head_dim] fp32 — get_3d_rotary_pos_embed_v2 (embeddings.py:641-707).  With a GPU `device` the tables
    are generated there (rope_3d_device); without one, on the host."""
    dim_t = head_dim // 4 if dim_t is None else dim_t
    dim_h = head_dim // 8 * 3 if dim_h is None else dim_h
    dim_w = head_dim // 8 * 3 if dim_w is None else dim_w
    if device is not None and torch.device(device).type == "cuda":
        return rope_3d_device(head_dim, grid_t, grid_h, grid_w, dim_t, dim_h, dim_w, torch.device(device))
    T, H, W = len(grid_t), len(grid_h), len(grid_w)
    at, ah, aw = _axis(dim_t, grid_t), _axis(dim_h, grid_h), _axis(dim_w, grid_w)
    out = []
    for i in (0, 1):
        tab = torch.empty(T, H, W, dim_t + dim_h + dim_w, dtype=torch.float32)
        tab[..., :dim_t] = at[i][:, None, None, :]
        tab[..., dim_t:dim_t + dim_h] = ah[i][None, :, None, :]
        tab[..., dim_t + dim_h:] = aw[i][None, None, :, :]
        tab = tab.reshape(T * H * W, -1)
        out.append(tab.to(device) if device is not None else tab)
    return out[0], out[1]


def rope_3d_crop(head_dim, start, stop, grid_size, device=None):
    """get_3d_rotary_pos_embed (embeddings.py:571-639): grids = linspace(start, stop, n, endpoint=False) in fp32."""
    gt, gh, gw = (np.linspace(start[i], stop[i], grid_size[i], endpoint=False, dtype=np.float32) for i in range(3))
    return rope_3d(head_dim, gt, gh, gw, device=device)
