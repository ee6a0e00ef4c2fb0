"""GPU: the reference-side ctypes stub printed in INTEGRATION.md section B is executed as written (only the library path is
resolved to the in-tree .so) and checked against torch's softmax attention — the documentation cannot rot unnoticed."""
import os
import re

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_ctypes_stub_runs():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "def tg_sdpa_pair" in b)
    stub = stub.replace('ctypes.CDLL("libtokensgen_hip.so")', f'ctypes.CDLL("{os.path.join(ROOT, "tokensgen_amd", "libtokensgen_hip.so")}")')
    ns = {}
    exec(compile(stub, "INTEGRATION.md:stub", "exec"), ns)
    B, N, H, NP = 2, 300, 4, 40
    g = torch.Generator().manual_seed(0)
    mk = lambda n: torch.randn(B, n, H * 64, generator=g).to("cuda", torch.bfloat16)
    q, k, v, q2, k2, v2 = mk(N), mk(N), mk(N), mk(N), mk(NP), mk(NP)
    out = ns["tg_sdpa_pair"](q, k, v, q2, k2, v2, 0.6)
    torch.cuda.synchronize()

    def ref(q, k, v):
        qh, kh, vh = (t.float().view(B, t.shape[1], H, 64).transpose(1, 2) for t in (q, k, v))
        return (torch.softmax(qh @ kh.transpose(-1, -2) / 8.0, -1) @ vh).transpose(1, 2).reshape(B, q.shape[1], H * 64)
    want = ref(q, k, v) + 0.6 * ref(q2, k2, v2)
    rel = ((out.float() - want).norm() / want.norm()).item()
    assert out.shape == want.shape and rel < 8e-3, rel


def test_integration_md_upsample_stub_runs():
    """The second stub of section B (CogVideoXUpsample3D as tg_conv3d_up2_subpixel), executed as written on top of the first one's `_tg`, against
    F.interpolate(nearest x2) + conv2d in fp32 — spatial only and with the time doubling of `compress_time`."""
    import torch.nn.functional as F
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    first = next(b for b in blocks if "def tg_sdpa_pair" in b)
    first = first.replace('ctypes.CDLL("libtokensgen_hip.so")', f'ctypes.CDLL("{os.path.join(ROOT, "tokensgen_amd", "libtokensgen_hip.so")}")')
    second = next(b for b in blocks if "def tg_upsample3d" in b)
    ns = {}
    exec(compile(first, "INTEGRATION.md:stub", "exec"), ns)
    exec(compile(second, "INTEGRATION.md:stub2", "exec"), ns)
    g = torch.Generator().manual_seed(1)
    T, H, W, C = 4, 40, 56, 256
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.03).to("cuda", torch.bfloat16)
    b = torch.randn(C, generator=g).to("cuda", torch.bfloat16)
    x = torch.randn(T, H, W, C, generator=g).to("cuda", torch.bfloat16)
    zeros = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    ph = ns["pack_phases"](w)
    for compress in (False, True):
        y = ns["tg_upsample3d"](x, ph, b, zeros, compress)
        torch.cuda.synchronize()
        xn = x.float().permute(0, 3, 1, 2)                                  # [T, C, H, W]
        want = F.conv2d(F.interpolate(xn, scale_factor=2.0, mode="nearest"), w.float(), b.float(), padding=1).permute(0, 2, 3, 1)
        if compress:
            want = want[torch.tensor([t for t in range(T) for _ in (0, 1)], device="cuda")]
        rel = ((y.float() - want).norm() / want.norm()).item()
        assert y.shape == want.shape and rel < 5e-3, (compress, rel)
