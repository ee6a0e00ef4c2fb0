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
