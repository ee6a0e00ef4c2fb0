#!/usr/bin/env python3
"""Socket power and shader clock (bench.PowerSampler: sysfs hwmon) while the main attention-backward call loops for a few seconds; TG_BENCH_ZERO=1: all-zero operands.
Prints one JSON line: ms per call, power, clock.  Same-box A/B of library variants through TG_LIB_PATH / TG_ATTN_BWD_PP."""
import importlib.util, json, math, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from tokensgen_amd import kernels as K  # noqa: E402
B, H, D, N1 = 2, 48, 3072, 17776
zero = os.environ.get("TG_BENCH_ZERO") == "1"
rnd = lambda *sh, scale=1.0: torch.zeros(*sh, device="cuda", dtype=torch.bfloat16) if zero else (torch.randn(*sh, device="cuda") * scale).to(torch.bfloat16)
qkv = rnd(B, N1, 3 * D, scale=0.6)
o, do = rnd(B, N1, D, scale=0.3), rnd(B, N1, D, scale=0.3)
dq, dk, dv = (torch.empty(B, N1, D, dtype=torch.float32, device="cuda") for _ in range(3))
kk = (qkv[:, :, D:2 * D].float() * (0.125 * 1.4426950408889634)).to(torch.bfloat16)
from tokensgen_amd import lib as L
pad = (N1 + 63) // 64 * 64
vt = K.transpose_v(qkv[:, :, 2 * D:], H, 0, N1, torch.zeros(B, H, 64, pad, dtype=torch.bfloat16, device="cuda"))
out = torch.empty(B, N1, D, dtype=torch.bfloat16, device="cuda")
_, lse = K.attention_lse(qkv[:, :, :D], kk, vt, N1, out, H, 0.125, k_prescaled=True)
fn = lambda: K.attention_bwd(qkv[:, :, :D], kk, qkv[:, :, 2 * D:], out, do, H, math.log(2.0), dq=dq, dk=dk, dv=dv, lse=lse)
for _ in range(3):
    fn()
torch.cuda.synchronize()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
ps = bench.PowerSampler(0, period_s=0.05)
with ps:
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(10):
            fn()
        torch.cuda.synchronize(); n += 10
    dt = time.perf_counter() - t0
print(json.dumps({"what": "attention_bwd main call (forward's lse given: statistics launch = D only)", "zero_operands": zero, "lib": os.path.basename(os.environ.get("TG_LIB_PATH", "product")),
                  "pp": os.environ.get("TG_ATTN_BWD_PP", "1"), "ms_per_call": 1e3 * dt / n, "tflops_5gemm": 5 * 2.0 * B * N1 * N1 * D / (dt / n) / 1e12, **ps.summary()}))
