#!/usr/bin/env python3
"""Reads what a `pplag` library variant (tools/patches/pp_lag.py) leaves in the exchange counters' unused words: the distance between consecutive key blocks of a head,
in tile times, and the number of exchange checks that had to poll.  GPU box only; TG_LIB_PATH must point at the variant."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tokensgen_amd import kernels as K, lib as L  # noqa: E402
assert "TG_LIB_PATH" in os.environ
B, H, D, N1 = 2, 48, 3072, 17776
rnd = lambda *sh, scale=1.0: (torch.randn(*sh, device="cuda") * scale).to(torch.bfloat16)
qkv = rnd(B, N1, 3 * D, scale=0.6)
o, do = rnd(B, N1, D, scale=0.3), rnd(B, N1, D, scale=0.3)
dq, dk, dv = (torch.empty(B, N1, D, dtype=torch.float32, device="cuda") for _ in range(3))
K.attention_bwd(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], o, do, H, 0.125, dq=dq, dk=dk, dv=dv)          # warm-up (and the device probe)
pr, outs, (ws, _) = K._bwd_problem(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], o, do, H, 0.125, dq=dq, dk=dk, dv=dv)
arr = (L.AttnBwdProblem * 1)(pr)
st = K.BwdDeviceState.get(qkv.device)
L.check(L.load().tg_attention_bwd_multi(arr, 1, H, B, st.flags(), K._p(st.status), K._stream()), "tg_attention_bwd_multi")
torch.cuda.synchronize()
nrow, ntile, nblk = B * H * N1, (N1 + 31) // 32, (N1 + 255) // 256
off = (6 * nrow + 8 + 3) & ~3
cnt = ws[off:off + B * H * ntile * 32].view(torch.int32).view(B * H, ntile, 32)[:, :nblk].cpu()
t100 = cnt[:, :, 16:18].contiguous().view(torch.int64)[..., 0].double()
t300 = cnt[:, :, 18:20].contiguous().view(torch.int64)[..., 0].double()
polls = cnt[:, :, 20]
tile = ((t300 - t100) / 200.0)
for hb in (0, 8, 40, 88):
    lag = (t300[hb, 1:] - t300[hb, :-1]) / tile[hb, 1:]
    print(json.dumps({"head_batch": hb, "ticks_per_tile_mean": round(float(tile[hb].mean()), 1), "lag_tiles_between_consecutive_key_blocks": [round(float(x), 1) for x in lag],
                      "polls_per_block": polls[hb].tolist()}))
lag = (t300[:, 1:] - t300[:, :-1]) / tile[:, 1:]
inner = torch.cat([lag[:, 0:31], lag[:, 32:63], lag[:, 64:]], dim=1)
print(json.dumps({"all_heads": True, "lag_tiles_median": round(float(lag.median()), 2), "lag_tiles_mean_within_a_round_of_32": round(float(inner.mean()), 2),
                  "polls_per_block_mean": round(float(polls.double().mean()), 1), "ticks_per_tile_mean": round(float(tile.mean()), 1)}))
