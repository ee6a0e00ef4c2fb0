#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: the transformer's share of one To2V training micro-step at the yaml's shapes (cogvideo_5b_vaevip_4x8x12_to2v.yaml:
per_gpu_batch_size 2, 13 latent frames of 60 x 90, 226 text tokens, 480 vip tokens) — forward with per-block checkpointing, loss, backward with
recompute, gradient accumulation, and (every --accum micro-steps) clip + AdamW.  Usage: python tools/bench_train.py [--layers 42] [--micro 2]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tokensgen_amd import optim, rope as R, train  # noqa: E402
from tokensgen_amd.scheduler import CogVideoXDPMScheduler  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=42)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--micro", type=int, default=2, help="timed micro-steps")
    ap.add_argument("--accum", type=int, default=9)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    bf = torch.bfloat16
    model = bench.build_model(dev, a.layers)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    tr = train.To2VTrainer(sd, 48, a.layers, patch_size=2, vip_scale=1.0)
    arena = optim.ParamArena({k: sd[k] for k in tr.trainable}, optim.arena_order(tr.trainable, a.layers), dev)
    tr.use_arena(arena)
    opt = optim.AdamW(arena, lr=2e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-4, max_grad_norm=1.0)
    sched = CogVideoXDPMScheduler(prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=1.0, timestep_spacing="trailing")
    step = train.To2VTrainStep(tr, arena, opt, torch.as_tensor(np.asarray(sched.alphas_cumprod), dtype=torch.float32), accumulation_steps=a.accum)
    g = torch.Generator(device=dev).manual_seed(7)
    B, nf, C, H, W = a.batch, 13, 16, 60, 90
    x0, noise = (torch.randn(B, nf, C, H, W, generator=g, device=dev, dtype=torch.float32).to(bf) for _ in range(2))
    text = (torch.randn(B, 226, 4096, generator=g, device=dev, dtype=torch.float32) * 0.1).to(bf)
    vip = torch.nn.functional.layer_norm(torch.randn(B, 5, 8, 12, 3072, generator=g, device=dev), (3072,)).permute(0, 1, 4, 2, 3).to(bf).contiguous()
    f32 = np.float32
    rope = R.rope_3d_crop(64, (0, 0, 0), (nf, 30, 45), (nf, 30, 45))
    vrope = rope
    crope = R.rope_3d(64, np.linspace(1000, 1016.25, 5, dtype=f32), np.linspace(0, 30, 8, endpoint=False, dtype=f32), np.linspace(0, 45, 12, endpoint=False, dtype=f32))
    ts = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(3))
    loss, _ = step.micro_step(x0, noise, ts, text, vip, rope, vrope, crope)          # warm-up
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.micro):
        loss, _ = step.micro_step(x0, noise, ts, text, vip, rope, vrope, crope)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / a.micro
    t0 = time.time()
    opt.step()
    torch.cuda.synchronize()
    t_opt = time.time() - t0
    print(json.dumps({"bench": "to2v_train_micro_step", "layers": a.layers, "batch": B, "s_per_micro_step": dt, "samples_per_s": B / dt, "optimizer_step_s": t_opt,
                      "trainable_params": int(sum(arena.views[n].numel() for n in arena.names)), "loss": float(loss),
                      "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}))


if __name__ == "__main__":
    main()
