#!/usr/bin/env python3
"""Headline benchmark: DiT denoising steps/sec, CogVideoX-5B To2V 720x480 (BASELINE.json configs[1] shape).

One "step" = the body of the reference's per-GPU FIFO worker (cogvideo_sampling_mp_fifo.py:491-550): a
CFG-batched (B=2) DiT forward over one 13-latent-frame window (226 text + 17 550 video + 480 condensed tokens,
42 layers, D=3072) + CFG combine + 13 per-frame DPM-solver++ updates.  Synthetic latents / embeddings and
random-init weights of the real architecture (no checkpoints offline).  Inputs are resident in HBM when the
timed region starts.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): every rank runs its own window of the same
FIFO iteration (weak scaling, windows are independent) and the kept half-windows (7 latent frames + 7 x0
frames per rank) are exchanged with ONE all_gather per step — the path's real exchange.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_STEP = 776.0e12          # 42 x 9.237 TFLOP/block/sample x 2 + embed/out
PEAK_BF16 = 2.5e15                # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
# dominant kernel = fused main attention (SDPA#1 + SDPA#2 of the To2V processor), per launch (B=2):
N1, NP, D_MODEL = 17776, 480, 3072
# + the vip-query attention (SDPA#3), whose workgroups ride in the same launch (tg_attention_fwd_multi)
ATTN_FLOP_PER_LAUNCH = 2 * (4.0 * N1 * N1 * D_MODEL + 4.0 * N1 * NP * D_MODEL + 4.0 * NP * (N1 + NP) * D_MODEL)


def build_model(device, layers):
    from tokensgen_amd.transformer import CogVideoXTransformer3DModel
    m = CogVideoXTransformer3DModel(num_attention_heads=48, attention_head_dim=64, num_layers=layers, time_embed_dim=512,
                                    text_embed_dim=4096, use_rotary_positional_embeddings=True, device=device)
    m.set_vip_layers(None, length=480, func_type="1", scale=[0.6],
                     resampler_params=dict(output_dim=3072, num_height_queries=8, num_width_queries=12, num_temporal_queries=4))
    g = torch.Generator(device=device).manual_seed(1234)
    for name, t in m._fused.items():       # random init at the real shapes, straight into the fused storages
        if t.dim() == 2 and t.shape[0] > 8:
            t.copy_(torch.randn(t.shape, generator=g, device=device, dtype=torch.float32) * 0.02)
        elif name.endswith(("ln", "qknorm", "vln", "vqknorm")):
            t.copy_(torch.randn(t.shape, generator=g, device=device, dtype=torch.float32) * 0.05)
            t[0::2] += 1.0
        else:
            t.copy_(torch.randn(t.shape, generator=g, device=device, dtype=torch.float32) * 0.02)
    return m


def cpu_baseline(seconds_budget=40.0):
    """The oracle (CPU restatement of the reference, validated against it) timed on this box's host cores on a
    bounded sample of the same workload: ONE full-width block forward (B=1) of the 84 a step needs."""
    from oracle import dit_ref as O
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    cfg = dict(num_attention_heads=48, attention_head_dim=64, num_layers=1, time_embed_dim=512)
    sd = {k: v.to(torch.bfloat16) for k, v in O.make_state_dict(cfg, n_vip_dim=3072, seed=500).items()
          if k.startswith("transformer_blocks.0.")}
    g = torch.Generator().manual_seed(501)
    hid = torch.randn(1, 17550, 3072, generator=g).bfloat16()
    enc = torch.randn(1, 706, 3072, generator=g).bfloat16()
    temb = torch.randn(1, 13, 512, generator=g).bfloat16()
    f32 = np.float32
    rope = O.rope_3d(64, np.arange(13, dtype=f32), np.arange(30, dtype=f32), np.arange(45, dtype=f32))
    crope = O.rope_3d(64, np.linspace(1000, 1016.25, 5, dtype=f32), np.linspace(0, 30, 8, endpoint=False, dtype=f32),
                      np.linspace(0, 45, 12, endpoint=False, dtype=f32))
    with torch.no_grad():
        t0 = time.perf_counter()
        O.block_forward(sd, "transformer_blocks.0", hid, enc, temb, 48, 480, [0.6], rope, rope, crope)
        dt = time.perf_counter() - t0
    return {"value": 1.0 / (dt * 84.0), "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"1 full-width CogVideoXBlock forward (B=1, bf16, {dt:.1f} s) of the 84 per step, extrapolated x84"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--layers", type=int, default=42, help="debug only: anything but 42 is not the benchmark")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    use_dist = "RANK" in os.environ            # launched by torch.distributed.run (also exercised with 1 rank)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from tokensgen_amd import kernels as K
    from tokensgen_amd import lib
    from tokensgen_amd.fifo import FifoWorker
    from tokensgen_amd.scheduler import CogVideoXDPMScheduler
    from tokensgen_amd import rope as R
    lib.load()

    model = build_model(device, a.layers)
    sched = CogVideoXDPMScheduler(prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=1.0,
                                  timestep_spacing="trailing")
    sched.set_timesteps(52)
    g = torch.Generator(device=device).manual_seed(42 + rank)
    nf, C, H, W = 13, 16, 60, 90
    bf = torch.bfloat16
    latents = torch.randn(1, nf, C, H, W, generator=g, device=device, dtype=torch.float32).to(bf)
    old_x0 = torch.randn(nf, C, H, W, generator=g, device=device, dtype=torch.float32).to(bf)
    prompt = (torch.randn(2, 226, 4096, generator=g, device=device, dtype=torch.float32) * 0.1).to(bf)
    emb = torch.nn.functional.layer_norm(torch.randn(1, 5, 8, 12, 3072, generator=g, device=device), (3072,))
    emb = emb.permute(0, 1, 4, 2, 3).to(bf).repeat(2, 1, 1, 1, 1).contiguous()
    f32 = np.float32
    rope = R.rope_3d_crop(64, (0, 0, 0), (nf, 30, 45), (nf, 30, 45))
    worker = FifoWorker(model, sched, prompt, rope, 6.0, np.arange(30, dtype=f32), np.arange(45, dtype=f32),
                        np.linspace(0, 30, 8, endpoint=False, dtype=f32), np.linspace(0, 45, 12, endpoint=False, dtype=f32))
    ts = sched.timesteps.tolist()
    # a steady-state window: rank r of 8 covers queue positions [13*(r//2)+6*(r%2), +13)  
    r8 = rank % 8
    start = 13 * (r8 // 2) + 6 * (r8 % 2)
    lvl = ([18] * 6 + ts[::-1])
    t = lvl[start:start + nf]
    prev_t = [(-1 if q <= 6 else lvl[q - 1]) for q in range(start, start + nf)]
    next_t = [(lvl[q + 1] if q + 1 < len(lvl) else -1) for q in range(start, start + nf)]
    has_old = [nt > 0 for nt in next_t]
    grid_t = np.arange(nf, dtype=f32) + f32(start)
    cond_t = np.linspace(1000, 1016.25, 5, dtype=f32)
    keep = torch.empty(2, 7, C, H, W, device=device, dtype=bf)
    gathered = torch.empty(world * 2, 7, C, H, W, device=device, dtype=bf) if use_dist else None   # rank-major concat

    def step():
        noise = torch.randn(nf, 2, C, H, W, generator=g, device=device, dtype=torch.float32).to(bf)
        x, x0 = worker.window_step(latents, old_x0, has_old, t, prev_t, next_t, noise, grid_t, cond_t, emb)
        if use_dist:
            keep[0].copy_(x[0, 6:]); keep[1].copy_(x0[6:])
            dist.all_gather_into_tensor(gathered, keep)
        return x

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    # inside the timed region only the dominant kernel's launches are bracketed by HIP events (on the launch stream); the other
    # kernels' durations come from one extra, untimed step afterwards, so their ~1800 event markers do not sit in the measurement
    ATTN = "attention_2seg+rider"
    K.PROFILE.clear(); K.PROFILE_FILTER[0] = {ATTN}; K.PROFILE_ON[0] = True
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    K.PROFILE_ON[0] = False
    attn_prof = K.profile_summary().get(ATTN, {"ms": float("nan"), "n": 0})
    K.PROFILE.clear(); K.PROFILE_FILTER[0] = None; K.PROFILE_ON[0] = True
    step()
    fence()
    K.PROFILE_ON[0] = False
    if use_dist:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        prof = K.profile_summary()
        attn = attn_prof
        attn_s = attn["ms"] * 1e-3
        achieved = ATTN_FLOP_PER_LAUNCH / attn_s / 1e12 if attn["n"] else float("nan")
        traffic = None          # HBM-side bytes per launch of the dominant kernel, from the committed rocprofv3 PMC passes
        try:
            with open(os.path.join(ROOT, "profiles", "r1n_pmc_summary.json")) as f:
                traffic = json.load(f)["_derived"]["attention_main_traffic_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "DiT denoising steps/sec (CFG-batched 13-frame window: DiT fwd + CFG + 13 DPM updates), CogVideoX-5B To2V 720x480",
            "value": world * a.steps / dt, "unit": "steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic latents/embeddings, random-init weights at CogVideoX-5B shapes",
            "config": {"workload": "To2V FIFO window step, CogVideoX-5B (42 layers, D=3072, 48x64 heads), 13x60x90 latent "
                                   "window = 226 text + 17550 video + 480 condensed tokens, CFG batch 2, DPM-solver++ (52 trailing steps)",
                       "layers": a.layers, "exchange": "RCCL all_gather of kept half-windows per step" if use_dist else "none"},
            "step_mfma_frac": FLOP_PER_STEP * (a.layers / 42.0) * (a.steps / dt) / PEAK_BF16,
            "roofline": {"bound": "mfma", "kernel": "attn_fwd_pp_kernel (SDPA#1+#2 fused, SDPA#3 riding in the last round)", "achieved": achieved,
                         "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s", "frac": achieved / (PEAK_BF16 / 1e12), "traffic": traffic,
                         "launch_ms": attn["ms"], "launches_timed": attn["n"]},
            "kernel_ms": {k: round(v["ms"], 4) for k, v in prof.items()},
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
