"""Rank body of tests/test_multirank_one_gpu.py::test_ddp_two_ranks...: TWO ranks on ONE MI355X over gloo run the To2V training step's data-parallel leg (BASELINE config 5's
DDP: train_cogvideo_to2v.py:1157-1164) with the REAL kernels — each rank one micro-batch, the flat fp32 gradient all-reduced in buckets handed over while the backward still runs
(optim.GradSync), the collective failure verdict, clip + AdamW — and compare the parameters after the optimizer step, bit for bit, with ONE process that accumulates the same two
micro-batches (accumulation_steps = 2): a/2 + b/2 either way."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

outdir = sys.argv[1]
rank = int(os.environ["RANK"])
DEV, BF = torch.device("cuda", 0), torch.bfloat16


def done(msg):
    with open(os.path.join(outdir, f"rank{rank}.txt"), "w") as f:
        f.write(msg)


def setup(sync_factory, accum):
    from oracle import dit_ref as O
    from oracle import scheduler_ref as S
    from tokensgen_amd import optim, train
    H = 2
    cfg = dict(num_attention_heads=H, attention_head_dim=64, num_layers=2, patch_size=2, time_embed_dim=128, text_embed_dim=64, in_channels=16, out_channels=16)
    sd = {k: v.to(BF).to(DEV).contiguous() for k, v in O.make_state_dict(cfg, n_vip_dim=128, seed=95, std=0.08).items()}
    tr = train.To2VTrainer(sd, H, 2, patch_size=2, vip_scale=1.0)
    arena = optim.ParamArena({k: sd[k] for k in tr.trainable}, optim.arena_order(tr.trainable, 2), DEV)
    tr.use_arena(arena)
    opt = optim.AdamW(arena, lr=2e-3, max_grad_norm=1.0)
    _, ac = S.alphas_cumprod()
    sync = sync_factory(arena)
    step = train.To2VTrainStep(tr, arena, opt, torch.as_tensor(ac, dtype=torch.float32), accumulation_steps=accum, sync=sync)
    return step, arena, opt, sync


def batches():
    from oracle import dit_ref as O
    f32 = np.float32
    out = []
    for b in range(2):
        g = torch.Generator().manual_seed(96 + b)
        x0, noise = (torch.randn(1, 4, 16, 10, 12, generator=g).to(BF).to(DEV) for _ in range(2))
        text, vip = torch.randn(1, 9, 64, generator=g).to(BF).to(DEV), torch.randn(1, 5, 128, 2, 3, generator=g).to(BF).to(DEV)
        out.append((x0, noise, torch.tensor([[500 + 7 * b, 520, 480, 510]]), text, vip))
    rope = O.rope_3d(64, np.arange(4, dtype=f32), np.arange(5, dtype=f32), np.arange(6, dtype=f32))
    vrope = O.rope_3d(64, np.arange(4, dtype=f32) + f32(3), np.arange(5, dtype=f32), np.arange(6, dtype=f32))
    crope = O.rope_3d(64, np.linspace(1000, 1016.25, 5, dtype=f32), np.arange(2, dtype=f32), np.arange(3, dtype=f32))
    return out, (rope, vrope, crope)


def main():
    import torch.distributed as dist
    from tokensgen_amd import optim
    from tokensgen_amd.runtime import init_distributed
    torch.cuda.set_device(0)
    sha = lambda t: hashlib.sha256(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:12]
    bt, ropes = batches()
    # ---- one process, two micro-batches accumulated ----
    step, arena, opt, _ = setup(lambda a: None, accum=2)
    start = arena.param.clone()
    l0, d0 = step.micro_step(*bt[0], *ropes)
    l1, d1 = step.micro_step(*bt[1], *ropes)
    assert (d0, d1) == (False, True) and opt.t == 1
    ref_param, ref_losses = arena.param.clone(), (float(l0), float(l1))
    assert not torch.equal(ref_param, start)
    # ---- two ranks, one micro-batch each, gradients exchanged in buckets ----
    r, world = init_distributed("gloo", timeout_s=300)
    assert (r, world) == (rank, 2)
    step2, arena2, opt2, sync = setup(lambda a: optim.GradSync(a.grad, bucket_elems=max(1024, a.grad.numel() // 7)), accum=1)
    assert torch.equal(arena2.param, start) and len(sync.bounds) >= 7 and sync.world == 2
    launched = []
    orig_ready = sync.ready

    def ready(end):
        orig_ready(end)
        launched.append(sync._next)
    sync.ready = ready
    l, did = step2.micro_step(*bt[rank], *ropes)
    torch.cuda.synchronize()
    res = {"stepped": bool(did) and opt2.t == 1, "loss": float(l) == ref_losses[rank], "params": torch.equal(arena2.param, ref_param),
           "buckets_during_backward": any(0 < n < len(sync.bounds) for n in launched)}       # some buckets went out before the backward had finished
    dist.barrier()
    dist.destroy_process_group()
    bad = [k for k, v in res.items() if not v]
    done(("ok " + " ".join(sorted(res)) if not bad else "mismatch: " + " ".join(bad)) + f" [param {sha(arena2.param)} ref {sha(ref_param)} buckets {len(sync.bounds)} launched {launched[:12]}]")


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:  # noqa: BLE001 — the parent test reads the file
        import traceback
        done("exception: " + "".join(traceback.format_exception(type(e), e, e.__traceback__))[-3000:])
        raise
