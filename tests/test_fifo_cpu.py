"""CPU: the product's FIFO driver (tokensgen_amd.fifo.cogvideo_fifo_mp_v2) against the oracle's driver with the
same injected denoiser and noise — single process, and sharded over 2 gloo ranks (the N>1 exchange path)."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import fifo_ref as Fq
from oracle import scheduler_ref as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BF = torch.bfloat16
NF, T, H, W, C = 13, 52, 2, 3, 4


def _fake_denoise(x, tt, grid_t, cond_t, vs, emb):
    """Cheap deterministic stand-in for the DiT that depends on every per-window input."""
    bias = 0.01 * float(np.sum(grid_t)) % 1.0 + 0.001 * float(np.sum(cond_t)) % 1.0 + 0.1 * vs
    y = 0.3 * x.float() + 0.05 * torch.tanh(x.float().mean(dim=1, keepdim=True)) + 0.001 * tt[..., None, None, None].float() / 100
    y = y + bias * 0.01 + emb.float().mean() * 0.1
    y[1] = y[1] * 1.1
    return y.to(x.dtype)


def _inputs(num_chunks=1):
    g = torch.Generator().manual_seed(7)
    f32 = np.float32
    d = dict(
        fifo_latents=torch.randn(1, T, C, H, W, generator=g).to(BF),
        fifo_old=[torch.randn(1, 1, C, H, W, generator=g).to(BF) for _ in range(T - 1)] + [None],
        emb=torch.randn(1, 4 * (num_chunks + 1), 8, 2, 3, generator=g).to(BF).repeat(2, 1, 1, 1, 1),
        grid_t=np.linspace(0, num_chunks * NF, num_chunks * NF, endpoint=False, dtype=f32),
        cond_t=np.concatenate([np.linspace(1000 + i * NF, 1000 + (i + 1) * NF, 4, endpoint=False, dtype=f32) for i in range(num_chunks + 1)]),
        num_frames=num_chunks * NF)
    return d


def _noise(i, tag, shape):
    g = torch.Generator().manual_seed(1000 * i + tag)
    return torch.randn(shape, generator=g).to(BF)


_COEF = {}
_orig_coef = S.step_coefficients


def _memo_coef(ac, t, prev_t, t_back):
    key = (t, prev_t, t_back)
    if key not in _COEF:
        _COEF[key] = _orig_coef(ac, t, prev_t, t_back)
    return _COEF[key]


S.step_coefficients = _memo_coef     # the tables are pure functions of (t, prev_t, t_back); keeps the CPU suite fast


def _oracle_run(d):
    betas, ac = S.alphas_cumprod()
    ts = S.trailing_timesteps(T)
    emb_ext = torch.cat([d["emb"]] + [d["emb"][:, -4:]] * (T // NF + 1), dim=1)
    trace = []
    out = Fq.run_fifo_prenoise(lambda x, tt, grid_t, cond_t, vs: _fake_denoise(x, tt, grid_t, cond_t, vs, emb_ext[:, vs:vs + 5]),
                               betas, ac, d["fifo_latents"], d["fifo_old"], ts, d["num_frames"], 6.0, d["grid_t"], d["cond_t"], 1000,
                               _noise, trace=trace)
    return out, trace


def _fake_decode(z):
    """Stand-in for vae.decode on one chunk (the sharding logic is what the CPU test checks): [1,nf,C,h,w] -> [1,3,4(nf-1)+1,2h,2w]."""
    x = z.float().permute(0, 2, 1, 3, 4)[:, :3]
    x = torch.nn.functional.interpolate(x, size=(4 * (z.shape[1] - 1) + 1, 2 * z.shape[3], 2 * z.shape[4]), mode="nearest")
    return (x * 0.5 + z.float().mean()).to(BF)


def _product_run(d, trace=None, output_type="latent", split=False, calls=None):
    from tokensgen_amd import fifo
    from tokensgen_amd.scheduler import CogVideoXDPMScheduler
    sched = CogVideoXDPMScheduler(prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=1.0, timestep_spacing="trailing")
    sched.set_timesteps(T)
    _, ac = S.alphas_cumprod()
    pipe = SimpleNamespace(device=torch.device("cpu"), scheduler=sched, transformer=None, guidance_scale=6.0)
    f32 = np.float32
    bo = SimpleNamespace(sampling_params=dict(use_adaptive_padding=True, num_partitions=4), fifo_latents=d["fifo_latents"],
                         fifo_old_pred_original_sample=d["fifo_old"], nf_per_chunk=NF, vip_nf_per_chunk=4, num_frames=d["num_frames"],
                         image_embeddings=d["emb"], timesteps=sched.timesteps, num_inference_steps=T, do_classifier_free_guidance=True,
                         use_separate_guidance=False, use_dynamic_cfg=False, prompt_embeds=None, image_rotary_emb=None,
                         vip_image_rotary_grid=[d["grid_t"].copy(), np.arange(1, dtype=f32), np.arange(1, dtype=f32)],
                         vip_condition_rotary_grid=[d["cond_t"].copy(), np.arange(1, dtype=f32), np.arange(1, dtype=f32)],
                         guidance_scale=6.0, cache_idx=[], video_ipadapter_start_frame_idx=1000, output_type=output_type, return_dict=False,
                         orig_latents=d["fifo_latents"][:, :NF])

    def both_branches(latents, t, grid_t, cond_grid_t, image_embeddings, **unused):
        vs_probe = float(cond_grid_t[0])
        vs = int(round((vs_probe - 1000) / 3.25))
        return _fake_denoise(torch.cat([latents] * 2), torch.as_tensor(np.asarray(t))[None].expand(2, -1), grid_t, cond_grid_t, vs, image_embeddings)

    def predict_fn(h, **kw):
        """One guidance branch of a window (what a rank of a split iteration computes): row h of the stand-in denoiser's batch."""
        if calls is not None:
            calls.append(h)
        return both_branches(**kw)[h]

    def finish_fn(preds, latents, old_x0, has_old, t, prev_t, next_t, noise, **unused):
        return solve(S.cfg_combine(preds, 6.0), latents, old_x0, has_old, t, prev_t, next_t, noise)

    def window_fn(latents, old_x0, has_old, t, prev_t, next_t, noise, grid_t, cond_grid_t, image_embeddings):
        """Same arithmetic as the oracle loop above (this test checks the DRIVER, not the kernels)."""
        pred = S.cfg_combine(both_branches(latents, t, grid_t, cond_grid_t, image_embeddings), 6.0)
        return solve(pred, latents, old_x0, has_old, t, prev_t, next_t, noise)

    def solve(pred, latents, old_x0, has_old, t, prev_t, next_t, noise):
        o_lat, o_x0 = latents.clone(), torch.zeros_like(old_x0)
        for j in range(NF):
            nxt = int(next_t[j]) if next_t[j] > 0 else None
            seq = iter([noise[j, 0][None, None], noise[j, 1][None, None]])
            x, x0 = S.dpm_step(ac, pred[:, [j]].float(), old_x0[j][None, None].float() if has_old[j] else None, int(t[j]), int(prev_t[j]), nxt,
                               latents[:, [j]].float(), lambda: next(seq).float())
            o_lat[:, [j]] = x.to(BF)
            o_x0[j] = x0.to(BF)[0, 0]
        return o_lat, o_x0

    res = fifo.cogvideo_fifo_mp_v2([pipe], bo, step_noise_fn=_noise, tail_noise_fn=lambda i, shape: _noise(i, 97, shape),
                                   window_fn=window_fn, trace=trace, decode_chunk_fn=_fake_decode,
                                   **(dict(predict_fn=predict_fn, finish_fn=finish_fn) if split else {}))
    return res[1] if output_type == "latent" else torch.cat([res[0], res[1]], dim=2)


def test_driver_single_process_matches_oracle():
    d = _inputs()
    ref, ref_trace = _oracle_run(d)
    trace = []
    out = _product_run(d, trace)
    assert trace == ref_trace                     # window geometry + condensed-token index, every iteration
    assert out.shape == ref.shape == (1, NF, C, H, W)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("num_chunks", [2, 3])
def test_driver_multi_chunk_long_video_matches_oracle(num_chunks):
    """The long-video case (BASELINE config 3's structure): more than one 13-frame chunk flows through the queue — condensed-token windows advance
    chunk by chunk, the rotary time grid keeps growing — same index trace and same latents as the oracle's driver, every iteration."""
    d = _inputs(num_chunks)
    ref, ref_trace = _oracle_run(d)
    trace = []
    out = _product_run(d, trace)
    assert trace == ref_trace and len({t[-1] for t in trace}) > 1          # the condensed-token window index really moves
    assert out.shape == ref.shape == (1, num_chunks * NF, C, H, W)
    assert torch.equal(out, ref)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = _product_run(_inputs())
        dec = _product_run(_inputs(2), output_type="pt")
        q.put((rank, out.float().numpy(), dec.float().numpy()))      # by value: a tensor travels as a shared-memory handle that dies with this process
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_driver_two_ranks_gloo_matches_single():
    import torch.multiprocessing as mp
    ref = _product_run(_inputs())          # single-process product result (itself checked against the oracle above)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from conftest import free_port
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ref_dec = _product_run(_inputs(2), output_type="pt")      # 2 chunks: one per rank in the sharded decode, + the 1-chunk orig video
    assert ref_dec.shape == (1, 3, 3 * 49, 2 * H, 2 * W)
    got = {r: (torch.from_numpy(o), torch.from_numpy(dcd)) for r, o, dcd in (q.get(timeout=280) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
    assert torch.equal(got[0][0], ref.float()) and torch.equal(got[1][0], ref.float())
    assert torch.equal(got[0][1], ref_dec.float()) and torch.equal(got[1][1], ref_dec.float())


def _split_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []
        out = _product_run(_inputs(), split=True, calls=calls)
        q.put((rank, out.float().numpy(), len(calls), sorted(set(calls))))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_small_iterations_split_by_guidance_branch_four_ranks_gloo():
    """Round 6: the ramp at the head of a run has 1, 2, 2, ... windows per iteration (cogvideo_sampling_mp_fifo.py:235-253 leaves the other workers idle).  With 4 ranks and 2
    guidance branches, the 7 iterations with <= 2 windows run as batch-1 forwards — rank 2k + h takes branch h of window k — one all_gather of the model outputs, the solver
    step replicated.  Same latents as the single-process run, bit for bit, on every rank; ranks 0..3 computed branch 0 / 1 / 0 / 1 only, ranks 2, 3 fewer times (the 1-window iteration)."""
    import torch.multiprocessing as mp
    from tokensgen_amd.fifo import window_plan
    ref = _product_run(_inputs())
    qs, small, one = T - (NF - NF // 2), 0, 0
    for _ in range(NF + T - NF):
        n = len(window_plan(qs, NF, 4))
        small += n <= 2
        one += n == 1
        qs = max(0, qs - 1)
    assert (small, one) == (7, 1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from conftest import free_port
    port = free_port()
    procs = [ctx.Process(target=_split_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    got = {r: (torch.from_numpy(o), n, br) for r, o, n, br in (q.get(timeout=280) for _ in range(4))}
    for p in procs:
        p.join(timeout=60)
    for r in range(4):
        assert torch.equal(got[r][0], ref.float()), r
        assert got[r][2] == [r % 2] and got[r][1] == (small if r < 2 else small - one), (r, got[r][1:])


def test_unsupported_modes_raise():
    from tokensgen_amd import fifo
    bo = SimpleNamespace(sampling_params=dict(use_sliding_window_embedding=True))
    with pytest.raises(NotImplementedError):
        fifo.cogvideo_fifo_mp_v2([SimpleNamespace(device="cpu")], bo)
    assert [w["start"] for w in fifo.window_plan(0)] == [0, 6, 13, 19, 26, 32, 39, 45]
    assert fifo.window_plan(45)[0]["rank"] == 7 and len(fifo.window_plan(45)) == 1
