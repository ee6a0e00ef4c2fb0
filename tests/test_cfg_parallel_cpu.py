"""CPU: the CFG-parallel exchange (tokensgen_amd/cfg_parallel.py) over gloo with 3 ranks — rank r computes half r % 2, every rank
ends up with [uncond, cond]; plus mode resolution.  The numerics of running the halves as batch-1 forwards are checked on the GPU
(tests/test_t2to_gpu.py, "emulate" mode == batched, bitwise)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _half(h):
    return torch.full((1, 3, 4), float(10 * (h + 1))) + torch.arange(12.0).reshape(1, 3, 4)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tokensgen_amd import cfg_parallel as CP
        calls = []
        mode = CP.resolve(None)

        def fh(h):
            calls.append(h)
            return _half(h)
        out = CP.predict(mode, fh, lambda: (_ for _ in ()).throw(AssertionError("batched forward must not run")))
        q.put((rank, mode, calls, out.numpy()))      # by value: a tensor travels as a shared-memory handle that dies with this process
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_cfg_parallel_three_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from conftest import free_port
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = {r: (m, c, o) for r, m, c, o in (q.get(timeout=100) for _ in range(3))}
    for p in procs:
        p.join(timeout=30)
    want = torch.cat([_half(0), _half(1)])
    for r in range(3):
        mode, calls, out = got[r]
        assert mode == "parallel" and calls == [r % 2] and torch.equal(torch.from_numpy(out), want)


def test_cfg_parallel_modes_single_process():
    from tokensgen_amd import cfg_parallel as CP
    assert CP.resolve(None) == "batched" and CP.resolve("auto") == "batched" and CP.resolve(False) == "batched" and CP.resolve("emulate") == "emulate"
    with pytest.raises(RuntimeError):
        CP.resolve(True)
    with pytest.raises(ValueError):
        CP.resolve("yes")
    both = torch.cat([_half(0), _half(1)])
    assert torch.equal(CP.predict("batched", None, lambda: both), both)
    assert torch.equal(CP.predict("emulate", _half, None), both)


# ---- the condensed-token encode of the source clips, sharded by chunk (pipeline.vae_encode_image -> cfg_parallel.map_chunks_sharded) --------------------------------
def _stub_pipeline(calls):
    """MPFIFOVideoIPAdapterCogVideoXPipeline around CPU stand-ins for the three HIP modules (VAE encoder, patch_embed.proj, Resampler): deterministic torch
    functions of their inputs, so that the test is about WHICH rank encodes WHICH chunk, the noise bookkeeping and the exchange — not about kernels."""
    from types import SimpleNamespace
    from tokensgen_amd.pipeline import MPFIFOVideoIPAdapterCogVideoXPipeline
    D = 8
    g = torch.Generator().manual_seed(5)
    wp, wr = torch.randn(16, D, generator=g), torch.randn(4, 5, generator=g)

    class Vae:
        config = SimpleNamespace(block_out_channels=(8, 8, 8, 8), temporal_compression_ratio=4, scaling_factor=1.15258426, latent_channels=16)

        def encode(self, x):                                       # [b, 3, 17, 16, 24] -> moments of [b, 16, 5, 2, 3]
            calls.append(float(x.float().sum()))
            m = x.float().mean(dim=1, keepdim=True)[:, :, ::4, ::8, ::8] * torch.arange(1, 17).view(1, 16, 1, 1, 1) / 16.0
            return SimpleNamespace(latent_dist=SimpleNamespace(mean=m.to(torch.bfloat16), std=(0.1 + m.abs() * 0.05).to(torch.bfloat16), mode=lambda: m.to(torch.bfloat16)))

    class Tr:
        device = torch.device("cpu")
        config = SimpleNamespace(patch_size=2, attention_head_dim=64, in_channels=16)

        def patch_embed_proj(self, lat):                           # b f c h w -> b f (h w) D
            return (lat.float().flatten(3).transpose(2, 3) @ wp).to(torch.bfloat16)

    class Rs:
        config = SimpleNamespace(dim_head=64, max_temporal_seq_len=5, max_height_seq_len=2, max_width_seq_len=3, num_temporal_queries=4, num_height_queries=2, num_width_queries=3)

        def __call__(self, tokens, image_rotary_emb=None, sampling_rotary_emb=None):      # b 5 (h w) D -> b 4 D 2 3
            b, f, hw, d = tokens.shape
            return torch.einsum("qf,bfnd->bqdn", wr, tokens.float()).reshape(b, 4, d, 2, 3).to(torch.bfloat16)
    return MPFIFOVideoIPAdapterCogVideoXPipeline(Tr(), SimpleNamespace(), vae=Vae(), resampler=Rs(), device="cpu")


def _frames():
    return torch.randn(1, 3 * 17, 3, 16, 24, generator=torch.Generator().manual_seed(9)).clamp(-1, 1)        # three 17-frame chunks (+ the padded fourth)


def _encode(pipe):
    gen = torch.Generator().manual_seed(21)
    emb = pipe.vae_encode_image(_frames(), nf_per_chunk=17, compressed_nf_per_chunk=5, generator=gen, sample_posterior=True)
    return emb, torch.randn(3, generator=gen)                     # + the generator's NEXT draw: its state after the call must not depend on the rank count


def _encode_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []
        emb, nxt = _encode(_stub_pipeline(calls))
        q.put((rank, len(calls), emb.float().numpy(), nxt.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_sharded_token_encode_three_ranks_gloo_is_bitwise_the_single_rank_run():
    """pipeline_cogvideox_mp_fifo.py:585-609 runs vae.encode -> patch_embed.proj -> Resampler chunk after chunk on GPU 0.  Here chunk c goes to rank c % 3
    (4 chunks incl. the padded one: ranks hold 2 / 1 / 1), one all_gather: every rank gets the single-rank tokens bit for bit, and the posterior-noise generator is
    left in the single-rank state."""
    import torch.multiprocessing as mp
    calls = []
    want, want_next = _encode(_stub_pipeline(calls))
    assert len(calls) == 4 and tuple(want.shape) == (2, 16, 8, 2, 3)           # CFG: the same tokens twice (:646)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from conftest import free_port
    port = free_port()
    procs = [ctx.Process(target=_encode_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = {r: (n, e, x) for r, n, e, x in (q.get(timeout=150) for _ in range(3))}
    for p in procs:
        p.join(timeout=30)
    assert [got[r][0] for r in range(3)] == [2, 1, 1]
    for r in range(3):
        assert torch.equal(torch.from_numpy(got[r][1]), want.float()) and torch.equal(torch.from_numpy(got[r][2]), want_next), r
