"""Rank body of tests/test_rccl_gpu.py: ONE rank with an `nccl` (= RCCL) process group on the MI355X, running every collective call site of the
package on GPU tensors and comparing each with the same code path run WITHOUT a process group, bit for bit:

  fifo        cogvideo_fifo_mp_v2 (the per-iteration all_gather_into_tensor of window outputs + failure flags, on the HIP DiT) and
              decode_chunks_sharded (all_gather_object + all_gather_into_tensor of decoded frames)
  broadcast   runtime.broadcast_weights over the fused bf16 storages (raw-byte views) and the coalesced small tensors
  gradsync    optim.GradSync bucketed async all_reduce on a flat fp32 CUDA tensor
  cfg         cfg_parallel.predict: "parallel" with fewer ranks than branches falls back to the batched forward; all_gather path with n = 1

Reference call sites: cogvideo_sampling_mp_fifo.py:195-221 (weights to workers), :284-334 (dispatch / merge per iteration), :373-376 (decode).
Writes "ok ..." or the failure into <outdir>/rank0.txt."""
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

outdir = sys.argv[1]
BF = torch.bfloat16
DEV = torch.device("cuda", 0)


def done(msg):
    with open(os.path.join(outdir, "rank0.txt"), "w") as f:
        f.write(msg)


def build():
    from oracle import dit_ref as O
    from tokensgen_amd.pipeline import MPFIFOVideoIPAdapterCogVideoXPipeline
    from tokensgen_amd.scheduler import CogVideoXDPMScheduler
    from tokensgen_amd.transformer import CogVideoXTransformer3DModel
    cfg = dict(num_attention_heads=2, attention_head_dim=64, num_layers=2, patch_size=2, time_embed_dim=128, text_embed_dim=64, in_channels=16,
               out_channels=16)
    vip = dict(length=30, func_type="1", scale=[0.6], resampler_params=dict(output_dim=128, num_height_queries=2, num_width_queries=3, num_temporal_queries=4))
    sd = {k: v.to(BF) for k, v in O.make_state_dict(cfg, 128, seed=31).items()}
    m = CogVideoXTransformer3DModel(num_attention_heads=2, attention_head_dim=64, num_layers=2, time_embed_dim=128, text_embed_dim=64,
                                    use_rotary_positional_embeddings=True, device=DEV)
    m.set_vip_layers(None, **vip)
    m.load_state_dict(sd, strict=True)
    sched = CogVideoXDPMScheduler(prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=1.0, timestep_spacing="trailing")
    pipe = MPFIFOVideoIPAdapterCogVideoXPipeline(m, sched, resampler_config=dict(num_temporal_queries=4, num_height_queries=2, num_width_queries=3))
    return m, sd, pipe


def _noise(i, tag, shape):
    g = torch.Generator().manual_seed(1000 * i + tag)
    return torch.randn(shape, generator=g).to(BF).to(DEV)


def fake_decode(z):
    """[1, nf, C, h, w] -> [1, 3, 4 (nf - 1) + 1, 2h, 2w] on the GPU (the sharding + gather is what is under test; the VAE has its own tests)."""
    x = z.float().permute(0, 2, 1, 3, 4)[:, :3]
    x = torch.nn.functional.interpolate(x, size=(4 * (z.shape[1] - 1) + 1, 2 * z.shape[3], 2 * z.shape[4]), mode="nearest")
    return (x * 0.5 + z.float().mean()).to(BF)


def fifo_run(pipe, output_type):
    from tokensgen_amd import fifo
    g = torch.Generator().manual_seed(5)
    H, W, nf, T, chunks = 4, 6, 13, 52, 2          # (the FIFO window plan needs T = nf * num_partitions: 52 = 13 x 4)
    lat0 = torch.randn(1, nf, 16, H, W, generator=g).to(BF)
    pe, ne = torch.randn(1, 8, 64, generator=g).to(BF), torch.randn(1, 8, 64, generator=g).to(BF)
    emb = torch.randn(1, 4 * chunks, 128, 2, 3, generator=g).to(BF)
    out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image_embeddings=emb, height=H * 8, width=W * 8, num_chunks=chunks, num_inference_steps=T,
               latents=lat0, step_noise=lambda i: _noise(i, 5, (nf, 2, 16, H, W)), output_type=output_type)
    res = fifo.cogvideo_fifo_mp_v2([pipe], out, step_noise_fn=_noise, tail_noise_fn=lambda i, shape: _noise(i, 97, shape), decode_chunk_fn=fake_decode)
    return res[1] if output_type == "latent" else torch.cat([res[0], res[1]], dim=2)


def gradsync_run(with_group):
    from tokensgen_amd.optim import GradSync
    g = torch.Generator().manual_seed(9)
    flat = torch.randn(1000, generator=g).to(DEV)
    want = flat.clone()
    sync = GradSync(flat, bucket_elems=300)
    for end in (100, 300, 650, 899):
        sync.ready(end)
    launched = sync._next
    sync.finish()
    torch.cuda.synchronize()
    assert launched == (2 if with_group else 0), launched          # buckets really went to the collective as the backward passed them
    assert sync._next == 0 and not sync._work
    return torch.equal(flat, want)                                   # SUM over one rank


def cfg_run():
    from tokensgen_amd import cfg_parallel as CP
    calls = []
    both = lambda: (calls.append("both"), torch.arange(6, dtype=torch.float32, device=DEV).view(2, 3))[1]
    half = lambda h: (calls.append(h), torch.full((1, 3), float(h), device=DEV))[1]
    a = CP.predict("parallel", half, both, n=2)                      # 1 rank < 2 branches: the batched forward
    b = CP.predict("parallel", half, both, n=1)                      # 1 branch on 1 rank: the all_gather path itself
    return calls == ["both", 0] and torch.equal(a, both()) and torch.equal(b, torch.zeros(1, 3, device=DEV))


def main():
    import torch.distributed as dist
    from tokensgen_amd.runtime import broadcast_weights, init_distributed
    torch.cuda.set_device(0)
    m, sd, pipe = build()
    # ---- the same calls with NO process group ----
    assert not dist.is_initialized()
    ref_lat = fifo_run(pipe, "latent")
    ref_vid = fifo_run(pipe, "pt")
    assert gradsync_run(False)
    # ---- 1-rank RCCL group ----
    rank, world = init_distributed("nccl", timeout_s=120, device=DEV)
    assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
    got_lat = fifo_run(pipe, "latent")
    got_vid = fifo_run(pipe, "pt")
    res = {"fifo_latents": torch.equal(got_lat, ref_lat), "fifo_decode": torch.equal(got_vid, ref_vid) and got_vid.shape[2] == 3 * 49}
    before = {k: v.clone() for k, v in m.state_dict().items()}
    nbytes = broadcast_weights(m, src=0)
    after = m.state_dict()
    res["broadcast"] = nbytes > 0 and sorted(after) == sorted(before) and all(torch.equal(after[k], before[k]) for k in before) \
        and all(torch.equal(after[k].cpu(), sd[k]) for k in sd)
    res["gradsync"] = gradsync_run(True)
    res["cfg_parallel"] = cfg_run()
    # the weights still compute the same thing after the broadcast wrote through the fused storages
    res["fifo_after_broadcast"] = torch.equal(fifo_run(pipe, "latent"), ref_lat)
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    bad = [k for k, v in res.items() if not v]
    done("ok " + " ".join(sorted(res)) if not bad else "mismatch: " + " ".join(bad))


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:  # noqa: BLE001 — the parent test reads the file
        import traceback
        done("exception: " + "".join(traceback.format_exception(type(e), e, e.__traceback__))[-3000:])
        raise
