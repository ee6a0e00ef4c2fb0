"""Rank body of tests/test_multirank_one_gpu.py::test_t2to_stage_and_token_encode_three_ranks...: THREE ranks on ONE MI355X over gloo (collectives staged through the host as in
rank_worker_gpu_multi.py) run, with the REAL kernels, the two multi-rank branches the four-rank worker does not reach:
  * the T2To stage CFG-parallel (infer_cogvideo_mp_fifo.py:262 runs it on one GPU; cfg_parallel.predict: rank r computes guidance half r % 2 as a batch-1 forward, one all_gather per
    step, the identical solver step with identically seeded noise on every rank), and
  * the condensed-token encode sharded by chunk (pipeline_cogvideox_mp_fifo.py:585-609 chunk after chunk on GPU 0; cfg_parallel.map_chunks_sharded: chunk c on rank c % 3 — 2 / 1 / 1
    of the 4 chunks — posterior noise for ALL chunks drawn on every rank), with `use_separate_guidance` (the all-zero video's tokens: a second sharded pass).
Every rank first runs the same calls with NO process group and compares bit for bit, the generator's final state included."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

outdir = sys.argv[1]
rank = int(os.environ["RANK"])
GOLD = os.path.join(ROOT, "tests", "golden")
DEV, BF = torch.device("cuda", 0), torch.bfloat16


def done(msg):
    with open(os.path.join(outdir, f"rank{rank}.txt"), "w") as f:
        f.write(msg)


def build_t2to():
    from oracle import dit_ref as O
    from tokensgen_amd.pca import PCA
    from tokensgen_amd.pipeline_t2to import LongVGenCogVideoXPipeline
    from tokensgen_amd.scheduler import CogVideoXDPMScheduler
    from tokensgen_amd.transformer import CogVideoXTransformer3DModel
    g = torch.load(os.path.join(GOLD, "t2to_tiny.pt"), weights_only=False)
    cfg = g["cfg"]
    m = CogVideoXTransformer3DModel(num_attention_heads=cfg["num_attention_heads"], attention_head_dim=64, num_layers=cfg["num_layers"], time_embed_dim=cfg["time_embed_dim"],
                                    text_embed_dim=cfg["text_embed_dim"], patch_size=1, use_rotary_positional_embeddings=True, device=DEV)
    m.load_state_dict({k: v.to(BF) for k, v in O.make_state_dict(cfg, seed=g["weight_seed"]).items()}, strict=True)
    pipe = LongVGenCogVideoXPipeline(m, CogVideoXDPMScheduler(prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=1.0, timestep_spacing="trailing"))
    pca = PCA()
    pca.register_buffer("mean_", g["pca_mean"]); pca.register_buffer("components_", g["pca_components16"])
    c = g["cases"]["torch.bfloat16"]

    def run():
        gen = torch.Generator().manual_seed(g["gen_seed"])
        out = pipe(prompt_embeds=c["prompt"], negative_prompt_embeds=c["negative"], height=g["H"], width=g["W"], num_frames_per_chunk=g["nfc"], num_chunks=g["chunks"],
                   num_inference_steps=g["steps"], use_dynamic_cfg=True, guidance_scale=g["guidance_scale"], generator=gen, longvgen_mean=g["mean"], longvgen_std=g["std"],
                   longvgen_pca=pca).frames
        return out, gen.get_state()
    return m, run


def build_encode():
    from oracle import dit_ref as O
    from oracle import resampler_ref as RR
    from oracle import vae_ref as V
    from tokensgen_amd.pipeline import MPFIFOVideoIPAdapterCogVideoXPipeline
    from tokensgen_amd.resampler import Resampler
    from tokensgen_amd.scheduler import CogVideoXDPMScheduler
    from tokensgen_amd.transformer import CogVideoXTransformer3DModel
    from tokensgen_amd.vae import AutoencoderKLCogVideoX
    gv = torch.load(os.path.join(GOLD, "vae_tiny.pt"), weights_only=False)
    vcfg = gv["cfg"]
    dcfg = dict(num_attention_heads=2, attention_head_dim=64, num_layers=1, patch_size=2, time_embed_dim=128, text_embed_dim=64, in_channels=16, out_channels=16)
    rcfg = dict(dim=128, depth=2, dim_head=64, heads=2, num_height_queries=2, num_width_queries=3, num_temporal_queries=4, embedding_dim=128, output_dim=128, ff_mult=4,
                max_height_seq_len=4, max_width_seq_len=6, max_temporal_seq_len=5)
    vae = AutoencoderKLCogVideoX(block_out_channels=vcfg["block_out_channels"], layers_per_block=1, sample_height=64, sample_width=96, device=DEV)
    vae.load_state_dict(V.make_state_dict(vcfg, seed=gv["weight_seed"]))
    m = CogVideoXTransformer3DModel(num_attention_heads=2, attention_head_dim=64, num_layers=1, time_embed_dim=128, text_embed_dim=64, use_rotary_positional_embeddings=True, device=DEV)
    m.load_state_dict({k: v.to(BF) for k, v in O.make_state_dict(dcfg, None, seed=21).items()}, strict=True)
    rs = Resampler(**rcfg, device=DEV)
    rs.load_state_dict(RR.make_state_dict(rcfg, seed=22))
    pipe = MPFIFOVideoIPAdapterCogVideoXPipeline(m, CogVideoXDPMScheduler(prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=1.0,
                                                                          timestep_spacing="trailing"), vae=vae, resampler=rs)
    frames = (torch.rand(1, 51, 3, 64, 96, generator=torch.Generator().manual_seed(23)) * 2 - 1).to(DEV)       # 3 chunks of 17 frames (+ the padded one: 4)

    def run():
        gen = torch.Generator(device=DEV).manual_seed(5)
        emb = pipe.vae_encode_image(frames, nf_per_chunk=17, compressed_nf_per_chunk=5, generator=gen, sample_posterior=True, use_separate_guidance=True)
        return emb, gen.get_state()
    return (vae, m, rs, pipe), run


def main():
    import torch.distributed as dist
    from tokensgen_amd import cfg_parallel as CP
    from tokensgen_amd.runtime import init_distributed
    torch.cuda.set_device(0)
    keep1, t2to = build_t2to()
    keep2, encode = build_encode()
    assert not dist.is_initialized()
    ref_t, ref_ts = t2to()
    ref_e, ref_es = encode()
    assert ref_e.shape[0] == 3 and ref_e.shape[1] == 4 * 4 and not torch.equal(ref_e[0], ref_e[1])        # [tokens, zero-video tokens, tokens], 4 queries x 4 chunks
    r, world = init_distributed("gloo", timeout_s=300)
    assert (r, world) == (rank, 3)
    orig_agit = dist.all_gather_into_tensor

    def staged_all_gather_into_tensor(out, inp, group=None, async_op=False):      # (see rank_worker_gpu_multi.py: gloo on host tensors, synchronous copies)
        assert not async_op
        if not inp.is_cuda:
            return orig_agit(out, inp, group=group)
        torch.cuda.synchronize()
        host_out = torch.empty(out.shape, dtype=out.dtype)
        orig_agit(host_out, inp.detach().cpu().contiguous(), group=group)
        out.copy_(host_out)
        torch.cuda.synchronize()
    dist.all_gather_into_tensor = staged_all_gather_into_tensor
    # what THIS rank computes: batch-1 halves in the T2To stage, its own chunks in the encode
    seen = {"half": [], "chunks": 0}
    orig_predict, orig_enc = CP.predict, keep2[0].encode

    def predict(mode, forward_half, forward_both, n=2):
        def fh(h):
            seen["half"].append(h)
            return forward_half(h)

        def fb():
            seen["half"].append("both")
            return forward_both()
        return orig_predict(mode, fh, fb, n)

    def enc(x, *a, **k):
        seen["chunks"] += 1
        return orig_enc(x, *a, **k)
    CP.predict, keep2[0].encode = predict, enc
    got_t, got_ts = t2to()
    got_e, got_es = encode()
    CP.predict, keep2[0].encode = orig_predict, orig_enc
    res = {"t2to_frames": torch.equal(got_t, ref_t), "t2to_generator": torch.equal(got_ts, ref_ts), "tokens": torch.equal(got_e, ref_e), "token_generator": torch.equal(got_es, ref_es),
           "halves": len(seen["half"]) > 0 and set(seen["half"]) == {rank % 2},                   # every step one batch-1 forward of this rank's half, never the batched one
           "chunks": seen["chunks"] == 2 * (2 if rank == 0 else 1)}                                 # two sharded passes (video, zero video) over 4 chunks on 3 ranks
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    bad = [k for k, v in res.items() if not v]
    done(("ok " + " ".join(sorted(res)) if not bad else "mismatch: " + " ".join(bad)) + f" [halves {seen['half'][:4]}.. x{len(seen['half'])}, encodes {seen['chunks']}]")


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:  # noqa: BLE001 — the parent test reads the file
        import traceback
        done("exception: " + "".join(traceback.format_exception(type(e), e, e.__traceback__))[-3000:])
        raise
