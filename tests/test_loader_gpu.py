"""GPU: the gen.yaml / edit.yaml LOADING sequence of the reference's entry script replayed against a synthetic checkpoint tree the
test writes (tiny shapes, the real file layout): infer_cogvideo_mp_fifo.py:101-132 (`init_pipeline`: transformer.from_pretrained ->
set_vip_layers(vip_path) -> Resampler.from_pretrained(vip_path, subfolder="resampler") -> resampler.set_pca(longvgen_pca) ->
Pipeline.from_pretrained(path, transformer=, resampler=) -> CogVideoXDPMScheduler.from_config(pipe.scheduler.config,
timestep_spacing="trailing") -> pipe.to(device) -> vae.enable_slicing/tiling) and :220-233 (the T2To pipeline), through the `longvgen`
import aliases — then one short run of each stage so that what was loaded is what computes."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import dit_ref as O
from oracle import resampler_ref as RR
from oracle import vae_ref as V

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16
DIT = dict(num_attention_heads=2, attention_head_dim=64, num_layers=2, patch_size=2, time_embed_dim=128, text_embed_dim=64, in_channels=16,
           out_channels=16, use_rotary_positional_embeddings=True)
VIP = dict(length=30, func_type="1", scale=[1.0], use_vae_as_encoder=True, video_ipadapter_start_frame_idx=1000,
           resampler_params=dict(dim=128, depth=2, dim_head=64, heads=2, num_height_queries=2, num_width_queries=3, num_temporal_queries=4,
                                 embedding_dim=128, output_dim=128, ff_mult=4, max_height_seq_len=4, max_width_seq_len=6, max_temporal_seq_len=5))
VAE = dict(block_out_channels=(64, 128), layers_per_block=1, latent_channels=16, sample_height=64, sample_width=96)
SCHED = dict(_class_name="CogVideoXDDIMScheduler", _diffusers_version="0.31.0.dev0", beta_end=0.012, beta_schedule="scaled_linear", beta_start=0.00085,
             clip_sample=False, num_train_timesteps=1000, prediction_type="v_prediction", rescale_betas_zero_snr=True, set_alpha_to_one=True,
             snr_shift_scale=1.0, steps_offset=0, timestep_spacing="trailing", trained_betas=None)


def _write_tree(root):
    from safetensors.torch import save_file
    from tokensgen_amd import compat
    from tokensgen_amd.pca import PCA
    from tokensgen_amd.resampler import Resampler
    from tokensgen_amd.transformer import CogVideoXTransformer3DModel
    from tokensgen_amd.vae import AutoencoderKLCogVideoX
    base, to2v, t2to = (os.path.join(root, n) for n in ("CogVideoX-5b", "TokensGen-To2V", "TokensGen-T2To"))
    full = {k: v.to(BF) for k, v in O.make_state_dict(DIT, n_vip_dim=128, seed=31).items()}
    plain = {k: v for k, v in full.items() if "vip_" not in k}
    for d, sd, cfg in ((os.path.join(base, "transformer"), plain, DIT),
                       (os.path.join(t2to, "transformer"), {k: v.to(BF) for k, v in O.make_state_dict(dict(DIT, patch_size=1), None, seed=32).items()},
                        dict(DIT, patch_size=1))):
        os.makedirs(d)
        with open(os.path.join(d, "config.json"), "w") as f:
            json.dump(dict(cfg, _class_name="CogVideoXTransformer3DModel"), f)
        save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(d, "diffusion_pytorch_model.safetensors"))
    os.makedirs(os.path.join(base, "scheduler"))
    with open(os.path.join(base, "scheduler", "scheduler_config.json"), "w") as f:
        json.dump(SCHED, f)
    vae = AutoencoderKLCogVideoX(block_out_channels=VAE["block_out_channels"], layers_per_block=1, sample_height=64, sample_width=96, device=DEV)
    vsd = V.make_state_dict(VAE, seed=33)
    vae.load_state_dict(vsd)
    vae.save_pretrained(os.path.join(base, "vae"))
    os.makedirs(to2v)
    torch.save({k: v.float() for k, v in full.items() if "vip_" in k}, os.path.join(to2v, "vip.pt"))      # save_vip_layers format (fp32, :627-634)
    rs = Resampler(**VIP["resampler_params"], device=DEV)
    rsd = RR.make_state_dict(VIP["resampler_params"], seed=34)
    rs.load_state_dict(rsd)
    rs.save_pretrained(os.path.join(to2v, "resampler"))
    compat.ensure_pca_module()
    g = torch.Generator().manual_seed(35)
    pca = PCA(None).fit(torch.randn(600, 128, generator=g) * torch.linspace(2.0, 0.1, 128))
    torch.save(pca, os.path.join(to2v, "pca.pt"))
    torch.save(torch.randn(1, 32, generator=g) * 0.5, os.path.join(to2v, "mean.pt"))
    torch.save(torch.rand(1, 32, generator=g) + 0.5, os.path.join(to2v, "std.pt"))
    return base, to2v, t2to, full, vsd, rsd, pca


class _Args(dict):
    __getattr__ = dict.__getitem__


@pytest.mark.timeout(600)
def test_entry_script_loading_sequence(tmp_path):
    import sys
    from tokensgen_amd import compat
    base, to2v, t2to, full, vsd, rsd, pca = _write_tree(str(tmp_path))
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "longvgen" or k.startswith("longvgen.")}
    compat.install_longvgen_alias(force=True)
    try:
        # ---- the script's own import lines (infer_cogvideo_mp_fifo.py:63-71) ----
        from longvgen.fifo_sampling import cogvideo_fifo_mp_v2
        from longvgen.models import CogVideoXTransformer3DModel
        from longvgen.pipeline import LongVGenCogVideoXPipeline, MPFIFOVideoIPAdapterCogVideoXPipeline
        from longvgen.schedulers import CogVideoXDPMScheduler
        from longvgen.video_ipadapter import Resampler
        args = _Args(pretrained_model_name_or_path=base, pretrained_resampler_name_or_path=to2v, pretrained_2nd_stage_model_name_or_path=t2to,
                     use_vip=True, video_ipadapter_params=_Args(VIP), longvgen_pca=os.path.join(to2v, "pca.pt"),
                     longvgen_mean=os.path.join(to2v, "mean.pt"), longvgen_std=os.path.join(to2v, "std.pt"))
        dtype = load_dtype = torch.bfloat16
        device = torch.device(DEV)
        # ---- init_pipeline body, :139-183 ----
        vip_params, vip_path = args.video_ipadapter_params, args.pretrained_resampler_name_or_path
        transformer = CogVideoXTransformer3DModel.from_pretrained(args.pretrained_model_name_or_path, subfolder="transformer", torch_dtype=load_dtype,
                                                                  revision=args.get("revision", None), variant=args.get("variant", None)).to(device)
        transformer.set_vip_layers(vip_path, **vip_params)
        transformer = transformer.to(dtype)
        resampler = Resampler.from_pretrained(vip_path, subfolder="resampler", torch_dtype=dtype).to(device)
        resampler.set_pca(args.get("longvgen_pca", None))
        pipe = MPFIFOVideoIPAdapterCogVideoXPipeline.from_pretrained(args.pretrained_model_name_or_path, transformer=transformer, resampler=resampler,
                                                                     torch_dtype=dtype)
        pipe.scheduler = CogVideoXDPMScheduler.from_config(pipe.scheduler.config, timestep_spacing="trailing")
        pipe.to(device)
        pipe.vae.enable_slicing()
        pipe.vae.enable_tiling()
        # ---- :220-233 ----
        tokens_transformer = CogVideoXTransformer3DModel.from_pretrained(args.pretrained_2nd_stage_model_name_or_path, subfolder="transformer", torch_dtype=dtype)
        pipe_2nd = LongVGenCogVideoXPipeline.from_pretrained(args.pretrained_model_name_or_path, transformer=tokens_transformer, torch_dtype=dtype)
        pipe_2nd.scheduler = CogVideoXDPMScheduler.from_config(pipe_2nd.scheduler.config, timestep_spacing="trailing")
        pipe_2nd.to("cuda:0")
    finally:
        for k in [k for k in sys.modules if k == "longvgen" or k.startswith("longvgen.")]:
            del sys.modules[k]
        sys.modules.update({k: v for k, v in saved.items() if v is not None})
    # what was loaded is what was written: every parameter incl. the vip.pt ones, the resampler, the VAE, the PCA buffers
    got = transformer.state_dict()
    assert sorted(got) == sorted(full)
    for k, v in full.items():
        assert torch.equal(got[k].cpu(), v), k
    for k, v in rsd.items():
        assert torch.equal(resampler.state_dict()[k].cpu(), v.to(BF)), k
    for k, v in vsd.items():
        assert torch.equal(pipe.vae.state_dict()[k].cpu(), v.to(BF)), k
    assert torch.equal(resampler.pca.components_, pca.components_) and resampler._pca_dev[0].shape == (16, 128)
    assert pipe.scheduler.config.timestep_spacing == "trailing" and pipe.scheduler.config.prediction_type == "v_prediction"
    assert pipe.vae.use_tiling and pipe.vae.use_slicing and pipe.resampler is resampler
    with pytest.raises(IOError):
        transformer.set_vip_layers(os.path.join(str(tmp_path), "nowhere"), **VIP)          # :611-612
    # ---- one short run of each stage on the loaded objects (gen.yaml flow: T2To -> tokens -> To2V base stage -> FIFO) ----
    g = torch.Generator().manual_seed(36)
    pe, ne = torch.randn(1, 8, 64, generator=g).to(DEV, BF), torch.randn(1, 8, 64, generator=g).to(DEV, BF)
    rp = vip_params["resampler_params"]
    emb = pipe_2nd(prompt_embeds=pe, negative_prompt_embeds=ne, height=rp["num_height_queries"], width=rp["num_width_queries"],
                   num_frames_per_chunk=rp["num_temporal_queries"], num_chunks=1, use_dynamic_cfg=True, guidance_scale=6.0, num_inference_steps=3,
                   generator=torch.Generator().manual_seed(42), longvgen_mean=args.longvgen_mean, longvgen_std=args.longvgen_std,
                   longvgen_pca=args.longvgen_pca).frames
    assert emb.shape == (1, 4, 128, 2, 3) and torch.isfinite(emb).all()
    base_out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image_embeddings=emb, height=8, width=12, num_frames_per_chunk=49, num_chunks=1,
                    num_inference_steps=52, guidance_scale=6.0, video_ipadapter_scale=vip_params["scale"], output_type="latent")
    assert base_out.fifo_latents.shape == (1, 52, 16, 4, 6) and torch.isfinite(base_out.fifo_latents).all()
    _, video, _ = cogvideo_fifo_mp_v2([pipe], base_out)
    assert video.shape == (1, 13, 16, 4, 6) and torch.isfinite(video).all()
    # and the Resampler branch with the PCA filter on (edit-style front end), against the oracle composition
    f32 = np.float32
    tok = torch.randn(1, 5, 24, 128, generator=g).to(BF)
    img = O.rope_3d(64, np.arange(5, dtype=f32), np.arange(4, dtype=f32), np.arange(6, dtype=f32))
    smp = O.rope_3d(64, np.linspace(1000, 1005, 4, endpoint=False, dtype=f32), np.linspace(0, 4, 2, endpoint=False, dtype=f32),
                    np.linspace(0, 6, 3, endpoint=False, dtype=f32))
    ref = RR.resampler_forward({k: v.to(BF) for k, v in rsd.items()}, rp, tok, img, smp, pca=(pca.components_, pca.mean_))
    y = resampler(tok.to(DEV), image_rotary_emb=img, sampling_rotary_emb=smp)
    assert ((y.float().cpu() - ref.float()).norm() / ref.float().norm()).item() < 2e-2
