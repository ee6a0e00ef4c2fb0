"""CPU: host logic of the product (no kernels run): RoPE tables, scheduler tables, packing of the fused weight
storages, and that the C-ABI library exports every symbol include/tokensgen_hip.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import dit_ref as O
from oracle import scheduler_ref as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rope_tables_bit_exact_vs_oracle():
    from tokensgen_amd import rope as R
    f32 = np.float32
    for gt, gh, gw in [(np.arange(13, dtype=f32), np.arange(30, dtype=f32), np.arange(45, dtype=f32)),
                       (np.linspace(1000, 1016.25, 5, dtype=f32), np.linspace(0, 30, 8, endpoint=False, dtype=f32),
                        np.linspace(0, 45, 12, endpoint=False, dtype=f32)),
                       (np.arange(13, dtype=f32) + f32(137.0), np.arange(4, dtype=f32), np.arange(6, dtype=f32))]:
        a, b = R.rope_3d(64, gt, gh, gw), O.rope_3d(64, gt, gh, gw)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    a, b = R.rope_3d_crop(64, (0, 0, 0), (13, 30, 45), (13, 30, 45)), O.rope_3d_crop(64, (0, 0, 0), (13, 30, 45), (13, 30, 45))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # T2To split 52/6/6 (pipeline_cogvideox_t2to.py:557-559)
    a = R.rope_3d(64, np.arange(4, dtype=f32), np.arange(8, dtype=f32), np.arange(12, dtype=f32), 52, 6, 6)
    b = O.rope_3d(64, np.arange(4, dtype=f32), np.arange(8, dtype=f32), np.arange(12, dtype=f32), 52, 6, 6)
    assert torch.equal(a[0], b[0]) and a[0].shape == (384, 64)


def test_scheduler_tables_vs_oracle(golden_dir):
    from tokensgen_amd.scheduler import CogVideoXDPMScheduler, dpm_coef_row
    s = CogVideoXDPMScheduler(prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=1.0,
                              timestep_spacing="trailing")
    s.set_timesteps(52)
    g = torch.load(os.path.join(golden_dir, "scheduler.pt"), weights_only=False)
    assert torch.equal(s.timesteps, g["timesteps"])
    assert (s.alphas_cumprod - g["alphas_cumprod"]).abs().max() < 1e-15 and s.alphas_cumprod[-1] == 0
    assert torch.equal(s.betas, g["betas"])
    _, ac = S.alphas_cumprod()
    ts = s.timesteps.tolist()
    for i in range(1, 51):
        row = dpm_coef_row(s.alphas_cumprod.numpy(), ts[i], ts[i + 1], ts[i - 1], True)
        c = S.step_coefficients(ac, ts[i], ts[i + 1], ts[i - 1])
        ref = [float(c[k]) for k in ("sa", "sb", "m1", "m2", "m3", "m4", "mn")]
        assert np.allclose(row[:7], ref, rtol=1e-12, atol=1e-14) and row[7] == 1.0
    # edges: t=999 first step (no x0 yet), prev_t=-1 final step
    assert dpm_coef_row(s.alphas_cumprod.numpy(), 999, 980, None, False)[2] == 0.0
    last = dpm_coef_row(s.alphas_cumprod.numpy(), 18, -1, 37, True)
    assert last[2] == 0.0 and last[3] == -1.0 and last[6] == 0.0 and last[7] == 0.0
    with pytest.raises(NotImplementedError):
        CogVideoXDPMScheduler(beta_schedule="vip_1")
    for sp, n in (("leading", 50), ("linspace", 7)):
        CogVideoXDPMScheduler(prediction_type="v_prediction", timestep_spacing=sp).set_timesteps(n)


def test_transformer_state_dict_contract():
    """Reference key names, fused storages shared with the named parameters, vip layers added later."""
    from tokensgen_amd.transformer import CogVideoXTransformer3DModel, VideoIPAdapterCogVideoXAttnProcessor2_0
    cfg = dict(num_attention_heads=2, attention_head_dim=64, num_layers=2, patch_size=2, time_embed_dim=128,
               text_embed_dim=64, in_channels=16, out_channels=16)
    m = CogVideoXTransformer3DModel(num_attention_heads=2, attention_head_dim=64, num_layers=2, time_embed_dim=128,
                                    text_embed_dim=64, use_rotary_positional_embeddings=True, device="cpu")
    assert sorted(m.state_dict()) == sorted(O.make_state_dict(cfg, None, seed=1))
    base = {k: v.to(torch.bfloat16) for k, v in O.make_state_dict(cfg, None, seed=1).items()}
    m.load_state_dict(base, strict=True)
    vipcfg = dict(length=30, func_type="1", scale=[0.6],
                  resampler_params=dict(output_dim=128, num_height_queries=2, num_width_queries=3, num_temporal_queries=4))
    m.set_vip_layers(None, **vipcfg)
    sd = {k: v.to(torch.bfloat16) for k, v in O.make_state_dict(cfg, 128, seed=1).items()}
    assert sorted(m.state_dict()) == sorted(sd)
    # set_vip_layers initialises the vip projections from the base ones (cogvideox_transformer_3d.py:207-218)
    assert torch.equal(m.state_dict()["transformer_blocks.1.attn1.processor.vip_to_k.weight"], base["transformer_blocks.1.attn1.to_k.weight"])
    # ... and keeps the already-loaded modulation weights when the fused matrix is re-laid-out
    assert torch.equal(m.state_dict()["transformer_blocks.1.norm2.linear.weight"], base["transformer_blocks.1.norm2.linear.weight"])
    m.load_state_dict(sd, strict=True)
    D = 128
    per, out_base, total = m._mod_cols()
    assert (per, out_base, total) == (18 * D, 36 * D, 38 * D)
    assert torch.equal(m._fused["mod.w"][per + 9 * D: per + 15 * D], sd["transformer_blocks.1.norm2.linear.weight"])
    assert torch.equal(m._fused["mod.w"][per + 6 * D: per + 9 * D], sd["transformer_blocks.1.vip_norm1.linear.weight"])
    assert torch.equal(m._fused["l0.vqkv.w"][2 * D:], sd["transformer_blocks.0.attn1.processor.vip_to_v.weight"])
    assert torch.equal(m._fused["patch.w"], sd["patch_embed.proj.weight"].reshape(D, 64))
    assert m._fused["proj_out.w"].shape == (128, D) and (m._fused["proj_out.w"][64:] == 0).all()
    procs = [mod for mod in m.modules() if mod.__class__.__name__ == "VideoIPAdapterCogVideoXAttnProcessor2_0"]
    assert len(procs) == 2 and isinstance(procs[0], VideoIPAdapterCogVideoXAttnProcessor2_0) and procs[0].scale == [0.6]
    # vip.pt round trip (save_vip_layers / set_vip_layers(vip_ckpt_dir), cogvideox_transformer_3d.py:603-634)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        m.save_vip_layers(d)
        saved = torch.load(os.path.join(d, "vip.pt"), weights_only=True)
        assert all("vip_" in k for k in saved) and len(saved) == 2 * 18 + 2
        m2 = CogVideoXTransformer3DModel(num_attention_heads=2, attention_head_dim=64, num_layers=2, time_embed_dim=128,
                                         text_embed_dim=64, use_rotary_positional_embeddings=True, device="cpu")
        m2.load_state_dict(base, strict=True)
        m2.set_vip_layers(d, **vipcfg)
        sd2 = m2.state_dict()
        assert all(torch.equal(sd2[k].float(), v) for k, v in saved.items())
        assert torch.equal(m2._fused["l1.vqkv.w"], m._fused["l1.vqkv.w"])
        assert torch.equal(sd2["transformer_blocks.0.norm1.linear.weight"], base["transformer_blocks.0.norm1.linear.weight"])
        with pytest.raises(IOError):
            m2.set_vip_layers(os.path.join(d, "missing"), **vipcfg) if os.makedirs(os.path.join(d, "missing")) is None else None


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU — never route through torch/CPU."""
    from tokensgen_amd.transformer import CogVideoXTransformer3DModel
    m = CogVideoXTransformer3DModel(num_attention_heads=2, attention_head_dim=64, num_layers=1, time_embed_dim=128,
                                    text_embed_dim=64, use_rotary_positional_embeddings=True, device="cpu")
    z = torch.zeros
    with pytest.raises(RuntimeError, match="GPU"):
        m(z(1, 13, 16, 4, 6, dtype=torch.bfloat16), z(1, 8, 64, dtype=torch.bfloat16), z(1, dtype=torch.long),
          image_rotary_emb=(z(78, 64), z(78, 64)))
    with pytest.raises(NotImplementedError):
        CogVideoXTransformer3DModel(num_attention_heads=2, attention_head_dim=32, use_rotary_positional_embeddings=True, device="cpu")


def test_c_abi_exports_every_declared_symbol():
    from tokensgen_amd import lib as L
    hdr = open(os.path.join(ROOT, "include", "tokensgen_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tg_[a-z0-9_]+)\s*\(", hdr))
    assert {"tg_gemm_bf16", "tg_attention_fwd", "tg_adaln_modulate", "tg_qk_layernorm_rope", "tg_cfg_dpm_step"} <= declared
    assert os.path.exists(L.LIB_PATH), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    so = ctypes.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(so, name), f"{name} declared in include/tokensgen_hip.h but not exported"
    lib = L.load()
    assert set(L.PROTOTYPES) | set(L.QUERIES) | set(L.OTHER_EXPORTS) == declared
    assert b"gfx950" in lib.tg_version()
    # ... and NOTHING ELSE: the library is built with -fvisibility=hidden, `nm -D` of its defined symbols == the header's declarations
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in "TWBDRV"}
    exported -= {"_init", "_fini", "__bss_start", "_edata", "_end"}
    exported = {n for n in exported if not n.startswith("__hip_")}          # the HIP fat-binary registration objects of every hipcc-built library
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))
    # the product sources read no environment variable, and the test-only cross-check kernels are not in the product library
    csrc = os.path.join(ROOT, "tokensgen_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".cpp", ".h")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f
    assert not os.path.exists(os.path.join(csrc, "attention_bwd_ref.hip"))
    # the debug knobs: defaults, set / get round trip, unknown names refused
    val = ctypes.c_long(-1)
    assert lib.tg_debug_get(b"TG_ATTN_PP_MIN_WG", ctypes.byref(val)) == 0 and val.value == 1024
    assert L.debug_set("TG_GEMM_W4", 0) == 1 and L.debug_set("TG_GEMM_W4", 1) == 0
    assert lib.tg_debug_set(b"TG_NO_SUCH_KNOB", 1) == -1 and b"unknown knob" in lib.tg_last_error_string()
    names = []
    while lib.tg_debug_knob_name(len(names)) is not None:
        names.append(lib.tg_debug_knob_name(len(names)).decode())
    assert names == ["TG_ATTN_PP_MIN_WG", "TG_ATTN_FIXEDM", "TG_ATTN_SPLIT", "TG_GEMM_W4", "TG_CONV_SPLITK", "TG_CONV_HALO", "TG_CONV_W4"]
    # argument validation happens before any launch, so it is testable without a GPU
    assert lib.tg_gemm_bf16(None, 0, 0, None, 0, None, None, 0, 0, 1, 128, 64, 1, 0, None, 0, 0, None, None) == -1
    assert b"null pointer" in lib.tg_last_error_string()
    buf = ctypes.create_string_buffer(4096)
    p = ctypes.addressof(buf) + (16 - ctypes.addressof(buf) % 16)
    assert lib.tg_gemm_bf16(p, 64, 0, p, 64, None, p, 100, 0, 4, 100, 64, 1, 0, None, 0, 0, None, None) == -2
    assert b"N%128" in lib.tg_last_error_string()
    # the epilogue enum of the header == the constants of the binding; the two training-step epilogues refuse shapes without the 4-wave kernel (no launch: testable here)
    raw = open(os.path.join(ROOT, "include", "tokensgen_hip.h")).read()
    enum = dict(re.findall(r"\b(TG_EPI_[A-Z_]+)\s*=\s*(\d+)", raw))
    assert {k: int(v) for k, v in enum.items()} == {"TG_EPI_BIAS": L.EPI_BIAS, "TG_EPI_BIAS_GELU": L.EPI_BIAS_GELU, "TG_EPI_BIAS_SILU": L.EPI_BIAS_SILU,
                                                    "TG_EPI_BIAS_GATE_RES": L.EPI_BIAS_GATE_RES, "TG_EPI_BIAS_KEEP_GELU": L.EPI_BIAS_KEEP_GELU,
                                                    "TG_EPI_BIAS_MUL_GELU_GRAD": L.EPI_BIAS_MUL_GELU_GRAD}
    for epi in (L.EPI_BIAS_KEEP_GELU, L.EPI_BIAS_MUL_GELU_GRAD):
        assert lib.tg_gemm_bf16(p, 64, 0, p, 64, None, p, 128, 0, 512, 128, 64, 1, epi, p, 128, 0, None, None) == -2 and b"4-wave kernel only" in lib.tg_last_error_string()
        assert lib.tg_gemm_bf16(p, 64, 0, p, 64, None, p, 128, 0, 512, 128, 64, 1, epi, None, 128, 0, None, None) == -1


def test_c_abi_rejects_null_arguments_everywhere():
    """Error convention of the boundary: EVERY export, called with null pointers and zero sizes, returns a negative code and leaves a
    message in tg_last_error_string() — it validates before it touches HIP, so this holds on a box without a GPU, and nothing crosses the ABI as a
    crash.  Run in a child process so that a missing check would show up as a failed test, not as a dead test session."""
    import subprocess
    import sys
    code = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
from tokensgen_amd import lib as L
lib = L.load()
bad = []
for name, argtypes in L.PROTOTYPES.items():
    if name == "tg_attention_bwd_probe_verdict":      # a predicate (0 / 1), not a launch
        assert lib.tg_attention_bwd_probe_verdict(None, 0) == 0
        continue
    args = [None if (t is C.c_void_p or (isinstance(t, type) and issubclass(t, C._Pointer))) else (0.0 if t is C.c_float else 0) for t in argtypes]
    rc = getattr(lib, name)(*args)
    if not (rc < 0 and lib.tg_last_error_string()):
        bad.append((name, rc))
print("checked", len(L.PROTOTYPES), "bad", bad)
assert not bad
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "bad []" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_longvgen_alias_and_vae_host_contract():
    import sys
    from tokensgen_amd import compat
    from tokensgen_amd.vae import AutoencoderKLCogVideoX
    from oracle import vae_ref as V
    assert "longvgen" not in sys.modules
    compat.install_longvgen_alias()
    try:
        from longvgen.models import CogVideoXTransformer3DModel, AutoencoderKLCogVideoX as A2
        from longvgen.schedulers import CogVideoXDPMScheduler
        from longvgen.fifo_sampling import cogvideo_fifo_mp_v2
        from longvgen.pipeline import MPFIFOVideoIPAdapterCogVideoXPipeline, LongVGenCogVideoXPipeline
        from longvgen.video_ipadapter import Resampler
        assert A2 is AutoencoderKLCogVideoX and callable(cogvideo_fifo_mp_v2) and callable(LongVGenCogVideoXPipeline) and callable(Resampler)
        with pytest.raises(ImportError):
            import longvgen.data  # noqa: F401  (out of scope: never silently stubbed)
    finally:
        for k in [k for k in sys.modules if k == "longvgen" or k.startswith("longvgen.") or k == "pca"]:
            del sys.modules[k]
    vae = AutoencoderKLCogVideoX(device="cpu")
    assert (vae.tile_sample_min_height, vae.tile_sample_min_width, vae.tile_latent_min_height, vae.tile_latent_min_width) == (240, 360, 30, 45)
    assert vae.config.scaling_factor == 1.15258426 and vae._frame_batches(13, 2) == V.frame_batches(13, 2) and vae._frame_batches(49, 8) == V.frame_batches(49, 8)
    cfg = dict(block_out_channels=(64, 128, 128, 128), layers_per_block=1, latent_channels=16, sample_height=64, sample_width=96)
    v2 = AutoencoderKLCogVideoX(block_out_channels=cfg["block_out_channels"], layers_per_block=1, sample_height=64, sample_width=96, device="cpu")
    sd = V.make_state_dict(cfg, seed=1)
    v2.load_state_dict(sd)
    assert sorted(v2.state_dict()) == sorted(sd)
    w = v2._packed["decoder.conv_in.conv.weight"]                 # [Cout_pad, taps, Cin_pad], taps ordered (dt, dh, dw)
    assert w.shape == (128, 27, 64) and torch.equal(w[:128, 5, :16].float(), sd["decoder.conv_in.conv.weight"][:, :, 0, 1, 2].to(torch.bfloat16).float())
    assert (w[:, :, 16:] == 0).all()
    with pytest.raises(RuntimeError, match="GPU"):
        v2.decode(torch.zeros(1, 16, 3, 4, 6))
    with pytest.raises(NotImplementedError):
        AutoencoderKLCogVideoX(latent_channels=8, device="cpu")


def test_pca_holder_matches_reference_fit_and_unpickles_under_the_reference_module_name(golden_dir, tmp_path):
    """tokensgen_amd.pca.PCA: the fit/inverse algebra (pca.py:11-66), and a pickle written under the module name `pca` (what the
    reference's torch.save of its own object produces) loads through the alias — the T2To pipeline torch.load()s such a file."""
    import sys
    from tokensgen_amd import compat
    from tokensgen_amd.pca import PCA
    g = torch.Generator().manual_seed(11)
    X = torch.randn(200, 12, generator=g) * torch.linspace(3, 0.2, 12)
    p = PCA(n_components=5).fit(X)
    assert p.components_.shape == (5, 12) and p.mean_.shape == (1, 12)
    assert torch.allclose(p.components_ @ p.components_.t(), torch.eye(5), atol=1e-5)
    U, Sv, Vt = torch.linalg.svd(X - X.mean(0, keepdim=True), full_matrices=False)
    assert torch.allclose(p.components_.abs(), Vt[:5].abs(), atol=1e-5)
    full = PCA().fit(X)
    assert torch.allclose(full.inverse_transform(full.transform(X)), X, atol=1e-4)
    # the fixture's 16 rows came from the reference class fitted on seeded data: same sign convention (largest |u| positive)
    t = torch.load(os.path.join(golden_dir, "t2to_tiny.pt"), weights_only=False)
    assert t["pca_components16"].shape == (16, 3072) and torch.allclose(t["pca_components16"] @ t["pca_components16"].t(), torch.eye(16), atol=1e-4)
    compat.install_longvgen_alias()
    try:
        assert sys.modules["pca"].PCA is PCA
        PCA.__module__ = "pca"                      # what a pickle of the reference's own class records
        try:
            torch.save(p, tmp_path / "pca.pt")
        finally:
            PCA.__module__ = "tokensgen_amd.pca"
        assert b"tokensgen_amd" not in (tmp_path / "pca.pt").read_bytes()
        q = torch.load(tmp_path / "pca.pt", weights_only=False)
        assert type(q) is PCA and torch.equal(q.components_, p.components_)
    finally:
        for k in [k for k in sys.modules if k == "longvgen" or k.startswith("longvgen.") or k == "pca"]:
            del sys.modules[k]


def test_scheduler_from_config_like_the_entry_script(tmp_path):
    """infer_cogvideo_mp_fifo.py:129,232 — from_config(existing.config, timestep_spacing="trailing"), also from a json file."""
    import json
    from tokensgen_amd.scheduler import CogVideoXDPMScheduler
    a = CogVideoXDPMScheduler(prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=1.0)
    b = CogVideoXDPMScheduler.from_config(a.config, timestep_spacing="trailing")
    assert b.config.timestep_spacing == "trailing" and b.config.snr_shift_scale == 1.0 and torch.equal(a.alphas_cumprod, b.alphas_cumprod)
    b.set_timesteps(52)
    assert b.timesteps[0] == 999 and len(b.timesteps) == 52
    f = tmp_path / "scheduler_config.json"
    f.write_text(json.dumps(dict(_class_name="CogVideoXDPMScheduler", _diffusers_version="0.31.0", prediction_type="v_prediction",
                                 rescale_betas_zero_snr=True, snr_shift_scale=3.0, beta_schedule="scaled_linear")))
    c = CogVideoXDPMScheduler.from_config(str(f), timestep_spacing="trailing")
    assert c.config.snr_shift_scale == 3.0 and c.alphas_cumprod[-1] == 0.0


def test_resampler_checkpoint_roundtrip_and_set_pca(tmp_path):
    """Host side of the gen.yaml loading contract (infer_cogvideo_mp_fifo.py:113-118): Resampler.from_pretrained(dir, subfolder="resampler")
    reads what save_pretrained wrote; set_pca(path) unpickles a `pca.PCA` (resampler.py:201-207) and keeps its first 16 components; strict
    loading reports missing keys.  No kernel is launched (device "cpu" holds the tensors only)."""
    from oracle import resampler_ref as RR
    from tokensgen_amd import compat
    from tokensgen_amd.pca import PCA
    from tokensgen_amd.resampler import Resampler
    cfg = dict(dim=128, depth=2, dim_head=64, heads=2, num_height_queries=2, num_width_queries=3, num_temporal_queries=4, embedding_dim=128,
               output_dim=128, ff_mult=4, max_height_seq_len=4, max_width_seq_len=6, max_temporal_seq_len=5)
    sd = RR.make_state_dict(cfg, seed=1)
    r = Resampler(**cfg, device="cpu")
    r.load_state_dict(sd)
    assert sorted(r.expected_keys()) == sorted(sd)
    r.save_pretrained(str(tmp_path / "resampler"))
    r2 = Resampler.from_pretrained(str(tmp_path), subfolder="resampler", torch_dtype=torch.bfloat16, device="cpu").to("cpu")
    assert vars(r2.config) == vars(r.config)
    assert all(torch.equal(r2.state_dict()[k], v.to(torch.bfloat16)) for k, v in sd.items())
    with pytest.raises(RuntimeError, match="missing"):
        Resampler(**cfg, device="cpu").load_state_dict({k: v for k, v in sd.items() if k != "latents"})
    compat.ensure_pca_module()
    p = PCA(None).fit(torch.randn(300, 128, generator=torch.Generator().manual_seed(2)))
    torch.save(p, str(tmp_path / "pca.pt"))
    r2.set_pca(str(tmp_path / "pca.pt"))
    assert torch.equal(r2._pca_dev[0], p.components_[:16]) and torch.equal(r2._pca_dev[1], p.mean_.reshape(-1))
    r2.set_pca(None)
    assert r2.pca is None and r2._pca_dev is None
    with pytest.raises(ValueError, match="width"):
        Resampler(**dict(cfg, output_dim=256), device="cpu").set_pca(str(tmp_path / "pca.pt"))


def test_vae_param_shapes_match_the_diffusers_layout():
    from oracle import vae_ref as V
    from tokensgen_amd.vae import AutoencoderKLCogVideoX
    for cfg in (dict(block_out_channels=(128, 256, 256, 512), layers_per_block=3, latent_channels=16),
                dict(block_out_channels=(64, 128), layers_per_block=1, latent_channels=16)):
        vae = AutoencoderKLCogVideoX(block_out_channels=cfg["block_out_channels"], layers_per_block=cfg["layers_per_block"], device="cpu")
        assert vae.param_shapes() == {k: tuple(v.shape) for k, v in V.make_state_dict(cfg).items()}


def test_stratified_timestep_sampling_per_rank():
    """train_cogvideo_to2v.py:1797-1813 (`use_explicit_uniform_sampling`): rank r draws from its own stratum, rank 0 also covers the remainder."""
    import torch
    from tokensgen_amd.train import sample_timesteps
    g = torch.Generator().manual_seed(0)
    for world in (1, 3, 8):
        interval, shift = 1000 // world, 1000 % (1000 // world)
        seen = []
        for r in range(world):
            t = sample_timesteps(4096, 1000, r, world, explicit_uniform=True, generator=g)
            lo, hi = (0, interval + shift) if r == 0 else (r * interval + shift, (r + 1) * interval + shift)
            assert t.dtype == torch.int64 and int(t.min()) >= lo and int(t.max()) < hi
            seen.append((int(t.min()), int(t.max())))
        assert seen[0][0] == 0 and seen[-1][1] == 999            # 4096 draws per stratum of <= 1000 values: the ends are hit
    t = sample_timesteps(8192, 1000, 5, 8, explicit_uniform=False, generator=g)
    assert int(t.min()) == 0 and int(t.max()) == 999
