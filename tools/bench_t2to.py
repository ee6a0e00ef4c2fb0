#!/usr/bin/env python3
"""T2To stage at the shipped size (gen.yaml, BASELINE config 3): patch-1 CogVideoX-5B DiT over 24 chunks x 4 x 8 x 12 condensed
tokens (N = 226 + 9216), CFG batch 2, DPM-solver++ with dynamic CFG, PCA tail.  Random-init weights, synthetic embeddings.
Prints one JSON line: ms per denoising step, the tail's time and the achieved MFMA rate (135.8 TFLOP per sample-step)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokensgen_amd.pca import PCA  # noqa: E402
from tokensgen_amd.pipeline_t2to import LongVGenCogVideoXPipeline  # noqa: E402
from tokensgen_amd.scheduler import CogVideoXDPMScheduler  # noqa: E402
from tokensgen_amd.transformer import CogVideoXTransformer3DModel  # noqa: E402


def main(steps=4, layers=42):
    dev = torch.device("cuda")
    m = CogVideoXTransformer3DModel(num_attention_heads=48, attention_head_dim=64, num_layers=layers, time_embed_dim=512, text_embed_dim=4096,
                                    patch_size=1, use_rotary_positional_embeddings=True, device=dev)
    g = torch.Generator(device=dev).manual_seed(1)
    for name, t in m._fused.items():
        t.copy_(torch.randn(t.shape, generator=g, device=dev, dtype=torch.float32) * 0.02)
        if name.endswith(("ln", "qknorm")):
            t[0::2] += 1.0
    sched = CogVideoXDPMScheduler(prediction_type="v_prediction", rescale_betas_zero_snr=True, snr_shift_scale=1.0, timestep_spacing="trailing")
    pipe = LongVGenCogVideoXPipeline(m, sched)
    cg = torch.Generator().manual_seed(2)
    pe, ne = torch.randn(1, 226, 4096, generator=cg) * 0.1, torch.randn(1, 226, 4096, generator=cg) * 0.1
    pca = PCA()
    q, _ = torch.linalg.qr(torch.randn(3072, 16, generator=cg))
    pca.register_buffer("mean_", torch.randn(1, 3072, generator=cg) * 0.1); pca.register_buffer("components_", q.t().contiguous())
    mean, std = torch.randn(1, 16, generator=cg), torch.rand(1, 16, generator=cg) + 0.5
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, height=8, width=12, num_frames_per_chunk=4, num_chunks=24, use_dynamic_cfg=True,
              guidance_scale=6.0, longvgen_mean=mean, longvgen_std=std, longvgen_pca=pca)
    pipe(num_inference_steps=1, generator=torch.Generator().manual_seed(3), **kw)          # warm-up (workspace, tables)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pipe(num_inference_steps=steps, generator=torch.Generator().manual_seed(3), **kw).frames
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert out.shape == (1, 96, 3072, 8, 12) and torch.isfinite(out).all()
    ms = 1e3 * dt / steps
    print(json.dumps({"stage": "T2To", "tokens": 226 + 9216, "layers": layers, "steps": steps, "ms_per_step": ms,
                      "tflops": 2 * 135.8 * (layers / 42.0) / (ms * 1e-3), "note": "includes the CPU-generator draws + H2D of the noise, and the PCA tail once"}))


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
