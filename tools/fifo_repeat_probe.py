#!/usr/bin/env python3
"""Is a single-process tiny To2V run (base stage + FIFO, tests/rank_worker_gpu.py's) bitwise repeatable while other processes share the GPU?  Runs it N times, hashes the base
stage's outputs and the queue after every FIFO iteration, and reports the first place a repeat leaves the first run.   python tools/fifo_repeat_probe.py N TAG"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.argv = [sys.argv[0], "/tmp"] + sys.argv[1:]
import torch  # noqa: E402
import rank_worker_gpu as W  # noqa: E402
from tokensgen_amd import fifo  # noqa: E402
n, tag = int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else "p"
sha = lambda t: hashlib.sha256(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:8]
torch.cuda.set_device(0)
m, sd, pipe = W.build()
BF, DEV = W.BF, W.DEV


def run():
    g = torch.Generator().manual_seed(5)
    H, Wd, nf, T, chunks = 4, 6, 13, 52, 2
    lat0 = torch.randn(1, nf, 16, H, Wd, generator=g).to(BF)
    pe, ne = torch.randn(1, 8, 64, generator=g).to(BF), torch.randn(1, 8, 64, generator=g).to(BF)
    emb = torch.randn(1, 4 * chunks, 128, 2, 3, generator=g).to(BF)
    out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image_embeddings=emb, height=H * 8, width=Wd * 8, num_chunks=chunks, num_inference_steps=T,
               latents=lat0, step_noise=lambda i: W._noise(i, 5, (nf, 2, 16, H, Wd)), output_type="latent")
    trail = [("base_fifo_latents", sha(out.fifo_latents)), ("base_orig", sha(out.orig_latents))]
    hook = lambda i, n_iter, lat, x0q: trail.append((f"iter{i}", sha(lat) + sha(x0q)))
    res = fifo.cogvideo_fifo_mp_v2([pipe], out, step_noise_fn=W._noise, tail_noise_fn=lambda i, shape: W._noise(i, 97, shape), decode_chunk_fn=W.fake_decode, iteration_hook=hook)
    trail.append(("video", sha(res[1])))
    return trail


ref = run()
bad = 0
for r in range(1, n):
    t = run()
    if t != ref:
        bad += 1
        first = next(i for i, (a, b) in enumerate(zip(t, ref)) if a != b)
        print(f"[{tag}] repeat {r}: first difference at {ref[first][0]} (entry {first} of {len(ref)}); later entries equal again: {sum(a == b for a, b in zip(t[first:], ref[first:]))}", flush=True)
print(f"[{tag}] FIFO_REPEAT {bad} of {n - 1} repeats differed; final {ref[-1]}")
