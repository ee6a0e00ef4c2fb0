"""CPU (gloo, 2 ranks): worker-failure surfacing of the multi-GPU runtime (tokensgen_amd/runtime.py) — the reference's parent blocks forever in
`output_queue.get()` when a worker dies (cogvideo_sampling_mp_fifo.py:308-311); here (a) a rank that raises inside an iteration makes every rank
raise RankFailure in that same iteration, (b) a rank that dies gets the others terminated by the launcher with its exit code reported,
(c) a rank that silently stops makes its peers' next collective time out."""
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "rank_worker.py")


def _read(d, r):
    with open(os.path.join(d, f"rank{r}.txt")) as f:
        return f.read()


@pytest.mark.timeout(280)
def test_exception_in_one_rank_raises_on_every_rank(tmp_path):
    from tokensgen_amd.runtime import launch
    launch(2, [sys.executable, WORKER, "raise", str(tmp_path)])
    for r in (0, 1):
        msg = _read(tmp_path, r)
        assert msg.startswith("RankFailure after 4 iterations"), msg          # iterations 0..3: raised in the iteration the failure happened
        assert "rank 1 failed: ValueError: injected failure in window of iteration 3" in msg


@pytest.mark.timeout(120)
def test_dead_rank_terminates_the_job(tmp_path):
    from tokensgen_amd.runtime import launch
    t0 = time.time()
    with pytest.raises(RuntimeError, match=r"rank 1 of 2 exited with code 3; the other ranks were terminated"):
        launch(2, [sys.executable, WORKER, "die", str(tmp_path)])
    assert time.time() - t0 < 60          # rank 0 sleeps 120 s: it was terminated, not waited for


@pytest.mark.timeout(120)
def test_silent_rank_times_out(tmp_path):
    from tokensgen_amd.runtime import launch
    try:
        launch(2, [sys.executable, WORKER, "silent", str(tmp_path)])
    except RuntimeError:
        pass                              # gloo may tear rank 1 down when rank 0 leaves; what matters is rank 0's report
    assert _read(tmp_path, 0).startswith("timeout surfaced after"), _read(tmp_path, 0)


@pytest.mark.timeout(120)
def test_gradient_all_reduce_buckets_two_ranks(tmp_path):
    """Training step, DDP leg (train_cogvideo_to2v.py:1157-1164): optim.GradSync on gloo — buckets are launched only once the backward has passed
    their end, every element is reduced exactly once, and the result is the rank average (scale folded in by the accumulation)."""
    from tokensgen_amd.runtime import launch
    launch(2, [sys.executable, WORKER, "gradsync", str(tmp_path)])
    assert _read(tmp_path, 0) == "ok" and _read(tmp_path, 1) == "ok"


@pytest.mark.timeout(120)
def test_invalid_attention_backward_on_one_rank_discards_the_window_on_every_rank(tmp_path):
    """Training step, failure leg (train_cogvideo_to2v.py:1995-2021 has no counterpart: autograd cannot fail this way): the one-kernel attention backward's
    sticky status word set on ONE rank — on the window's last micro-step, with gradient buckets already handed to the exchange, and in the middle of a
    window — makes EVERY rank drain the exchange, zero the gradient arena, roll the micro counter back to the window's start and raise; feeding the window
    again then steps the optimizer once with the rank sum (To2VTrainStep._apply_or_discard; gloo, 2 ranks)."""
    from tokensgen_amd.runtime import launch
    launch(2, [sys.executable, WORKER, "verdict", str(tmp_path)])
    assert _read(tmp_path, 0) == "ok" and _read(tmp_path, 1) == "ok", (_read(tmp_path, 0), _read(tmp_path, 1))


@pytest.mark.timeout(180)
def test_weight_broadcast_two_ranks(tmp_path):
    """runtime.broadcast_weights (north_star: "RCCL broadcast of weights"; replaces the reference's whole-pipeline CUDA-IPC pickling,
    cogvideo_sampling_mp_fifo.py:195-221) on gloo: rank 1 ends up with rank 0's state dict bit for bit — through the explicit call on the fused
    storages (large tensors one by one, small ones in coalesced buckets) and through from_pretrained(broadcast=True), where rank 1's checkpoint
    directory holds config.json ONLY (it cannot have read the weights from disk)."""
    import json
    import torch
    from safetensors.torch import save_file
    from oracle import dit_ref as O
    from tokensgen_amd.runtime import launch
    cfg = dict(num_attention_heads=2, attention_head_dim=64, num_layers=2, time_embed_dim=128, text_embed_dim=64, use_rotary_positional_embeddings=True)
    sd = {k: v.to(torch.bfloat16).contiguous() for k, v in O.make_state_dict(dict(cfg, patch_size=2, in_channels=16, out_channels=16), n_vip_dim=128, seed=77).items()
          if "vip_" not in k}
    for r in (0, 1):
        d = tmp_path / f"ckpt{r}"
        d.mkdir()
        (d / "config.json").write_text(json.dumps(dict(cfg, _class_name="CogVideoXTransformer3DModel")))
    save_file(sd, str(tmp_path / "ckpt0" / "diffusion_pytorch_model.safetensors"))
    from oracle import resampler_ref as RR
    rcfg = dict(dim=128, depth=2, dim_head=64, heads=2, num_height_queries=2, num_width_queries=3, num_temporal_queries=4, embedding_dim=128, output_dim=128, ff_mult=4)
    for r in (0, 1):
        d = tmp_path / f"rs{r}"
        d.mkdir()
        (d / "config.json").write_text(json.dumps(dict(rcfg, _class_name="Resampler")))
    save_file({k: v.to(torch.bfloat16).contiguous() for k, v in RR.make_state_dict(rcfg, seed=78).items()}, str(tmp_path / "rs0" / "diffusion_pytorch_model.safetensors"))
    launch(2, [sys.executable, WORKER, "broadcast", str(tmp_path)])
    assert _read(tmp_path, 0) == "ok" and _read(tmp_path, 1) == "ok"


def test_arena_order_is_backward_order_and_keeps_qkv_adjacent():
    from tokensgen_amd.optim import arena_order
    names = [f"transformer_blocks.{i}.{n}" for i in range(3) for n in
             ("vip_norm1.linear.weight", "attn1.processor.vip_to_v.weight", "attn1.processor.vip_to_q.weight", "attn1.processor.vip_to_k.weight",
              "attn1.processor.vip_to_k.bias", "attn1.processor.vip_to_q.bias", "attn1.processor.vip_to_v.bias")]
    names += ["patch_embed.vip_proj.weight", "resampler.latents", "resampler.proj_in.weight"]
    order = arena_order(names, 3)
    assert order[0].startswith("transformer_blocks.2.") and order[7].startswith("transformer_blocks.1.") and order[14].startswith("transformer_blocks.0.")
    assert [n.split("processor.")[1] for n in order[:6]] == ["vip_to_q.weight", "vip_to_k.weight", "vip_to_v.weight", "vip_to_q.bias", "vip_to_k.bias", "vip_to_v.bias"]
    assert order[-3:] == ["patch_embed.vip_proj.weight", "resampler.latents", "resampler.proj_in.weight"]
