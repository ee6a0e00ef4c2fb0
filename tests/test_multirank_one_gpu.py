"""GPU: the multi-rank branches of the product executed with the real HIP kernels — four ranks on ONE MI355X over gloo (tests/rank_worker_gpu_multi.py): CFG-parallel base
stage, FIFO iterations split by guidance branch + round-robin windows, chunk-sharded decode gather — each rank bitwise equal to its own no-process-group run.
(The N > 1 RCCL runs belong to the driver's multi-GPU node; RCCL itself is exercised by tests/test_rccl_gpu.py on a one-rank group.)"""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_four_ranks_on_one_gpu_over_gloo(tmp_path):
    sys.path.insert(0, ROOT)
    from tokensgen_amd.runtime import launch
    env = dict(os.environ, TG_DIST_TIMEOUT_S="300")
    launch(4, [sys.executable, os.path.join(ROOT, "tests", "rank_worker_gpu_multi.py"), str(tmp_path)], env=env)
    msgs = [(tmp_path / f"rank{r}.txt").read_text() if (tmp_path / f"rank{r}.txt").exists() else "(no result file)" for r in range(4)]
    print("\n".join(msgs))
    for r, msg in enumerate(msgs):
        assert msg.startswith("ok "), f"rank {r}: {msg}\n(all ranks: {msgs})"
        for part in ("fifo_latents", "fifo_decode", "split_counts", "whole_windows"):
            assert part in msg


@pytest.mark.timeout(900)
def test_ddp_two_ranks_on_one_gpu_over_gloo(tmp_path):
    """BASELINE config 5's data-parallel leg with the real kernels (tests/rank_worker_gpu_ddp.py): two ranks, one micro-batch each, bucketed gradient all-reduce overlapping the
    backward, collective verdict, clip + AdamW — parameters after the step bitwise equal to one process accumulating the two micro-batches."""
    sys.path.insert(0, ROOT)
    from tokensgen_amd.runtime import launch
    launch(2, [sys.executable, os.path.join(ROOT, "tests", "rank_worker_gpu_ddp.py"), str(tmp_path)], env=dict(os.environ, TG_DIST_TIMEOUT_S="300"))
    msgs = [(tmp_path / f"rank{r}.txt").read_text() if (tmp_path / f"rank{r}.txt").exists() else "(no result file)" for r in range(2)]
    print("\n".join(msgs))
    for r, msg in enumerate(msgs):
        assert msg.startswith("ok "), f"rank {r}: {msg}\n(all ranks: {msgs})"
        for part in ("stepped", "loss", "params", "buckets_during_backward"):
            assert part in msg


@pytest.mark.timeout(900)
def test_t2to_stage_and_token_encode_three_ranks_on_one_gpu_over_gloo(tmp_path):
    """The two multi-rank branches the four-rank test does not reach, with the real kernels (tests/rank_worker_gpu_stages.py): the T2To stage CFG-parallel and the condensed-token
    encode sharded by chunk over three ranks (2 / 1 / 1 chunks, separate guidance) — frames, tokens and generator states bitwise equal to the no-process-group run on every rank."""
    sys.path.insert(0, ROOT)
    from tokensgen_amd.runtime import launch
    launch(3, [sys.executable, os.path.join(ROOT, "tests", "rank_worker_gpu_stages.py"), str(tmp_path)], env=dict(os.environ, TG_DIST_TIMEOUT_S="300"))
    msgs = [(tmp_path / f"rank{r}.txt").read_text() if (tmp_path / f"rank{r}.txt").exists() else "(no result file)" for r in range(3)]
    print("\n".join(msgs))
    for r, msg in enumerate(msgs):
        assert msg.startswith("ok "), f"rank {r}: {msg}\n(all ranks: {msgs})"
        for part in ("t2to_frames", "t2to_generator", "tokens", "token_generator", "halves", "chunks"):
            assert part in msg
