"""GPU: the RCCL path executed on hardware.  The 8-GPU scaling runs belong to the driver; what a 1-GPU box CAN do is run every collective call
site of the package through a real `nccl` (= RCCL) process group of one rank on CUDA tensors — `init_process_group("nccl", device_id=...)`,
`all_gather_into_tensor` on bf16 views, `broadcast` of raw-byte views of the fused weight storages, async `all_reduce` buckets, `barrier` — and
check each against the no-process-group path bit for bit (tests/rank_worker_gpu.py), plus bench.py's `use_dist` branch.
Multi-rank logic (round-robin windows, rank-major gather layout, failure flags, weight broadcast to an empty rank) is covered on CPU with 2-3 gloo
ranks: tests/test_fifo_cpu.py, test_runtime_cpu.py, test_cfg_parallel_cpu.py."""
import json
import os
import subprocess
import sys

import pytest
from conftest import free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank_env():
    return dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
                HSA_ENABLE_IPC_MODE_LEGACY="0")


@pytest.mark.timeout(900)
def test_collective_call_sites_on_a_one_rank_rccl_group(tmp_path):
    """FIFO exchange + sharded decode gather + weight broadcast + gradient buckets + CFG-parallel, each bitwise equal to the no-`dist` path."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rank_worker_gpu.py"), str(tmp_path)], env=_rank_env(), capture_output=True, text=True,
                       timeout=850, cwd=ROOT)
    path = tmp_path / "rank0.txt"
    msg = path.read_text() if path.exists() else "(no result file)"
    assert r.returncode == 0 and msg.startswith("ok "), msg + "\n--- stderr ---\n" + r.stderr[-3000:]
    for part in ("fifo_latents", "fifo_decode", "broadcast", "gradsync", "cfg_parallel", "fifo_after_broadcast"):
        assert part in msg


@pytest.mark.timeout(900)
def test_bench_use_dist_branch_under_a_one_rank_group():
    """`bench.py --gpus 1` with RANK / WORLD_SIZE in the environment (what torch.distributed.run exports): init_distributed("nccl"), the per-step
    all_gather_into_tensor of the window outputs, barrier-fenced timing and the all_gather of per-rank times all execute (2 layers: a debug shape,
    this is a test of the branch, not a measurement)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--layers", "2", "--no-vae",
                        "--no-cpu-baseline", "--no-train"], env=_rank_env(), capture_output=True, text=True, timeout=850, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["finite"] and rec["config"]["exchange"].startswith("RCCL")
    assert rec["value"] == rec["value_aggregate"] == rec["value_per_gpu"] and len(rec["rank_ms_per_step"]) == 1
