"""Rank body for tests/test_runtime_cpu.py (started through tokensgen_amd.runtime.launch; gloo on CPU)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

torch.set_num_threads(1)
mode, outdir = sys.argv[1], sys.argv[2]
rank = int(os.environ["RANK"])


def done(msg):
    with open(os.path.join(outdir, f"rank{rank}.txt"), "w") as f:
        f.write(msg)


if mode == "die":                       # rank 1 dies without raising; rank 0 would wait "forever"
    if rank == 1:
        os._exit(3)
    time.sleep(120)
    sys.exit(0)

import torch.distributed as dist  # noqa: E402
from tokensgen_amd.runtime import RankFailure, init_distributed  # noqa: E402

if mode == "gradsync":                  # DDP gradient exchange of the training step (optim.GradSync): bucketed, incremental, averaged by the scale
    init_distributed("gloo", timeout_s=120)
    from tokensgen_amd.optim import GradSync
    n = 1000
    g = torch.Generator().manual_seed(7)
    both = torch.randn(2, n, generator=g)
    flat = both[rank].clone() * 0.5                    # the accumulation scale carries 1 / world_size
    sync = GradSync(flat, bucket_elems=300)
    assert sync.bounds == [(0, 300), (300, 600), (600, 900), (900, 1000)] and sync.world == 2
    launched = []
    for end in (100, 299, 300, 650, 650, 899):         # the backward passing arena offsets: only complete buckets may go
        sync.ready(end)
        launched.append(sync._next)
    assert launched == [0, 0, 1, 2, 2, 2], launched
    sync.finish()
    want = both.sum(0) * 0.5
    ok = torch.allclose(flat, want, atol=1e-6) and sync._next == 0 and not sync._work
    sync.ready(1000); sync.finish()                    # a second window works on the same object
    ok = ok and torch.allclose(flat, want * 2, atol=1e-6)
    done("ok" if ok else "mismatch")
    dist.destroy_process_group()
    sys.exit(0)

if mode == "raise":                     # rank 1's denoiser raises in FIFO iteration 3: EVERY rank must get RankFailure in that iteration
    init_distributed("gloo", timeout_s=120)
    import test_fifo_cpu as T
    from tokensgen_amd import fifo
    calls = [0]
    orig = fifo.window_plan

    def plan(qs, nf=13, num_partitions=4):
        calls[0] += 1
        return orig(qs, nf, num_partitions)
    fifo.window_plan = plan
    real_noise = T._noise

    def noise(i, tag, shape):
        if rank == 1 and i == 3 and tag != 97:
            raise ValueError("injected failure in window of iteration 3")
        return real_noise(i, tag, shape)
    T._noise = noise
    t0 = time.time()
    try:
        T._product_run(T._inputs())
        done("no exception")
    except RankFailure as e:
        done(f"RankFailure after {calls[0]} iterations in {time.time() - t0:.1f}s: {e}")
    dist.destroy_process_group()
elif mode == "silent":                  # rank 1 stops taking part without dying: rank 0's collective must time out, not hang
    init_distributed("gloo", timeout_s=8)
    x = torch.zeros(4)
    dist.all_reduce(x)
    if rank == 1:
        time.sleep(40)
        sys.exit(0)
    t0 = time.time()
    try:
        dist.all_reduce(x)
        done("no exception")
    except Exception as e:              # noqa: BLE001 — gloo raises RuntimeError / DistBackendError depending on the version
        done(f"timeout surfaced after {time.time() - t0:.1f}s: {type(e).__name__}")
