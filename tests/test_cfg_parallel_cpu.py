"""CPU: the CFG-parallel exchange (tokensgen_amd/cfg_parallel.py) over gloo with 3 ranks — rank r computes half r % 2, every rank
ends up with [uncond, cond]; plus mode resolution.  The numerics of running the halves as batch-1 forwards are checked on the GPU
(tests/test_t2to_gpu.py, "emulate" mode == batched, bitwise)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _half(h):
    return torch.full((1, 3, 4), float(10 * (h + 1))) + torch.arange(12.0).reshape(1, 3, 4)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tokensgen_amd import cfg_parallel as CP
        calls = []
        mode = CP.resolve(None)

        def fh(h):
            calls.append(h)
            return _half(h)
        out = CP.predict(mode, fh, lambda: (_ for _ in ()).throw(AssertionError("batched forward must not run")))
        q.put((rank, mode, calls, out.numpy()))      # by value: a tensor travels as a shared-memory handle that dies with this process
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_cfg_parallel_three_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    from conftest import free_port
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = {r: (m, c, o) for r, m, c, o in (q.get(timeout=100) for _ in range(3))}
    for p in procs:
        p.join(timeout=30)
    want = torch.cat([_half(0), _half(1)])
    for r in range(3):
        mode, calls, out = got[r]
        assert mode == "parallel" and calls == [r % 2] and torch.equal(torch.from_numpy(out), want)


def test_cfg_parallel_modes_single_process():
    from tokensgen_amd import cfg_parallel as CP
    assert CP.resolve(None) == "batched" and CP.resolve("auto") == "batched" and CP.resolve(False) == "batched" and CP.resolve("emulate") == "emulate"
    with pytest.raises(RuntimeError):
        CP.resolve(True)
    with pytest.raises(ValueError):
        CP.resolve("yes")
    both = torch.cat([_half(0), _half(1)])
    assert torch.equal(CP.predict("batched", None, lambda: both), both)
    assert torch.equal(CP.predict("emulate", _half, None), both)
