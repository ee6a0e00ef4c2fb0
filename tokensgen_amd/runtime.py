"""Persistent one-process-per-GPU runtime — what replaces the reference's per-item `mp.Process` spawn with the whole
pipeline pickled as CUDA-IPC handles (cogvideo_sampling_mp_fifo.py:195-221) and its join (:361-365).

* `launch(nprocs, argv)`: start `nprocs` ranks of a script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the
  environment, exactly what `torch.distributed.run` exports, so a script works under either), stream rank 0's stdout through, and
  SUPERVISE them: the first rank that exits non-zero gets the others terminated and its stderr tail raised as RuntimeError in the
  parent.  The reference's parent blocks forever in `output_queue.get()` when a worker dies (:308-311).
* `init_distributed(...)`: `init_process_group` with an explicit timeout (backend nccl = RCCL on the GPU, gloo on CPU), so that a
  rank that died WITHOUT raising (killed, device lost) makes its peers' next collective fail after `timeout_s` instead of hanging.
* `broadcast_weights(model, src)`: ONE rank reads the checkpoint, the others receive the weights over RCCL / xGMI — the collective that
  replaces the reference's whole-pipeline CUDA-IPC pickling (:195-221); a few large broadcasts over the models' fused storages.
* `RankGuard`: exception propagation between live ranks.  A rank that raised inside its part of an iteration still takes part in the
  iteration's exchange and flags it; every rank then raises `RankFailure` naming the rank and its message — within one iteration, not after
  a timeout.
"""
import datetime
import os
import signal
import subprocess
import sys
import tempfile
import time

import torch

DEFAULT_TIMEOUT_S = float(os.environ.get("TG_DIST_TIMEOUT_S", "600"))


class RankFailure(RuntimeError):
    """Raised on EVERY rank when one of them failed inside a guarded iteration."""


def init_distributed(backend=None, timeout_s=None, device=None):
    """Rank / world from the environment (torch.distributed.run's or `launch`'s); returns (rank, world).  No-op for a lone process."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1 and "RANK" not in os.environ:
        return 0, 1
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = dict(rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s or DEFAULT_TIMEOUT_S))
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, **kw)
    return rank, world


def _weight_storages(model):
    """The tensors that ARE a model's weights, in a deterministic order: the fused storages of CogVideoXTransformer3DModel (its nn.Parameters
    are views of them), the state-dict tensors of the VAE / Resampler mirrors, or a plain nn.Module's parameters + buffers."""
    if hasattr(model, "_fused"):
        return [model._fused[k] for k in sorted(model._fused)]
    if hasattr(model, "_sd") and isinstance(model._sd, dict):
        return [model._sd[k] for k in sorted(model._sd)]
    if isinstance(model, torch.nn.Module):
        return [t for _, t in sorted(list(model.named_parameters()) + list(model.named_buffers()), key=lambda kv: kv[0])]
    raise TypeError(f"broadcast_weights: do not know the weight storages of {type(model).__name__}")


@torch.no_grad()
def broadcast_weights(model, src=0, group=None, bucket_bytes=1 << 30):
    """Every rank ends up with rank `src`'s weights (north_star: "RCCL broadcast of weights").  Large storages (the per-layer [3D, D] / [4D, D]
    GEMM weights, the fused modulation matrix) go as one broadcast each; the many small ones (biases, LayerNorm affines) are coalesced into flat
    buckets of at most `bucket_bytes` so that the 42-layer model is ~260 collectives instead of ~1 300 — xGMI rings are per-link bound, few large
    messages.  bf16 travels as raw bytes (gloo knows neither bf16 nor int16, and a broadcast does not care).  Shapes / dtypes must already
    agree on every rank (construct the model from the same config; `from_pretrained(..., broadcast=True)` does).  Afterwards the receiver's
    derived state is refreshed (`_after_weight_update` when the model has one).  Returns the number of bytes received / sent."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0                                       # (a 1-rank group still runs the broadcasts: the same calls, RCCL to itself)
    tensors = _weight_storages(model)
    total = 0

    def raw(t):
        return t.view(torch.uint8) if t.dtype == torch.bfloat16 else t
    small = []
    for t in tensors:
        if not t.is_contiguous():
            raise ValueError("broadcast_weights: weight storages must be contiguous")
        nbytes = t.numel() * t.element_size()
        total += nbytes
        if nbytes >= (4 << 20):
            dist.broadcast(raw(t), src=src, group=group)
        else:
            small.append(t)
    by_dtype = {}
    for t in small:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dt, ts in by_dtype.items():
        i = 0
        while i < len(ts):
            chunk, nb = [], 0
            while i < len(ts) and (not chunk or nb + ts[i].numel() * ts[i].element_size() <= bucket_bytes):
                chunk.append(ts[i]); nb += ts[i].numel() * ts[i].element_size(); i += 1
            flat = torch.cat([c.reshape(-1) for c in chunk])
            dist.broadcast(raw(flat), src=src, group=group)
            off = 0
            for c in chunk:
                c.copy_(flat[off:off + c.numel()].view(c.shape))
                off += c.numel()
    hook = getattr(model, "_after_weight_update", None)
    if hook is not None:
        hook()
    return total


class RankGuard:
    """Usage, once per iteration on every rank:

        g = RankGuard()
        with g:                     # the rank-local work; an exception is caught and remembered
            ... compute ...
        flag = g.flag(device)       # 1-element tensor to ride in the iteration's own exchange (no extra collective on the good path)
        ... all_gather(...) of the payload + flag ...
        g.check(all_flags)          # every rank: raises RankFailure(rank, message) if any flag is set

    `check` runs one extra `all_gather_object` only on the failure path (every rank knows by then that one is needed)."""

    def __init__(self, what=""):
        self.exc = None
        self.what = what

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if ev is not None and isinstance(ev, Exception):
            self.exc = ev
            return True
        return False

    def flag(self, device, dtype=torch.bfloat16):
        return torch.full((1,), 1.0 if self.exc is not None else 0.0, dtype=dtype, device=device)

    def check(self, flags):
        """flags: tensor of every rank's flag (rank order).  Lone process: re-raises its own exception."""
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if world == 1:
            if self.exc is not None:
                raise self.exc
            return
        bad = [r for r, f in enumerate(flags.float().flatten().tolist()) if f != 0.0]
        if not bad:
            return
        msgs = [None] * world
        dist.all_gather_object(msgs, None if self.exc is None else f"{type(self.exc).__name__}: {self.exc}")
        raise RankFailure(f"{self.what}: rank {bad[0]} failed: {msgs[bad[0]]}" + (f" (also ranks {bad[1:]})" if len(bad) > 1 else "")) from self.exc


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(nprocs, argv, env=None, poll_s=0.2, grace_s=5.0, stdout=None):
    """Run `argv` (e.g. [sys.executable, "bench.py", ...]) as `nprocs` ranks on this node and supervise them.  Returns rank 0's exit
    code (0).  Raises RuntimeError as soon as any rank exits non-zero — after terminating the others — with that rank's stderr tail."""
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base["MASTER_PORT"] = str(base.get("TG_MASTER_PORT") or _free_port())
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs, errs = [], []
    for r in range(nprocs):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nprocs), LOCAL_WORLD_SIZE=str(nprocs))
        ef = tempfile.TemporaryFile(mode="w+")
        errs.append(ef)
        procs.append(subprocess.Popen(argv, env=e, stdout=(stdout if r == 0 else subprocess.DEVNULL), stderr=ef, start_new_session=True))

    def stop_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except ProcessLookupError:
                    pass
        t_end = time.time() + grace_s
        for p in procs:
            try:
                p.wait(max(0.0, t_end - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass

    try:
        while True:
            codes = [p.poll() for p in procs]
            failed = [r for r, c in enumerate(codes) if c not in (None, 0)]
            if failed:
                stop_all()
                r = failed[0]
                errs[r].seek(0)
                tail = errs[r].read()[-4000:]
                raise RuntimeError(f"rank {r} of {nprocs} exited with code {codes[r]}; the other ranks were terminated.\n--- rank {r} stderr tail ---\n{tail}")
            if all(c == 0 for c in codes):
                errs[0].seek(0)
                sys.stderr.write(errs[0].read()[-2000:])
                return 0
            time.sleep(poll_s)
    except BaseException:
        stop_all()
        raise
    finally:
        for f in errs:
            f.close()
