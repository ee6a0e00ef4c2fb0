"""Classifier-free-guidance parallelism for the stages the reference runs on ONE GPU while the others idle (the 52-step
base stage on GPU 0, infer_cogvideo_mp_fifo.py:300, and the T2To stage, :262): the unconditional and the conditional forward of a
step are independent, so with >= 2 ranks rank r runs half r % 2 (batch 1) and ONE all_gather per step (2.8 MB at the To2V shape)
gives every rank both halves; every rank then applies the identical fused CFG + solver step with identically seeded noise, so all
ranks leave the stage holding the same state — which the FIFO stage needs anyway (the reference pickles it to its workers)."""
import torch


def world_and_rank():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def resolve(mode):
    """mode: None/"auto" (parallel iff >= 2 ranks), False (batched forward on every rank), True (require >= 2 ranks),
    "emulate" (one process runs both batch-1 halves in turn: the numerics of the parallel path without a second GPU)."""
    world, _ = world_and_rank()
    if mode in (None, "auto"):
        return "parallel" if world >= 2 else "batched"
    if mode is False:
        return "batched"
    if mode == "emulate":
        return "emulate"
    if mode is True:
        if world < 2:
            raise RuntimeError("cfg_parallel=True needs torch.distributed with at least 2 ranks")
        return "parallel"
    raise ValueError(f"cfg_parallel={mode!r}")


def predict(mode, forward_half, forward_both, n=2):
    """forward_half(h) -> prediction [1, ...] of guidance branch h (0 = unconditional, 1 = conditional; with `use_separate_guidance` n = 3:
    uncond_txt, uncond_img, txt_img); forward_both() -> [n, ...].  Returns the [n, ...] prediction tensor on every rank."""
    world, me = world_and_rank()
    if mode == "batched" or (mode == "parallel" and world < n):
        return forward_both()
    if mode == "emulate":
        return torch.cat([forward_half(h) for h in range(n)], dim=0)
    import torch.distributed as dist
    mine = forward_half(me % n).contiguous()
    allp = torch.empty((world,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(allp, mine)
    return allp[:n].contiguous()          # rank h < n holds branch h


def map_chunks_sharded(n_chunks, fn, device=None):
    """[fn(0), ..., fn(n_chunks - 1)] on EVERY rank, each rank having run only the chunks c with c % world == rank, brought together by ONE all_gather_into_tensor
    (rank-major slots, like fifo.decode_chunks_sharded).  fn(c) -> tensor of one fixed shape / dtype on `device` (this rank's; needed by a rank that holds no chunk).
    For work that is independent per chunk and that the reference runs chunk after chunk on one GPU: the condensed-token encoding of the source clips
    (pipeline_cogvideox_mp_fifo.py:585-609: vae.encode -> patch_embed.proj -> Resampler per 49-frame chunk).  With no process group: a plain loop; with one — of any
    size — the collective runs.  fn must not depend on state the other chunks' calls would have changed (the caller draws any random numbers for ALL chunks itself, in
    chunk order, so every rank's generator stays in step)."""
    import torch.distributed as dist
    dist_on = dist.is_available() and dist.is_initialized()
    world, me = world_and_rank()
    mine = [fn(c) for c in range(n_chunks) if c % world == me]
    if not dist_on:
        return mine
    meta = [None] * world
    dist.all_gather_object(meta, (tuple(mine[0].shape), mine[0].dtype) if mine else None)     # a rank may hold no chunk
    shape, dtype = next(m for m in meta if m is not None)
    dev = mine[0].device if mine else torch.device(device if device is not None else "cpu")
    per = (n_chunks + world - 1) // world
    buf = torch.zeros((per,) + shape, dtype=dtype, device=dev)
    for slot, t in enumerate(mine):
        buf[slot] = t
    allbuf = torch.empty((world * per,) + shape, dtype=dtype, device=dev)
    dist.all_gather_into_tensor(allbuf, buf)
    return [allbuf[(c % world) * per + c // world] for c in range(n_chunks)]
