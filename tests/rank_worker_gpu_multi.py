"""Rank body of tests/test_multirank_one_gpu.py: FOUR ranks on ONE MI355X over a `gloo` process group (RCCL refuses several ranks on one device; gloo moves CUDA tensors
through the host) — so that the multi-rank branches of the product run with the REAL HIP kernels, not stand-ins: the base stage CFG-parallel (rank r computes guidance half
r % 2), the FIFO driver with its small iterations split by guidance branch (rank 2k + h: branch h of window k as a batch-1 forward, one all_gather of the model outputs) and
its ordinary iterations dealt round-robin, the chunk-sharded decode gather.  Every rank first runs the same calls with NO process group and compares bit for bit.
Reference: cogvideo_sampling_mp_fifo.py:195-221, 230-334, 373-376; infer_cogvideo_mp_fifo.py:300."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import rank_worker_gpu as W  # noqa: E402  (build / fifo_run of the 1-rank RCCL test: tiny To2V DiT + pipeline on cuda:0)

outdir = sys.argv[1]
rank = int(os.environ["RANK"])


def done(msg):
    with open(os.path.join(outdir, f"rank{rank}.txt"), "w") as f:
        f.write(msg)


def main():
    import torch.distributed as dist
    from tokensgen_amd import fifo
    from tokensgen_amd.runtime import init_distributed
    torch.cuda.set_device(0)
    m, sd, pipe = W.build()
    assert not dist.is_initialized()
    ref_lat = W.fifo_run(pipe, "latent")
    ref_vid = W.fifo_run(pipe, "pt")
    r, world = init_distributed("gloo", timeout_s=300)
    assert (r, world) == (rank, 4) and dist.get_backend() == "gloo"
    # gloo's own handling of CUDA tensors is documented for broadcast / all_reduce only (all_gather on device tensors ran here, and returned stale rows in ~1 of 6 runs): the
    # test stages the two gather collectives through the host itself — synchronous copies, gloo on CPU tensors — so that what is under test is the product's rank logic on the
    # real kernels, not gloo's device path.  (On a multi-GPU node these calls are RCCL's.)
    orig_agit = dist.all_gather_into_tensor

    def staged_all_gather_into_tensor(out, inp, group=None, async_op=False):
        assert not async_op
        if not inp.is_cuda:
            return orig_agit(out, inp, group=group)
        torch.cuda.synchronize()
        host_in = inp.detach().cpu().contiguous()
        host_out = torch.empty(out.shape, dtype=out.dtype)
        orig_agit(host_out, host_in, group=group)
        out.copy_(host_out)
        torch.cuda.synchronize()
    dist.all_gather_into_tensor = staged_all_gather_into_tensor
    # count what THIS rank computes: whole windows (ordinary iterations) and single guidance branches (split iterations)
    calls = {"window": 0, "branch": []}
    orig_step, orig_pred = fifo.FifoWorker.window_step, fifo.FifoWorker.predict

    def step(self, *a, **k):
        calls["window"] += 1
        return orig_step(self, *a, **k)

    def pred(self, branch, *a, **k):
        if branch is not None:
            calls["branch"].append(int(branch))
        return orig_pred(self, branch, *a, **k)
    fifo.FifoWorker.window_step, fifo.FifoWorker.predict = step, pred
    got_lat = W.fifo_run(pipe, "latent")
    n_branch, n_window = len(calls["branch"]), calls["window"]
    got_vid = W.fifo_run(pipe, "pt")
    fifo.FifoWorker.window_step, fifo.FifoWorker.predict = orig_step, orig_pred
    res = {"fifo_latents": torch.equal(got_lat, ref_lat), "fifo_decode": torch.equal(got_vid, ref_vid),
           # 2 chunks: 65 iterations, 7 of them with <= 2 windows (1 with a single one): ranks 0, 1 run 7 branches, ranks 2, 3 six; every rank only ITS branch
           "split_counts": n_branch == (7 if rank < 2 else 6) and set(calls["branch"]) == {rank % 2},
           # the other 58 iterations: windows dealt round-robin (none of them has fewer than 3 windows, at most 8)
           "whole_windows": 0 < n_window <= 2 * 58}
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    import hashlib
    sha = lambda t: hashlib.sha256(t.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:10]
    detail = f" sha ref_lat {sha(ref_lat)} got_lat {sha(got_lat)}"
    if not res["fifo_latents"]:
        d = (got_lat.float() - ref_lat.float()).abs()
        pf = d.amax(dim=(0, 2, 3, 4))
        detail += f" latent diff max {float(d.max()):.4g} frames {[int(i) for i in torch.nonzero(pf > 0).flatten()[:30]]} of {pf.numel()}"
    if not res["fifo_decode"]:
        d = (got_vid.float() - ref_vid.float()).abs()
        per_frame = d.amax(dim=(0, 1, 3, 4))
        detail = f" decode diff: shape {tuple(got_vid.shape)} max {float(d.max()):.4g}, frames differing {[int(i) for i in torch.nonzero(per_frame > 0).flatten()[:12]]} of {per_frame.numel()}"
    bad = [k for k, v in res.items() if not v]
    done(("ok " + " ".join(sorted(res)) if not bad else "mismatch: " + " ".join(bad)) + f" [branches {n_branch}, windows {n_window}]" + detail)


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:  # noqa: BLE001 — the parent test reads the file
        import traceback
        done("exception: " + "".join(traceback.format_exception(type(e), e, e.__traceback__))[-3000:])
        raise
