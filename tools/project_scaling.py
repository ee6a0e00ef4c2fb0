#!/usr/bin/env python3
"""Projected multi-GPU scaling of the two shipped runs (edit.yaml: 12 chunks, BASELINE config 2; gen.yaml: 24 chunks, BASELINE config 3) from the PRODUCT's own
window schedule and MEASURED one-GPU phase seconds (VERDICT r5 missing #1 / next #5).  CPU only; no GPU, no oracle.

What is walked, not assumed: `tokensgen_amd.fifo.window_plan` for every FIFO iteration of the run (queue_start = T - ceil(nf/2) counting down, as
`cogvideo_fifo_mp_v2` does), windows dealt round-robin exactly like the driver (`k % world == rank`): an iteration costs max-over-ranks(windows) window-forwards.
What is measured (profiles/*.json, one MI355X): seconds per window forward inside the FIFO phase, the base stage, the T2To stage, a clip's VAE decode, and the
batch-1 : batch-2 forward ratio (the CFG-parallel stages run one CFG half per rank).  What is a stated placeholder: the per-iteration all_gather (<= 4.4 MB per rank
over xGMI) and the per-step CFG all_gather (2.8 MB): 1.0 ms / 0.5 ms each at N > 1 — three orders of magnitude below a window forward, listed so that the
table's sensitivity to them is visible.

    python tools/project_scaling.py [--b1-ratio R] [--out profiles/r6_scaling_projection.json]

Reference: cogvideo_sampling_mp_fifo.py:230-306 (the windows of an iteration and their workers), :373-376 (decode on GPU 0), infer_cogvideo_mp_fifo.py:262,300
(T2To stage and base stage on one GPU while the others idle)."""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fifo_iterations(chunks, nf=13, T=52, num_partitions=4):
    """Windows per iteration of one run: list of len(window_plan(queue_start)) over the bo.num_frames + T - nf iterations (fifo.py: the loop of cogvideo_fifo_mp_v2)."""
    from tokensgen_amd.fifo import window_plan
    l = nf - nf // 2
    qs = T - l
    out = []
    for _ in range(chunks * nf + T - nf):
        out.append(len(window_plan(qs, nf, num_partitions)))
        qs = max(0, qs - 1)
    return out


def project(chunks, meas, n, with_t2to, b1_ratio, b1_ratio_t2to=None, t_xchg=1.0e-3, t_cfg_xchg=0.5e-3, split=True, nb=2):
    per_iter = fifo_iterations(chunks)
    fwd1 = sum(per_iter)
    # an iteration costs max-over-ranks(windows) window forwards — or, when the ranks number at least nb x windows, ONE batch-1 forward (fifo.py: the windows split by
    # guidance branch, round 6)
    fwdn = sum((b1_ratio if (split and n >= nb * w and n > 1) else math.ceil(w / n)) for w in per_iter)
    steps = 52
    cfg_par = n >= 2                                       # cfg_parallel.resolve("auto"): ranks r % 2 take one CFG half each
    stage = lambda t1, r: t1 * r + steps * t_cfg_xchg if cfg_par else t1
    t = {"t2to": stage(meas["t2to_s"], b1_ratio_t2to or b1_ratio) if with_t2to else 0.0, "base": stage(meas["base_s"], b1_ratio),
         "fifo": fwdn * meas["window_s"] + (len(per_iter) * t_xchg if n > 1 else 0.0),
         # decode_chunks_sharded twice: the video's `chunks` clips dealt round-robin + the base clip (one rank)
         "decode": (math.ceil(chunks / n) + 1) * meas["decode_clip_s"]}
    t["total"] = sum(t.values())
    return {"n_gpus": n, "window_forwards_total": fwd1, "window_forwards_on_the_critical_rank": round(fwdn, 2), "seconds": {k: round(v, 2) for k, v in t.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b1-ratio", type=float, default=None, help="measured seconds(batch-1 forward) / seconds(batch-2 forward); default: profiles/r6_b1_forward.json, else 0.5 (flagged)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r6_scaling_projection.json"))
    a = ap.parse_args()
    P = lambda f: json.load(open(os.path.join(ROOT, "profiles", f)))
    gen, edit = P("r6_gen_1gpu_24clips.json"), P("r6_e2e_1gpu_12clips.json")
    try:
        vae_s = P("r6_bench_detail.json")["vae"]["decode"]["seconds"]; vae_src = "r6_bench_detail.json"
    except (OSError, KeyError, TypeError):
        vae_s = P("r5_bench.json")["vae"]["decode"]["seconds"]; vae_src = "r5_bench.json"
    b1, b1_src, b1t = a.b1_ratio, "--b1-ratio", None
    if b1 is None:
        try:
            r = P("r6_b1_forward.json")
            b1, b1t, b1_src = r["to2v"]["b1_over_b2"], r["t2to"]["b1_over_b2"], "profiles/r6_b1_forward.json (measured: To2V window shape and T2To shape, tools/bench_b1.py)"
        except (OSError, KeyError):
            b1, b1_src = 0.5, "ASSUMED 0.5 (MFMA-bound forward: half the batch, half the time) — unmeasured"
    runs = {}
    for name, rec, chunks, with_t2to in (("edit.yaml (12 chunks, To2V)", edit, 12, False), ("gen.yaml (24 chunks, T2To + To2V)", gen, 24, True)):
        n_fwd = sum(fifo_iterations(chunks))
        clips = chunks + 1
        meas = {"t2to_s": (rec["seconds"].get("t2to_stage") or 0.0), "base_s": rec["seconds"]["base_stage"], "decode_clip_s": vae_s,
                "window_s": (rec["seconds"]["fifo_and_decode"] - clips * vae_s) / n_fwd, "fifo_and_decode_s": rec["seconds"]["fifo_and_decode"], "window_forwards": n_fwd}
        assert n_fwd == rec["steps"] - 52, (n_fwd, rec["steps"])        # the walked schedule IS the measured run's
        rows = [project(chunks, meas, n, with_t2to, b1, b1t) for n in (1, 2, 4, 8)]
        whole = [project(chunks, meas, n, with_t2to, b1, b1t, split=False) for n in (1, 2, 4, 8)]       # round 5's driver: small iterations run whole windows
        t1, f1 = rows[0]["seconds"]["total"], rows[0]["seconds"]["fifo"]
        for r in rows:
            n = r["n_gpus"]
            r["speedup_end_to_end"] = round(t1 / r["seconds"]["total"], 3)
            r["efficiency_end_to_end"] = round(t1 / r["seconds"]["total"] / n, 4)
            r["efficiency_fifo_phase"] = round(f1 / r["seconds"]["fifo"] / n, 4)
            r["steps_per_s_per_gpu_end_to_end"] = round((52 + n_fwd) / r["seconds"]["total"] / n, 4)
        for r, w_ in zip(rows, whole):
            r["without_branch_split"] = {"fifo_s": w_["seconds"]["fifo"], "efficiency_fifo_phase": round(f1 / w_["seconds"]["fifo"] / r["n_gpus"], 4),
                                         "efficiency_end_to_end": round(t1 / w_["seconds"]["total"] / r["n_gpus"], 4)}
        runs[name] = {"measured_one_gpu": {k: round(v, 4) for k, v in meas.items()}, "projection": rows}
    out = {"what": "PROJECTION, not a measurement: product window schedule x one-GPU phase seconds; no N > 1 hardware was available to the builder",
           "sources": {"phase_seconds": ["profiles/r6_gen_1gpu_24clips.json", "profiles/r6_e2e_1gpu_12clips.json"], "vae_decode_clip_s": vae_src, "b1_over_b2": b1_src},
           "b1_over_b2": b1, "b1_over_b2_t2to": b1t, "placeholders_s": {"fifo_all_gather_per_iteration": 1.0e-3, "cfg_all_gather_per_step": 0.5e-3},
           "not_modelled": "T5 / checkpoint loading (out of scope), the weight broadcast (once per process, ~14 GB over xGMI), the condensed-token encode of edit.yaml "
                           "(chunks + 1 VAE encodes + Resampler calls, sharded round-robin over all ranks since round 6: ceil(13 / N) x ~0.27 s)",
           "runs": runs}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    for name, r in runs.items():
        print(f"\n{name}: window {r['measured_one_gpu']['window_s'] * 1e3:.1f} ms, base {r['measured_one_gpu']['base_s']:.1f} s, t2to {r['measured_one_gpu']['t2to_s']:.1f} s, "
              f"decode {r['measured_one_gpu']['decode_clip_s']:.3f} s/clip, b1/b2 {b1:.3f}")
        print("| N | T2To s | base s | FIFO s | decode s | total s | speed-up | FIFO-phase eff. | end-to-end eff. | steps/s/GPU e2e | FIFO / e2e eff. without the branch split |")
        print("|---|---|---|---|---|---|---|---|---|---|---|")
        for p in r["projection"]:
            s = p["seconds"]
            print(f"| {p['n_gpus']} | {s['t2to']} | {s['base']} | {s['fifo']} | {s['decode']} | {s['total']} | {p['speedup_end_to_end']} | {p['efficiency_fifo_phase']} | "
                  f"{p['efficiency_end_to_end']} | {p['steps_per_s_per_gpu_end_to_end']} | {p['without_branch_split']['efficiency_fifo_phase']} / {p['without_branch_split']['efficiency_end_to_end']} |")


if __name__ == "__main__":
    main()
