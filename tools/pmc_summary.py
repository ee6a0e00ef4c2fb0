#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc passes of tools/profile_pmc.sh into one JSON: per kernel (name + grid), the mean of every counter
over its dispatches, plus the HBM-side traffic of the dominant kernel corrected as MI355X_MICROARCH.md prescribes
((2*FETCH_SIZE + WRITE_SIZE) KiB on gfx950).   tools/pmc_summary.py gpurun_out/prof_TAG profiles/TAG_pmc_summary.json"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def main(src, dst):
    acc = defaultdict(lambda: defaultdict(list))
    for p in ("pmc1", "pmc2", "pmc3"):
        per_dispatch = defaultdict(float)      # (dispatch, counter) summed over the rows rocprofv3 emits per dimension
        meta = {}
        with open(f"{src}/{p}/pmc_counter_collection.csv") as f:
            for r in csv.DictReader(f):
                key = (r["Dispatch_Id"], r["Counter_Name"])
                per_dispatch[key] += float(r["Counter_Value"])
                meta[r["Dispatch_Id"]] = f'{short(r["Kernel_Name"])}(grid={r["Grid_Size"]})'
        for (d, c), v in per_dispatch.items():
            acc[meta[d]][c].append(v)
    out = {}
    for k, cs in acc.items():
        if not any(t in k for t in ("attn", "gemm", "adaln", "qk_norm", "transpose_v", "cfg_dpm", "patchify")):
            continue
        out[k] = {c: {"dispatches": len(v), "mean": sum(v) / len(v)} for c, v in sorted(cs.items())}
    # the attention call of a block is several launches since round 3: the whole workgroups, the key-split tail, its combine, the (normally
    # empty) retry grid — `main` is the one with the largest grid, the traffic figure sums all of them per block
    parts = [k for k in out if k.startswith(("attn_fwd_pp_kernel", "attn_split_combine_kernel"))]
    if parts:
        grid = lambda k: int(re.search(r"grid=(\d+)", k).group(1))
        main_attn = [max(parts, key=grid)]
        a = out[main_attn[0]]
        n1, nv, d = 17776, 480, 3072
        # bench.py also runs ONE step on the other softmax path (fewer dispatches than `main`): those kernels are not part of the timed call
        ratio = lambda k: out[k]["FETCH_SIZE"]["dispatches"] / a["FETCH_SIZE"]["dispatches"]
        parts = [k for k in parts if ratio(k) >= 0.99]
        per_block = lambda k: min(1.0, ratio(k))
        out["_derived"] = {
            "attention_main_kernel": main_attn[0],
            "attention_launches_per_block": {k: per_block(k) for k in parts},
            "attention_main_traffic_bytes_per_launch": sum((2 * out[k]["FETCH_SIZE"]["mean"] + out[k]["WRITE_SIZE"]["mean"]) * 1024 * per_block(k) for k in parts),
            # B=2, bf16: q1,k1,v1,q2 read + out written by segment 1, read and re-written by segment 2 (counted once each way) + k2,v2
            # + the rider (vip queries, SDPA#3): q 480 rows, k and v of all N rows, out 480 rows
            "attention_main_algorithmic_bytes_per_launch": 2 * 2 * (6 * n1 * d + 2 * nv * d) + 2 * 2 * (2 * nv * d + 2 * (n1 + nv) * d),
            "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024  [gfx950: FETCH_SIZE reports half of a wide coalesced read]",
            # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs (so /8 = the
            # kernel's wall in shader cycles: 16.3 M cycles for 8.2 ms = 1.99 GHz under this load)
            "attention_main_shader_clock_cycles": a["GRBM_GUI_ACTIVE"]["mean"] / 8,
            "attention_main_mfma_busy_frac": a["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (1024 * a["GRBM_GUI_ACTIVE"]["mean"] / 8),
        }
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out.get("_derived"), indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
