#!/bin/bash
# tile-order group size (TG_GEMM_GROUP_M) against time, HBM-side read traffic (FETCH_SIZE; one counter per PMC pass — two
# in one pass exceed what the hardware collects) and the clock the kernel gets under the power cap (tools/clock_probe.py).
# On the GPU box:  bash tools/ab_gemm_traffic.sh "1 4 8" -> gpurun_out/gemm_group_ab.json
R=$PWD; out=$R/gpurun_out/gemm_group_ab; mkdir -p $out
export TMPDIR=/tmp
for g in ${1:-1 4 8}; do
  date +%T
  TG_GEMM_GROUP_M=$g timeout 200 python tools/bench_kernels.py gemm 2>/dev/null | grep '"ms"' > $out/time_g${g}_r3.jsonl
  date +%T
  (cd /tmp && TG_GEMM_GROUP_M=$g timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_g$g -o pmc -- python $R/tools/bench_kernels.py gemm > /dev/null 2> $out/pmc_g$g.err)
  date +%T
  [ -f $out/clock_g$g.txt ] || TG_GEMM_GROUP_M=$g timeout 60 python tools/clock_probe.py tg 3 > $out/clock_g$g.txt 2>&1
done
python3 - <<PY
import csv, glob, json, collections
out = {}
for g in (1, 2, 4, 8, 16):
    rec = {}
    ms = collections.defaultdict(list)
    for fn in glob.glob("$out/time_g%d_r*.jsonl" % g):
        for l in open(fn):
            d = json.loads(l); ms[d["kernel"]].append(d["ms"])
    if not ms: continue
    rec["ms"] = {k: round(min(v), 4) for k, v in ms.items()}
    per = collections.defaultdict(float); meta = {}
    for fn in glob.glob("$out/pmc_g%d/**/*counter_collection.csv" % g, recursive=True):
        for r in csv.DictReader(open(fn)):
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); meta[r["Dispatch_Id"]] = r["Kernel_Name"]
    ids = sorted({d for d, _ in per if "gemm256w4" in meta[d]}, key=int)      # launch order: 7 launches of each of the six shapes
    names = ["qkv", "out(bias)", "ff1", "ff2(bias)", "out(gate)", "ff2(gate)"]
    rec["read_GB"] = {n: round(sum(2 * per[(d, "FETCH_SIZE")] * 1024 for d in ids[i * 7:(i + 1) * 7]) / max(1, len(ids[i * 7:(i + 1) * 7])) / 1e9, 3)
                      for i, n in enumerate(names) if ids[i * 7:(i + 1) * 7]}
    try:
        rec["clock_probe"] = [l for l in open("$out/clock_g%d.txt" % g).read().strip().splitlines() if "amdgpu.ids" not in l][:5]
    except OSError:
        pass
    out[g] = rec
json.dump(out, open("$R/gpurun_out/gemm_group_ab.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:8000])
PY
rm -rf $out/pmc_g*/
