#!/bin/bash
# rocprofv3 SQ counters of one tools/bench_kernels.py target (own pass: no trace domains with --pmc):  tools/pmc_sq.sh WHAT TAG  -> gpurun_out/TAG_sq.json
what=$1; tag=$2
R=$PWD
out=$R/gpurun_out/prof_${tag}_sq
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $out -o pmc -- python $R/tools/bench_kernels.py $what > /dev/null 2> $out/err.log
cd $R
python - <<PY
import csv, json, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
per = collections.defaultdict(float); meta = {}
for r in csv.DictReader(open("$out/pmc_counter_collection.csv")):
    per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); meta[r["Dispatch_Id"]] = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0] + " grid=" + r["Grid_Size"]
for (d, c), v in per.items(): acc[meta[d]][c].append(v)
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items() if "attn" in k or "gemm" in k or "conv" in k}
for k, m in res.items():
    if m.get("SQ_WAVE_CYCLES"):
        m["_mfma_busy_frac_of_cu_cycles"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (m["GRBM_GUI_ACTIVE"] / 8 * 256 * 4) if m.get("GRBM_GUI_ACTIVE") else None
json.dump(res, open("$R/gpurun_out/${tag}_sq.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
