#!/bin/bash
# rocprofv3 SQ counters of the attention-backward kernels (tools/bench_kernels.py attn_bwd): MFMA busy, VALU / LDS instruction activity, issue stalls, bank conflicts.
#   tools/pmc_attn_bwd_sq.sh  -> gpurun_out/s3_attn_bwd_pmc3.json   (own pass: no trace domains with --pmc)
R=$PWD
out=$R/gpurun_out/prof_s3_attn_bwd_pmc3
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $out -o pmc -- python $R/tools/bench_kernels.py attn_bwd > /dev/null 2> $out/err.log
cd $R
python - <<PY
import csv, json, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
per = collections.defaultdict(float); meta = {}
for r in csv.DictReader(open("$out/pmc_counter_collection.csv")):
    per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); meta[r["Dispatch_Id"]] = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
for (d, c), v in per.items(): acc[meta[d]][c].append(v)
res = {}
for k, cs in acc.items():
    if "attn_bwd" not in k: continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    # SQ_BUSY_CYCLES sums over 8 XCDs x SEs...; MFMA busy fraction as in pmc_summary: MFMA_BUSY / (4 SIMDs * CU cycles)
    res[k] = m
json.dump(res, open("$R/gpurun_out/s3_attn_bwd_pmc3.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
