#!/bin/bash
# rocprofv3 evidence for the attention backward (tools/bench_kernels.py attn_bwd): kernel stats + FETCH_SIZE / WRITE_SIZE passes.
#   tools/profile_attn_bwd.sh TAG  -> gpurun_out/prof_TAG/{stats,pmc1,pmc2}
tag=${1:-r3_attn_bwd}
R=$PWD
out=$R/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python $R/tools/bench_kernels.py attn_bwd > $out/bench.log 2> $out/stats.err
timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc1 -o pmc -- python $R/tools/bench_kernels.py attn_bwd > /dev/null 2> $out/pmc1.err
timeout 300 rocprofv3 --pmc WRITE_SIZE SQ_BUSY_CYCLES --output-format csv -d $out/pmc2 -o pmc -- python $R/tools/bench_kernels.py attn_bwd > /dev/null 2> $out/pmc2.err
cd $R
find $out -name "*kernel_trace.csv" -delete
python - <<PY
import csv, json, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("pmc1", "pmc2"):
    per = collections.defaultdict(float); meta = {}
    for r in csv.DictReader(open("$out/%s/pmc_counter_collection.csv" % p)):
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); meta[r["Dispatch_Id"]] = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
    for (d, c), v in per.items(): acc[meta[d]][c].append(v)
res = {}
for k, cs in acc.items():
    if "attn_bwd" not in k: continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    res[k] = dict(m, l2_miss_traffic_bytes_per_launch=(2 * m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0)) * 1024)
n, d = 17776, 3072
res["_algorithmic_bytes"] = {"dkdv (q, k, v, dO bf16 in; dk, dv fp32 out), B=2": 2 * (4 * n * d * 2 + 2 * n * d * 4), "dq (q, k, v, dO in; dq fp32 out), B=2": 2 * (4 * n * d * 2 + n * d * 4)}
json.dump(res, open("$R/gpurun_out/%s_pmc.json" % "$tag", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_kernel_stats.csv
