#!/bin/bash
# PMC passes over tools/conv_micro.py (own runs, no trace domains): where the VAE convolution kernels' cycles go.
#   tools/pmc_conv.sh TAG case...   -> gpurun_out/TAG_conv_pmc.json
tag=${1:-r4}; shift
cases=${@:-c128 c256 c256_128}
R=$PWD
out=$R/gpurun_out/prof_${tag}_conv
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python $R/tools/conv_micro.py 10 $cases > $out/bench.log 2> $out/stats.err
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/pmc1 -o pmc -- python $R/tools/conv_micro.py 4 $cases > /dev/null 2> $out/pmc1.err
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM --output-format csv -d $out/pmc2 -o pmc -- python $R/tools/conv_micro.py 4 $cases > /dev/null 2> $out/pmc2.err
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc3 -o pmc -- python $R/tools/conv_micro.py 4 $cases > /dev/null 2> $out/pmc3.err
cd $R
find $out -name "*kernel_trace.csv" -delete
python - <<PY
import csv, json, re, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("pmc1", "pmc2", "pmc3"):
    fs = glob.glob("$out/%s/**/*counter_collection.csv" % p, recursive=True)
    if not fs: continue
    per = collections.defaultdict(float); meta = {}
    for r in csv.DictReader(open(fs[0])):
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        meta[r["Dispatch_Id"]] = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0] + " grid=" + r["Grid_Size"]
    for (d, c), v in per.items(): acc[meta[d]][c].append(v)
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items() if "conv3d" in k}
for k, m in res.items():
    if m.get("SQ_BUSY_CYCLES") and m.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        m["_mfma_busy_frac (MFMA_BUSY / (4 SIMD x 256 CU x GRBM_GUI_ACTIVE))"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * m["GRBM_GUI_ACTIVE"]) if m.get("GRBM_GUI_ACTIVE") else None
    if m.get("SQ_LDS_IDX_ACTIVE"): m["_lds_conflict_frac"] = m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"]
json.dump(res, open("gpurun_out/${tag}_conv_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
cat $out/bench.log | grep case
