#!/bin/bash
# Why do the dQ stores of the one-kernel attention backward leave the L2?  TCC counters of the fused kernel, one pass per group (4 TCC slots):
#   tools/pmc_bwd_tcc.sh [variant]   -> gpurun_out/prof_tcc/summary.json
R=$PWD
export TMPDIR=/tmp
v=${1:-}
[ -n "$v" ] && export TG_LIB_PATH=$R/tokensgen_amd/csrc/variants/$v.so
out=$R/gpurun_out/prof_tcc; mkdir -p $out
i=0
for grp in "TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_WRITEBACK_sum" \
           "TCC_RW_REQ_sum TCC_NC_REQ_sum TCC_UC_REQ_sum TCC_CC_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_WRITE_sum TCC_READ_sum" \
           "TCC_EA0_WR_UNCACHED_32B_sum TCC_STREAMING_REQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_sum" \
           "TCC_ALL_TC_OP_INV_EVICT_sum TCC_REQ_sum TCC_ATOMIC_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1)); cd /tmp
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d $out/p$i -o pmc -- python $R/tools/bench_kernels.py attn_bwd > $out/p$i.log 2> $out/p$i.err
  cd $R
done
python - <<PY
import csv, collections, glob, json
res = {}
for f in sorted(glob.glob("$out/p*/pmc_counter_collection.csv")):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "fused_pp" not in r["Kernel_Name"]: continue
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    acc = collections.defaultdict(list)
    for (d, c), v in per.items(): acc[c].append(v)
    for c, v in acc.items(): res[c] = sum(v) / len(v)
json.dump(res, open("$out/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
