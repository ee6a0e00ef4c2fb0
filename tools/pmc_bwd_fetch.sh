#!/bin/bash
# FETCH_SIZE pass (own pass: 3 TCC slots) of the one-kernel attention backward for library variants:  tools/pmc_bwd_fetch.sh NAME...
R=$PWD
export TMPDIR=/tmp
for v in "$@"; do
  out=$R/gpurun_out/prof_fetch_$v; mkdir -p $out; cd /tmp
  TG_LIB_PATH=$R/tokensgen_amd/csrc/variants/$v.so timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -o pmc -- python $R/tools/bench_kernels.py attn_bwd > $out/bench.log 2> $out/err.log
  cd $R
  python - <<PY
import csv, collections
per = collections.defaultdict(float)
for r in csv.DictReader(open("$out/pmc_counter_collection.csv")):
    if "fused_pp" not in r["Kernel_Name"]: continue
    per[r["Dispatch_Id"]] += float(r["Counter_Value"])
v = list(per.values())
print("$v", "fetch GB (2 x FETCH_SIZE KiB)", round(2 * sum(v) / len(v) * 1024 / 1e9, 2))
PY
done
