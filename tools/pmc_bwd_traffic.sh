#!/bin/bash
# WRITE_SIZE (FETCH_SIZE needs a pass of its own: 3 + 2 TCC slots do not fit one pass — rocprofv3 aborts) and duration of the one-kernel attention backward for library variants:  tools/pmc_bwd_traffic.sh NAME...  (csrc/variants/NAME.so)
R=$PWD
export TMPDIR=/tmp
for v in "$@"; do
  out=$R/gpurun_out/prof_traffic_$v; mkdir -p $out; cd /tmp
  TG_LIB_PATH=$R/tokensgen_amd/csrc/variants/$v.so timeout 120 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out -o pmc -- python $R/tools/bench_kernels.py attn_bwd > $out/bench.log 2> $out/err.log
  cd $R
  python - <<PY
import csv, collections
per = collections.defaultdict(float); meta = {}
for r in csv.DictReader(open("$out/pmc_counter_collection.csv")):
    if "fused_pp" not in r["Kernel_Name"]: continue
    per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
acc = collections.defaultdict(list)
for (d, c), v in per.items(): acc[c].append(v)
m = {c: sum(v) / len(v) for c, v in acc.items()}
print("$v", "fetch GB(x2)", round(2 * m.get("FETCH_SIZE", 0) * 1024 / 1e9, 2), "write GB", round(m.get("WRITE_SIZE", 0) * 1024 / 1e9, 2), "cycles(M)", round(m.get("GRBM_GUI_ACTIVE", 0) / 8e6, 2), open("$out/bench.log").read().strip()[:90])
PY
done
