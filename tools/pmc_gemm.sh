#!/bin/bash
# PMC pass over the GEMM micro-benchmark (own run, no trace domains): tools/pmc_gemm.sh TAG [ENV=VAL ...] -> gpurun_out/pmc_gemm_TAG/
tag=$1; shift
R=$PWD; out=$R/gpurun_out/pmc_gemm_$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
env "$@" rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS \
  --output-format csv -d $out -o pmc -- python $R/tools/bench_kernels.py gemm > $out/log.txt 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"][:60] + "|grid=" + r.get("Grid_Size", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "gemm" not in k: continue
    n = max(1, cnt[(k, "SQ_BUSY_CYCLES")])
    print(k, "launches", n, {c: round(v / n) for c, v in sorted(d.items())})
PY
