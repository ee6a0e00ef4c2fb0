#!/bin/bash
# rocprofv3 kernel stats of one whole To2V run (bench.py --mode e2e --chunks 1): where the end-to-end step time goes beyond the steady-state window step.
#   tools/profile_e2e.sh TAG  -> gpurun_out/TAG_e2e_kernel_stats.csv, gpurun_out/TAG_e2e_under_rocprof.json
tag=${1:-r3}
R=$PWD
out=$R/gpurun_out/prof_${tag}_e2e
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python $R/bench.py --mode e2e --chunks 1 > $R/gpurun_out/${tag}_e2e_under_rocprof.json 2> $out/err.log
cd $R
cp $(find $out -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_e2e_kernel_stats.csv
find $out -name "*kernel_trace.csv" -delete
python - <<PY
import csv, json
rows = list(csv.DictReader(open("gpurun_out/${tag}_e2e_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e9
d = json.load(open("gpurun_out/${tag}_e2e_under_rocprof.json"))
print("kernel seconds", tot, "wall", d["seconds"], "forwards", d["steps"])
for r in rows[:28]:
    print("%9.1f ms %6.2f%% calls %7s avg %9.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"]), r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
