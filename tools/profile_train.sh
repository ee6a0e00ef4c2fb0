#!/bin/bash
# rocprofv3 kernel trace of the training bench, summed per (kernel, grid size) so that GEMM shapes show up separately.
#   tools/profile_train.sh TAG [steps]  -> gpurun_out/TAG_train_kernel_stats.csv, gpurun_out/TAG_train_by_grid.json (ms per micro-step)
tag=${1:-r3}
steps=${2:-5}
R=$PWD
out=$R/gpurun_out/prof_${tag}_train
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python $R/bench.py --mode train --steps $steps --warmup 1 > $out/bench.json 2> $out/err.log
cd $R
cp $(find $out -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_train_kernel_stats.csv
python - <<PY
import csv, glob, json, re, collections
f = glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:60]
    key = "%s grid=%sx%sx%s wg=%s" % (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    acc[key][0] += 1
    acc[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
n = $steps + 1
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
res = {"micro_steps_traced": n, "kernel_ms_per_micro_step": tot / n,
       "by_kernel_and_grid": [{"kernel": k, "calls_per_micro_step": v[0] / n, "ms_per_micro_step": v[1] / n, "avg_us": 1e3 * v[1] / v[0]} for k, v in rows[:60]]}
json.dump(res, open("gpurun_out/${tag}_train_by_grid.json", "w"), indent=1)
for e in res["by_kernel_and_grid"][:45]:
    print("%8.1f ms %7.1f calls %9.1f us  %s" % (e["ms_per_micro_step"], e["calls_per_micro_step"], e["avg_us"], e["kernel"]))
print("total", tot / n)
PY
find $out -name "*kernel_trace.csv" -delete
