#!/bin/bash
# rocprofv3 kernel trace of tools/bench_vae.py --plain summed per (kernel, grid): which layer shapes the VAE time goes to.
#   tools/profile_vae_by_grid.sh TAG [decode|encode]  -> gpurun_out/TAG_vae_by_grid.json
tag=${1:-r3}
what=${2:-decode}
R=$PWD
out=$R/gpurun_out/prof_${tag}_vaegrid
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python $R/tools/bench_vae.py --plain $what > $out/bench.json 2> $out/err.log
cd $R
python - <<PY
import csv, glob, json, re, collections
f = glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:50]
    key = "%s grid=%sx%sx%s" % (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    acc[key][0] += 1
    acc[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
n = 5.0      # bench_vae.py runs the op 1 + 4 times
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
json.dump({"runs": n, "kernel_ms_per_run": tot / n, "by_kernel_and_grid": [{"kernel": k, "calls_per_run": v[0] / n, "ms_per_run": v[1] / n, "avg_us": 1e3 * v[1] / v[0]} for k, v in rows[:80]]},
          open("gpurun_out/${tag}_vae_${what}_by_grid.json", "w"), indent=1)
for k, v in rows[:40]:
    print("%8.1f ms %8.1f calls %9.1f us  %s" % (v[1] / n, v[0] / n, 1e3 * v[1] / v[0], k))
print("total per run", tot / n, open("$out/bench.json").read()[:200])
PY
find $out -name "*kernel_trace.csv" -delete
