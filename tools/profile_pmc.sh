#!/bin/bash
# rocprofv3 evidence for profiles/: one --kernel-trace --stats run of bench.py, then PMC passes (own runs, no trace domains).  --no-zero-control: the zero-operand steps run the
# same kernels 20 % faster and would pull every per-kernel average down.
# Usage (on the GPU box, from the repo root):  tools/profile_pmc.sh TAG   -> gpurun_out/prof_TAG/{stats,pmc1,pmc2,pmc3}
set -u
tag=${1:-r1}
R=$PWD
out=$R/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vae --no-train --no-zero-control > $out/bench_under_rocprof.json 2> $out/stats.err
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc1 -o pmc -- python $R/bench.py --steps 1 --warmup 0 --layers 4 --no-cpu-baseline --no-vae --no-zero-control > /dev/null 2> $out/pmc1.err
rocprofv3 --pmc WRITE_SIZE SQ_BUSY_CYCLES --output-format csv -d $out/pmc2 -o pmc -- python $R/bench.py --steps 1 --warmup 0 --layers 4 --no-cpu-baseline --no-vae --no-zero-control > /dev/null 2> $out/pmc2.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $out/pmc3 -o pmc -- python $R/bench.py --steps 1 --warmup 0 --layers 4 --no-cpu-baseline --no-vae --no-zero-control > /dev/null 2> $out/pmc3.err
cd $R
find $out -name "*.csv" | head -20
# the traces are large: keep only the stats and counter tables
find $out -name "*kernel_trace.csv" -delete
ls -la $out/*
