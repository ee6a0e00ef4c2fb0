#!/bin/bash
# per-kernel times of one tools/bench_kernels.py run under rocprofv3:  tools/prof_kernels.sh TAG WHAT [env...]  -> gpurun_out/prof_TAG_kernel_stats.csv
tag=$1; what=$2; shift 2
R=$PWD
export TMPDIR=/tmp
cd /tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o s -- python $R/tools/bench_kernels.py $what > $R/gpurun_out/prof_$tag.log 2>&1
cd $R
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/prof_${tag}_kernel_stats.csv
find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete
head -8 gpurun_out/prof_${tag}_kernel_stats.csv | cut -c1-200
