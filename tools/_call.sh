export TMPDIR=/tmp
timeout 900 python bench.py --mode train --steps 9 --warmup 1 > gpurun_out/train_bench.json 2> gpurun_out/train_bench.err; tail -1 gpurun_out/train_bench.json | cut -c1-1500
R=$PWD; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train -o stats -- python $R/bench.py --mode train --steps 2 --warmup 1 --accum 2 > $R/gpurun_out/train_bench_under_rocprof.json 2> $R/gpurun_out/prof_train.err
rm -rf /tmp/p_ab; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ab -o stats -- python $R/tools/bench_kernels.py attn_bwd > /dev/null 2>&1; cp /tmp/p_ab/stats_kernel_stats.csv $R/gpurun_out/attn_bwd_kernel_stats.csv
cd $R; find gpurun_out/prof_train -name "*kernel_trace.csv" -delete
