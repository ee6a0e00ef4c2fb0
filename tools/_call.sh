export TMPDIR=/tmp
timeout 900 python bench.py --mode train --steps 9 --warmup 1 > gpurun_out/train_bench.json 2> gpurun_out/train_bench.err; tail -1 gpurun_out/train_bench.json | cut -c1-1800
R=$PWD; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train -o stats -- python $R/bench.py --mode train --steps 2 --warmup 1 --accum 2 > $R/gpurun_out/train_bench_under_rocprof.json 2> $R/gpurun_out/prof_train.err
cd $R; find gpurun_out/prof_train -name "*kernel_trace.csv" -delete
head -22 gpurun_out/prof_train/stats_kernel_stats.csv | cut -c1-160
