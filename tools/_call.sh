export TMPDIR=/tmp
timeout 120 python tools/attn_bwd_check.py 2>&1 | grep "dq"
timeout 300 python -m pytest tests/test_train_gpu.py -m gpu -x -q 2>&1 | tail -3
R=$PWD; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab -o stats -- python $R/tools/bench_kernels.py attn_bwd > $R/gpurun_out/prof_ab.log 2>&1
cd $R; find gpurun_out/prof_ab -name "*kernel_trace.csv" -delete
grep ms gpurun_out/prof_ab.log | tail -1
head -5 gpurun_out/prof_ab/stats_kernel_stats.csv | cut -c1-150
