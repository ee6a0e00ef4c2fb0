set -x
export TMPDIR=/tmp
timeout 300 python tools/bench_kernels.py attn_bwd 2>&1 | tail -1
R=$PWD; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab -o stats -- python $R/tools/bench_kernels.py attn_bwd > /dev/null 2>&1
cd $R; find gpurun_out/prof_ab -name "*kernel_trace.csv" -delete
head -8 gpurun_out/prof_ab/stats_kernel_stats.csv | cut -c1-150
timeout 300 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "attention_bwd or processor" 2>&1 | tail -3
