rm -f gpurun_out/parity_report.json
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r2i_bench.json 2> gpurun_out/r2i_bench.err; tail -c 300 gpurun_out/r2i_bench.json
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -o stats -- python $R/tools/bench_vae.py --plain > /dev/null 2> $R/gpurun_out/r2i_vaestats.err
cp /tmp/pv/*/stats_kernel_stats.csv $R/gpurun_out/r2i_vae_kernel_stats.csv 2>/dev/null || find /tmp/pv -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r2i_vae_kernel_stats.csv \;
ls -la $R/gpurun_out/r2i_*
