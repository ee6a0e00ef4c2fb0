set -x
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.json
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r2a_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r2a_tests.log
tail -30 gpurun_out/r2a_tests.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err; echo rc=$?
cat gpurun_out/r2a_bench.json
timeout 600 python tools/bench_vae.py > gpurun_out/r2a_vae.json 2>&1
cat gpurun_out/r2a_vae.json
timeout 900 tools/profile_vae.sh r2a > gpurun_out/r2a_profvae.log 2>&1
tail -5 gpurun_out/r2a_profvae.log
