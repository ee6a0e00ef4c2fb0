set -x
for cfg in "1 0" "1 1" "3 1" "4 1" "9 1" "4 0"; do set -- $cfg; echo "streams=$1 graphs=$2"; TG_VAE_STREAMS=$1 TG_VAE_GRAPHS=$2 timeout 300 python tools/bench_vae.py --plain 2>&1 | grep -v amdgpu.ids | tail -4 ; done > gpurun_out/r2c_streams.log 2>&1
cat gpurun_out/r2c_streams.log
timeout 900 python -m pytest tests/test_vae_full_gpu.py tests/test_vae_gpu.py -m gpu -x -q 2>&1 | tail -15
