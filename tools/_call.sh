timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_vae_full_gpu.py -m gpu -x -q 2>&1 | tail -3
for h in 1 0 1 0; do echo "TG_CONV_HALO=$h"; TG_CONV_HALO=$h timeout 300 python tools/bench_vae.py --plain 2>&1 | grep vae_ | sed 's/, "out_shape.*//'; done
