timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_vae_full_gpu.py -m gpu -x -q 2>&1 | tail -6
