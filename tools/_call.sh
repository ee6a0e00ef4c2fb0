set -x
timeout 1200 python -m pytest tests/test_loader_gpu.py tests/test_dit_gpu.py tests/test_vae_full_gpu.py tests/test_vae_gpu.py tests/test_t2to_gpu.py -m gpu -x -q 2>&1 | tail -30
