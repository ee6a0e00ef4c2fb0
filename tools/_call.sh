set -x
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -x -q 2>&1 | tail -40
