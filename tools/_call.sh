set -x
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "optimizer_step or training_steps" 2>&1 | tail -40
