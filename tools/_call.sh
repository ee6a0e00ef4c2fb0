timeout 300 python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "training_steps" 2>&1 | tail -12
