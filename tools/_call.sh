timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -x -q 2>&1 | tail -8
timeout 900 python bench.py --mode train --steps 3 --warmup 1 --accum 3 2>/dev/null | tail -1 | cut -c1-330
