timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -x -q 2>&1 | tail -8
timeout 600 python bench.py --mode train --layers 4 --steps 2 --warmup 1 --accum 2 2>&1 | tail -1 | cut -c1-330
