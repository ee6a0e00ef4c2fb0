set -x
timeout 900 python bench.py --mode train --layers 4 --steps 2 --warmup 1 --accum 2 2>&1 | tail -5
