timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -x -q 2>&1 | tail -3
for sel in 2 3 4; do TG_ATTN_BWD_DKDV=$sel timeout 120 python tools/attn_bwd_check.py 2>&1 | grep "513"; done
