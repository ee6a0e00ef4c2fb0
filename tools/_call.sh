timeout 300 python -m pytest tests/test_train_gpu.py -m gpu -x -q 2>&1 | tail -3
# one rank under torch.distributed.run: the RCCL gradient buckets of the training step (init, async all-reduce of arena slices, wait) on a real process group
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --mode train --layers 2 --steps 2 --warmup 1 --accum 2 2>&1 | tail -2 | cut -c1-600
