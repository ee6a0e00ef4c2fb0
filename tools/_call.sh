timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_vae_full_gpu.py tests/test_train_gpu.py -m gpu -x -q 2>&1 | tail -8
timeout 300 python tools/bench_vae.py --plain 2>&1 | grep vae_ | sed 's/, "out_shape.*//'
TG_VAE_STREAMS=1 TG_VAE_GRAPHS=0 timeout 300 python tools/bench_vae.py 2>&1 | grep vae_ | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['op'], d['seconds'], {k:v for k,v in d['kernel_total_ms'].items() if 'norm' in k})"
