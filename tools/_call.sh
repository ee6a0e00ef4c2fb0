timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_vae_full_gpu.py -m gpu -x -q 2>&1 | tail -4
for o in 1 0 1 0; do echo "TG_CONV_ORDER=$o"; TG_CONV_ORDER=$o timeout 300 python tools/bench_vae.py --plain 2>&1 | grep vae_ | sed 's/, "out_shape.*//'; done
for o in 1 0; do echo "single stream TG_CONV_ORDER=$o"; TG_CONV_ORDER=$o TG_VAE_STREAMS=1 TG_VAE_GRAPHS=0 timeout 300 python tools/bench_vae.py decode 2>&1 | grep vae_ | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['op'], d['seconds'], {k:v for k,v in d['kernel_total_ms'].items() if 'conv3d' in k})"; done
