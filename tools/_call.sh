cat gpurun_out/r2h_e2e_chunks1.json 2>/dev/null | head -c 900
TG_CONV_HALO=0 python tools/conv_halo_check.py /tmp 2>&1 | grep -v amdgpu.ids | tail -8
TG_CONV_HALO=2 python tools/conv_halo_check.py /tmp 2>&1 | grep -v amdgpu.ids | tail -14
for h in 1 0; do echo "TG_CONV_HALO=$h"; TG_CONV_HALO=$h TG_VAE_STREAMS=1 TG_VAE_GRAPHS=0 timeout 300 python tools/bench_vae.py 2>&1 | grep vae_ | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['op'], d['seconds'], {k:v for k,v in d['kernel_total_ms'].items() if 'conv3d' in k})"; done
for h in 1 0; do echo "streams TG_CONV_HALO=$h"; TG_CONV_HALO=$h timeout 300 python tools/bench_vae.py --plain 2>&1 | grep vae_ | sed 's/, "out_shape.*//'; done
