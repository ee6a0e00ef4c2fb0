run() { echo "cfg: $*"; env "$@" timeout 300 python tools/bench_vae.py --plain 2>&1 | grep vae_ | sed 's/, "out_shape.*//'; }
run TG_VAE_STREAMS=3
run TG_VAE_STREAMS=3 TG_VAE_ASSIGN=0,1,2,1,0,2,0,1,2
run TG_VAE_STREAMS=3 TG_VAE_ENQ=2,0,1,3,4,5,6,7,8
run TG_VAE_STREAMS=3 TG_VAE_ASSIGN=0,1,1,1,0,2,2,2,2 TG_VAE_ENQ=0,2,5,1,3,4,6,7,8
run TG_VAE_STREAMS=3 TG_VAE_ASSIGN=0,1,2,0,1,2,2,2,2
run TG_VAE_STREAMS=4 TG_VAE_ASSIGN=0,1,2,1,0,2,3,3,3
run TG_VAE_STREAMS=3 TG_VAE_ENQ=8,7,6,5,2,0,1,3,4
run TG_VAE_STREAMS=3
