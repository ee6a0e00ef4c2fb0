export TMPDIR=/tmp
TG_ATTN_BWD_DKDV=5 timeout 60 python tools/attn_bwd_check.py 2>&1 | tail -4; echo "rc $?"
TG_ATTN_BWD_DKDV=5 timeout 60 python tools/bench_kernels.py attn_bwd 2>&1 | tail -1
