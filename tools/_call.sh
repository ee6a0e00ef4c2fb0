export TMPDIR=/tmp
timeout 120 python tools/attn_bwd_check.py 2>&1 | grep "dq"
(cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_base -o stats -- python $OLDPWD/tools/bench_kernels.py attn_bwd > /dev/null 2>&1)
echo "dkdv2 $(grep dkdv2 /tmp/p_base/stats_kernel_stats.csv | cut -d, -f4)  dq2 $(grep dq2 /tmp/p_base/stats_kernel_stats.csv | cut -d, -f4)"
